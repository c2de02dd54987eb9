#!/usr/bin/env python
"""A/B timing of runtime switches on the GPU box: per-kernel ms for representative shapes (torch events on the
current stream) and whole-step ms for each option combination.  usage: python tools/ab_bench.py"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rgb_no_more_amd as rg
from rgb_no_more_amd import lib as L

DEV = "cuda"


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3  # us


def main():
    M = 256 * 196
    dt = torch.bfloat16
    lib = L.lib()
    x = torch.randn(M, 192, device=DEV).to(dt)
    x4 = torch.randn(M, 768, device=DEV).to(dt)
    x3 = torch.randn(M, 576, device=DEV).to(dt)
    res = {}
    for name, (A, N, K, epi) in {"qkv": (x, 576, 192, 0), "fc1_gelu": (x, 768, 192, 2), "fc2_res": (x4, 192, 768, 1),
                                 "dgelu": (x, 768, 192, 4), "dxn_k768": (x4, 192, 768, 0), "dxn_k576": (x3, 192, 576, 0)}.items():
        W = (torch.randn(N, K, device=DEV) * 0.05).to(dt)
        b = torch.randn(N, device=DEV)
        Cc = torch.empty(M, N, device=DEV, dtype=dt)
        C2 = torch.empty(M, N, device=DEV, dtype=dt)
        R = torch.randn(M, N, device=DEV).to(dt)
        for staged in (0, 1):
            lib.rgbnm_set_option(b"nt_staged", staged)
            f = lambda: L.check(lib.rgbnm_gemm_nt(1, epi, A.data_ptr(), K, W.data_ptr(), K, Cc.data_ptr(), N, b.data_ptr(),
                                                  R.data_ptr(), N, C2.data_ptr(), N, None, 0, M, N, K, 0, L.stream()))
            res[f"nt {name} staged={staged}"] = timeit(f)
    lib.rgbnm_set_option(b"nt_staged", 1)
    for name, (dY, X) in {"dw1": (x4, x), "dw2": (x, x4), "dwqkv": (x3, x), "dwproj": (x, x)}.items():
        No, Ki = dY.shape[1], X.shape[1]
        dW = torch.empty(No, Ki, device=DEV)
        db = torch.empty(No, device=DEV)
        wsb = lib.rgbnm_gemm_tn_workspace(M, No, Ki)
        ws = torch.empty(wsb, device=DEV, dtype=torch.uint8)
        for tr in (0, 1):
            lib.rgbnm_set_option(b"tn_tr", tr)
            f = lambda: L.check(lib.rgbnm_gemm_tn(1, dY.data_ptr(), No, X.data_ptr(), Ki, dW.data_ptr(), db.data_ptr(), M, No,
                                                  Ki, 0, 0, ws.data_ptr(), wsb, L.stream()))
            res[f"tn {name} tr={tr} (incl. reduce)"] = timeit(f)
    lib.rgbnm_set_option(b"tn_tr", 1)
    for k, v in res.items():
        print(f"{k:40s} {v:9.1f} us")


if __name__ == "__main__":
    main()
