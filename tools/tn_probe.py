#!/usr/bin/env python
"""The four weight-gradient GEMMs (dW = dY^T X, fp32 out, + db) of one encoder block through rgbnm_gemm_tn at a given width / row
count, with the algorithmic bytes (both operands read once) and the rate they correspond to.
usage: python tools/tn_probe.py E M [option=value ...]      (SwinV2-T stages at B = 256: 192 524288 (stage 1, row-paired) / 192 262144 / 384 65536 / 768 16384)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rgb_no_more_amd import lib as L

DEV = "cuda"


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    E, M = int(sys.argv[1]), int(sys.argv[2])
    lib = L.lib()
    for kv in sys.argv[3:]:                      # library options, e.g. tn_pack=0
        k, v = kv.split("=")
        L.check(lib.rgbnm_set_option(k.encode(), int(v)))
    dt = torch.bfloat16
    tot = 0.0
    for name, No, Ki in (("dW qkv", 3 * E, E), ("dW proj", E, E), ("dW fc1", 4 * E, E), ("dW fc2", E, 4 * E)):
        dY = torch.randn(M, No, device=DEV).to(dt)
        X = torch.randn(M, Ki, device=DEV).to(dt)
        dW = torch.empty(No, Ki, device=DEV)
        db = torch.empty(No, device=DEV)
        wsb = lib.rgbnm_gemm_tn_workspace(M, No, Ki)
        ws = torch.empty(wsb, device=DEV, dtype=torch.uint8)
        f = lambda: L.check(lib.rgbnm_gemm_tn(1, dY.data_ptr(), No, X.data_ptr(), Ki, dW.data_ptr(), db.data_ptr(), M, No, Ki, 0, 0,  # noqa: E731
                                              ws.data_ptr(), wsb, L.stream()))
        t = timeit(f)
        lib_t = timeit(lambda: torch.matmul(dY.t(), X))
        mb = 2.0 * M * (No + Ki) / 1e6
        gf = 2.0 * M * No * Ki / 1e9
        tot += t
        print(f"{name:8s} No={No:5d} Ki={Ki:5d}  ours {t:7.1f} us  lib {lib_t:7.1f}  {mb:7.1f} MB  {mb / t:5.2f} TB/s  {gf / t * 1e3:6.1f} TF/s  workspace {wsb / 1e6:.1f} MB")
    print(f"sum {tot:.1f} us")


if __name__ == "__main__":
    main()
