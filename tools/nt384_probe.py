#!/usr/bin/env python
"""The eight activation GEMMs of one JPEG-S encoder block (E = 384, B = 256: M = 50176 rows) through rgbnm_gemm_nt, with the
library GEMM torch dispatches to (hipBLASLt / rocBLAS) beside each as a yardstick -- what a tuned plain GEMM of that shape takes
on this GPU.  The yardstick has no fused epilogue: its time is a lower bound for the plain part only.
usage: python tools/nt384_probe.py [E] [out.json | -] [M] [option=value ...]     (SwinV2-T stages at B = 256: E, M = 96, 1048576 / 192, 262144 / 384, 65536 / 768, 16384)"""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rgb_no_more_amd import lib as L

DEV = "cuda"
M = 256 * 196


def timeit(fn, n=30):
    for _ in range(4):
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
    global M
    E = int(sys.argv[1]) if len(sys.argv) > 1 else 384
    if len(sys.argv) > 3:
        M = int(sys.argv[3])
    lib = L.lib()
    L.check(lib.rgbnm_gelu_table_init(L.stream()))      # as the model constructors do: the fc1 + GELU epilogue's table form
    for kv in sys.argv[4:]:                      # library options, e.g. kp_persist=0
        k, v = kv.split("=")
        L.check(lib.rgbnm_set_option(k.encode(), int(v)))
    dt = torch.bfloat16
    g = torch.Generator(device=DEV)
    g.manual_seed(0)
    rn = lambda *s: torch.randn(*s, device=DEV, generator=g).to(dt)  # noqa: E731
    rows = []

    def nt(name, N, K, epi, res=False, c2=False):
        A, W = rn(M, K), (torch.randn(N, K, device=DEV, generator=g) * 0.05).to(dt)
        b = torch.randn(N, device=DEV, generator=g)
        Cc = torch.empty(M, N, device=DEV, dtype=dt)
        R = rn(M, N) if res else None
        C2 = torch.empty(M, N, device=DEV, dtype=dt) if c2 else None
        f = lambda: L.check(lib.rgbnm_gemm_nt(1, epi, A.data_ptr(), K, W.data_ptr(), K, Cc.data_ptr(), N, b.data_ptr(),  # noqa: E731
                                              L.ptr(R), N, L.ptr(C2), N, None, 0, M, N, K, 0, L.stream()))
        ours = timeit(f)
        Wt = W.t().contiguous()
        lib_t = timeit(lambda: torch.matmul(A, Wt, out=Cc))
        lib_nt = timeit(lambda: torch.nn.functional.linear(A, W))
        gf = 2.0 * M * N * K / 1e9
        mb = 2.0 * M * (K + N + (N if res else 0) + (N if c2 else 0)) / 1e6      # algorithmic bytes: A in, C (+ R in, + C2) out
        rows.append(dict(name=name, N=N, K=K, epi=epi, ours_us=round(ours, 1), torch_mm_us=round(lib_t, 1),
                         torch_linear_us=round(lib_nt, 1), gflop=round(gf, 1), MB=round(mb, 1), ours_TBps=round(mb / ours, 2), ours_tflops=round(gf / ours * 1e3, 1),
                         lib_tflops=round(gf / min(lib_t, lib_nt) * 1e3, 1)))
        print(rows[-1], flush=True)

    I = E * 3 if E != 1024 else 768 * 3
    nt("qkv", I, E, 0)
    nt("proj+res", E, I // 3, 1, res=True)
    nt("fc1+gelu", 4 * E, E, 2, c2=True)
    nt("fc2+res", E, 4 * E, 1, res=True)
    nt("dgelu (dX of fc2)", 4 * E, E, 4, res=True)
    nt("dX of fc1", E, 4 * E, 0)
    nt("dX of proj", I // 3, E, 0)
    nt("dX of qkv", E, I, 0)
    tot = sum(r["ours_us"] for r in rows)
    print("sum ours %.1f us, library (plain) %.1f us" % (tot, sum(min(r["torch_mm_us"], r["torch_linear_us"]) for r in rows)))
    if len(sys.argv) > 2 and sys.argv[2] != "-":
        json.dump(rows, open(sys.argv[2], "w"), indent=1)


if __name__ == "__main__":
    main()
