#!/usr/bin/env python
"""gemm_nt_kstream.hip against the row-panel kernels of gemm_nt_kpipe.hip (option nt_kstream 1 / 0): bit equality of every output
and the launch times, shape by shape.  usage: python tools/kstream_check.py [M]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rgb_no_more_amd import lib as L

M = int(sys.argv[1]) if len(sys.argv) > 1 else 50176
lib = L.lib()
L.check(lib.rgbnm_gelu_table_init(L.stream()))
dt = torch.bfloat16


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


for name, N, K, epi in (("fc1+gelu", 1536, 384, 2), ("qkv", 1152, 384, 0), ("dX fc1", 384, 1536, 0), ("dX qkv", 384, 1152, 0),
                        ("dX proj", 384, 384, 0), ("fc2+res", 384, 1536, 1), ("dgelu", 1536, 384, 4)):
    torch.manual_seed(1)
    A = torch.randn(M, K, device="cuda").to(dt)
    W = (torch.randn(N, K, device="cuda") * 0.05).to(dt)
    b = torch.randn(N, device="cuda")
    R = torch.randn(M, N, device="cuda").to(dt)
    outs = {}
    t = {}
    for opt in (0, 1):
        L.check(lib.rgbnm_set_option(b"nt_kstream", 2 * opt))
        Cc = torch.full((M, N), 7.0, device="cuda", dtype=dt)
        C2 = torch.full((M, N), 7.0, device="cuda", dtype=dt)
        f = lambda: L.check(lib.rgbnm_gemm_nt(1, epi, A.data_ptr(), K, W.data_ptr(), K, Cc.data_ptr(), N, b.data_ptr(), R.data_ptr(), N,  # noqa: E731
                                              C2.data_ptr(), N, None, 0, M, N, K, 0, L.stream()))
        f()
        torch.cuda.synchronize()
        outs[opt] = (Cc.clone(), C2.clone())
        t[opt] = timeit(f)
    same = torch.equal(outs[0][0], outs[1][0]) and (epi != 2 or torch.equal(outs[0][1], outs[1][1]))
    nbad = int((outs[0][0] != outs[1][0]).sum())
    print(f"{name:9s} N {N:5d} K {K:5d} epi {epi}: kpipe {t[0]:7.1f} us  kstream {t[1]:7.1f} us  ({t[0] / t[1]:.2f}x)  bit-equal {same}  differing {nbad}  finite {bool(torch.isfinite(outs[1][0].float()).all())}")
