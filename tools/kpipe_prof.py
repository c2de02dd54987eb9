#!/usr/bin/env python
"""Experiment only (build with RGBNM_HIPCC_FLAGS=-DKP_PROF): where a workgroup of the row-panel GEMM (kp7) spends its life --
launch to first k-tile landed (prologue), k-loop, epilogue.  usage: python tools/kpipe_prof.py N K [epi] [M]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from rgb_no_more_amd import lib as L

N, K = int(sys.argv[1]), int(sys.argv[2])
epi = int(sys.argv[3]) if len(sys.argv) > 3 else 0
M = int(sys.argv[4]) if len(sys.argv) > 4 else 50176
lib = L.lib()
L.check(lib.rgbnm_set_option(b"kp_persist", 0))        # the stamps sit in the one-tile-per-workgroup kernel
dt = torch.bfloat16
A = torch.randn(M, K, device="cuda").to(dt)
W = (torch.randn(N, K, device="cuda") * 0.05).to(dt)
b = torch.randn(N, device="cuda")
R = torch.randn(M, N, device="cuda").to(dt)
Cc = torch.empty(M, N, device="cuda", dtype=dt)
C2 = torch.empty_like(Cc)
f = lambda: L.check(lib.rgbnm_gemm_nt(1, epi, A.data_ptr(), K, W.data_ptr(), K, Cc.data_ptr(), N, b.data_ptr(), R.data_ptr(), N,  # noqa: E731
                                      C2.data_ptr(), N, None, 0, M, N, K, 0, L.stream()))
for _ in range(5):
    f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    f()
e1.record()
torch.cuda.synchronize()
print("launch us", e0.elapsed_time(e1) / 20 * 1e3)
buf = np.zeros(4096 * 4, dtype=np.uint64)
fn = lib.rgbnm_debug_kp_prof
fn.restype = C.c_int
fn.argtypes = [C.c_void_p]
assert fn(buf.ctypes.data) == 0
nwg = ((M + 223) // 224 + 7) // 8 * 8 * (N // 192)
p = buf.reshape(4096, 4).astype(np.int64)[:min(nwg, 4096)]
p = p[p[:, 3] > p[:, 0]]
d = np.diff(p, axis=1)
print(f"{len(p)} workgroups (100 MHz ticks x 24 = cycles at 2.4 GHz): prologue {d[:, 0].mean():.0f}  k-loop {d[:, 1].mean():.0f}  epilogue {d[:, 2].mean():.0f}  total {(p[:, 3] - p[:, 0]).mean():.0f}")
print("kernel span ticks", p[:, 3].max() - p[:, 0].min())
