#!/usr/bin/env python
"""Experiment only (build with RGBNM_HIPCC_FLAGS=-DKS_PROF): where the waves of gemm_nt_kstream.hip spend a k-tile.
usage: kstream_prof.py N K [epi] [M]"""
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
L.check(lib.rgbnm_set_option(b"nt_kstream", 1))
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
buf = np.zeros(256 * 8 * 6, dtype=np.uint64)
fn = C.CDLL(L.LIB_PATH).rgbnm_debug_ks_prof
fn.restype = C.c_int
fn.argtypes = [C.c_void_p]
assert fn(buf.ctypes.data) == 0
p = buf.reshape(256, 8, 6).astype(np.float64)
for wg in (0, 9, 255):
    for w in range(7):
        q = p[wg, w]
        n = max(q[4], 1)
        print(f"wg {wg:3d} wave {w}: k-tiles {int(q[4]):3d}  per k-tile: vmcnt wait {q[0] / n:6.0f}  barrier {q[1] / n:6.0f}  slice in {q[2] / n:6.0f}  fragments + MFMAs + DMA {q[3] / n:6.0f}  slice out {q[5] / n:6.0f}")
q = p[:, :7].reshape(-1, 6)
q = q[q[:, 4] > 0]
n = q[:, 4:5]
print("mean over all waves, per k-tile: vmcnt wait %.0f  barrier %.0f  slice in %.0f  MFMA block %.0f  slice out %.0f" % (*(q[:, :4] / n).mean(0), (q[:, 5:6] / n).mean()))
