#!/usr/bin/env python
"""Experiment only (build with RGBNM_HIPCC_FLAGS=-DKP_PROF): stamps of wave 0 per unit of the persistent row-panel GEMM (kp7):
unit start -> k-loop done -> staging written (pass 1) -> stores issued + final barrier (pass 2).  usage: kpipe_prof_persist.py N K [epi] [M]"""
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
p = buf[:256 * 8 * 4].reshape(256, 8, 4).astype(np.int64)
t0 = p[:, 0, 0].min()
for wg in (0, 1, 100, 255):
    print(f"workgroup {wg} (s_memtime ticks = shader cycles): unit start | k-loop | pass 1 | pass 2 (stores issued, barrier) | gap to next unit")
    for ui in range(8):
        q = p[wg, ui]
        if q[3] <= q[0]:
            continue
        nxt = p[wg, ui + 1, 0] if ui + 1 < 8 and p[wg, ui + 1, 3] > p[wg, ui + 1, 0] else q[3]
        print(f"  unit {ui}: start {q[0] - t0:6d}  k-loop {q[1] - q[0]:5d}  pass1 {q[2] - q[1]:5d}  pass2 {q[3] - q[2]:5d}  gap {nxt - q[3]:4d}")
ok = p[:, :, 3] > p[:, :, 0]
d = np.stack([p[:, :, 1] - p[:, :, 0], p[:, :, 2] - p[:, :, 1], p[:, :, 3] - p[:, :, 2]], -1)[ok]
print("mean over all units (ticks): k-loop %.0f  pass1 %.0f  pass2 %.0f ; units %d ; launch span %d ticks" % (*d.mean(0), ok.sum(), p[:, :, 3].max() - t0))
