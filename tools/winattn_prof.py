#!/usr/bin/env python
"""Experiment only (build with RGBNM_HIPCC_FLAGS=-DWIN_PROF): cycle stamps inside win_attn_bwd_kernel, second window of every wave.
usage: RGBNM_HIPCC_FLAGS=-DWIN_PROF python rgb-no-more_amd/build.py && python tools/winattn_prof.py"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from rgb_no_more_amd import lib as L

B, res, Cc, heads, shift = 256, 64, 96, 3, int(sys.argv[1]) if len(sys.argv) > 1 else 0
FWD = len(sys.argv) > 2 and sys.argv[2] == "fwd"         # built with -DWIN_PROF=2
lib = L.lib()
M = B * res * res
dt = torch.bfloat16
qkv = torch.randn(M, 3 * Cc, device="cuda").to(dt)
bias = torch.randn(heads, 64, 64, device="cuda") * 0.5
scale = torch.full((heads,), 10.0, device="cuda")
out = torch.empty(M, Cc, device="cuda", dtype=dt)
nwin = B * (res // 8) ** 2
lse = torch.empty(nwin * heads * 64, device="cuda")
dout = torch.randn(M, Cc, device="cuda").to(dt)
dqkv = torch.empty_like(qkv)
dbias = torch.empty_like(bias)
dsp = torch.empty(nwin * heads, device="cuda")
wsb = lib.rgbnm_window_attention_bwd_workspace(B, res, heads)
ws = torch.empty(wsb, device="cuda", dtype=torch.uint8)
L.check(lib.rgbnm_window_attention_fwd(1, qkv.data_ptr(), bias.data_ptr(), scale.data_ptr(), out.data_ptr(), lse.data_ptr(), B, res, Cc,
                                       heads, shift, L.stream()))
for _ in range(5):
    L.check(lib.rgbnm_window_attention_bwd(1, qkv.data_ptr(), out.data_ptr(), dout.data_ptr(), bias.data_ptr(), None,
                                           scale.data_ptr(), lse.data_ptr(), dqkv.data_ptr(), dbias.data_ptr(), dsp.data_ptr(), B, res, Cc,
                                           heads, shift, ws.data_ptr(), wsb, L.stream()))
torch.cuda.synchronize()
buf = np.zeros(1024 * 4 * 8, dtype=np.uint64)
f = lib.rgbnm_debug_win_prof
f.restype = C.c_int
f.argtypes = [C.c_void_p]
assert f(buf.ctypes.data) == 0
p = buf.reshape(1024, 4, 8).astype(np.int64)[:255 if not FWD else 510]
names = ["", "park", "prefetch issue + LDS drain", "query tile 0 (S, softmax, PV, park O)", "query tile 1", "store O"] if FWD else ["", "park (unpack, norms, LDS writes)", "prefetch issue + LDS drain", "phase A (dq, dbias, dscale) + stores", "phase B (dk, dv) + stores"]
for i in range(1, len(names)):
    d = p[:, :, i] - p[:, :, i - 1]
    print(f"{names[i]:44s} mean {d.mean():8.0f}  min {d.min():8.0f}  max {d.max():8.0f}")
print("window total", (p[:, :, len(names) - 1] - p[:, :, 0]).mean())
