#!/usr/bin/env python
"""Window of the table GELU on this device, and the table checked against torch's erf GELU (sanity, not bit equality)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rgb_no_more_amd import lib as L
lib = L.lib()
torch.cuda.set_device(0)
L.check(lib.rgbnm_gelu_table_init(L.stream()))
win = (C.c_int * 16)()
full = np.zeros(65536, dtype=np.uint32)
L.check(lib.rgbnm_gelu_table_info(win, full.ctypes.data))
w = list(win)
print("valid", w[0], "A0 %#x P1 %#x N1 %#x" % (w[1], w[2], w[3]), "image dwords", w[4], "bytes", 4 * w[4], "gp(-large) %#x" % w[5])
def f(bits): return torch.from_numpy(bits.astype(np.int32) << 16).view(torch.float32)
u = f(np.arange(65536, dtype=np.uint32))
g, gp = f(full & 0xFFFF), f(full >> 16)
fin = torch.isfinite(u)
ref = torch.nn.functional.gelu(u.double()).float()
sel = fin & (u.abs() > 1e-3) & (u.abs() < 8)
err = ((g - ref).abs() / ref.abs())[sel]
refp = (0.5 * (1 + torch.erf(u.double() / 2 ** 0.5)) + u.double() * torch.exp(-0.5 * u.double() ** 2) / (2 * np.pi) ** 0.5).float()
print("1e-3 < |u| < 8: max rel err of table gelu vs erf gelu %.3e (half a bf16 ulp = 3.9e-3), max abs err of gelu' %.3e" % (err.max().item(), (gp - refp).abs()[sel].max().item()))
print("A0 = %g, P1 = %g, N1 = %g" % (f(np.array([w[1]], dtype=np.uint32)).item(), f(np.array([w[2]], dtype=np.uint32)).item(), f(np.array([w[3]], dtype=np.uint32)).item()))
A = np.arange(0x100, w[1], dtype=np.uint32)
assert (full[A] == ((A - 0x80) | (0x3F00 << 16))).all() and (full[0x8000 | A] == ((0x8000 | (A - 0x80)) | (0x3F00 << 16))).all()
A = np.arange(w[2], 0x7F80, dtype=np.uint32)
assert (full[A] == (A | (0x3F80 << 16))).all()
A = np.arange(w[3], 0x7F80, dtype=np.uint32)
assert (full[0x8000 | A] == (0x8000 | (w[5] << 16))).all()
print("closed forms outside the window reproduce the arithmetic for every finite input with |u| >= 2.36e-38")
