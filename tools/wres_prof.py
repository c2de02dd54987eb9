"""Experiment only (build with RGBNM_HIPCC_FLAGS=-DWRES_PROF): cycle stamps inside gemm_nt_wres_kernel."""
import ctypes as C, sys
import numpy as np, torch
from rgb_no_more_amd import lib as L
M, N, epi = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
A = torch.randn(M, 192, device="cuda").bfloat16(); W = torch.randn(N, 192, device="cuda").bfloat16() * 0.1
b = torch.randn(N, device="cuda"); R = torch.randn(M, N, device="cuda").bfloat16()
Cc = torch.empty(M, N, device="cuda", dtype=torch.bfloat16); C2 = torch.empty_like(Cc)
def run():
    L.check(L.lib().rgbnm_gemm_nt(1, epi, A.data_ptr(), 192, W.data_ptr(), 192, Cc.data_ptr(), N, b.data_ptr(), R.data_ptr(), N,
                                  C2.data_ptr(), N, None, 0, M, N, 192, 0, L.stream()))
import time
t_end = time.time() + 3.0
while time.time() < t_end:
    for _ in range(50): run()
    torch.cuda.synchronize()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200): run()
e1.record(); torch.cuda.synchronize()
print('avg kernel+launch us:', e0.elapsed_time(e1) * 1000 / 200)
out = np.zeros(320 * 8 * 32, dtype=np.uint64)
f = L.lib().rgbnm_debug_wres_prof; f.restype = C.c_int; f.argtypes = [C.c_void_p]
assert f(out.ctypes.data) == 0
p = out.reshape(320, 8, 32).astype(np.int64)
ok = p[:, :7, 0] > 0
names = ["start", "after_barrier"] + [f"s{s}_{n}" for s in range(6) for n in ("parked+issued", "mfma_done", "epi_done", "stores_drained")]
prev = p[:, :7, 0]
for i in range(1, 26):
    cur = p[:, :7, i]
    m = ok & (cur > 0)
    if not m.any(): break
    d = (cur - prev)[m]
    print(f"{names[i]:22s} n={d.size:5d} delta mean={d.mean():8.0f} min={d.min():8.0f} max={d.max():8.0f}   since start mean={(cur - p[:, :7, 0])[m].mean():9.0f}")
    prev = np.where(cur > 0, cur, prev)
