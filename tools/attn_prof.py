"""Experiment only (build with RGBNM_HIPCC_FLAGS=-DATTN_PROF): cycle stamps inside attn2_bwd_kernel."""
import ctypes as C, time
import numpy as np, torch
from rgb_no_more_amd import lib as L
B, N, H = 256, 196, 3
qkv = torch.randn(B, N, 3 * H * 64, device="cuda").bfloat16()
out = torch.empty(B, N, H * 64, device="cuda", dtype=torch.bfloat16); dout = torch.randn_like(out)
lse = torch.empty(B * H * N, device="cuda"); dqkv = torch.empty_like(qkv)
scale = 1.0 / (192 ** 0.5)
L.check(L.lib().rgbnm_attention_fwd(1, qkv.data_ptr(), out.data_ptr(), lse.data_ptr(), B, N, H, scale, L.stream()))
def run():
    L.check(L.lib().rgbnm_attention_bwd(1, qkv.data_ptr(), out.data_ptr(), dout.data_ptr(), lse.data_ptr(), dqkv.data_ptr(), B, N, H, scale, L.stream()))
t_end = time.time() + 3.0
while time.time() < t_end:
    for _ in range(50): run()
    torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200): run()
e1.record(); torch.cuda.synchronize()
print('avg kernel+launch us:', e0.elapsed_time(e1) * 1000 / 200)
buf = np.zeros(768 * 8 * 8, dtype=np.uint64)
f = L.lib().rgbnm_debug_attn_prof; f.restype = C.c_int; f.argtypes = [C.c_void_p]
assert f(buf.ctypes.data) == 0
p = buf.reshape(768, 8, 8).astype(np.int64)[:256, :7, :]      # 256 persistent workgroups x 7 waves x 8 stamps
names = ["top wait done", "-", "phase A + park dQ (+own kf/vf)", "mid + mid2", "phase B loop", "park dK",
         "end barrier", "park dV + own loads issue + end2"]
print("persistent kernel: second (image, head) pair of every workgroup; per-wave cycle deltas")
for i in range(1, 8):
    d = p[:, :, i] - p[:, :, i - 1]
    per_wave = " ".join(f"{d[:, w].mean():7.0f}" for w in range(7))
    print(f"{names[i]:24s} mean={d.mean():8.0f} max={d.max():8.0f} | per wave: {per_wave}")
print("pair total", (p[:, :, 7] - p[:, :, 0]).mean())
