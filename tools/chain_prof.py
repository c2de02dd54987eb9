"""Cycle stamps inside the one-launch encoder forward (build with RGBNM_HIPCC_FLAGS=-DCHAIN_PROF [-DCHAIN_PROF_BLK=n]).
Prints, per step of one block, the work time (barrier release -> arrival at the next barrier) and the barrier wait of every wave
of workgroup 0, in cycles of s_memtime, plus the kernel's wall time from s_memrealtime (100 MHz)."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import rgb_no_more_amd as rg
from rgb_no_more_amd import detfill, lib as L

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
m = rg.ViT(3, 16, 192, depth=12, n_classes=1000, drop_p=0.0, device="cuda", num_heads=3, head_size=64, pixel_space="DCT", ver=1)
shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
m.load_state_dict({k: torch.from_numpy(v) for k, v in detfill.fill_state_dict(shapes, 1).items()})
m.compute_dtype = torch.bfloat16
y = torch.from_numpy(detfill.normalish((B, 1, 28, 28, 8, 8), 71)).cuda()
c = torch.from_numpy(detfill.normalish((B, 2, 14, 14, 8, 8), 72)).cuda()
with torch.no_grad():
    for _ in range(3):
        m(y, c)
torch.cuda.synchronize()
dll = C.CDLL(L.LIB_PATH)
buf = (C.c_ulonglong * (8 * 8 * 128))()
assert dll.rgbnm_chain_prof_read(buf) == 0
t = np.array(buf, dtype=np.uint64).reshape(8, 8, 128).astype(np.int64)
names = {}
for h in range(3):
    names[4 * h] = f"q{h}"; names[4 * h + 1] = f"k{h}"; names[4 * h + 2] = f"v{h}"; names[4 * h + 3] = f"attn{h}"
for h in range(3):
    names[12 + h] = f"proj{h}"
names[15] = "(14b) LN2"
for ch in range(12):
    names[16 + ch] = f"mlp{ch}"
names[28] = "tail"
for wg in (0, 5):
    T = t[wg]
    print(f"== workgroup {wg}: kernel wall {(T[0, 127] - T[0, 126]) / 100.0:.1f} us; block span {T[0, 58] - T[0, 1]} cycles "
          f"(wave 0, behind barrier 0 -> end of block)")
    print("step        " + "".join(f"   w{w}:work/wait" for w in range(8)))
    tot = np.zeros((8, 2))
    for s in range(29):
        row = f"{names[s]:10s}"
        for w in range(8):
            rel, arr = T[w, 2 * s + 1], T[w, 2 * s + 2] if s < 28 else T[w, 58]
            nxt_rel = T[w, 2 * s + 3] if s < 28 else arr
            work, wait = arr - rel, nxt_rel - arr
            if w == 7 and s == 28:
                work = wait = 0
            tot[w] += (work, wait)
            row += f" {work:7d}/{wait:6d}"
        print(row)
    print("total     " + "".join(f" {int(a):7d}/{int(b):6d}" for a, b in tot))
