"""Cycle stamps inside the one-launch encoder backward (build with RGBNM_HIPCC_FLAGS=-DCHAINB_PROF [-DCHAINB_PROF_BLK=n]): per
barrier interval of one block, work (release -> arrival at the next barrier) and wait of every wave of two workgroups."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import rgb_no_more_amd as rg
from rgb_no_more_amd import lib as L
from test_chain_fwd import build

m, y, c, tgt = build(12, 256)
for _ in range(3):
    m.zero_grad()
    rg.cls_transforms.cross_entropy(m(y, c), tgt, grad_dtype=torch.bfloat16).backward()
torch.cuda.synchronize()
dll = C.CDLL(L.LIB_PATH)
buf = (C.c_ulonglong * (8 * 8 * 128))()
assert dll.rgbnm_chainb_prof_read(buf) == 0
t = np.array(buf, dtype=np.uint64).reshape(8, 8, 128).astype(np.int64)
names = [f"mlp{c}" for c in range(12)] + [f"LN2epi{k}" for k in range(7)] + ["P:frags"] + [f"P{h}" for h in range(3)] + ["P:stored", "A:start"]
for h in range(3):
    names += [f"A{h}:phaseA", f"A{h}:mid2", f"A{h}:phaseB", f"A{h}:end2"]
names += [f"X{j}" for j in range(9)] + [f"LN1epi{k}" for k in range(7)] + ["END"]
# names[s] = what runs AFTER barrier s-1 and BEFORE barrier s?  (interval s = release of barrier s -> arrival at barrier s + 1)
for wg in (0, 5):
    T = t[wg]
    print(f"== workgroup {wg}: block span {T[0, 2 * 53] - T[0, 1]} ticks (wave 0, release of barrier 0 -> arrival at the last)")
    print("after barrier".ljust(14) + "".join(f"   w{w}:work/wait" for w in range(8)))
    groups = {}
    for s in range(53):
        row = f"{s:2d} {names[s]:10s}"
        for w in range(8):
            rel, arr, nrel = T[w, 2 * s + 1], T[w, 2 * s + 2], T[w, 2 * s + 3]
            row += f" {arr - rel:7d}/{nrel - arr:6d}"
        print(row)
        key = names[s].split(":")[0].rstrip("0123456789")
        groups[key] = groups.get(key, 0) + (T[4, 2 * s + 3] - T[4, 2 * s + 1])
    print("wave 4, release to release, by phase:", {k: int(v) for k, v in groups.items()})
