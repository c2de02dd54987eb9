import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import rgb_no_more_amd as rg
from rgb_no_more_amd import lib as L
from test_chain_fwd import build

def run(m, y, c, tgt, chain):
    L.lib().rgbnm_set_option(b"bwd_chain", chain)
    m.train(); m.zero_grad()
    logits = m(y, c)
    arena = logits.grad_fn.st.arena
    loss = rg.cls_transforms.cross_entropy(logits, tgt, grad_dtype=torch.bfloat16)
    loss.backward(); torch.cuda.synchronize()
    L.lib().rgbnm_set_option(b"bwd_chain", 1)
    f = lambda t: t.float().cpu().numpy().copy()
    if chain:
        return dict(du=f(arena.du_blk[0]), dx_mid=f(arena.dxmid_blk[0]), dattn=f(arena.dattn_chain), dqkv=f(arena.dqkv_blk[0]), dx=f(arena.dx_blk[0]))
    return dict(du=f(arena.du), dx_mid=f(arena.dx_mid), dattn=f(arena.dattn), dqkv=f(arena.dqkv), dx=f(arena.dx[0]), dx1=f(arena.dx[1]))

import sys as _s
m, y, c, tgt = build(1, int(_s.argv[1]) if len(_s.argv) > 1 else 256)
a = run(m, y, c, tgt, 1); b = run(m, y, c, tgt, 0)
for k in a:
    ref = b[k]
    if k == "dx":
        d0 = np.abs(a[k] - b["dx"]).max(); d1 = np.abs(a[k] - b["dx1"]).max()
        ref = b["dx"] if d0 <= d1 else b["dx1"]
    ne = a[k] != ref
    print(k, "mismatching elements:", int(ne.sum()), "of", ne.size, "max abs", float(np.abs(a[k] - ref).max()), "scale", float(np.abs(ref).max()))
    if ne.any():
        idx = np.argwhere(ne)
        rows = idx[:, 0] % 196
        print("   token rows (mod 196) histogram of first mismatches:", np.bincount(rows, minlength=196).nonzero()[0][:40], " cols:", np.unique(idx[:, 1])[:40], " images:", np.unique(idx[:, 0] // 196)[:20])

print("== reproducibility of the chain path")
a2 = run(m, y, c, tgt, 1)
for k in a: print(k, "run-to-run mismatches:", int((a[k] != a2[k]).sum()))
print("== chain vs per-op with the fused epilogues off (mlp_bwd=0, ln_fuse=0)")
L.lib().rgbnm_set_option(b"mlp_bwd", 0); L.lib().rgbnm_set_option(b"ln_fuse", 0)
b2 = run(m, y, c, tgt, 0)
L.lib().rgbnm_set_option(b"mlp_bwd", 1); L.lib().rgbnm_set_option(b"ln_fuse", 1)
for k in ("du", "dx_mid", "dattn", "dqkv"):
    print(k, "mismatches vs unfused per-op:", int((a[k] != b2[k]).sum()), " fused per-op vs unfused per-op:", int((b[k] != b2[k]).sum()))
