#!/usr/bin/env python
"""Run-to-run bit identity of the one-launch encoder forward / backward at the bench configuration (B = 256, depth 12):
every saved tensor of every block, the logits (train mode and no-grad mode: the no-grad arena ping-pongs two x buffers and shares
one block's activation buffers), and every gradient.  Prints, per differing tensor, how many elements differ and which rows
(token index inside the image -> owning wave) they belong to.  usage: chain_determinism.py [B] [depth] [runs]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import rgb_no_more_amd as rg
from rgb_no_more_amd import lib as L
from test_chain_fwd import build, SAVED

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
D = int(sys.argv[2]) if len(sys.argv) > 2 else 12
R = int(sys.argv[3]) if len(sys.argv) > 3 else 4
m, y, c, tgt = build(D, B)
lib = L.lib()
lib.rgbnm_set_option(b"fwd_chain", 1)
lib.rgbnm_set_option(b"bwd_chain", 1)


def describe(name, a, b):
    d = a != b
    if a.dtype.is_floating_point:
        d &= ~(torch.isnan(a) & torch.isnan(b))
    n = int(d.sum())
    if n == 0:
        return 0
    rows = d.reshape(d.shape[0], -1).any(1).nonzero().flatten().cpu().numpy()
    msg = f"  DIFF {name}: {n} elements in {rows.size} rows"
    if a.shape[0] == B * 196:
        tok = rows % 196
        waves = np.bincount(tok // 32, minlength=7)
        msg += f"; images {np.unique(rows // 196)[:8]}; rows per wave {waves.tolist()}"
        cols = d.reshape(d.shape[0], -1).any(0).nonzero().flatten().cpu().numpy()
        msg += f"; cols {cols[:6]}..{cols[-3:]} ({cols.size})"
    print(msg)
    return n


def train_pass():
    m.train()
    m.zero_grad()
    logits = m(y, c)
    arena = logits.grad_fn.st.arena
    saved = {f"blk{i}.{k}": arena.blk[i][k].clone() for i in range(D) for k in SAVED}
    for i in range(D + 1):
        saved[f"x{i}"] = arena.x[i].clone()
    saved["logits"] = logits.detach().clone()
    loss = rg.cls_transforms.cross_entropy(logits, tgt, grad_dtype=torch.bfloat16)
    loss.backward()
    torch.cuda.synchronize()
    for n, p in m.named_parameters():
        saved["grad." + n] = p.grad.clone()
    return saved


ref = train_pass()
total = 0
for r in range(1, R):
    cur = train_pass()
    bad = sum(describe(k, ref[k], cur[k]) for k in ref)
    print(f"train run {r}: {bad} differing elements")
    total += bad
m.eval()
with torch.no_grad():
    a0 = m(y, c).clone()
    for r in range(1, R):
        a1 = m(y, c).clone()
        bad = describe("logits(no_grad)", a0, a1)
        print(f"no-grad run {r}: {bad} differing elements; vs train-mode logits: {int((a1 != ref['logits']).sum())}")
        total += bad
print("TOTAL differing:", total)
