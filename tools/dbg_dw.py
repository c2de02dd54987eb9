import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import rgb_no_more_amd as rg
from rgb_no_more_amd import lib as L
from test_chain_fwd import build
from test_chain_bwd import step
for depth in (3, 6, 12):
    m, y, c, tgt = build(depth, 256)
    gc = step(m, y, c, tgt, True); gp = step(m, y, c, tgt, False)
    worst = max((float(np.abs(gc[n] - gp[n]).max() / (np.abs(gp[n]).max() + 1e-30)), n) for n in gp)
    zeros = [n for n in gp if np.abs(gc[n]).max() == 0]
    print("depth", depth, "worst rel", worst, "all-zero grads:", len(zeros), zeros[:4], flush=True)
