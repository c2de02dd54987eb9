#!/usr/bin/env python
"""What the two waves of a SIMD share (csrc/calib.hip calib_pipes_kernel): time per loop round for MFMA / VALU instruction kinds
alone, paired with the same kind on the SIMD's other wave, paired with the other pipe, and both in one instruction stream.
Only RATIOS between the lines mean anything (the counter is not the shader clock).  Findings recorded in DESIGN.md section 4."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rgb_no_more_amd import lib as L
lib = L.lib()
ITERS = 400
NAMES = {0: "idle", 1: "8 MFMA", 2: "64 v_fma", 3: "32 v_pk_fma", 4: "16 exp + 16 rcp", 5: "32 cvt_pk_bf16", 6: "8 MFMA + 64 v_fma (one stream)", 7: "8 GELU pairs", 8: "8 ds_read+MFMA"}
PER = {1: 8, 2: 64, 3: 32, 4: 32, 5: 32, 6: 8, 7: 8, 8: 8}
sink = torch.zeros(4, device="cuda")


def run(roles):
    r = torch.tensor(roles + [0] * (8 - len(roles)), dtype=torch.int32, device="cuda")
    out = torch.zeros(256 * 8, dtype=torch.int64, device="cuda")
    for _ in range(2):
        L.check(lib.rgbnm_calib_pipes(r.data_ptr(), 8, ITERS, 256, out.data_ptr(), sink.data_ptr(), L.stream()))
    torch.cuda.synchronize()
    return out.view(256, 8).double().mean(0).cpu().numpy()


print("wave w and wave w + 4 share a SIMD; cycles per loop round (mean over 256 workgroups) and per instruction")
for title, roles in [
        ("MFMA alone (wave 0)", [1, 0, 0, 0, 0, 0, 0, 0]),
        ("MFMA on both waves of SIMD 0", [1, 0, 0, 0, 1, 0, 0, 0]),
        ("v_fma alone", [2, 0, 0, 0, 0, 0, 0, 0]),
        ("v_fma on both waves of SIMD 0", [2, 0, 0, 0, 2, 0, 0, 0]),
        ("MFMA (w0) beside v_fma (w4), same SIMD", [1, 0, 0, 0, 2, 0, 0, 0]),
        ("MFMA (w0) beside v_fma (w1), different SIMDs", [1, 2, 0, 0, 0, 0, 0, 0]),
        ("MFMA + v_fma in one stream (w0)", [6, 0, 0, 0, 0, 0, 0, 0]),
        ("v_pk_fma alone", [3, 0, 0, 0, 0, 0, 0, 0]),
        ("exp + rcp alone", [4, 0, 0, 0, 0, 0, 0, 0]),
        ("cvt_pk_bf16 alone", [5, 0, 0, 0, 0, 0, 0, 0]),
        ("MFMA (w0) beside exp+rcp (w4)", [1, 0, 0, 0, 4, 0, 0, 0]),
        ("MFMA (w0) beside v_pk_fma (w4)", [1, 0, 0, 0, 3, 0, 0, 0]),
        ("MFMA (w0) beside cvt_pk (w4)", [1, 0, 0, 0, 5, 0, 0, 0]),
        ("GELU pairs alone", [7, 0, 0, 0, 0, 0, 0, 0]),
        ("GELU pairs on both waves of SIMD 0", [7, 0, 0, 0, 7, 0, 0, 0]),
        ("MFMA (w0) beside GELU pairs (w4)", [1, 0, 0, 0, 7, 0, 0, 0]),
        ("GELU pairs (w0) beside MFMA (w4)", [7, 0, 0, 0, 1, 0, 0, 0]),
        ("ds_read+MFMA alone", [8, 0, 0, 0, 0, 0, 0, 0]),
        ("ds_read+MFMA (w0) beside GELU pairs (w4)", [8, 0, 0, 0, 7, 0, 0, 0]),
        ("3 x (ds_read+MFMA | GELU) + lone GELU on SIMD 3", [8, 8, 8, 7, 7, 7, 7, 0]),
        ("all 8 waves MFMA", [1] * 8),
        ("all 8 waves v_fma", [2] * 8),
        ("4 waves MFMA + 4 waves v_fma (paired per SIMD)", [1, 1, 1, 1, 2, 2, 2, 2])]:
    c = run(roles) / ITERS
    desc = "  ".join(f"w{w}[{NAMES[r]}]: {c[w]:7.1f} cyc/round = {c[w] / PER[r]:5.2f} per instr" for w, r in enumerate(roles) if r)
    print(f"{title:50s} {desc}")
