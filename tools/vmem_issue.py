#!/usr/bin/env python
"""How long does one wave take to ISSUE 16 back-to-back 1 KB global stores / loads (8 rows x 128 B each)?"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rgb_no_more_amd import lib as L

lib = L.lib()
st = torch.cuda.current_stream().cuda_stream
for waves in (1, 7):
    for wgs in (8, 256):
        for ld in (128, 1152):
            wave_bytes = 128 * ld
            buf = torch.zeros(wgs * waves * wave_bytes, dtype=torch.uint8, device="cuda")
            out = torch.zeros(wgs * waves * 2, dtype=torch.int64, device="cuda")
            for mode, name in ((0, "store"), (1, "load")):
                for _ in range(3):
                    L.check(lib.rgbnm_calib_vmem_issue(mode, wgs, waves, buf.data_ptr(), wave_bytes, ld, out.data_ptr(), st))
                torch.cuda.synchronize()
                o = out.view(-1, 2).double()
                print(f"{name:5s} waves/WG={waves} WGs={wgs:3d} ld={ld:4d}: issue 16 instr = {o[:, 0].mean():7.0f} cycles "
                      f"({o[:, 0].mean() / 16:5.0f}/instr), all done after {o[:, 1].mean():7.0f}")
