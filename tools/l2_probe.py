#!/usr/bin/env python
"""Per-CU throughput of L2-resident re-reads (csrc/calib.hip calib_l2_kernel): plain loads vs LDS-DMA, by waves per CU."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rgb_no_more_amd import lib as L
lib = L.lib()
buf = torch.empty(256 * 131072, device="cuda", dtype=torch.uint8).random_(0, 255)
sink = torch.zeros(4, device="cuda", dtype=torch.uint8)
for mode, name in ((0, "plain global_load_dwordx4"), (1, "LDS-DMA global_load_lds_dwordx4")):
    for waves in (1, 2, 4, 8, 16):
        slice_bytes, iters = 131072 if waves != 16 else 131072, 50
        f = lambda: L.check(lib.rgbnm_calib_l2(buf.data_ptr(), slice_bytes, iters, mode, 256, waves, sink.data_ptr(), L.stream()))
        f(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5): f()
        b.record(); torch.cuda.synchronize()
        us = a.elapsed_time(b) / 5 * 1e3
        tot = 256 * slice_bytes * iters
        print(f"{name:34s} waves/CU {waves:2d}: {tot / us / 1e6:6.2f} TB/s aggregate = {tot / us / 1e3 / 256:6.1f} GB/s per CU")
