#!/usr/bin/env python
"""Timeline of mlp_fwd_kernel from in-kernel stamps (build with RGBNM_HIPCC_FLAGS=-DMLP_TRACE; experiments only)."""
import ctypes
import importlib
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
rg = importlib.import_module("rgb-no-more_amd")
lib = rg.lib.lib()
import bench  # noqa: E402

if __name__ == "__main__":
    sys.argv = ["bench.py", "--steps", "3", "--warmup", "2", "--no-cpu-baseline", "--no-parity-check"]
    bench.main()
    torch.cuda.synchronize()
    buf = np.zeros(16 * 8 * 80, dtype=np.uint64)
    lib.rgbnm_mlp_trace_read.argtypes = [ctypes.c_void_p]
    lib.rgbnm_mlp_trace_read.restype = ctypes.c_int
    assert lib.rgbnm_mlp_trace_read(buf.ctypes.data) == 0
    t = buf.reshape(16, 8, 80).astype(np.int64)
    for wg in (0, 5):
        base = t[wg, :, 0].min()
        real = (t[wg, 0, 78] - t[wg, 0, 79]) * 10.0          # s_memrealtime: 100 MHz
        ticks = t[wg, 0, 63] - t[wg, 0, 0]
        print(f"wg {wg}: wave 0 lifetime {ticks} ticks = {real:.0f} ns  ({ticks / max(real, 1):.3f} ticks/ns)")
        for w in (0, 3, 4, 6, 7):
            r = t[wg, w] - base
            print(f"  wave {w}: start {r[0]}  X-loaded {r[1]}  loop-end {r[62]}  end {r[63] if w < 7 else 0}   epilogue stamps 64.. {[int(v) for v in (r[64:69] - r[62])]}")
            rows = []
            for c in range(12):
                s = r[2 + 5 * c:7 + 5 * c]
                if w < 7:
                    rows.append(f"c{c}: wait {s[1]-s[0]:5d} h0 {s[2]-s[1]:5d} h1 {s[3]-s[2]:5d} st {s[4]-s[3]:5d}")
                else:
                    rows.append(f"c{c}: land@{s[0]:6d} wait {s[1]-s[0]:5d} issue {s[2]-s[1]:5d}")
            print("    " + "\n    ".join(rows))
