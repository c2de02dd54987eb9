#!/usr/bin/env python
"""Two identical runs of the bench step must give the same loss bits (the library has no floating-point atomics; the only
work-stealing counter, gemm_nt_wres's tile queue, does not change which arithmetic a tile gets)."""
import json, subprocess, sys
outs = []
for i in range(2):
    r = subprocess.run([sys.executable, "bench.py", "--steps", "150", "--warmup", "5", "--no-cpu-baseline", "--no-trace",
                        "--prewarm-sec", "0"] + sys.argv[1:], capture_output=True, text=True)
    d = json.loads(r.stdout.strip().splitlines()[-1])
    outs.append(d["config"]["loss"])
print("losses", outs, "identical" if outs[0] == outs[1] else "DIFFERENT")
