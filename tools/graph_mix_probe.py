#!/usr/bin/env python
"""What does an eager step cost in between HIP-graph replays of the bench step (with and without the HIP-event trace)?"""
import sys, time, json
sys.argv = ["bench.py", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-parity-check", "--no-trace", "--prewarm-sec", "0.5"]
import torch
src = open("bench.py").read()
src = src.replace("    t_pre = time.perf_counter()\n    while time.perf_counter() - t_pre < a.prewarm_sec:", "    globals()['_STEP'] = step\n    globals()['_LIB'] = lib\n    t_pre = time.perf_counter()\n    while time.perf_counter() - t_pre < a.prewarm_sec:", 1)
g = {"__name__": "bench_patched", "__file__": "bench.py"}
exec(compile(src, "bench.py", "exec"), g)
g["main"]()
step, lib = g["_STEP"], g["_LIB"]

def run(pattern, n=96):
    for _ in range(16):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        mode = pattern(i)
        if mode == 2:
            lib.rgbnm_set_option(b"trace", 2)
        step(eager=mode > 0)
        if mode == 2:
            lib.rgbnm_set_option(b"trace", 0)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

print("replay only            ", round(run(lambda i: 0), 4))
print("eager only             ", round(run(lambda i: 1), 4))
print("1 eager in 8           ", round(run(lambda i: 1 if i % 8 == 0 else 0), 4))
print("1 traced eager in 8    ", round(run(lambda i: 2 if i % 8 == 0 else 0), 4))
print("eager, 1 traced in 8   ", round(run(lambda i: 2 if i % 8 == 0 else 1), 4))
