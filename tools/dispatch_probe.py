#!/usr/bin/env python
"""GPU-side dispatch interval: a HIP graph of N dependent tiny kernels replayed, timed with events -> microseconds per dispatch.
The AQL packets and kernel arguments of every dispatch are read from HOST memory; on a loaded host this interval grows and a step
of ~45 launches pays it ~45 times (bench.py's ms_per_step moves with it while every kernel's own time stays put)."""
import json
import sys
import torch

N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
x = torch.zeros(64, device="cuda")
s = torch.cuda.Stream()
torch.cuda.set_stream(s)
for _ in range(3):
    x.add_(1.0)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    for _ in range(N):
        x.add_(1.0)
res = []
for rep in range(7):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    res.append(e0.elapsed_time(e1) * 1e3 / N)
# eager launches of the same kernel, host far ahead is impossible here (python is slower than the GPU): reported for reference
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(N):
    x.add_(1.0)
e1.record()
torch.cuda.synchronize()
print(json.dumps({"graph_us_per_dispatch": [round(r, 2) for r in res], "eager_us_per_launch": round(e0.elapsed_time(e1) * 1e3 / N, 2),
                  "loadavg": open("/proc/loadavg").read().split()[:3]}))
