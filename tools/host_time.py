#!/usr/bin/env python
"""How long the HOST needs to enqueue one train step (bench.py's step), against how long the GPU needs to run it: if the two
are close, a slow or shared host makes the step launch-bound."""
import sys, time, json, subprocess
sys.argv = ["bench.py", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-parity-check", "--no-trace", "--prewarm-sec", "0.5"] + sys.argv[1:]
import torch
# re-use bench.main's construction by running it once, then grab its step closure through a hook
orig_sync = torch.cuda.synchronize
steps = {}
def main():
    import types
    src = open("bench.py").read()
    # expose `step` of main(): patch the timed loop to stash the closure
    assert "    t_pre = time.perf_counter()\n" in src
    src = src.replace("    t_pre = time.perf_counter()\n", "    globals()['_STEP'] = step\n    t_pre = time.perf_counter()\n", 1)
    g = {"__name__": "bench_patched", "__file__": "bench.py"}
    exec(compile(src, "bench.py", "exec"), g)
    g["main"]()
    return g["_STEP"]
import os
step_ = main()
if os.environ.get("HOST_TIME_TRACED"):        # the event-bracketed eager steps bench.py runs every 8th step
    import ctypes
    from rgb_no_more_amd import lib as _L
    _L.lib().rgbnm_set_option(b"trace", (1 << 6) | (1 << 5) | (1 << 2) | (1 << 1))
    step = lambda: step_(eager=True)
else:
    step = step_
for _ in range(20):
    step()
torch.cuda.synchronize()
N = 100
t0 = time.perf_counter()
for _ in range(N):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(json.dumps({"host_enqueue_ms_per_step": round((t1 - t0) / N * 1e3, 3), "wall_ms_per_step": round((t2 - t0) / N * 1e3, 3)}))
import cProfile, pstats, io
pr = cProfile.Profile()
pr.enable()
for _ in range(50):
    step()
pr.disable()
torch.cuda.synchronize()
st = io.StringIO()
pstats.Stats(pr, stream=st).sort_stats("tottime").print_stats(30)
print("\n".join(l[:150] for l in st.getvalue().splitlines()[:70]))
