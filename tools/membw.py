"""Calibrate achievable HBM bandwidth with plain torch ops (fill = write only, sum = read only, copy = 1R + 1W)."""
import time, torch
def bench(f, nbytes, name):
    for _ in range(5): f()
    torch.cuda.synchronize()
    t_end = time.time() + 1.5
    while time.time() < t_end:
        for _ in range(20): f()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / 50
    print(f"{name:28s} {nbytes/1e6:8.1f} MB  {us:8.1f} us  {nbytes/us/1e6:6.2f} TB/s")
for mb in (77, 154, 1024):
    n = mb * 1000 * 1000 // 2
    a = torch.empty(n, device="cuda", dtype=torch.bfloat16); b = torch.empty_like(a)
    bench(lambda: a.fill_(1.0), n * 2, f"fill {mb}MB (W)")
    bench(lambda: b.copy_(a), n * 4, f"copy {mb}MB (R+W)")
    bench(lambda: a.float().sum() if False else torch.sum(a, dtype=torch.float32), n * 2, f"sum {mb}MB (R)")
