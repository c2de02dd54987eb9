#!/usr/bin/env python
"""Does one JPEG-Ti forward+backward (no augment, no optimizer) capture into a HIP graph, and what does replay save?"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rgb_no_more_amd as rg

dev, cdt, B = "cuda", torch.bfloat16, 256
model = rg.ViT(3, 16, 192, depth=12, n_classes=1000, drop_p=0.0, device=dev, num_heads=3, head_size=64,
               pixel_space="DCT", ver=1, use_subblock=True)
model.compute_dtype = cdt
y = torch.randn(B, 1, 28, 28, 8, 8, device=dev).to(cdt)
c = torch.randn(B, 2, 14, 14, 8, 8, device=dev).to(cdt)
lab = torch.randint(0, 999, (B,), device=dev)


def fb():
    for p in model.parameters():
        p.grad = None
    loss = rg.cls_transforms.cross_entropy(model(y, c), lab, grad_dtype=cdt)
    loss.backward()
    return loss


def timeit(fn, n=50):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


t_end = time.time() + 3
while time.time() < t_end:
    fb()
torch.cuda.synchronize()
print("eager ms:", timeit(fb))
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        fb()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    loss = fb()
torch.cuda.synchronize()
g.replay()
torch.cuda.synchronize()
print("loss after replay", loss.item())
print("graph ms:", timeit(g.replay))
print("eager ms again:", timeit(fb))
