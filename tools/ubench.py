#!/usr/bin/env python
"""Launch each hot kernel of one encoder block a few times at BASELINE config-2 shapes (B=256, JPEG-Ti, bf16).
Meant to run under rocprofv3 (--kernel-trace [--pmc ...]); timings come from the profiler, not from here."""
import ctypes as C
import math
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rgb_no_more_amd as rg
from rgb_no_more_amd import lib as L

DEV = "cuda"


def main(reps=3, B=256, emb=192, heads=3):
    torch.manual_seed(0)
    m = rg.ViT(3, 16, emb, depth=1, n_classes=1000, drop_p=0.0, device=DEV, num_heads=heads, head_size=64,
               pixel_space="DCT", ver=1)
    m.compute_dtype = torch.bfloat16
    y = torch.randn(B, 1, 28, 28, 8, 8, device=DEV)
    c = torch.randn(B, 2, 14, 14, 8, 8, device=DEV)
    lab = torch.randint(0, 999, (B,), device=DEV)
    for _ in range(reps):
        m.zero_grad(set_to_none=True)
        loss = rg.cls_transforms.cross_entropy(m(y, c), lab, grad_dtype=torch.bfloat16)
        loss.backward()
    torch.cuda.synchronize()
    print("ubench done")


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 3)
