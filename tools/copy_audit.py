#!/usr/bin/env python
"""Where do the small device copies / fills of one train step come from?  torch.profiler over 3 bench-style steps,
grouped by Python stack.  usage (GPU box): python tools/copy_audit.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import rgb_no_more_amd as rg
from rgb_no_more_amd import custom_transforms as CT
import bench

dev = "cuda"
cdt = torch.bfloat16
B = 256
model = rg.ViT(3, 16, 192, depth=12, n_classes=1000, drop_p=0.0, device=dev, num_heads=3, head_size=64,
               pixel_space="DCT", ver=1, use_subblock=True)
model.compute_dtype = cdt
opt = rg.custom_optims.FusedClipAdamWWD(model, lr=1e-3, eps=1e-8, weight_decay=1e-4, max_norm=1.0)
mix = rg.cls_transforms.RandomMixup_DCT(1000, alpha=0.2)
mix.out_dtype = cdt
lab = torch.randint(0, 999, (B,), device=dev)
Yq, Cq, quant = bench.synth_coefficients(B, dev, 1234)
aug = CT.TrainTransform_DCT(out_dtype=cdt)
sampler = CT.FastParamSampler(aug, seed=1234)


def step():
    opt.zero_grad(set_to_none=True)
    packed, nops = sampler.sample(B, 64, 64)
    y, c = CT.apply_packed(aug, Yq, Cq, quant, packed, nops)
    (my, mc), mt = mix((y, c), lab)
    loss = rg.cls_transforms.cross_entropy(model(my, mc), mt, grad_dtype=cdt)
    loss.backward()
    opt.step()


for _ in range(5):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(3):
        step()
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_stack_n=6)
        if any(k in e.key for k in ("copy", "fill", "zero", "clone", "to", "empty", "cat", "mul", "add", "div", "sort", "roll"))
        and e.key.startswith("aten::") and e.count >= 3]
rows.sort(key=lambda e: -e.count)
for e in rows[:40]:
    stack = [s for s in e.stack if "rgb" in s or "bench" in s or "tools" in s][:3]
    print(f"{e.key:28s} x{e.count / 3:5.1f}/step  dev {e.device_time_total / 3:8.1f} us/step  {' <- '.join(s.split('/')[-1] for s in stack)}")
