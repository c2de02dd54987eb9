#!/usr/bin/env python
"""Per-mode cost of the augment stage: time TrainTransform_DCT on batches whose crop sides are all /2, all identity, all x2
or the sampler's mix, with 0 or 2 ops (rocprofv3 --kernel-trace --stats around this script gives the per-kernel split)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from rgb_no_more_amd import custom_transforms as CT

dev = "cuda"
B = 256
Yq, Cq, quant = bench.synth_coefficients(B, dev, 1)
aug = CT.TrainTransform_DCT(size=28, out_dtype=torch.bfloat16)
sampler = CT.FastParamSampler(aug, seed=3)
packed, nops = sampler.sample(B, 64, 64)
res = {}
for name, side in (("half", 56), ("identity", 28), ("double", 14), ("mix", 0)):
    for ops in (0, 2):
        pk = packed.copy()
        if side:
            pk["crop"][:, 2] = side
            pk["crop"][:, 3] = side
            pk["crop"][:, 0] = 4
            pk["crop"][:, 1] = 2
        if ops == 0:
            pk["op"][:] = 0
        for _ in range(5):
            CT.apply_packed(aug, Yq, Cq, quant, pk, ops)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 50
        e0.record()
        for _ in range(n):
            CT.apply_packed(aug, Yq, Cq, quant, pk, ops)
        e1.record()
        torch.cuda.synchronize()
        res[f"{name}_ops{ops}"] = round(e0.elapsed_time(e1) / n * 1e3, 1)
print(json.dumps(res))
