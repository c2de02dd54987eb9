#!/usr/bin/env python
"""HIP-graph replay of a SwinV2-T train step (eval mode) against the eager pass, per parameter; brackets on."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rgb_no_more_amd as rg
from rgb_no_more_amd import detfill
dev = "cuda"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
hold = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ws = torch.cuda.Stream(); torch.cuda.set_stream(ws)
if len(sys.argv) > 3:        # like bench.py: another model instance runs a step first and is dropped
    m0 = rg.SwinTransformerV2(img_size=256, patch_size=4, embed_dim=96, depths=[2, 2, 6, 2], num_heads=[3, 6, 12, 24], window_size=8,
                              drop_path_rate=0.0, device=dev, pixel_space="dct")
    m0.compute_dtype = torch.bfloat16
    y0 = torch.randn(B, 1, 32, 32, 8, 8, device=dev); c0 = torch.randn(B, 2, 16, 16, 8, 8, device=dev)
    m0(y0, c0).float().square().mean().backward()
    del m0, y0, c0
    torch.cuda.empty_cache()
    import gc; gc.collect(); gc.freeze()
m = rg.SwinTransformerV2(img_size=256, patch_size=4, embed_dim=96, depths=[2, 2, 6, 2], num_heads=[3, 6, 12, 24], window_size=8,
                         drop_path_rate=0.0, device=dev, pixel_space="dct")
m.eval(); m.compute_dtype = torch.bfloat16
m.group_dw_backward, m.hold_reductions = True, bool(hold)
y = torch.from_numpy(detfill.normalish((B, 1, 32, 32, 8, 8), 171)).to(dev).bfloat16()
c = torch.from_numpy(detfill.normalish((B, 2, 16, 16, 8, 8), 172)).to(dev).bfloat16()
t = detfill.uniform((B, 1000), 173, 0.0, 1.0); t = torch.from_numpy(t / t.sum(1, keepdims=True)).to(dev)
names = [n for n, _ in m.named_parameters()]
def part():
    loss = rg.cls_transforms.cross_entropy(m(y, c), t, grad_dtype=torch.bfloat16)
    loss.backward()
    return loss
m.zero_grad(set_to_none=True); part(); torch.cuda.synchronize()
ref = [p.grad.clone() for p in m.parameters()]
for _ in range(2):
    m.zero_grad(set_to_none=True); part()
m.zero_grad(set_to_none=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=ws):
    gl = part()
m2 = None
for r in range(4):
    g.replay(); torch.cuda.synchronize()
    if r == 0:
        rep = [p.grad.clone() for p in m.parameters()]
    bad = [(n, (a - p.grad).abs().max().item()) for n, a, p in zip(names, ref, m.parameters()) if not torch.equal(a, p.grad)]
    print(f"replay {r}: {len(bad)} parameters differ", sorted(bad, key=lambda kv: -kv[1])[:8], flush=True)

m.zero_grad(set_to_none=True); part(); torch.cuda.synchronize()
bad = [(n, (a - p.grad).abs().max().item()) for n, a, p in zip(names, ref, m.parameters()) if not torch.equal(a, p.grad)]
print("eager again vs first eager:", len(bad), sorted(bad, key=lambda kv: -kv[1])[:6])
bad = [(n, (a - p.grad).abs().max().item()) for n, a, p in zip(names, rep, m.parameters()) if not torch.equal(a, p.grad)]
print("eager again vs replay:", len(bad), sorted(bad, key=lambda kv: -kv[1])[:6])
