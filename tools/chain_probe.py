"""Time the one-launch encoder forward against the per-operation forward (B = 256, depth 12, bf16) and print both."""
import sys
import time
import numpy as np
import torch
import os
import sys as _s
_s.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rgb_no_more_amd as rg
from rgb_no_more_amd import detfill, lib as L

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
m = rg.ViT(3, 16, 192, depth=12, n_classes=1000, drop_p=0.0, device="cuda", num_heads=3, head_size=64, pixel_space="DCT", ver=1)
shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
m.load_state_dict({k: torch.from_numpy(v) for k, v in detfill.fill_state_dict(shapes, 1).items()})
m.compute_dtype = torch.bfloat16
y = torch.from_numpy(detfill.normalish((B, 1, 28, 28, 8, 8), 71)).cuda()
c = torch.from_numpy(detfill.normalish((B, 2, 14, 14, 8, 8), 72)).cuda()
tgt = torch.from_numpy(detfill.integers((B,), 74, 0, 998, np.int64)).cuda()
out = {}
for chain in (1, 0, 1, 0):
    L.lib().rgbnm_set_option(b"fwd_chain", chain)
    for mode in ("fwd", "fwdbwd"):
        def one():
            m.zero_grad(set_to_none=False)
            lo = m(y, c)
            if mode == "fwdbwd":
                rg.cls_transforms.cross_entropy(lo, tgt, grad_dtype=torch.bfloat16).backward()
            return lo
        for _ in range(5):
            lo = one()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            lo = one()
        e1.record()
        torch.cuda.synchronize()
        print(f"chain={chain} {mode}: {e0.elapsed_time(e1) / 20:.3f} ms/iter", flush=True)
        out[(chain, mode)] = lo.detach().float().cpu().numpy()
print("max |dlogit| chain vs per-op:", np.abs(out[(1, "fwd")] - out[(0, "fwd")]).max())
