#!/usr/bin/env python
"""BASELINE config 5 (SwinV2-T --domain DCT, window 8, bf16): model-only train step on S-randn inputs
(Y (B,1,32,32,8,8), CbCr (B,2,16,16,8,8); reference benchmark.py:146-148 semantics), torch AdamW on the HIP gradients.
First-generation kernels (generic GEMM tiles, VALU window attention): a functional figure, not a tuned one."""
import argparse
import json
import sys
import os
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rgb_no_more_amd as rg  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--dtype", default="bf16")
    a = ap.parse_args()
    dev = "cuda:0"
    torch.manual_seed(0)
    m = rg.SwinTransformerV2(img_size=256, patch_size=4, embed_dim=96, depths=[2, 2, 6, 2], num_heads=[3, 6, 12, 24],
                             window_size=8, drop_path_rate=0.2, device=dev, pixel_space="dct")
    m.compute_dtype = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    m.train()
    opt = torch.optim.AdamW(m.parameters(), lr=1e-3, weight_decay=0.05)
    B = a.batch
    y = torch.randn(B, 1, 32, 32, 8, 8, device=dev)
    c = torch.randn(B, 2, 16, 16, 8, 8, device=dev)
    lab = torch.randint(0, 999, (B,), device=dev)

    def step():
        opt.zero_grad(set_to_none=True)
        loss = rg.cls_transforms.cross_entropy(m(y, c), lab, grad_dtype=m.compute_dtype)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)
        opt.step()
        return loss

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"metric": "images/sec SwinV2-T DCT 256x256 train step (model only)", "value": round(B * a.steps / dt, 1),
                      "unit": "images/sec", "n_gpus": 1, "ms_per_step": round(1e3 * dt / a.steps, 2), "per_gpu_batch": B,
                      "dtype": a.dtype, "loss": round(float(loss.item()), 4),
                      "params": sum(p.numel() for p in m.parameters())}), flush=True)


if __name__ == "__main__":
    main()
