#!/usr/bin/env python
"""bench.py -- images/sec of the JPEG-Ti DCT train step on N MI355X (one process per GPU, RCCL).

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N>1 launched under torch.distributed.run.
One "step" = one pass of the hot path over one per-GPU batch of synthetic input resident in HBM:
  mixup -> ViT forward (HIP) -> soft-label cross entropy -> backward (HIP) [-> DDP all-reduce over RCCL]
  -> global-norm clip + AdamW + WeightDecay (HIP), i.e. the reference's `Model F/B pass` benchmark
  (benchmark.py:125-197) with the train-loop optimizer tail (train.py:153-176).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_IMG = {"vitti": 7.4025e9, "vits": 27.2813e9}   # train step = 3x forward
ARCH = {"vitti": (192, 3), "vits": (384, 6)}
MFMA_PEAK_BF16 = 2500.0   # TFLOP/s dense, MI355X_MICROARCH.md
MFMA_PEAK_F32 = 157.3


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch (BASELINE config 2: 256)")
    ap.add_argument("--arch", default="vitti", choices=list(ARCH))
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-images", type=int, default=96)
    ap.add_argument("--trace", action="store_true", help="per-kernel HIP-event timing of the timed region")
    ap.add_argument("--opt", action="append", default=[], help="library option name=value (A/B experiments)")
    return ap.parse_args()


def cpu_baseline(arch, n_images, batch=8):
    """The oracle (CPU port of the reference path, oracle/vit_torch.py) timed on this host: config-1 style
    train step (fwd + soft-label CE + bwd + clip + AdamW + WeightDecay), fp32, batch 8."""
    import torch
    from oracle import vit_torch as V
    from rgb_no_more_amd import detfill
    emb, heads = ARCH[arch]
    depth = 12
    shapes = V.param_shapes(depth, emb, heads)
    p = {k: torch.from_numpy(v).requires_grad_(True) for k, v in detfill.fill_state_dict(shapes, 1).items()}
    params = list(p.values())
    opt = torch.optim.AdamW(params, lr=1e-3, weight_decay=0, eps=1e-8)
    wd = [v for k, v in p.items() if (".weight" in k) and ("lrnorm" not in k)]
    y = torch.randn(batch, 1, 28, 28, 8, 8)
    c = torch.randn(batch, 2, 14, 14, 8, 8)
    oh = torch.nn.functional.one_hot(torch.randint(0, 999, (batch,)), 1000).float()

    def step():
        opt.zero_grad()
        my, mc, mt = V.mixup(y, c, oh, 0.8, 0.2)
        loss = V.soft_xent(V.vit_forward(p, my, mc, depth, heads, emb), mt)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(params, 1.0)
        opt.step()
        with torch.no_grad():
            torch._foreach_mul_(wd, 1.0 - 1e-4)

    step()
    nsteps = max(1, n_images // batch)
    t0 = time.perf_counter()
    for _ in range(nsteps):
        step()
    dt = time.perf_counter() - t0
    return {"value": round(nsteps * batch / dt, 2), "unit": "images/sec", "cores": torch.get_num_threads(),
            "kind": "port", "sample": f"{nsteps} steps x batch {batch}, fp32 torch-CPU oracle of the same train step "
                                      f"(model part of BASELINE config 1), {arch}"}


def main():
    a = parse()
    import torch
    import torch.distributed as dist
    import rgb_no_more_amd as rg
    from rgb_no_more_amd import lib as L

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("for --gpus N>1 launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
    torch.cuda.set_device(local)
    dev = f"cuda:{local}"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)   # "nccl" is RCCL on ROCm

    for o in a.opt:
        k, v = o.split("=")
        L.check(L.lib().rgbnm_set_option(k.encode(), int(v)), f"option {k}")
    emb, heads = ARCH[a.arch]
    cdt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    torch.manual_seed(1234 + rank)
    model = rg.ViT(3, 16, emb, depth=12, n_classes=1000, drop_p=0.0, device=dev, num_heads=heads, head_size=64,
                   pixel_space="DCT", ver=1, use_subblock=True)
    model.compute_dtype = cdt
    net = model
    if world > 1:
        from torch.nn.parallel import DistributedDataParallel as DDP
        net = DDP(model, device_ids=[local], output_device=local, bucket_cap_mb=8, gradient_as_bucket_view=False)
    opt = rg.custom_optims.FusedClipAdamWWD(model, lr=1e-3, eps=1e-8, weight_decay=1e-4, max_norm=1.0)
    mix = rg.cls_transforms.RandomMixup_DCT(1000, alpha=0.2)
    mix.out_dtype = cdt
    B = a.batch
    # S-randn synthetic input of the reference benchmark (benchmark.py:146-148), resident in HBM
    y = torch.randn(B, 1, 28, 28, 8, 8, device=dev)
    c = torch.randn(B, 2, 14, 14, 8, 8, device=dev)
    lab = torch.randint(0, 999, (B,), device=dev)

    def step():
        opt.zero_grad(set_to_none=True)
        (my, mc), mt = mix((y, c), lab)
        logits = net(my, mc)
        loss = rg.cls_transforms.cross_entropy(logits, mt, grad_dtype=cdt)
        loss.backward()
        opt.step()
        return loss

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    ms = dt / a.steps * 1e3
    value = world * B * a.steps / dt
    out = None
    if rank == 0:
        peak = MFMA_PEAK_BF16 if a.dtype == "bf16" else MFMA_PEAK_F32
        tfl = value / world * FLOP_PER_IMG[a.arch] / 1e12
        out = {
            "metric": "images/sec JPEG-Ti DCT 512x512 train step" if a.arch == "vitti" else f"images/sec {a.arch} DCT train step",
            "value": round(value, 1), "unit": "images/sec", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": f"JPEG-{'Ti' if a.arch == 'vitti' else 'S'} --domain DCT {a.dtype}, HIP ViT fwd/bwd + mixup + "
                                   f"soft-CE + clip/AdamW/WD, per-GPU batch {B} (BASELINE config 2), S-randn inputs in HBM",
                       "per_gpu_batch": B, "global_batch": B * world, "parallelism": f"dp{world}",
                       "loss": round(float(loss.item()), 5)},
            "roofline": {"bound": "mfma", "achieved": round(tfl, 2), "peak": peak, "unit": "TFLOP/s",
                         "frac": round(tfl / peak, 4), "traffic": None,
                         "note": "whole train step: 3x forward FLOPs/img (SURVEY 8d) / step time, per GPU"},
        }
        if not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.arch, a.cpu_baseline_images)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
