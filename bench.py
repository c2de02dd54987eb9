#!/usr/bin/env python
"""bench.py -- images/sec of the JPEG-Ti DCT train step on N MI355X (one process per GPU, RCCL over xGMI).

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N>1 launched under torch.distributed.run.
One "step" = one pass of the hot path over one per-GPU batch of synthetic input already resident in HBM
(BASELINE.json config 2: JPEG-Ti --domain DCT bf16, batch 256 per GPU):

  raw int16 DCT coefficients of 512x512 4:2:0 images (S-coef)
    -> HIP dequant/crop/resize/flip/RandAugment/ToRange   (reference: datasets.py:286-293,354-361 on CPU workers)
    -> HIP mixup                                           (cls_transforms.py:163-176)
    -> HIP ViT forward, soft-label CE, backward            (models/plainvit.py; train.py:153-170)
    -> [DDP gradient all-reduce over RCCL, N > 1]          (train.py:137)
    -> HIP global-norm clip + AdamW + WeightDecay          (train.py:163-165, custom_optims.py:37-42)

Prints ONE JSON line on rank 0 (metric/value/.../roofline/cpu_baseline).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# train step = 3x forward FLOPs.  SwinV2-T DCT recounted from the architecture: per block 12 C^2 + 128 C MACs per
# token (qkv, proj, MLP, 64-key window attention), patch merging 8 C^2 per merged token, embed 24 x 96, head: 5.9116 GMAC forward
FLOP_PER_IMG = {"vitti": 7.4025e9, "vits": 27.2813e9, "swinv2t": 35.47e9}
ARCH = {"vitti": (192, 3), "vits": (384, 6), "swinv2t": (96, 3)}
NAMES = {"vitti": "JPEG-Ti", "vits": "JPEG-S", "swinv2t": "SwinV2-T"}
CONFIG_OF = {"vitti": "BASELINE config 2", "vits": "BASELINE config 4 model", "swinv2t": "BASELINE config 5 model"}
MFMA_PEAK = {"bf16": 2500.0, "fp32": 157.3}             # dense TFLOP/s, MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0
Q90_LUMA = [3, 2, 2, 3, 5, 8, 10, 12, 2, 2, 3, 4, 5, 12, 12, 11, 3, 3, 3, 5, 8, 11, 14, 11, 3, 3, 4, 6, 10, 17, 16, 12, 4, 4,
            7, 11, 14, 22, 21, 15, 5, 7, 11, 13, 16, 21, 23, 18, 10, 13, 16, 17, 21, 24, 24, 20, 14, 18, 19, 20, 22, 20, 21, 20]
Q90_CHROMA = [3, 4, 5, 9, 20, 20, 20, 20, 4, 4, 5, 13, 20, 20, 20, 20, 5, 5, 11, 20, 20, 20, 20, 20, 9, 13, 20, 20, 20, 20,
              20, 20] + [20] * 32


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--prewarm-sec", type=float, default=2.0,
                    help="untimed steps run for this long before the W warm-up steps so the GPU leaves its idle clocks")
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch (BASELINE configs 2-5: 256)")
    ap.add_argument("--arch", default="vitti", choices=list(ARCH))
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-augment", action="store_true", help="model-only step on S-randn inputs (benchmark.py:146-148)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity-check", action="store_true", help="skip the golden-vector check of the timed kernel mix")
    ap.add_argument("--cpu-baseline-images", type=int, default=512)     # ~10 s of CPU work at JPEG-Ti (17.7 ms / image on 16 threads)
    ap.add_argument("--no-trace", action="store_true", help="skip the per-kernel HIP-event trace of the timed region")
    ap.add_argument("--opt", action="append", default=[], help="library option name=value (A/B experiments)")
    ap.add_argument("--rccl-channels", type=int, default=0,
                    help="N > 1: cap RCCL at this many channels (NCCL_MAX_NCHANNELS = NCCL_MIN_NCHANNELS, set before the process group "
                         "is created -- RCCL reads them once per process, so this is a launch-time knob, not a calibrated candidate): "
                         "every channel is a workgroup that holds a CU while the 256-workgroup compute kernels of this library want "
                         "all of them; 0 = RCCL's default")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel of every step from the host (no HIP graph replay)")
    ap.add_argument("--no-defer-reduce", action="store_true",
                    help="run every block's gradient reductions inside its backward instead of one launch per step")
    ap.add_argument("--grad-sync", default="flat", choices=["flat", "ddp"],
                    help="N > 1: zero-copy slice-wise all-reduce of the flat gradient buffer (self-checked, falls back "
                         "to ddp) or torch DistributedDataParallel")
    return ap.parse_args()


def _flat_sync_selfcheck(model, fsync, cdt, dev, B, dist, S=28):
    """One backward with the overlapped slice-wise exchange vs the same backward followed by ONE blocking all-reduce of
    the whole flat gradient buffer: must agree on every rank (summation order inside RCCL may differ: tolerance).
    (The caller puts a model with DropPath into eval mode: the two passes must draw the same masks.)"""
    import torch
    g = torch.Generator(device=dev)
    g.manual_seed(99 + dist.get_rank())
    y = torch.randn(B, 1, S, S, 8, 8, device=dev, generator=g)
    c = torch.randn(B, 2, S // 2, S // 2, 8, 8, device=dev, generator=g)

    def backward():
        model.zero_grad(set_to_none=True)
        model(y, c).float().square().mean().backward()

    backward()
    fsync.wait()
    base = model.flat_grad_base()
    if base is None:
        return False
    g1 = torch.cat([p.grad.reshape(-1) for p in model.parameters()]).clone()
    model._grad_sync = None
    backward()
    g2 = torch.cat([p.grad.reshape(-1) for p in model.parameters()]).clone()
    dist.all_reduce(g2, op=dist.ReduceOp.SUM)
    g2 /= dist.get_world_size()
    model._grad_sync = fsync
    model.zero_grad(set_to_none=True)
    err = (g1 - g2).abs().max().item()
    ref = g2.abs().max().item()
    return bool(torch.isfinite(g1).all()) and err <= 1e-4 * ref + 1e-8 and fsync.collectives >= 2


def synth_coefficients(B, dev, seed):
    """S-coef: de-quantised DC ~ N(0,300), AC[u,v] ~ Laplace(0, 60/(1+u+v)), clamped to [-1024,1016],
    stored QUANTISED with the libjpeg quality-90 tables, int16, laid out like dct_manip.read_coefficients."""
    import torch
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    u = torch.arange(8, device=dev).view(8, 1).float()
    v = torch.arange(8, device=dev).view(1, 8).float()
    scale = 60.0 / (1 + u + v)
    ql = torch.tensor(Q90_LUMA, device=dev, dtype=torch.float32).view(8, 8)
    qc = torch.tensor(Q90_CHROMA, device=dev, dtype=torch.float32).view(8, 8)

    def plane(shape, q):
        e1 = torch.empty(shape, device=dev).exponential_(1.0, generator=g)
        e2 = torch.empty(shape, device=dev).exponential_(1.0, generator=g)
        x = (e1 - e2) * scale
        x[..., 0, 0] = torch.randn(shape[:-2], device=dev, generator=g) * 300.0
        return (x.clamp(-1024, 1016) / q).round().to(torch.int16).contiguous()

    Y = plane((B, 1, 64, 64, 8, 8), ql)
    Cc = plane((B, 2, 32, 32, 8, 8), qc)
    quant = torch.stack([ql, qc, qc]).to(torch.int16).unsqueeze(0).repeat(B, 1, 1, 1).contiguous()
    return Y, Cc, quant


def cpu_baseline(arch, n_images, batch=8):
    """The oracle (CPU port of the reference path: oracle/dct_np.py + oracle/vit_torch.py) timed on this host on a
    bounded sample of the same workload shape (BASELINE config 1 style: batch 8, fp32):
    per image dequant+crop+resize+flip+2 RandAugment ops+ToRange (numpy, one thread), per batch mixup + JPEG-Ti
    fwd + soft CE + bwd + clip + AdamW + WeightDecay (torch CPU, all intra-op threads)."""
    import numpy as np
    import torch
    from oracle import dct_np as O
    from oracle import vit_torch as V
    from rgb_no_more_amd import detfill
    from rgb_no_more_amd import custom_transforms as CT
    emb, heads = ARCH[arch]
    depth = 12
    ncpu = min(os.cpu_count() or 1, _cgroup_cpu_quota() or (os.cpu_count() or 1))      # honour a cgroup CPU quota (the GPU box: 16 of 256 hardware threads)
    torch.set_num_threads(min(32, ncpu))   # batch-8 fp32 GEMMs stop scaling well before 128 threads
    swin = arch == "swinv2t"
    size = 32 if swin else 28
    if swin:
        from oracle import swin_torch as SW
        sdepths, sheads = [2, 2, 6, 2], [3, 6, 12, 24]
        p = {k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v).clone().requires_grad_(True)
             for k, v in SW.fill_params(SW.param_shapes(sdepths, sheads)).items()}
    else:
        shapes = V.param_shapes(depth, emb, heads)
        p = {k: torch.from_numpy(v).requires_grad_(True) for k, v in detfill.fill_state_dict(shapes, 1).items()}
    params = list(p.values())
    opt = torch.optim.AdamW(params, lr=1e-3, weight_decay=0, eps=1e-8)
    wd = [v for k, v in p.items() if (".weight" in k) and ("lrnorm" not in k)]
    rng = np.random.default_rng(0)
    uu, vv = np.meshgrid(np.arange(8), np.arange(8), indexing="ij")
    ql = np.array(Q90_LUMA, dtype=np.int16).reshape(8, 8)
    qc = np.array(Q90_CHROMA, dtype=np.int16).reshape(8, 8)
    Yq = np.rint(rng.laplace(0, 1, (batch, 1, 64, 64, 8, 8)) * 60.0 / (1 + uu + vv) / ql).astype(np.int16)
    Cq = np.rint(rng.laplace(0, 1, (batch, 2, 32, 32, 8, 8)) * 60.0 / (1 + uu + vv) / qc).astype(np.int16)
    quant = np.stack([ql, qc, qc]).astype(np.int16)
    t = CT.TrainTransform_DCT(size=size)
    torch.manual_seed(0)
    oh = torch.nn.functional.one_hot(torch.randint(0, 999, (batch,)), 1000).float()

    def data(step_params):
        ys, cs = [], []
        for b, sp in enumerate(step_params):
            oy, oc = O.train_transform(Yq[b], Cq[b], quant, sp["box"], sp["flip"], sp["ops"], size=size)
            ys.append(oy)
            cs.append(oc)
        return torch.from_numpy(np.stack(ys)), torch.from_numpy(np.stack(cs))

    def model_step(y, c):
        opt.zero_grad()
        my, mc, mt = V.mixup(y, c, oh, 0.8, 0.2)
        logits = SW.swin_forward(p, my, mc, sdepths, sheads) if swin else V.vit_forward(p, my, mc, depth, heads, emb)
        loss = V.soft_xent(logits, mt)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(params, 1.0)
        opt.step()
        with torch.no_grad():
            torch._foreach_mul_(wd, 1.0 - 1e-4)

    y, c = data(t.sample_params(batch, 64, 64))
    model_step(y, c)   # warm-up
    nsteps = max(1, n_images // batch)
    t_data = t_model = 0.0
    for _ in range(nsteps):
        sp = t.sample_params(batch, 64, 64)
        t0 = time.perf_counter()
        y, c = data(sp)
        t1 = time.perf_counter()
        model_step(y, c)
        t_model += time.perf_counter() - t1
        t_data += t1 - t0
    n = nsteps * batch
    return {"value": round(n / (t_data + t_model), 2), "unit": "images/sec", "cores": torch.get_num_threads(),
            "kind": "port",
            "sample": f"{n} synthetic 512x512 coefficient images, batch {batch}, fp32, {NAMES[arch]}: numpy oracle data path "
                      f"({1e3 * t_data / n:.2f} ms/img, 1 thread) + torch-CPU oracle train step "
                      f"({1e3 * t_model / n:.2f} ms/img, {torch.get_num_threads()} threads); entropy decode excluded "
                      f"(inputs are coefficients, as on the GPU side)"}


BF16_LOGIT_TOL = 1e-2      # bf16 operands / fp32 accumulate vs the fp32 reference (tests/test_fastpath_model.py uses the same bar)


def parity_check(arch, cdt, dev):
    """The exact kernel mix of the timed step against the reference: a second model instance with the deterministic detfill
    weights on detfill inputs at the fast-path batch sizes, compared with golden vectors captured from the reference model
    itself -- EVERY logit, at the batch that is timed (tests/golden/g20_fullsize.npz: JPEG-Ti B = 256 depth 12;
    g21_b256.npz: JPEG-S depth 12 and SwinV2-T at B = 256).  Data only -- nothing of oracle/ is imported here."""
    import numpy as np
    import torch
    import rgb_no_more_amd as rg
    from rgb_no_more_amd import detfill
    # JPEG-Ti: g20 (B = 256 = the timed batch); JPEG-S / SwinV2-T: g21 (make_golden_r4.py), also at the timed batch 256
    path = os.path.join(ROOT, "tests", "golden", "g20_fullsize.npz" if arch == "vitti" else "g21_b256.npz")
    if not os.path.exists(path):
        return None
    g = np.load(path)
    bf16 = cdt == torch.bfloat16
    if arch == "swinv2t":
        tag, B, hard = "swt_b256", 256, False
        depths, sheads = [2, 2, 6, 2], [3, 6, 12, 24]
        m = rg.SwinTransformerV2(img_size=256, patch_size=4, embed_dim=96, depths=depths, num_heads=sheads, window_size=8,
                                 mlp_ratio=4.0, drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.0, qkv_bias=True, ape=False,
                                 patch_norm=True, pretrained_window_sizes=[0] * 4, device=dev, pixel_space="dct")
        # the reference test vectors' parameter fill (a data generator shared with the golden script, not oracle code)
        names = [str(n) for n in g[tag + "_names"]]
        shapes = {n: tuple(p.shape) for n, p in m.named_parameters()}
        m.load_state_dict({k: torch.from_numpy(v) for k, v in detfill.fill_swin_params({n: shapes[n] for n in names}).items()},
                          strict=False)
        y = torch.from_numpy(detfill.normalish((B, 1, 32, 32, 8, 8), 171)).to(dev)
        c = torch.from_numpy(detfill.normalish((B, 2, 16, 16, 8, 8), 172)).to(dev)
        t = detfill.uniform((B, 1000), 173, 0.0, 1.0)
        tol = 2.5e-2 if bf16 else 1e-3                   # cosine attention with logit scales up to 30 (tests/test_swin.py; measured 1.9e-2)
        desc = f"reference SwinV2-T DCT, detfill weights, B={B}, drop_path 0"
    else:
        tag, emb, heads, depth, B, hard = {"vitti": ("ti_d12_b256", 192, 3, 12, 256, True),
                                           "vits": ("s_d12_b256", 384, 6, 12, 256, False)}[arch]
        m = rg.ViT(3, 16, emb, depth=depth, n_classes=1000, drop_p=0.0, device=dev, num_heads=heads, head_size=64,
                   pixel_space="DCT", ver=1, use_subblock=True)
        shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
        m.load_state_dict({k: torch.from_numpy(v) for k, v in detfill.fill_state_dict(shapes, base_seed=1).items()})
        y = torch.from_numpy(detfill.normalish((B, 1, 28, 28, 8, 8), 71)).to(dev)
        c = torch.from_numpy(detfill.normalish((B, 2, 14, 14, 8, 8), 72)).to(dev)
        t = detfill.uniform((B, 1000), 73, 0.0, 1.0)
        tol = BF16_LOGIT_TOL if bf16 else 1e-3
        desc = f"reference ViT, detfill weights, B={B}, depth {depth}"
    if tag + "_logits" not in g.files:
        return None
    if hard:
        tgt = torch.from_numpy(detfill.integers((B,), 74, 0, 998, np.int64)).to(dev)
    else:
        tgt = torch.from_numpy(t / t.sum(1, keepdims=True)).to(dev)
    m.compute_dtype = cdt
    m.train()
    logits = m(y, c)
    loss = rg.cls_transforms.cross_entropy(logits, tgt, grad_dtype=cdt)
    loss.backward()
    gn = np.array([p.grad.double().norm().item() for _, p in m.named_parameters()])
    rel = np.abs(gn - g[tag + "_gradnorms"]) / (g[tag + "_gradnorms"] + 1e-12)
    ref = g[tag + "_logits"]
    got = logits.detach().float().cpu().numpy()
    err = float(np.abs(got - ref).max()) if got.shape == ref.shape else float("inf")
    out = {"golden": f"tests/golden/{os.path.basename(path)}:{tag} ({desc}; all {ref.shape[0]} x {ref.shape[1]} logits)",
           "max_abs_dlogit": round(err, 6), "tol": tol, "loss": round(float(loss.item()), 6),
           "loss_reference": round(float(g[tag + "_loss"]), 6), "gradnorm_rel_err_median": round(float(np.median(rel)), 6),
           "ok": bool(err <= tol and abs(float(loss.item()) - float(g[tag + "_loss"])) < 5e-3 and np.median(rel) < 2e-2)}
    del m
    torch.cuda.empty_cache()
    return out


def decode_leg(n=16):
    """Host entropy-decode cost of S-jpeg files (512x512, 4:2:0, q90) with the PRODUCT reader (a1 stays host C),
    one thread: lets the CPU baseline be quoted decode-inclusive, like the reference's own end-to-end figure.  (The reference's own
    reader, oracle/_ref, is only loaded by the CPU test tests/test_reader_cpu.py: nothing built from the reference runs here.)"""
    try:
        import tempfile
        import numpy as np
        from PIL import Image
        from rgb_no_more_amd import dct_manip as dm
    except Exception:       # noqa: BLE001  (no JPEG encoder on the box)
        return None
    rng = np.random.default_rng(0)
    d = tempfile.mkdtemp(prefix="sjpeg_")
    paths = []
    for i in range(n):
        small = rng.integers(0, 256, (32, 32, 3), dtype=np.uint8)
        img = np.asarray(Image.fromarray(small).resize((512, 512), Image.BICUBIC), dtype=np.float32)
        img = np.clip(img + rng.normal(0, 8, img.shape), 0, 255).astype(np.uint8)
        p = os.path.join(d, f"s{i:03d}.jpg")
        Image.fromarray(img).save(p, quality=90, subsampling="4:2:0")
        paths.append(p)
    for p in paths[:2]:
        dm.read_coefficients(p)
    t0 = time.perf_counter()
    for p in paths:
        dm.read_coefficients(p)
    prod = (time.perf_counter() - t0) / n
    return prod


def _cgroup_cpu_quota():
    """CPUs the cgroup grants (cpu.max), or None."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        return None if q == "max" else max(1, int(int(q) / int(per)))
    except (OSError, ValueError):
        return None


def _cgroup_throttle():
    """(nr_throttled, throttled_usec) of this cgroup so far."""
    try:
        d = dict(ln.split() for ln in open("/sys/fs/cgroup/cpu.stat").read().splitlines())
        return int(d.get("nr_throttled", 0)), int(d.get("throttled_usec", 0))
    except (OSError, ValueError):
        return 0, 0


def main():
    a = parse()
    # The GPU box shows 256 hardware threads and grants 16 (cgroup cpu.max).  Left alone, torch / OpenMP size their pools by the 256:
    # one parallel CPU op (the parity check in front of the timed region has several) leaves hundreds of spinning workers, the
    # cgroup burns its quota in a few milliseconds and EVERY thread of the process -- the one that feeds the GPU included -- is
    # frozen for the rest of the 100 ms period: the 16-step queue drains and the step time jumps by 5 - 20 % from run to run with
    # unchanged kernel times (profiles/r04_host_throttle.txt).  Size the pools by the quota.
    quota = _cgroup_cpu_quota()
    ranks_here = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1"))))      # the ranks of this node share the quota
    nthr = max(1, min(os.cpu_count() or 1, (quota or (os.cpu_count() or 1)) // ranks_here or 1, 16))
    os.environ.setdefault("OMP_NUM_THREADS", str(nthr))
    os.environ.setdefault("MKL_NUM_THREADS", str(nthr))
    if os.environ.get("RGBNM_BENCH_OMP_PASSIVE", "0") == "1":          # (experiment switch; the default keeps OpenMP's own policy so that
        os.environ.setdefault("OMP_WAIT_POLICY", "passive")            #  the cpu_baseline leg is not slowed by sleeping workers)
    import numpy as np
    import torch
    torch.set_num_threads(nthr)
    import torch.distributed as dist
    import rgb_no_more_amd as rg
    from rgb_no_more_amd import lib as L
    from rgb_no_more_amd import custom_transforms as CT

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1 and a.gpus > 1:
        raise SystemExit("for --gpus N>1 launch: python -m torch.distributed.run --nnodes=1 --nproc-per-node N "
                         "--master-addr 127.0.0.1 bench.py --gpus N ...")
    if os.environ.get("RGBNM_BENCH_SAME_DEVICE") == "1":   # smoke test of the N > 1 code path on a 1-GPU box (gloo)
        local = 0
    torch.cuda.set_device(local)
    dev = f"cuda:{local}"
    # Graph mode (one GPU): the whole run uses ONE non-default stream -- a HIP graph cannot be captured on the legacy
    # default stream, and autograd ties every parameter's AccumulateGrad node to the stream of its first use: capturing on a
    # side stream while the eager steps run on the default one costs ~150 cross-stream event waits per eager backward
    work_stream = None
    if not a.no_graph:
        work_stream = torch.cuda.Stream()
        torch.cuda.set_stream(work_stream)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.rccl_channels > 0:
            os.environ["NCCL_MAX_NCHANNELS"] = os.environ["NCCL_MIN_NCHANNELS"] = str(a.rccl_channels)
        # "nccl" IS RCCL on ROCm; RGBNM_BENCH_BACKEND=gloo only for the 1-GPU smoke test above
        dist.init_process_group(os.environ.get("RGBNM_BENCH_BACKEND", "nccl"), rank=rank, world_size=world)

    lib = L.lib()
    for o in a.opt:
        k, v = o.split("=")
        L.check(lib.rgbnm_set_option(k.encode(), int(v)), f"option {k}")
    emb, heads = ARCH[a.arch]
    cdt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    torch.manual_seed(1234 + rank)
    pcheck = None
    if rank == 0 and not a.no_parity_check:
        pcheck = parity_check(a.arch, cdt, dev)
        if pcheck is not None and not pcheck["ok"]:
            raise SystemExit(f"bench.py: parity check FAILED, refusing to time a wrong step: {json.dumps(pcheck)}")
    swin = a.arch == "swinv2t"
    if swin:
        model = rg.SwinTransformerV2(img_size=256, patch_size=4, embed_dim=96, depths=[2, 2, 6, 2], num_heads=[3, 6, 12, 24],
                                     window_size=8, drop_path_rate=0.2, device=dev, pixel_space="dct")
        model.train()
    else:
        model = rg.ViT(3, 16, emb, depth=12, n_classes=1000, drop_p=0.0, device=dev, num_heads=heads, head_size=64,
                       pixel_space="DCT", ver=1, use_subblock=True)
    model.compute_dtype = cdt
    # one GPU, or one all-reduce after the backward: no gradient is read before the backward pass is over, so the encoder
    # blocks' split-sum reductions are held and run as one launch (ViT.defer_grad_reduction; with overlapped slices or torch
    # DDP the model ignores / must not get the flag)
    if not swin and not a.no_defer_reduce:
        model.defer_grad_reduction = True
    if swin and world == 1 and not a.no_defer_reduce:
        model.group_dw_backward = True          # SwinTransformerV2: the weight-gradient GEMMs of the whole backward in one bracket
        model.hold_reductions = os.environ.get("RGBNM_SWIN_HOLD", "1") == "1"     # ... and its split-sum reductions as one launch
    net = model
    grad_sync = "none"
    if world > 1:
        grad_sync = a.grad_sync
        if grad_sync == "flat":
            # zero-copy exchange: ~4 MB slices of the flat gradient buffer are all-reduced (RCCL, AVG) from inside the
            # backward as soon as a block's gradients are final (rgb_no_more_amd/parallel.py); verified below against
            # one blocking all-reduce, with torch DDP as the fallback.  SwinV2's backward hands autograd separate gradient
            # tensors: GatheredFlatGradSync gathers each 16 MB bucket into the flat buffer when its last gradient arrives
            # (post-accumulate hooks) and all-reduces the slice -- the copy the fused optimizer would do anyway, instead of
            # DDP's 2 x 221 bucket copies
            try:
                if swin:
                    fsync = rg.parallel.GatheredFlatGradSync(model, bucket_bytes=16 << 20)
                    model.eval()
                else:
                    fsync = rg.parallel.FlatGradSync(model, bucket_bytes=4 << 20)
                try:
                    ok = _flat_sync_selfcheck(model, fsync, cdt, dev, min(a.batch, 32) if swin else a.batch, dist, 32 if swin else 28)
                finally:
                    if swin:
                        model.train()
            except Exception as e:          # noqa: BLE001
                print(f"[rank {rank}] flat gradient exchange unavailable ({type(e).__name__}: {e}); using torch DDP", file=sys.stderr)
                ok = False
            flag = torch.tensor([1 if ok else 0], device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if flag.item() == 0:
                if swin and model._grad_sync is not None:
                    try:
                        model._grad_sync.detach()
                    except Exception:           # noqa: BLE001
                        pass
                model._grad_sync = None
                grad_sync = "ddp"
            elif swin:
                grad_sync = "flat-gathered: 16 MB buckets gathered into the flat gradient buffer and all-reduced from post-accumulate hooks"
        if grad_sync == "ddp":
            model.defer_grad_reduction = False          # DDP's reducer reads .grad from hooks during the backward
            from torch.nn.parallel import DistributedDataParallel as DDP
            # several ~4 MB buckets so the all-reduce of late layers overlaps the backward of early ones
            net = DDP(model, device_ids=[local], output_device=local, bucket_cap_mb=4, gradient_as_bucket_view=False)
    # clip_grad_norm_(1) + AdamW(weight_decay=0) + the name-filtered WeightDecay of train.py (pipeline_utils.py:535-537) as the one
    # fused launch over the flat parameter buffer, for every arch (SwinV2's per-parameter gradients are gathered into the flat
    # buffer first; tests/test_swin.py compares it with train.py's own torch objects)
    opt = rg.custom_optims.FusedClipAdamWWD(model, lr=1e-3, eps=1e-8, weight_decay=1e-4, max_norm=1.0)
    mix = rg.cls_transforms.RandomMixup_DCT(1000, alpha=0.2)
    mix.out_dtype = cdt
    # the ViT mixes the batch while its sub-block kernel loads it (cls_transforms.LazyMixed: same bits, the mixed batch never
    # exists in memory); needs the augment stage's output in the compute dtype (it is) -- SwinV2 mixes the ordinary way
    mix.lazy = (not swin) and not a.no_augment and os.environ.get("RGBNM_BENCH_LAZY_MIX", "1") == "1"
    # ... and the loss mixes the target where it reads it (cls_transforms.LazyTarget: labels + lambda; same bits, no mixup_target launch)
    mix.lazy_target = os.environ.get("RGBNM_BENCH_LAZY_TARGET", "1") == "1"
    B = a.batch
    lab = torch.randint(0, 999, (B,), device=dev)
    S = 32 if swin else 28
    if a.no_augment:
        y_in = torch.randn(B, 1, S, S, 8, 8, device=dev)
        c_in = torch.randn(B, 2, S // 2, S // 2, 8, 8, device=dev)
    else:
        Yq, Cq, quant = synth_coefficients(B, dev, 1234 + rank)
        aug = CT.TrainTransform_DCT(size=S, out_dtype=cdt)
        sampler = CT.FastParamSampler(aug, seed=1234 + rank)

    def data_part(out=None):
        """out: (y, c, target, lambda) static buffers of the captured graph.  Lazy mixing: the augment stage writes y, c itself and
        the batch stays un-mixed; the mixed target and the lambda the model's first kernel reads go to their static places."""
        if a.no_augment:
            y, c = y_in, c_in
        else:
            packed, nops = sampler.sample(B, 64, 64)
            y, c = CT.apply_packed(aug, Yq, Cq, quant, packed, nops, out=out[:2] if (out is not None and mix.lazy) else None)
        if mix.lazy:
            lam = mix.sample_lambda(lab.device, out=None if out is None else out[3])
            return mix((y, c), lab, lam=lam, out=None if out is None else (None, None, None if mix.lazy_target else out[2]))
        if mix.lazy_target:
            lam = mix.sample_lambda(lab.device, out=None if out is None else out[3])
            return mix((y, c), lab, lam=lam, out=None if out is None else (out[0], out[1], None))
        return mix((y, c), lab, out=None if out is None else out[:3])

    def model_part(my, mc, mt):
        logits = net(my, mc)
        loss = rg.cls_transforms.cross_entropy(logits, mt, grad_dtype=cdt)
        loss.backward()
        return loss

    # One GPU: mixup output -> model forward -> loss -> backward (~110 of the step's ~150 launches, ~2.5 ms of host time next to
    # 5 ms of GPU time) is captured ONCE into a HIP graph and replayed; sampling, augment, mixup and the optimizer stay eager
    # (their arguments change every step).  Same kernels, same order, same bits (checked below against an eager pass); what it
    # buys is that a busy host cannot make the step launch-bound.  Steps whose kernels are bracketed with HIP events for the
    # roofline figure run eagerly.  N > 1 with collectives inside the backward stays eager.  SwinV2 is captured too (round 4): its ~600
    # launches per step cost the host 25 ms, as long as the GPU needs to run them; DropPath's masks come from captured Philox offsets that
    # advance with every replay, so its replay is checked against the eager pass's scale, not its bits.
    graph = None
    use_graph = world == 1          # N > 1: decided by the schedule calibration below (replay + ONE all-reduce after it)
    fs_saved = model._grad_sync
    if work_stream is not None and grad_sync in ("none", "flat"):
        model._grad_sync = None     # the captured backward contains no collective: at N > 1 the exchange follows the replay
        try:
            sy = torch.empty(B, 1, S, S, 8, 8, device=dev, dtype=cdt)
            sc = torch.empty(B, 2, S // 2, S // 2, 8, 8, device=dev, dtype=cdt)
            smt = torch.empty(B, 1000, device=dev, dtype=torch.float32)
            slam = torch.empty(2, device=dev, dtype=torch.float32)
            static = (sy, sc, smt, slam)
            data_part(out=static)
            if mix.lazy:
                sy, sc = rg.cls_transforms.LazyMixed(sy, slam), rg.cls_transforms.LazyMixed(sc, slam)
            if mix.lazy_target:         # the captured loss reads the labels and the lambda the data stage refreshes in place
                smt = rg.cls_transforms.LazyTarget(lab, slam, 1000)
            if swin:
                # DropPath draws new masks in every pass (captured Philox offsets advance with the replays), so a train-mode replay
                # cannot be compared bit for bit.  The capture is therefore validated FIRST with DropPath off (eval mode: same ~600
                # launches, same buffers, same order): loss and every gradient element of the replay must equal the eager pass;
                # only then is the train-mode graph captured for timing (and held to the eager pass's scale below).
                model.eval()
                try:
                    opt.zero_grad(set_to_none=True)
                    v_loss = model_part(sy, sc, smt).detach().clone()
                    v_grad = torch.cat([p.grad.reshape(-1) for p in model.parameters()]).clone()
                    for _ in range(2):
                        opt.zero_grad(set_to_none=True)
                        model_part(sy, sc, smt)
                    opt.zero_grad(set_to_none=True)
                    gv = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(gv, stream=work_stream):
                        gv_loss = model_part(sy, sc, smt)
                    gv.replay()
                    torch.cuda.synchronize()
                    gv_grad = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
                    if not torch.equal(gv_loss.detach(), v_loss) or not torch.equal(gv_grad, v_grad):
                        off, worst = 0, []
                        for n_, p_ in model.named_parameters():          # which gradients: the message names the three worst
                            d_ = (gv_grad[off:off + p_.numel()] - v_grad[off:off + p_.numel()]).abs().max().item()
                            off += p_.numel()
                            if d_ > 0:
                                worst.append((d_, n_))
                        worst.sort(reverse=True)
                        raise RuntimeError("graph replay (DropPath off) does not reproduce the eager pass bit for bit: max |d grad| "
                                           f"{(gv_grad - v_grad).abs().max().item():.3e}; {len(worst)} gradients differ, worst {worst[:3]}; "
                                           f"loss {float(gv_loss):.6f} vs {float(v_loss):.6f}")
                    del gv, gv_loss, gv_grad, v_grad
                finally:
                    model.train()
            opt.zero_grad(set_to_none=True)
            ref_loss = model_part(sy, sc, smt).detach().clone()
            ref_grad = torch.cat([p.grad.reshape(-1) for p in model.parameters()]).clone()
            for _ in range(2):
                opt.zero_grad(set_to_none=True)
                model_part(sy, sc, smt)
            opt.zero_grad(set_to_none=True)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=work_stream):      # the stream everything here runs on (see work_stream above)
                gloss = model_part(sy, sc, smt)
            g.replay()
            torch.cuda.synchronize()
            got_grad = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
            if swin:
                # DropPath draws new masks in every pass (captured Philox offsets advance with the replays), so the replay is held to
                # the eager pass's scale, not to its bits
                if not (torch.isfinite(got_grad).all() and abs(float(gloss) - float(ref_loss)) < 0.2
                        and 0.5 < float(got_grad.norm() / ref_grad.norm()) < 2.0):
                    raise RuntimeError("graph replay does not resemble the eager pass")
            elif model.flat_grad_base() is None or not torch.equal(gloss.detach(), ref_loss) or not torch.equal(got_grad, ref_grad):
                raise RuntimeError("graph replay does not reproduce the eager pass")
            graph = (g, gloss, static)
            graph_inputs = (sy, sc, smt)
        except Exception as e:          # noqa: BLE001
            print(f"[bench] HIP graph capture unavailable ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
            graph = None
            torch.cuda.synchronize()
        model._grad_sync = fs_saved
    if world > 1:                   # every rank must take the same path
        gflag = torch.tensor([1 if graph is not None else 0], device=dev)
        dist.all_reduce(gflag, op=dist.ReduceOp.MIN)
        if gflag.item() == 0:
            if graph is not None:
                print(f"[bench rank {rank}] another rank could not capture the step; running eagerly", file=sys.stderr)
            graph = None

    def exchange_after_replay():
        """N > 1 in graph mode: ONE all-reduce (average) of the flat gradient buffer the replayed backward has just written."""
        if dist.get_backend() == "nccl":
            dist.all_reduce(model._gflat, op=dist.ReduceOp.AVG)
        else:
            dist.all_reduce(model._gflat, op=dist.ReduceOp.SUM)
            model._gflat.mul_(1.0 / world)

    # N > 1, schedule "all-reduce under the next step's data stage" (parallel.DeferredFlatExchange): the exchange of step i is
    # issued behind its backward and waited for in front of optimizer step i, which is queued BEHIND the data stage of step i + 1
    deferred = None if world == 1 else rg.parallel.DeferredFlatExchange(lambda: model._gflat)
    use_deferred = False

    def step(eager=False):
        if use_deferred:
            if graph is not None and use_graph and not eager:
                data_part(out=graph[2])
                deferred.finish(opt.step)
                graph[0].replay()
                loss = graph[1]
            else:
                (my, mc), mt = data_part()
                deferred.finish(opt.step)
                opt.zero_grad(set_to_none=True)
                fs_, model._grad_sync = model._grad_sync, None     # no exchange from inside the backward: the same ONE collective
                try:
                    loss = model_part(my, mc, mt)
                finally:
                    model._grad_sync = fs_
            deferred.issue()
            return loss
        if graph is not None and use_graph and not eager:
            data_part(out=graph[2])
            graph[0].replay()
            if world > 1:
                exchange_after_replay()
            opt.step()
            return graph[1]
        opt.zero_grad(set_to_none=True)
        (my, mc), mt = data_part()
        if world > 1 and graph is not None and use_graph:
            # an eager step (rank 0's event-bracketed ones) inside the "graph replay, then one all-reduce" schedule must issue the
            # SAME collective as the replaying ranks: no exchange from inside the backward, one all-reduce of the flat buffer after it
            fs_, model._grad_sync = model._grad_sync, None
            try:
                loss = model_part(my, mc, mt)
            finally:
                model._grad_sync = fs_
            exchange_after_replay()
        else:
            loss = model_part(my, mc, mt)
        opt.step()
        return loss

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    t_pre = time.perf_counter()
    while True:                                            # DVFS ramp: cold clocks cost up to 40 % on the first runs
        go = time.perf_counter() - t_pre < a.prewarm_sec
        if world > 1:                                      # every rank runs the SAME number of steps (each one holds collectives)
            gf = torch.tensor([1 if go else 0], device=dev)
            dist.all_reduce(gf, op=dist.ReduceOp.MIN)
            go = gf.item() == 1
        if not go:
            break
        step()
        torch.cuda.synchronize()
    sync_schedule = None
    calib = None
    if world > 1 and grad_sync == "flat":
        # How to place the collectives is decided by measurement, on every rank alike, before the timed region: slices
        # all-reduced from inside the backward (overlapped) or ONE all-reduce of the flat buffer after it.  The compute
        # kernels of this library run one workgroup per CU with nearly all of its LDS, so an RCCL kernel that holds a few CUs
        # while a backward kernel launches can push that kernel into a second round -- which schedule wins depends on the
        # RCCL channel count and the link speed, i.e. on the node.
        fs = model._grad_sync
        best = None
        calib = {}
        cands = [("overlapped 4 MB slices", (4 << 20) // 4, False), ("overlapped 16 MB slices", (16 << 20) // 4, False),
                 ("one all-reduce after the backward", 1 << 60, False)]
        if graph is not None:
            # the replayed backward + one explicit all-reduce must give the gradients of the eager step with the same inputs
            fs.bucket_elems = 1 << 60
            opt.zero_grad(set_to_none=True)
            model_part(*graph_inputs)
            torch.cuda.synchronize()
            g_eager = model._gflat.clone()
            graph[0].replay()
            exchange_after_replay()
            torch.cuda.synchronize()
            err = (model._gflat - g_eager).abs().max().item()
            okf = torch.tensor([1 if (model.flat_grad_base() is not None and err <= 1e-4 * g_eager.abs().max().item() + 1e-8) else 0],
                               device=dev)
            dist.all_reduce(okf, op=dist.ReduceOp.MIN)
            if okf.item() != 1:
                print(f"[bench rank {rank}] graph replay + one all-reduce differs from the eager exchange "
                      f"(max |d| {err:.3e} of {g_eager.abs().max().item():.3e}); schedule not a candidate", file=sys.stderr)
            if okf.item() == 1:
                cands.append(("HIP graph replay, then one all-reduce", 1 << 60, True))
                cands.append(("HIP graph replay, all-reduce under the next step's data stage, optimizer behind it", 1 << 60, True))
        cands.append(("one all-reduce after the backward, under the next step's data stage, optimizer behind it", 1 << 60, False))
        for name, elems, ug in cands:
            fs.bucket_elems = elems
            use_graph = ug
            use_deferred = "next step's data stage" in name
            for _ in range(2):
                step()
            barrier()
            tc = time.perf_counter()
            for _ in range(6):
                step()
            if use_deferred:
                deferred.finish(opt.step)              # the candidate pays for its last optimizer step too
            barrier()
            tt = torch.tensor([time.perf_counter() - tc], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            calib[name] = round(tt.item() / 6 * 1e3, 3)
            if best is None or tt.item() < best[0]:
                best = (tt.item(), name, elems, ug)
        fs.bucket_elems = best[2]
        use_graph = best[3]
        sync_schedule = best[1]
        use_deferred = "next step's data stage" in best[1]
        if os.environ.get("RGBNM_BENCH_SCHEDULE"):                   # tests: pin a schedule by (a substring of) its name
            pick = [c_ for c_ in cands if os.environ["RGBNM_BENCH_SCHEDULE"] in c_[0]]
            if pick:
                sync_schedule, fs.bucket_elems, use_graph = pick[0][0], pick[0][1], pick[0][2]
                use_deferred = "next step's data stage" in sync_schedule
    # every traced kernel class: (tag, key in profiles/pmc_traffic*.json / pmc_mfma*.json, the roof SURVEY 8d names for it, description)
    TRACED = ((6, "chain_bwd", "mfma", "vit_chain_bwd_kernel: the data path of the whole encoder backward, one launch, one workgroup per image"),
              (5, "chain_fwd", "mfma", "vit_chain_fwd_kernel: the whole encoder forward, one launch, one workgroup per image"),
              (2, "gemm_tn", "mfma", "gemm_tn_pipe: weight-gradient GEMMs as grouped launches"),
              (1, "gemm_nt", "mfma", "gemm_nt family (gemm_nt_wres / gemm_nt_kpipe / gemm_nt / fused MLP: nn.Linear forward + dX GEMMs with their fused epilogues)"),
              (3, "attn_fwd", "mfma", "attention forward (per-operation path)"),
              (4, "attn_bwd", "mfma", "attention backward (per-operation path)"),
              (7, "augment", "hbm", "dct_resize + dct_randaug: dequantise, crop, resize, flip, two RandAugment ops, ToRange (SURVEY 8d: crop box in, S x S image out)"),
              (8, "subblock_embed", "hbm", "subblock_embed: batch mixup on load + sub-block reshuffle (SURVEY 8d: blocks in, 196 x 384 features out)"),
              (9, "optimizer", "hbm", "sqnorm + adamw: clip_grad_norm_ + AdamW + WeightDecay over the flat buffers (28 B per parameter + the gradient again for the norm)"))
    TRACE_MASK = sum(1 << t for t, _, _, _ in TRACED)
    trace_on = (not a.no_trace) and rank == 0
    # The set-up above left a few hundred thousand long-lived Python objects (modules, golden vectors, ctypes tables).  A full
    # collection of that heap takes the interpreter 70 - 100 ms, and the allocation count of the eager steps triggered one at a fixed
    # step of the timed region (step 64 of 80: faulthandler showed the main thread inside a plain attribute loop) -- longer than the
    # 16 queued steps last, so the GPU ran dry: +0.1 - 0.4 ms per step over an 80-step region, differing from run to run.  Collect
    # now and move the survivors to the permanent generation: later collections only look at what the loop itself allocates.
    import gc
    gc.collect()
    gc.freeze()
    if trace_on:
        # every event the traced steps of the timed region will record exists before it starts (rgbnm.h rgbnm_trace_reserve: a
        # signal-pool growth inside a traced step froze the host for 70 - 85 ms and the GPU ran dry)
        L.check(lib.rgbnm_trace_reserve((a.steps // 8 + 2) * 64), "trace_reserve")
    for i in range(a.warmup):
        if trace_on and i == 0:            # one traced step outside the timed region (first-use costs of the bracketing itself)
            lib.rgbnm_set_option(b"trace", TRACE_MASK)
        step(eager=trace_on and i == 0)
        if trace_on and i == 0:
            lib.rgbnm_set_option(b"trace", 0)
            for t_, _, _, _ in TRACED:
                lib.rgbnm_trace_collect(t_, None, None, None, None)
    if use_deferred:
        deferred.finish(opt.step)             # the timed region starts with nothing in flight: K exchanges and K optimizer steps in it
    traced_steps = 0
    step_times = [] if os.environ.get("RGBNM_BENCH_STEPTIMES") else None      # debug: host time of every timed step
    barrier()
    L.HOST_WAIT["sec"] = 0.0
    thr0, cpu0 = _cgroup_throttle(), time.process_time()
    t0 = time.perf_counter()
    for i in range(a.steps):
        # HIP-event brackets around the dominant kernel class on every 8th timed step (keeps the probe's own cost,
        # ~200 event records per traced step, below 1 % of the timed region)
        tr = trace_on and (i % 8 == 0)
        if tr:
            lib.rgbnm_set_option(b"trace", TRACE_MASK)
            traced_steps += 1
        if step_times is not None:
            ts0, hw0 = time.perf_counter(), L.HOST_WAIT["sec"]
            if tr:
                import faulthandler
                faulthandler.dump_traceback_later(0.03, exit=False)       # a traced step that takes > 30 ms shows where it sits
        loss = step(eager=tr)
        if step_times is not None and tr:
            faulthandler.cancel_dump_traceback_later()
        if tr:
            lib.rgbnm_set_option(b"trace", 0)
        if step_times is not None:
            step_times.append((time.perf_counter() - ts0 - (L.HOST_WAIT["sec"] - hw0), i, tr))
    if use_deferred:
        deferred.finish(opt.step)             # the last step's exchange and optimizer step belong to the timed region
    t_enq = time.perf_counter() - t0          # the host has enqueued every step (it may be up to 16 steps ahead of the GPU)
    host_wait = L.HOST_WAIT["sec"]
    barrier()
    dt = time.perf_counter() - t0
    thr1, cpu1 = _cgroup_throttle(), time.process_time()
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    ms = dt / a.steps * 1e3
    value = world * B * a.steps / dt
    if rank == 0 and step_times:
        worst = sorted(step_times, reverse=True)[:6]
        print("[bench] slowest steps on the host (ms, index, traced): " + ", ".join(f"({1e3 * t:.2f}, {i}, {int(tr)})" for t, i, tr in worst)
              + f"; median {1e3 * sorted(step_times)[len(step_times) // 2][0]:.2f}", file=sys.stderr)
    if rank == 0:
        lib.rgbnm_set_option(b"trace", 0)
        peak = MFMA_PEAK[a.dtype]
        step_tflops = value / world * FLOP_PER_IMG[a.arch] / 1e12
        roof = None
        roof_all = []
        whole = None
        if not a.no_trace:
            # HBM bytes per launch and the MFMA-op counters cannot be read inside this process: they come from separate `rocprofv3
            # --pmc` passes of this same command (tools/gpu.sh pass, corrected as MI355X_MICROARCH.md prescribes) whose summaries are
            # committed under profiles/ WITH the hash of the kernel sources they were measured on; a figure whose hash differs from
            # the tree this process runs from is marked "stale": true
            here = L.source_hash()
            tpath = os.path.join(ROOT, "profiles", f"pmc_traffic_{a.arch}.json")
            if not os.path.exists(tpath) and a.arch == "vitti":
                tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            tjs = json.load(open(tpath)) if os.path.exists(tpath) else {}
            mpath = os.path.join(ROOT, "profiles", "pmc_mfma.json" if a.arch == "vitti" else f"pmc_mfma_{a.arch}.json")
            mfile = json.load(open(mpath)) if os.path.exists(mpath) else {}
            mjs = mfile.get("classes") or {}

            def sourced(js, path):
                at = (js.get("measured_at") or {}) if js else {}
                return {"file": os.path.relpath(path, ROOT), "measured_at": at or None,
                        "stale": (at.get("csrc_sha16") != here) if js else None}
            tsrc, msrc = sourced(tjs, tpath), sourced(mfile, mpath)
            # attainable peaks measured on an MI355X box by tools/calib.py (nominal peaks are what `frac` is priced against)
            attain = None
            cpath = os.path.join(ROOT, "profiles", "calibration.json")
            if os.path.exists(cpath):
                cj = json.load(open(cpath))
                attain = {"hbm_copy_GBs": cj.get("hbm_copy_GBps[1 GiB]"), "hbm_read_GBs": cj.get("hbm_read_GBps[1 GiB]"),
                          "hbm_write_GBs": cj.get("hbm_write_GBps[1 GiB]"),
                          "mfma_bf16_TFLOPs": cj.get("mfma_bf16_tflops[1 workgroup (4 waves) per CU]"),
                          "source": sourced(cj, cpath)}

            def counter_bytes(key):
                if key == "augment":
                    parts = [tjs.get("dct_resize_bytes_per_launch"), tjs.get("dct_randaug_bytes_per_launch")]
                elif key == "optimizer":
                    parts = [tjs.get("sqnorm_bytes_per_launch"), tjs.get("adamw_bytes_per_launch")]
                else:
                    return tjs.get(key + "_bytes_per_launch")
                return sum(parts) if all(v is not None for v in parts) else None
            byts_step = {}
            for tag, key, bound, desc in TRACED:
                tms, fl, by, cnt = C.c_double(), C.c_double(), C.c_double(), C.c_int()
                L.check(lib.rgbnm_trace_collect(tag, C.byref(tms), C.byref(fl), C.byref(by), C.byref(cnt)))
                if not cnt.value:
                    continue
                sec = tms.value / 1e3
                gbs, tfs = by.value / sec / 1e9, fl.value / sec / 1e12
                traffic = counter_bytes(key)
                hbm = {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                       "algorithmic_bytes_per_launch": round(by.value / cnt.value),
                       "counter_frac": round(traffic * cnt.value / sec / 1e9 / HBM_PEAK_GBS, 4) if traffic else None}
                mf = {"bound": "mfma", "achieved": round(tfs, 1), "peak": peak, "unit": "TFLOP/s", "frac": round(tfs / peak, 4),
                      "algorithmic_flops_per_launch": round(fl.value / cnt.value),
                      # MFMA ops issued (hardware counter of the committed --pmc pass: padding rows and recomputed products included)
                      "counter_frac": (mjs.get(key) or {}).get("mfma_util_vs_2.5PF"), "counter_source": msrc if key in mjs else None}
                # the model step's kernels are priced against the dense MFMA roof (FLOPs: 2 x MAC, backward = 2 x forward,
                # recompute NOT counted), the augment / embed / optimizer classes against HBM; the other view rides along
                head_, other = (mf, hbm) if bound == "mfma" else (hbm, None)
                rec = dict(head_)
                rec.update({"traffic": traffic, "traffic_source": tsrc if traffic else None, "kernel": desc,
                            "launches_per_step": round(cnt.value / max(1, traced_steps), 2), "traced_steps": traced_steps,
                            "avg_launch_us": round(1e3 * tms.value / cnt.value, 2),
                            "us_per_step": round(1e3 * tms.value / max(1, traced_steps), 1)})
                if other is not None:
                    rec["hbm"] = other
                roof_all.append(rec)
                byts_step[key] = by.value / max(1, traced_steps)
            roof_all.sort(key=lambda r: -r["us_per_step"])
            roof = roof_all[0] if roof_all else None       # the dominant kernel (largest share of the step)
            if roof is not None:
                roof["attainable_peaks_measured"] = attain
            # the whole step against both roofs: algorithmic bytes = the traced kernel classes (their C entries state them) + the
            # patch-embedding GEMM; counter bytes = every dispatch of a step in the committed FETCH / WRITE passes (file-sourced)
            alg = sum(byts_step.values())
            if not swin:
                alg += B * 196 * (384 + 2 * emb) * 2.0
            cnt_b = tjs.get("whole_step_bytes")
            whole = {"algorithmic_bytes": round(alg), "counter_bytes": cnt_b, "counter_source": tsrc if cnt_b else None,
                     "frac_hbm": round(alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                     "frac_hbm_counter": round(cnt_b / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if cnt_b else None,
                     "frac_mfma": round(step_tflops / peak, 4)}
        out = {
            "metric": f"images/sec {NAMES[a.arch]} DCT 512x512 train step",
            "value": round(value, 1), "unit": "images/sec", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": ("%s --domain DCT %s, per-GPU batch %d (%s): %s + mixup + HIP model "
                                    "fwd/bwd + soft-CE + clip/AdamW/WD") %
                                   (NAMES[a.arch], a.dtype, B, CONFIG_OF[a.arch],
                                    "model-only on S-randn inputs" if a.no_augment else
                                    "HIP DCT-augment of S-coef 512x512 coefficient batches resident in HBM"),
                       "per_gpu_batch": B, "global_batch": B * world, "parallelism": f"dp{world}", "launch": "HIP graph replay of mixup-out -> forward -> loss -> backward; data stage and optimizer eager" if (graph is not None and use_graph) else "eager", "grad_sync": grad_sync if sync_schedule is None else f"{grad_sync}: {sync_schedule}", "grad_sync_calibration_ms_per_step": calib,
                       "rccl_channels": a.rccl_channels or "default",
                       "loss": round(float(loss.item()), 5)},
            "parity_mode": ("bf16 operands, fp32 accumulate: every logit within 1e-2 of the fp32 reference (torch's own bf16 autocast of "
                            "the reference deviates 6e-3); the fp32 strict mode (--dtype fp32) carries the 1e-3 north-star "
                            "tolerance (tests/test_fastpath_model.py)") if a.dtype == "bf16" else "fp32 strict mode: logits within 1e-3 of the reference",
            "parity_check": pcheck,
            # host side of the timed loop: time to enqueue all steps minus the time blocked on the 16-slot rings = what the host
            # needs per step; when that approaches ms_per_step the host, not the GPU, paces the loop
            "host_ms_per_step": round((t_enq - host_wait) / a.steps * 1e3, 3),
            "host_blocked_on_rings_ms_per_step": round(host_wait / a.steps * 1e3, 3),
            # CPU side of the timed region: cores this process kept busy, and how often / how long the cgroup was frozen for
            # exceeding its quota (any freeze longer than the queued work idles the GPU)
            "host_cpu": {"threads": nthr, "cgroup_quota_cpus": quota, "process_cores_busy": round((cpu1 - cpu0) / dt, 2),
                         "cgroup_throttled_periods": thr1[0] - thr0[0], "cgroup_throttled_ms": round((thr1[1] - thr0[1]) / 1e3, 1)},
            "mfma_pct_whole_step": round(100 * step_tflops / peak, 2),
            "step_tflops_per_gpu": round(step_tflops, 1),
            "roofline": roof,
            "whole_step": whole,
            "roofline_kernels": roof_all[1:],            # the other traced kernel classes, by time per step
        }
        if not a.no_cpu_baseline and world == 1:
            cb = cpu_baseline(a.arch, a.cpu_baseline_images)
            dsec = decode_leg()
            if dsec is not None and cb.get("value"):
                cb["decode_ms_per_img_1thread"] = round(1e3 * dsec, 3)
                cb["value_incl_entropy_decode"] = round(1.0 / (1.0 / cb["value"] + dsec), 2)
                cb["sample"] += ("; value_incl_entropy_decode adds the product reader's libjpeg coefficient read of 16 S-jpeg "
                                 "512x512 4:2:0 q90 files, one thread (BASELINE config 1 reads JPEG files)")
            # the reference ITSELF cannot run on the GPU box; tools/time_reference_cpu.py times it (BASELINE config 1) in the build
            # container and commits the result -- quoted here as data, with where and when it was measured
            rpath = os.path.join(ROOT, "profiles", "reference_cpu.json")
            if a.arch == "vitti" and os.path.exists(rpath):
                rj = json.load(open(rpath))
                cb["reference_itself"] = {k: rj.get(k) for k in ("value", "unit", "cores", "kind", "sample", "where", "model_only_value",
                                                                 "data_path_ms_per_img_1thread", "measured_at")}
                cb["reference_itself"]["source"] = "profiles/reference_cpu.json (tools/time_reference_cpu.py)"
            out["cpu_baseline"] = cb
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
