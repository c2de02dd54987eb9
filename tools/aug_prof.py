#!/usr/bin/env python
"""Experiment only (build with RGBNM_HIPCC_FLAGS=-DAUG_PROF): per-wave cycle stamps of dct_resize_kernel on the bench's data stage.
usage: RGBNM_HIPCC_FLAGS=-DAUG_PROF python rgb-no-more_amd/build.py && python tools/aug_prof.py
Prints the kernel's span, the distribution of the waves' lifetimes, the time per item of each resize mode (least squares over the
waves) -- i.e. what the cost weights AUG_K_HALF / AUG_K_ID / AUG_K_DBL should be -- and where the late waves sit."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench as BN
from rgb_no_more_amd import lib as L
import rgb_no_more_amd as rg

CT = rg.custom_transforms
dev = torch.device("cuda:0")
B = 256
Yq, Cq, quant = BN.synth_coefficients(B, dev, 1234)
aug = CT.TrainTransform_DCT(size=28, out_dtype=torch.bfloat16)
sampler = CT.FastParamSampler(aug, seed=1234)
lib = L.lib()
f = lib.rgbnm_debug_aug_prof
f.restype = C.c_int
f.argtypes = [C.c_void_p]
f2 = lib.rgbnm_debug_aug_prof2
f2.restype = C.c_int
f2.argtypes = [C.c_void_p]
rows = []
for it in range(12):
    packed, nops = sampler.sample(B, 64, 64)
    y, c = CT.apply_packed(aug, Yq, Cq, quant, packed, nops)
    torch.cuda.synchronize()
    buf = np.zeros(4096 * 8, dtype=np.uint64)
    assert f(buf.ctypes.data) == 0
    p = buf.reshape(4096, 8).astype(np.int64)
    if it < 4:
        continue
    t0, t1, t2 = p[:, 0], p[:, 1], p[:, 2]
    n = np.stack([(p[:, 3] >> 40) & 0xFFFFF, (p[:, 3] >> 20) & 0xFFFFF, p[:, 3] & 0xFFFFF], 1).astype(np.float64)
    cyc = p[:, 4:7].astype(np.float64)
    start = t0.min()
    span = t2.max() - start
    life = t2 - t0
    work = t2 - t1
    xcc = (p[:, 7] >> 32) & 15
    with np.errstate(divide="ignore", invalid="ignore"):
        per = np.where(n > 0, cyc / n, np.nan)
    # least squares: work cycles = k0 n0 + k1 n1 + k2 n2 + c
    A = np.concatenate([n, np.ones((4096, 1))], 1)
    k, *_ = np.linalg.lstsq(A, work.astype(np.float64), rcond=None)
    rows.append((span, life.mean(), np.percentile(life, 50), np.percentile(life, 95), life.max(), (t1 - t0).mean(), (t0 - start).max(), k))
    print(f"pass {it}: span {span} cyc; wave life mean {life.mean():.0f} p50 {np.percentile(life, 50):.0f} p95 {np.percentile(life, 95):.0f} "
          f"max {life.max()}; prefix+sync {np.mean(t1 - t0):.0f}; latest start {np.max(t0 - start)}; end spread p5 {np.percentile(t2 - start, 5):.0f} "
          f"p50 {np.percentile(t2 - start, 50):.0f} p95 {np.percentile(t2 - start, 95):.0f}")
    print(f"   cycles / item by mode (mean over the waves that had some): /2 {np.nanmean(per[:, 0]):.0f}  id {np.nanmean(per[:, 1]):.0f}  x2 {np.nanmean(per[:, 2]):.0f}"
          f"   items total {n.sum(0)}   lstsq per item {k[0]:.0f} {k[1]:.0f} {k[2]:.0f} + {k[3]:.0f}")
    late = np.argsort(t2)[-40:]
    print("   the 40 latest waves: items", n[late].sum(0), " mean work", work[late].mean(), " xcc histogram", np.bincount(xcc[late], minlength=8),
          " mean end by xcc", [int(np.mean((t2 - start)[xcc == x])) for x in range(8)])
    pure = [(n[:, m] > 0) & (n.sum(1) == n[:, m]) for m in range(3)]
    print("   waves with ONE mode only: count / mean work / items:", [(int(q.sum()), int(work[q].mean()) if q.any() else 0, float(n[q].sum(1).mean()) if q.any() else 0) for q in pure])
    b2 = np.zeros(4096 * 8, dtype=np.uint64)
    assert f2(b2.ctypes.data) == 0
    q = b2.reshape(256, 16, 8).astype(np.int64)
    ld, o1, o2, stt, tot = q[:, :, 1] - q[:, :, 0], q[:, :, 2] - q[:, :, 1], q[:, :, 3] - q[:, :, 2], q[:, :, 4] - q[:, :, 3], q[:, :, 4] - q[:, :, 0]
    print(f"   kernel 2 (per wave, cycles): load+sync mean {ld.mean():.0f} max {ld.max()}; op slot 0 mean {o1.mean():.0f} max {o1.max()}; op slot 1 mean {o2.mean():.0f} max {o2.max()}; "
          f"ToRange+store mean {stt.mean():.0f} max {stt.max()}; wave life mean {tot.mean():.0f} p95 {np.percentile(tot, 95):.0f} max {tot.max()}")
    byop = {}
    for b in range(256):
        byop.setdefault(int(q[b, 0, 5]), []).append(o1[b].max())
        byop.setdefault(int(q[b, 0, 6]), []).append(o2[b].max())
    print("   cycles by op id (slot time, slowest wave of the workgroup):", {k: int(np.mean(v)) for k, v in sorted(byop.items())})
