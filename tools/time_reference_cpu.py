#!/usr/bin/env python
"""Time the REFERENCE ITSELF on BASELINE.json config 1, in the build container (it never travels to the GPU box).

    python tools/time_reference_cpu.py            -> profiles/reference_cpu.json  (bench.py quotes it as cpu_baseline.reference_itself)

Workload: JPEG-Ti (vitti, embed_type 1, sub-block on), --domain DCT, fp32, batch 8, 64 synthetic
512x512 baseline 4:2:0 q90 JPEGs (S-jpeg), 8 steps, one process, torch intra-op threads = the container's CPUs.  Per image the
reference's own code: dct_manip.read_coefficients (its dct_manip.cpp compiled by oracle/build_ref.py) -> dequantise + clamp
(datasets.py:288-293) -> RandomResizedCrop_DCT / RandomFlip_DCT / RandAugment_dct / ToRange exactly as get_transform('imagenet_dct',
'train') composes them (datasets.py:354-361) with the vitti op list (utils/configs.py:93); per batch RandomMixup_DCT(alpha 0.2)
(pipeline_utils.py:181) -> models.plainvit.ViT forward -> CrossEntropyLoss -> backward -> clip_grad_norm_(1) -> AdamW(eps 1e-8,
wd 0) + WeightDecay(1e-4) (pipeline_utils.py:535-537, train.py:153-172).

The reference is IMPORTED from /root/reference like tests/golden/make_golden*.py do (torchvision stubbed: its symbols are only
dereferenced by out-of-scope functions); nothing of it is copied.  Only numbers are written.
"""
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
REF = "/root/reference"
VITTI_OPS = ("AutoContrast,Posterize,SolarizeAdd,Color,Contrast,Brightness,MidfreqAug,Cutout,TranslateX,TranslateY,Rotate90,"
             "AutoSaturation,Grayscale,ChromaDrop").split(",")          # utils/configs.py:93


def s_jpeg(d, n=64):
    from PIL import Image
    rng = np.random.default_rng(0)
    paths = []
    for i in range(n):
        small = rng.integers(0, 256, (32, 32, 3), dtype=np.uint8)
        img = np.asarray(Image.fromarray(small).resize((512, 512), Image.BICUBIC), dtype=np.float32)
        img = np.clip(img + rng.normal(0, 8, img.shape), 0, 255).astype(np.uint8)
        p = os.path.join(d, f"s{i:03d}.jpg")
        Image.fromarray(img).save(p, quality=90, subsampling="4:2:0")
        paths.append(p)
    return paths


def main(out=os.path.join(ROOT, "profiles", "reference_cpu.json"), batch=8, nfiles=64):
    if not os.path.isdir(REF):
        raise SystemExit("needs /root/reference (the build container)")
    import make_golden as MG                    # the stub set-up of the golden scripts (torchvision; dct_manip = oracle/_ref)
    dm = MG._stub_modules()
    if dm is None:
        raise SystemExit("oracle/_ref is not built: python oracle/build_ref.py")
    sys.path.insert(0, REF)
    import utils.custom_transforms as ctrans
    import utils.cls_transforms as cls
    import utils.custom_optims as coptim
    import models.plainvit as pvit

    ncpu = os.cpu_count() or 1
    torch.set_num_threads(ncpu)
    torch.manual_seed(0)
    # datasets.py:354-361 (transforms.Compose is torchvision's: the same chain, applied in order)
    chain = [ctrans.RandomResizedCrop_DCT(28, scale=(0.05, 1.0), ratio=(1, 1), dtype_resize=torch.float32),
             ctrans.RandomFlip_DCT(p=0.5, direction="horizontal"),
             ctrans.RandAugment_dct(num_ops=2, magnitude=3, num_magnitude_bins=11, ops_list=VITTI_OPS),
             ctrans.ToRange(val_min=-1, val_max=1, orig_min=-1024, orig_max=1016, dtype=torch.float32)]
    model = pvit.ViT(in_channels=3, patch_size=16, emb_size=192, depth=12, n_classes=1000, drop_p=0.0, device="cpu",
                     dtype=torch.float32, num_heads=3, head_size=64, pixel_space="DCT", ver=1, use_subblock=True)
    mixup = cls.RandomMixup_DCT(1000, alpha=0.2)
    criterion = torch.nn.CrossEntropyLoss()
    optimizer = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=0, eps=1e-8)
    decayer = coptim.WeightDecay([p for n, p in model.named_parameters() if (".weight" in n) and ("lrnorm" not in n)], lr=1e-3,
                                 weight_decay=1e-4)
    tdir = tempfile.mkdtemp(prefix="sjpeg_ref_")
    paths = s_jpeg(tdir, nfiles)
    labels = torch.randint(0, 999, (nfiles,))

    t_read = t_aug = 0.0

    def sample(i):                               # datasets.py:286-297
        nonlocal t_read, t_aug
        t0 = time.perf_counter()
        dim, quant, Y, cbcr = dm.read_coefficients(paths[i])
        t1 = time.perf_counter()
        Y = torch.clamp(Y * quant[0], min=-2 ** 10, max=2 ** 10 - 8)
        cbcr = torch.clamp(cbcr * quant[1:3].unsqueeze(1).unsqueeze(1), min=-2 ** 10, max=2 ** 10 - 8)
        co = (Y, cbcr)
        for t in chain:
            co = t(co)
        t2 = time.perf_counter()
        t_read += t1 - t0
        t_aug += t2 - t1
        return co

    def step(idx):
        ys, cs = zip(*[sample(i) for i in idx])
        t0 = time.perf_counter()
        y, c, lab = torch.stack(ys), torch.stack(cs), labels[idx]
        (y, c), tgt = mixup((y, c), lab)                                     # pipeline_utils.py:70-76
        optimizer.zero_grad()
        loss = criterion(model(y, c), tgt)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=1)       # train.py:163-165
        optimizer.step()
        decayer.step()
        return time.perf_counter() - t0, float(loss.detach())

    order = list(range(nfiles))
    step(order[:batch])                           # warm-up (first-use costs of MKL / the JIT-free reference ops)
    t_read = t_aug = 0.0
    t_model, t0 = 0.0, time.perf_counter()
    for s in range(nfiles // batch):
        dt, loss = step(order[s * batch:(s + 1) * batch])
        t_model += dt
    total = time.perf_counter() - t0
    n = nfiles // batch * batch
    try:
        head = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    except Exception:       # noqa: BLE001
        head = None
    cpu = ""
    try:
        cpu = [ln.split(":", 1)[1].strip() for ln in open("/proc/cpuinfo") if ln.startswith("model name")][0]
    except Exception:       # noqa: BLE001
        pass
    res = {"value": round(n / total, 2), "unit": "images/sec", "cores": ncpu, "kind": "reference",
           "sample": f"{n} S-jpeg files (512x512, 4:2:0, q90), batch {batch}, fp32, JPEG-Ti, {nfiles // batch} steps after one warm-up step: "
                     f"entropy decode {1e3 * t_read / n:.2f} ms/img + dequantise / crop / resize / flip / RandAugment / ToRange "
                     f"{1e3 * t_aug / n:.2f} ms/img (per sample, one process) + mixup / forward / loss / backward / clip / AdamW / "
                     f"WeightDecay {1e3 * t_model / n:.2f} ms/img ({ncpu} intra-op threads)",
           "where": f"build container ({ncpu} vCPU, {cpu}), torch {torch.__version__} CPU; the reference's own dct_manip.cpp + Python path "
                    f"imported from /root/reference; it cannot run on the GPU box",
           "model_only_value": round(n / t_model, 2),
           "data_path_ms_per_img_1thread": round(1e3 * (t_read + t_aug) / n, 3),
           "final_loss": round(loss, 4),
           "measured_at": {"head": head, "date": time.strftime("%Y-%m-%d")}}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:2])
