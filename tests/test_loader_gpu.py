"""DCTBatchLoader on the GPU: pinned ring -> one H2D copy per tensor on a side stream -> fused HIP transform.  What it
yields must be exactly what the same files give when read one by one with `dct_manip.read_coefficients` (the reference's
per-sample entry point, datasets.py:287) and pushed through the same transform."""
import numpy as np
import pytest
import torch

import rgb_no_more_amd as rg
from rgb_no_more_amd import custom_transforms as CT
from rgb_no_more_amd import dct_manip as dm
from rgb_no_more_amd.loader import DCTBatchLoader

pytestmark = pytest.mark.gpu


def test_loader_feeds_the_fused_eval_transform(tmp_path):
    PIL = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(0)
    paths, labels = [], []
    for i in range(10):
        small = rng.integers(0, 256, (16, 16, 3), dtype=np.uint8)
        img = np.asarray(PIL.fromarray(small).resize((512, 512), PIL.BICUBIC), dtype=np.float32)
        img = np.clip(img + rng.normal(0, 8, img.shape), 0, 255).astype(np.uint8)
        p = tmp_path / f"s{i}.jpg"
        PIL.fromarray(img).save(str(p), quality=90, subsampling="4:2:0")
        paths.append(str(p))
        labels.append(i)
    ev = rg.datasets.get_transform("imagenet_dct", "val", dtype=torch.float32, fused=True)
    ld = DCTBatchLoader(paths, labels, batch_size=4, device="cuda", threads=4, prefetch=2, shuffle=False, transform=ev)
    got_labels = []
    for (Y, C), lab in ld:
        assert Y.is_cuda and Y.dtype == torch.float32 and Y.shape[1:] == (1, 28, 28, 8, 8) and C.shape[1:] == (2, 14, 14, 8, 8)
        for k, li in enumerate(lab.tolist()):
            dim, quant, y1, c1 = dm.read_coefficients(paths[li])
            wy, wc = ev(y1.unsqueeze(0).cuda(), c1.unsqueeze(0).cuda(), quant.unsqueeze(0).cuda())
            assert torch.equal(Y[k], wy[0]) and torch.equal(C[k], wc[0]), li
        got_labels += lab.tolist()
    assert got_labels == list(range(10))
    # the train transform path: device tensors of the right shape, finite, in [-1, 1]
    tr = rg.datasets.get_transform("imagenet_dct", "train", ops_list=CT.VITTI_OPS, ops_magnitude=3, dtype=torch.bfloat16, fused=True)
    ld2 = DCTBatchLoader(paths, labels, batch_size=5, device="cuda", threads=4, seed=1, transform=tr)
    n = 0
    for (Y, C), lab in ld2:
        assert Y.dtype == torch.bfloat16 and float(Y.float().abs().max()) <= 1.0 and torch.isfinite(C.float()).all()
        n += Y.shape[0]
    assert n == 10


def test_crop_on_host_ships_the_crop_and_gives_the_same_bits(tmp_path):
    """crop_on_host=True: the crop boxes are drawn before the decode, only the boxes cross PCIe (packed back to back), the
    augment kernels read them in place -- and every output equals the whole-grid path run with the same parameters."""
    PIL = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(1)
    paths, labels = [], []
    for i in range(24):
        small = rng.integers(0, 256, (16, 16, 3), dtype=np.uint8)
        img = np.asarray(PIL.fromarray(small).resize((512, 512), PIL.BICUBIC), dtype=np.float32)
        img = np.clip(img + rng.normal(0, 8, img.shape), 0, 255).astype(np.uint8)
        p = tmp_path / f"c{i}.jpg"
        if i % 7 == 3:
            PIL.fromarray(img[..., 0]).save(str(p), quality=90)                  # a grayscale file: zero chroma
        else:
            PIL.fromarray(img).save(str(p), quality=90, subsampling="4:2:0")
        paths.append(str(p))
        labels.append(i)
    tr = rg.datasets.get_transform("imagenet_dct", "train", ops_list=CT.VITTI_OPS, ops_magnitude=3, dtype=torch.float32, fused=True)
    ld = DCTBatchLoader(paths, labels, batch_size=8, device="cuda", threads=4, seed=5, shuffle=False, transform=tr, crop_on_host=True)
    seen, sides, shipped = 0, set(), 0
    for (Y, C), lab in ld:
        packed, nops = ld.last_packed
        li = lab.tolist()
        Yf, Cf, Qf = dm.read_coefficients_batch([paths[i] for i in li], threads=4)
        wy, wc = CT.apply_packed(tr, Yf.cuda(), Cf.cuda(), Qf.cuda(), packed, nops)
        assert torch.equal(Y, wy) and torch.equal(C, wc)
        sides |= set(packed["crop"][:, 2].tolist())
        shipped += ld.h2d_bytes
        seen += len(li)
    assert seen == 24 and len(sides) >= 2
    assert shipped < 0.8 * seen * (64 * 64 + 2 * 32 * 32) * 128, shipped           # less than the whole grids (787 KB each)
    print(f"crop_on_host: {shipped / seen / 1024:.0f} KB per image shipped (whole grid: 768 KB), crop sides {sorted(sides)}")
