"""The reference's per-transform classes (utils/custom_transforms.py:406-1138) as device modules: each class against the
numpy oracle with explicit parameters, the get_transform() Compose chain against the fused batch transform (bit for bit),
and the calling conventions of the reference (single tensor / (Y, CbCr) tuple, per sample / batched)."""
import numpy as np
import pytest
import torch

import rgb_no_more_amd as rg
from rgb_no_more_amd import custom_transforms as CT
from rgb_no_more_amd import detfill
from oracle import dct_np as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def coeffs(B, H, W, seed, lo=-1024, hi=1016):
    Y = detfill.integers((B, 1, H, W, 8, 8), seed, lo, hi)
    C = detfill.integers((B, 2, H // 2, W // 2, 8, 8), seed + 1, lo, hi)
    return Y, C


def dev(a):
    return torch.from_numpy(a).to(DEV)


def test_flip_crop_torange_are_bit_exact_vs_oracle():
    Y, C = coeffs(3, 28, 28, 11)
    oy, oc = CT.RandomFlip_DCT(p=1.0)((dev(Y), dev(C)))
    for b in range(3):
        assert np.array_equal(oy[b].cpu().numpy(), O.flip(Y[b])) and np.array_equal(oc[b].cpu().numpy(), O.flip(C[b]))
    ny, nc = CT.RandomFlip_DCT(p=0.0)((dev(Y), dev(C)))
    assert np.array_equal(ny.cpu().numpy(), Y) and np.array_equal(nc.cpu().numpy(), C)
    # values beyond the clamp range survive a flip unclamped (the reference's flip_dct does not clamp, dct_ops.py:601-621)
    Yb = Y.copy()
    Yb[0, 0, 3, 5, 2, 3] = 1020
    Yb[0, 0, 3, 6, 1, 1] = -1030
    fy = CT.RandomFlip_DCT(p=1.0)(dev(Yb))
    assert np.array_equal(fy.cpu().numpy()[0], O.flip(Yb[0]))
    # vertical (custom_transforms.py:919; dct_ops.py:617-620) = horizontal flip + half turn on the kernels.  The half turn runs as
    # RandAugment ops, whose per-op clamp (custom_transforms.py:1019-1020) bounds what a sign flip can produce: a coefficient in
    # [-1024, -1017] that the flip negates comes out as 1016 where flip_dct gives up to 1024 -- the value RandAugment's entry
    # clamp (:1106-1108) makes of it one stage later in every reference pipeline.  Bit exact on [-1016, 1016]; stated, not hidden
    Ys, Cs = coeffs(3, 28, 28, 31, lo=-1016)
    for flags in ((True, True, True), (True, False, True)):
        outs = [CT.RandomFlip_DCT(p=1.0, direction="vertical")((dev(Ys[b:b + 1]), dev(Cs[b:b + 1])), flip=flags[b]) for b in range(3)]
        for b in range(3):
            wy, wc = (O.flip(Ys[b], "vertical"), O.flip(Cs[b], "vertical")) if flags[b] else (Ys[b], Cs[b])
            assert np.array_equal(outs[b][0][0].cpu().numpy(), wy) and np.array_equal(outs[b][1][0].cpu().numpy(), wc)
    fv = CT.RandomFlip_DCT(p=1.0, direction="vertical")(dev(Yb))
    assert np.array_equal(fv.cpu().numpy()[0], np.clip(O.flip(np.clip(Yb[0], O.CMIN, O.CMAX), "vertical"), O.CMIN, O.CMAX))
    # crops: 64 -> 28 window, no resize
    Y2, C2 = coeffs(2, 64, 64, 21)
    cy, cc = CT.CenterCrop_DCT(28)((dev(Y2), dev(C2)))
    assert np.array_equal(cy.cpu().numpy(), Y2[:, :, 18:46, 18:46]) and np.array_equal(cc.cpu().numpy(), C2[:, :, 9:23, 9:23])
    ry, rc = CT.RandomCrop_DCT(28)((dev(Y2), dev(C2)), box=(6, 30, 28, 28))
    assert np.array_equal(ry.cpu().numpy(), Y2[:, :, 6:34, 30:58]) and np.array_equal(rc.cpu().numpy(), C2[:, :, 3:17, 15:29])
    torch.manual_seed(3)
    i, j, h, w = CT.RandomCrop_DCT(28).get_params(64, 64)
    assert i % 2 == 0 and j % 2 == 0 and (h, w) == (28, 28) and 0 <= i <= 36 and 0 <= j <= 36
    # ToRange: fp32 values of the reference formula, bf16 = its rounding
    ty, tc = CT.ToRange(-1, 1, -1024, 1016, torch.float32)((dev(Y), dev(C)))
    assert np.array_equal(ty.cpu().numpy(), O.to_range(Y)) and np.array_equal(tc.cpu().numpy(), O.to_range(C))
    by = CT.ToRange(-1, 1, -1024, 1016, torch.bfloat16)(dev(Y))
    assert torch.equal(by.cpu(), torch.from_numpy(O.to_range(Y)).bfloat16())
    # any other range -- the class default (orig_max = 1024) included -- and any grid: the reference's two fp32 statements
    for args in ((), (0.0, 1.0, -1024, 1024), (-2.0, 3.0, -1000, 1000)):
        gy, gc = CT.ToRange(*args)((dev(Y), dev(C)))
        kw = dict(zip(("val_min", "val_max", "orig_min", "orig_max"), args)) if args else dict(orig_max=1024)
        assert np.array_equal(gy.cpu().numpy(), O.to_range(Y, **kw)) and np.array_equal(gc.cpu().numpy(), O.to_range(C, **kw))
    Yr, _ = coeffs(1, 20, 36, 5)
    assert np.array_equal(CT.ToRange(-1, 1, -1024, 1016)(dev(Yr)).cpu().numpy(), O.to_range(Yr))      # non-square grid
    # bf16 through the fallback: the reference casts FIRST and evaluates both statements in bf16 (custom_transforms.py:447-451)
    gb = CT.ToRange(-2.0, 3.0, -1000, 1000, torch.bfloat16)(dev(Y))
    xb = torch.from_numpy(Y).to(torch.bfloat16)
    xb = (xb - (-1000)) / (1000 - (-1000))
    xb = -2.0 + (xb * (3.0 - (-2.0)))
    assert gb.dtype == torch.bfloat16 and torch.equal(gb.cpu(), xb)


@pytest.mark.parametrize("size,side", [(28, 56), (28, 28), (28, 14), (32, 64), (32, 16)])
def test_resized_crop_classes_vs_oracle(size, side):
    Y, C = coeffs(2, 64, 64, 31)
    box = (4, 6, side, side) if side < 64 else (0, 0, 64, 64)
    oy, oc = CT.RandomResizedCrop_DCT(size)((dev(Y), dev(C)), box=box)
    assert oy.dtype == torch.int16 and oy.shape == (2, 1, size, size, 8, 8) and oc.shape == (2, 2, size // 2, size // 2, 8, 8)
    for b in range(2):
        ry = O.resize(O.crop(Y[b], *box), size)
        rc = O.resize(O.crop(C[b], box[0] // 2, box[1] // 2, side // 2, side // 2), size // 2)
        dy = np.abs(oy[b].cpu().numpy().astype(np.int32) - ry.astype(np.int32))
        dc = np.abs(oc[b].cpu().numpy().astype(np.int32) - rc.astype(np.int32))
        assert dy.max() <= 1 and dc.max() <= 1                     # <= 1 LSB on exact .5 ties
        assert (dy != 0).mean() < 0.02 and (dc != 0).mean() < 0.02
    if side == 2 * size:                                           # the eval transforms are these two boxes
        if size == 28:
            ey, ec = CT.ResizedCenterCrop_DCT(32, 28)((dev(Y), dev(C)))
            wy, wc = CT.RandomResizedCrop_DCT(28)((dev(Y), dev(C)), box=(4, 4, 56, 56))
        else:
            ey, ec = CT.Resize_DCT(32)((dev(Y), dev(C)))
            wy, wc = CT.RandomResizedCrop_DCT(32)((dev(Y), dev(C)), box=(0, 0, 64, 64))
        assert torch.equal(ey, wy) and torch.equal(ec, wc)


def test_randaugment_class_vs_oracle_ops():
    Y, C = coeffs(2, 28, 28, 41)
    meta = CT.magnitude_table(11, (28, 28))
    cases = [("Posterize", float(meta["Posterize"][0][3]), None), ("TranslateX", -float(meta["TranslateX"][0][3]), None),
             ("Cutout", float(meta["Cutout"][0][3]), (6, 10)), ("Rotate90", 1.0, None), ("Color", 0.27, None),
             ("SolarizeAdd", float(meta["SolarizeAdd"][0][3]), None), ("Grayscale", 0.0, None), ("Brightness", -0.27, None)]
    ra = CT.RandAugment_dct(num_ops=2, magnitude=3, num_magnitude_bins=11, ops_list=CT.VITTI_OPS)
    for k in range(0, len(cases), 2):
        ops = cases[k:k + 2]
        oy, oc = ra((dev(Y), dev(C)), ops=ops)
        for b in range(2):
            ry, rc = np.clip(Y[b], -1024, 1016), np.clip(C[b], -1024, 1016)
            for name, mag, aux in ops:
                ry, rc = O.apply_op(ry, rc, name, mag, aux)
            assert np.array_equal(oy[b].cpu().numpy(), ry), ops
            assert np.array_equal(oc[b].cpu().numpy(), rc), ops
    # num_ops > 2 (custom_transforms.py:1024: any count): chained two at a time, every per-op clamp where the reference has it
    for ops in (cases[:3], cases[2:7], cases[3:7]):
        oy, oc = CT.RandAugment_dct(num_ops=len(ops), magnitude=3, ops_list=CT.VITTI_OPS)((dev(Y), dev(C)), ops=ops)
        for b in range(2):
            ry, rc = np.clip(Y[b], -1024, 1016), np.clip(C[b], -1024, 1016)
            for name, mag, aux in ops:
                ry, rc = O.apply_op(ry, rc, name, mag, aux)
            assert np.array_equal(oy[b].cpu().numpy(), ry) and np.array_equal(oc[b].cpu().numpy(), rc), ops
    torch.manual_seed(1)
    s5y, _ = CT.RandAugment_dct(num_ops=5, magnitude=3, ops_list=CT.VITTI_OPS)((dev(Y), dev(C)))
    assert s5y.dtype == torch.int16 and int(s5y.max()) <= 1016 and int(s5y.min()) >= -1024
    # sampling path: structure, dtype, range
    torch.manual_seed(0)
    sy, sc = ra((dev(Y), dev(C)))
    assert sy.dtype == torch.int16 and sy.shape == (2, 1, 28, 28, 8, 8) and int(sy.max()) <= 1016 and int(sy.min()) >= -1024
    with pytest.raises(NotImplementedError):
        CT.RandAugment_dct(ops_list=["Warp"])
    with pytest.raises(NotImplementedError):
        CT.TrainTransform_DCT(size=28, ops_list=["Rotate"])          # the fused transform has no DFT-plane ops


def test_dft_plane_rotate_and_shear_vs_oracle():
    """Rotate / ShearX / ShearY of RandAugment_dct (custom_transforms.py:949-968, dct_ops.py:367-434, 957-1013) against
    oracle/dft_np.py -- PARITY UNPINNED (both restate torchvision's nearest-neighbour rotate / affine, which could not be run next
    to the reference); chained with kernel ops on either side, per-sample magnitudes, the reference's sqrt(2) padding.  The complex
    matrix products are summed in another order on the device: <= 1 LSB on a handful of rounding ties (3e-4 of the coefficients
    after one such op, 5e-3 after two)."""
    from oracle import dft_np as D
    Y, C = coeffs(2, 28, 28, 77)
    meta = CT.magnitude_table(11, (28, 28))
    assert float(meta["Rotate"][0][10]) == 30.0 and float(meta["ShearX"][0][10]) == 17.0 and meta["ShearY"][1]
    ra = CT.RandAugment_dct(num_ops=2, magnitude=3)
    assert "Rotate" in ra.ops_list and "ShearX" in ra.ops_list and "ShearY" in ra.ops_list      # the reference's default list
    cases = [[("Rotate", 9.0, None), ("Brightness", 0.27, None)],
             [("Contrast", -0.27, None), ("ShearX", float(meta["ShearX"][0][3]), None)],
             [("ShearY", -float(meta["ShearY"][0][3]), None), ("Rotate", -100.0, None)],
             [("Rotate", 90.0, None), ("TranslateX", float(meta["TranslateX"][0][3]), None)]]
    for ops in cases:
        oy, oc = ra((dev(Y), dev(C)), ops=ops)
        assert oy.dtype == torch.int16 and oc.shape == (2, 2, 14, 14, 8, 8)
        for b in range(2):
            ry, rc = np.clip(Y[b], -1024, 1016), np.clip(C[b], -1024, 1016)
            for name, mag, aux in ops:
                if name in CT.DFT_OPS:
                    ry, rc = D.apply_dft_op(ry, rc, name, mag, pad=2 ** 0.5)
                else:
                    ry, rc = O.apply_op(ry, rc, name, mag, aux)
            for got, ref in ((oy[b].cpu().numpy(), ry), (oc[b].cpu().numpy(), rc)):
                d = np.abs(got.astype(np.int32) - ref.astype(np.int32))
                # (a second DFT-plane op spreads every 1-LSB difference of its input over its whole output: more ties flip)
                bound = 2e-3 if sum(o[0] in CT.DFT_OPS for o in ops) < 2 else 2e-2
                assert d.max() <= 1 and (d > 0).mean() < bound, (ops, d.max(), (d > 0).mean())
    # quarter turns are exact index work: the same bits as the oracle (which, like the reference, turns the PADDED grid -- 39 blocks,
    # margins 5 and 6 -- so the result is the exact Rotate90 only where the padded size is even), and exactly Rotate90 without padding
    oy, oc = ra((dev(Y), dev(C)), ops=[("Rotate", 90.0, None)])
    assert np.array_equal(oy[0].cpu().numpy(), D.rotate_block(np.clip(Y[0], -1024, 1016), 90.0, 2 ** 0.5))
    oy, oc = CT.RandAugment_dct(num_ops=1, pad=False)((dev(Y), dev(C)), ops=[("Rotate", -90.0, None)])
    assert np.array_equal(oy[1].cpu().numpy(), O.rotate90(np.clip(Y[1], -1024, 1016), -1))
    assert np.array_equal(oc[1].cpu().numpy(), O.rotate90(np.clip(C[1], -1024, 1016), -1))
    # sampled path with the default list: runs, int16, per-sample ops
    torch.manual_seed(5)
    sy, sc = CT.RandAugment_dct(num_ops=2, magnitude=5, ops_list=["Rotate", "ShearX", "ShearY", "Color"])((dev(Y), dev(C)))
    assert sy.dtype == torch.int16 and sy.shape == (2, 1, 28, 28, 8, 8) and sc.dtype == torch.int16


def test_get_transform_chain_equals_fused_transform_bitwise():
    """datasets.get_transform('imagenet_dct', 'train') as the reference composes it == the fused TrainTransform_DCT on the
    same explicit parameters, bit for bit (fp32 and bf16 outputs), per sample and batched."""
    B = 4
    Y, C = coeffs(B, 64, 64, 51)
    boxes = [(0, 0, 56, 56), (10, 4, 28, 28), (30, 40, 14, 14), (8, 8, 56, 56)]
    flips = [True, False, True, True]
    meta = CT.magnitude_table(11, (28, 28))
    opss = [[("Contrast", 0.27, None), ("TranslateY", float(meta["TranslateY"][0][3]), None)],
            [("Cutout", float(meta["Cutout"][0][3]), (4, 20)), ("AutoContrast", 0.0, None)],
            [("MidfreqAug", -0.27, None), ("ChromaDrop", 0.0, True)],
            [("Rotate90", -1.0, None), ("AutoSaturation", 0.0, None)]]
    for odt in (torch.float32, torch.bfloat16):
        fused = CT.TrainTransform_DCT(size=28, out_dtype=odt)
        fy, fc = fused(dev(Y), dev(C), fused._unit_quant(B, DEV),
                       params=[dict(box=boxes[b], flip=flips[b], ops=opss[b]) for b in range(B)])
        chain = rg.datasets.get_transform("imagenet_dct", "train", ops_list=CT.VITTI_OPS, num_ops=2, ops_magnitude=3, dtype=odt)
        rrc, flip, ra, tr = chain.transforms
        assert [type(t).__name__ for t in chain.transforms] == ["RandomResizedCrop_DCT", "RandomFlip_DCT", "RandAugment_dct", "ToRange"]
        for b in range(B):                                      # the reference's convention: one sample (C,H,W,8,8) at a time
            x = (dev(Y[b]), dev(C[b]))
            x = rrc(x, box=boxes[b])
            x = flip(x, flip=flips[b])
            x = ra(x, ops=opss[b])
            cy, cc = tr(x)
            assert cy.shape == (1, 28, 28, 8, 8) and cy.dtype == odt
            assert torch.equal(cy, fy[b]) and torch.equal(cc, fc[b]), (odt, b)
    # sampled end to end: runs, right shapes/dtypes, values in [-1, 1]
    torch.manual_seed(1)
    oy, oc = rg.datasets.get_transform("imagenet_dct", "train", ops_list=CT.VITTI_OPS, ops_magnitude=3)((dev(Y), dev(C)))
    assert oy.shape == (B, 1, 28, 28, 8, 8) and oc.shape == (B, 2, 14, 14, 8, 8) and oy.dtype == torch.float32
    assert float(oy.abs().max()) <= 1.0 and float(oc.abs().max()) <= 1.0
    ey, ec = rg.datasets.get_transform("imagenet_dct", "val")((dev(Y), dev(C)))
    fe = CT.EvalTransform_DCT()
    wy, wc = fe(dev(Y), dev(C), fe._unit_quant(B, DEV))
    assert torch.equal(ey, wy) and torch.equal(ec, wc)
    sy, sc = rg.datasets.get_transform("imagenet_dct_swin", "test")((dev(Y), dev(C)))
    assert sy.shape == (B, 1, 32, 32, 8, 8) and sc.shape == (B, 2, 16, 16, 8, 8)
    # luma-only input (grayscale JPEG: CbCr is None in the reference) keeps its structure
    gy = rg.datasets.get_transform("imagenet_dct", "val")(dev(Y[0]))
    assert torch.is_tensor(gy) and gy.shape == (1, 28, 28, 8, 8) and torch.equal(gy, ey[0])


@pytest.mark.gpu
def test_resize_dct_general_factors_vs_reference_golden(golden):
    """Resize_DCT for grids the HIP augment kernels do not cover (any factor: dct_ops.resize_dct, device tensor ops restating
    utils/dct_ops.py:436-580) against the reference itself (golden g23: 20 -> 28, 36 -> 28, 24 x 20 -> 28, chroma 10 x 12 -> 14,
    6 -> 4, and the class on a (Y, CbCr) pair): <= 1 LSB, exact off .5 ties; batched = per sample; the factors 1/2, 1, 2 of the
    28-block pipeline still take the kernels (same results as the oracle, as test_resized_crop_classes_vs_oracle checks)."""
    from rgb_no_more_amd import dct_ops as D
    g = golden("g23_resize_general.npz")
    for nm in [str(s) for s in g["case_names"]]:
        x, size = torch.from_numpy(g[nm + "_in"]).to(DEV), int(g[nm + "_size"])
        out = D.resize_dct(x, size)
        assert out.dtype == torch.int16 and out.is_cuda
        ref, raw = g[nm + "_f32"], g[nm + "_f64raw"].astype(np.float64)
        diff = np.abs(out.cpu().numpy().astype(np.int32) - ref.astype(np.int32))
        assert diff.max() <= 1, nm
        frac = np.abs(raw - np.floor(raw) - 0.5)
        assert (diff[frac > 1e-3] == 0).all(), nm
        both = D.resize_dct(torch.stack([x, x.flip(1)]), size)
        assert torch.equal(both[0], out)
    Y, C = torch.from_numpy(g["pair_Y"]).to(DEV), torch.from_numpy(g["pair_C"]).to(DEV)
    oy, oc = CT.Resize_DCT(28)((Y, C))                           # 20 x 20 / 10 x 10: not a kernel case
    for got, want in ((oy, g["pair_oY"]), (oc, g["pair_oC"])):
        d = np.abs(got.cpu().numpy().astype(np.int32) - want.astype(np.int32))
        assert got.shape == want.shape and d.max() <= 1 and (d > 0).mean() < 1e-3
    # a kernel case goes on taking the kernels and agrees with the general path to the rounding of ties
    Y56 = torch.from_numpy(detfill.integers((2, 1, 56, 56, 8, 8), 91, -1024, 1016)).to(DEV)
    C28 = torch.from_numpy(detfill.integers((2, 2, 28, 28, 8, 8), 92, -1024, 1016)).to(DEV)
    ky, kc = CT.Resize_DCT(28)((Y56, C28))
    gy = D.resize_dct(Y56, 28)
    assert ky.shape == gy.shape and (ky.int() - gy.int()).abs().max().item() <= 1
