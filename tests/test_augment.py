"""GPU parity of the batched DCT-domain data path (csrc/augment.hip via custom_transforms.TrainTransform_DCT) against
the numpy oracle (oracle/dct_np.py, itself pinned to the reference by tests/golden/g2..g8).

Bar: integer and index work bit exact; the fp32 resize may differ by one LSB only where the
exact (fp64) pre-round value sits on a .5 tie."""
import numpy as np
import pytest
import torch

import rgb_no_more_amd as rg
from rgb_no_more_amd import detfill, custom_transforms as CT
from oracle import dct_np as O

pytestmark = pytest.mark.gpu
DEV = "cuda"
Q_LUMA = np.array([3, 2, 2, 3, 5, 8, 10, 12, 2, 2, 3, 4, 5, 12, 12, 11, 3, 3, 3, 5, 8, 11, 14, 11, 3, 3, 4, 6, 10, 17, 16,
                   12, 4, 4, 7, 11, 14, 22, 21, 15, 5, 7, 11, 13, 16, 21, 23, 18, 10, 13, 16, 17, 21, 24, 24, 20, 14, 18,
                   19, 20, 22, 20, 21, 20], dtype=np.int16).reshape(8, 8)   # libjpeg quality-90 luma table


def synth(B, Hy=64, Wy=64, seed=0, gray=False):
    """S-coef style synthetic quantised coefficients: DC ~ wide, AC decaying, a few overflow cases."""
    rng = np.random.default_rng(seed)
    u, v = np.meshgrid(np.arange(8), np.arange(8), indexing="ij")
    scale = 60.0 / (1 + u + v)
    Y = np.rint(rng.laplace(0, 1, (B, 1, Hy, Wy, 8, 8)) * scale / Q_LUMA).astype(np.int16)
    Y[..., 0, 0] = np.rint(rng.normal(0, 300, (B, 1, Hy, Wy)) / Q_LUMA[0, 0]).astype(np.int16)
    Y[0, 0, 3, 5, 0, 0] = 20000      # wraps in int16 after * q
    Cc = None
    if not gray:
        Cc = np.rint(rng.laplace(0, 1, (B, 2, Hy // 2, Wy // 2, 8, 8)) * scale / (Q_LUMA + 2)).astype(np.int16)
        Cc[..., 0, 0] = np.rint(rng.normal(0, 120, (B, 2, Hy // 2, Wy // 2)) / 5).astype(np.int16)
    quant = np.stack([Q_LUMA, Q_LUMA + 2, Q_LUMA + 2])[None].repeat(B, 0).astype(np.int16)
    return Y, Cc, quant


def run_hip(Y, Cc, quant, params, eval_mode=False, out_dtype=torch.float32, size=28):
    t = CT.TrainTransform_DCT(size=size, eval_mode=eval_mode, out_dtype=out_dtype)
    oy, oc = t(torch.from_numpy(Y).to(DEV), None if Cc is None else torch.from_numpy(Cc).to(DEV),
               torch.from_numpy(quant).to(DEV), params=params)
    torch.cuda.synchronize()
    return oy.float().cpu().numpy(), oc.float().cpu().numpy()


def run_oracle(Y, Cc, quant, params, size=28):
    ys, cs = [], []
    for b, p in enumerate(params):
        oy, oc = O.train_transform(Y[b], None if Cc is None else Cc[b], quant[b], p["box"], p["flip"], p["ops"], size)
        ys.append(oy)
        cs.append(oc)
    return np.stack(ys), np.stack(cs)


ALL_OPS = [("Identity", 0.0, None), ("AutoContrast", 0.0, None), ("Posterize", 2.0, None), ("SolarizeAdd", 264.9, None),
           ("Color", 0.27, None), ("Color", -0.27, None), ("Contrast", 0.27, None), ("Contrast", -0.27, None),
           ("Brightness", 0.27, None), ("Brightness", -0.27, None), ("MidfreqAug", 0.27, None),
           ("MidfreqAug", -0.27, None), ("Cutout", 1.8, (0, 0)), ("Cutout", 1.8, (12, 26)), ("Cutout", 1.8, (26, 4)),
           ("TranslateX", 3.75, None), ("TranslateX", -3.75, None), ("TranslateY", 3.75, None),
           ("TranslateY", -3.75, None), ("Rotate90", 1.0, None), ("Rotate90", -1.0, None), ("AutoSaturation", 0.0, None),
           ("Grayscale", 0.0, None), ("ChromaDrop", 0.0, True), ("ChromaDrop", 0.0, False), ("Sharpness", 0.27, None),
           ("Sharpness", -0.27, None), ("Invert", 0.0, None), ("Solarize", 327.2, None), ("Solarize", -163.6, None),
           ("FreqEnhance", 0.27, None), ("FreqEnhance", -0.27, None), ("Equalize", 0.0, None)]


def test_every_op_bit_exact_on_identity_resize():
    """crop side 28 = no resampling: the whole chain is integer/index work + exact DC arithmetic -> bit exact."""
    B = len(ALL_OPS)
    Y, Cc, quant = synth(B, 40, 48, seed=1)
    params = []
    for b, op in enumerate(ALL_OPS):
        second = ALL_OPS[(b * 7 + 3) % len(ALL_OPS)]
        params.append(dict(box=(2 * (b % 5), 2 * (b % 9), 28, 28), flip=bool(b & 1), ops=[op, second]))
    hy, hc = run_hip(Y, Cc, quant, params)
    ry, rc = run_oracle(Y, Cc, quant, params)
    for b in range(B):
        assert np.array_equal(hy[b], ry[b]), (b, params[b]["ops"], np.abs(hy[b] - ry[b]).max())
        assert np.array_equal(hc[b], rc[b]), (b, params[b]["ops"], np.abs(hc[b] - rc[b]).max())


def test_magnitudes_match_reference_table(golden):
    g = golden("g6_photo.npz")
    mine = CT.magnitude_table(11, (6, 6))
    for n, v in zip([str(s) for s in g["mag_names"]], g["mag_vals"]):
        if n in mine:
            m = mine[n][0]
            assert (float(m[3]) if m.ndim > 0 else float(m)) == v, n
    bank = CT._FilterBank()
    assert CT.encode_op("TranslateX", 3.75, None, bank)[2] == 2 and CT.encode_op("TranslateX", -3.75, None, bank)[2] == -4
    assert CT.encode_op("Cutout", 1.8, (4, 6), bank)[2:] == (2, 4, 6)
    assert CT.encode_op("Posterize", 2.0, None, bank)[2:4] == (2, 511)


@pytest.mark.parametrize("side", [56, 14])
def test_resize_within_one_lsb_exact_off_ties(side):
    B = 4
    Y, Cc, quant = synth(B, 64, 64, seed=2)
    params = [dict(box=(2 * b, 8 - 2 * b, side, side), flip=bool(b & 1), ops=[("Identity", 0.0, None)]) for b in range(B)]
    hy, hc = run_hip(Y, Cc, quant, params)
    ry, rc = run_oracle(Y, Cc, quant, params)
    lsb = 2.0 / 2040.0
    for h, r, nm in ((hy, ry, "Y"), (hc, rc, "C")):
        d = np.abs(h - r) / lsb
        assert d.max() <= 1.0 + 1e-3, (nm, d.max())
        frac = (d > 0.5).mean()
        print(f"side {side} {nm}: {100 * frac:.3f}% of coefficients differ by one LSB")
        assert frac < 0.06
    # exactness away from ties, checked against the fp64 pre-round values of the oracle
    for b in range(B):
        i, j, hh, ww = params[b]["box"]
        Yd, _ = O.dequantize(Y[b], Cc[b], quant[b])
        raw = O.resize_raw(O.crop(Yd, i, j, hh, ww), 28, np.float64)
        if params[b]["flip"]:
            raw = O.flip(raw)
        near_tie = np.abs(raw - np.floor(raw) - 0.5) < 2e-3
        exp = np.clip(np.rint(raw), -1024, 1016)
        got = np.rint((hy[b] + 1) / 2 * 2040 - 1024)
        assert np.array_equal(got[~near_tie], exp[~near_tie])


def test_eval_transform_center_crop_downsample():
    B = 3
    Y, Cc, quant = synth(B, 64, 64, seed=3)
    t = CT.EvalTransform_DCT()
    params = t.sample_params(B, 64, 64)
    assert params[0]["box"] == (4, 4, 56, 56) and params[0]["ops"] == []
    hy, hc = run_hip(Y, Cc, quant, params, eval_mode=True)
    lsb = 2.0 / 2040.0
    for b in range(B):
        ry, rc = O.eval_transform(Y[b], Cc[b], quant[b])
        assert np.abs(hy[b] - ry).max() <= lsb * 1.001 and (np.abs(hy[b] - ry) > lsb / 2).mean() < 0.06
        assert np.abs(hc[b] - rc).max() <= lsb * 1.001


def test_grayscale_jpeg_and_bf16_output():
    Y, _, quant = synth(2, 32, 32, seed=4, gray=True)
    params = [dict(box=(0, 2, 28, 28), flip=False, ops=[("Brightness", 0.27, None), ("AutoSaturation", 0.0, None)]),
              dict(box=(4, 0, 28, 28), flip=True, ops=[("Rotate90", -1.0, None), ("Color", 0.27, None)])]
    hy, hc = run_hip(Y, None, quant, params)
    ry, rc = run_oracle(Y, None, quant, params)
    assert np.array_equal(hy, ry) and np.array_equal(hc, rc)
    by, bc = run_hip(Y, None, quant, params, out_dtype=torch.bfloat16)
    assert np.array_equal(by, torch.from_numpy(ry).bfloat16().float().numpy())
    assert np.array_equal(bc, torch.from_numpy(rc).bfloat16().float().numpy())


def test_sampled_params_follow_reference_distribution_and_run():
    torch.manual_seed(0)
    t = CT.TrainTransform_DCT()
    params = t.sample_params(400, 64, 64)
    sides = np.array([p["box"][3] for p in params])
    assert set(np.unique(sides)) <= {14, 28, 56}
    f56 = (sides == 56).mean()
    assert 0.5 < f56 < 0.75          # reference: 62.1 % / 31.8 % / 6.0 %
    assert all(p["box"][0] % 2 == 0 and p["box"][1] % 2 == 0 for p in params)
    for p in params:
        names = [o[0] for o in p["ops"]]
        assert len(names) == 2
        if names[0] == "Grayscale":
            assert names[1] not in CT.CHROMA_OPS
        if names[0] in ("Color", "AutoSaturation", "ChromaDrop"):
            assert names[1] != "Grayscale"
    Y, Cc, quant = synth(8, 64, 64, seed=5)
    oy, oc = t(torch.from_numpy(Y).to(DEV), torch.from_numpy(Cc).to(DEV), torch.from_numpy(quant).to(DEV))
    assert oy.shape == (8, 1, 28, 28, 8, 8) and oc.shape == (8, 2, 14, 14, 8, 8)
    assert torch.isfinite(oy).all() and oy.min() >= -1 and oy.max() <= 1


def test_invalid_crop_is_rejected():
    Y, Cc, quant = synth(1, 64, 64, seed=6)
    with pytest.raises(rg.lib.RgbnmError):
        run_hip(Y, Cc, quant, [dict(box=(0, 0, 42, 42), flip=False, ops=[])])
    with pytest.raises(rg.lib.RgbnmError):
        run_hip(Y, Cc, quant, [dict(box=(40, 40, 56, 56), flip=False, ops=[])])


def test_swin_pipeline_32_block_grid_every_op_and_resize_cases():
    """get_transform('imagenet_dct_swin') (datasets.py:370-382): the same kernels instantiated for a 32 x 32 output grid
    (image kept in L2 instead of LDS).  Identity-resize crops: every op bit exact; crop 16 (x2) / 64 (/2): <= 1 LSB;
    eval = Resize_DCT(32) of the whole 64 x 64 grid."""
    B = len(ALL_OPS)
    Y, Cc, quant = synth(B, 44, 52, seed=7)
    params = []
    for b, op in enumerate(ALL_OPS):
        second = ALL_OPS[(b * 5 + 1) % len(ALL_OPS)]
        params.append(dict(box=(2 * (b % 6), 2 * (b % 10), 32, 32), flip=bool(b & 1), ops=[op, second]))
    hy, hc = run_hip(Y, Cc, quant, params, size=32)
    assert hy.shape[1:] == (1, 32, 32, 8, 8) and hc.shape[1:] == (2, 16, 16, 8, 8)
    ry, rc = run_oracle(Y, Cc, quant, params, size=32)
    for b in range(B):
        assert np.array_equal(hy[b], ry[b]), (b, params[b]["ops"], np.abs(hy[b] - ry[b]).max())
        assert np.array_equal(hc[b], rc[b]), (b, params[b]["ops"], np.abs(hc[b] - rc[b]).max())
    lsb = 2.0 / 2040.0
    Y2, C2, q2 = synth(4, 64, 64, seed=8)
    for side in (16, 64):
        pr = [dict(box=(2 * b, 6 - 2 * b, side, side) if side == 16 else (0, 0, 64, 64), flip=bool(b & 1),
                   ops=[("Identity", 0.0, None)]) for b in range(4)]
        h2, c2 = run_hip(Y2, C2, q2, pr, size=32)
        r2, rc2 = run_oracle(Y2, C2, q2, pr, size=32)
        for h, r in ((h2, r2), (c2, rc2)):
            d = np.abs(h - r) / lsb
            assert d.max() <= 1.0 + 1e-3 and (d > 0.5).mean() < 0.06, (side, d.max())
    t = CT.TrainTransform_DCT(size=32, eval_mode=True)
    pe = t.sample_params(4, 64, 64)
    assert pe[0]["box"] == (0, 0, 64, 64) and pe[0]["ops"] == []
    he, ce = run_hip(Y2, C2, q2, pe, eval_mode=True, size=32)
    for b in range(4):
        Yd, Cd = O.dequantize(Y2[b], C2[b], q2[b])
        ref = O.to_range(O.resize(Yd, 32))
        assert np.abs(he[b] - ref).max() <= lsb * 1.001
    # the reference crop-side distribution for size 32 on 64 x 64 grids: {16, 32, 64}
    tt = CT.TrainTransform_DCT(size=32)
    sides = {p["box"][3] for p in tt.sample_params(400, 64, 64)}
    assert sides <= {16, 32, 64} and len(sides) == 3


def test_torange_quotient_is_exact_for_every_int16():
    """The kernel's ToRange divides by 2040 with a reciprocal and one residual step instead of the division sequence; the class
    passes whatever coefficients the preceding transform left (no clamp), so EVERY int16 value must come out with the bits of
    the reference's statements (custom_transforms.py:447-452): fp32 exactly, bf16 = the rounded fp32 chain."""
    vals = np.arange(-32768, 32768, dtype=np.int16)
    Y = np.zeros((2, 1, 28, 28, 8, 8), np.int16)
    Y.reshape(-1)[:65536] = vals                       # 2 x 50176 luma slots hold all 65536 values
    Y.reshape(-1)[65536:] = vals[::-1][:2 * 50176 - 65536]
    C = np.zeros((2, 2, 14, 14, 8, 8), np.int16)
    C.reshape(-1)[:] = vals[7000:7000 + 2 * 25088]
    for dt in (torch.float32, torch.bfloat16):
        oy, oc = CT.ToRange(-1, 1, -1024, 1016, dtype=dt)((torch.from_numpy(Y).to(DEV), torch.from_numpy(C).to(DEV)))
        torch.cuda.synchronize()
        for got, src in ((oy, Y), (oc, C)):
            x = torch.from_numpy(src).float()
            want = (-1.0 + ((x - (-1024)) / torch.full((), 2040.0)) * 2.0).to(dt)      # the fused pipeline's output cast (see the bf16 case above)
            assert got.dtype == dt and torch.equal(got.cpu(), want.reshape(got.shape)), dt


def test_batches_beyond_one_prefix_table_equal_their_chunks():
    """Kernel 1's cost prefix table (kernel arguments) covers 512 images; launch() loops beyond.  A batch of 600 with the sampler's
    mix of crop sides must come out with the bits of the same images run as batches of 300."""
    torch.manual_seed(1)
    B = 600
    t = CT.TrainTransform_DCT()
    params = t.sample_params(B, 64, 64)
    Y8, C8, q8 = synth(8, 64, 64, seed=9)
    idx = np.arange(B) % 8
    Y = torch.from_numpy(Y8[idx]).to(DEV)
    Cc = torch.from_numpy(C8[idx]).to(DEV)
    quant = torch.from_numpy(q8[idx]).to(DEV)
    oy, oc = t(Y, Cc, quant, params=params)
    for lo in (0, 300):
        py, pc = t(Y[lo:lo + 300], Cc[lo:lo + 300], quant[lo:lo + 300], params=params[lo:lo + 300])
        assert torch.equal(oy[lo:lo + 300], py) and torch.equal(oc[lo:lo + 300], pc)
    ident = [b for b in range(64) if params[b]["box"][3] == 28][:6]            # identity-resize images: every op is exact
    ry, rc = run_oracle(Y8[idx[ident]], C8[idx[ident]], q8[idx[ident]], [params[b] for b in ident])
    assert np.array_equal(oy[ident].cpu().numpy(), ry) and np.array_equal(oc[ident].cpu().numpy(), rc)
