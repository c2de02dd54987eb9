"""Pin the oracle (oracle/dct_np.py, oracle/vit_torch.py) against golden vectors captured from the
reference itself (tests/golden/make_golden.py).  CPU only."""
import math

import numpy as np
import pytest
import torch

from oracle import dct_np as O
from oracle import vit_torch as V
from rgb_no_more_amd import detfill


def test_g2_dequant(golden):
    g = golden("g2_dequant.npz")
    oY, oC = O.dequantize(g["Y"], g["C"], g["quant"])
    assert np.array_equal(oY, g["oY"]) and np.array_equal(oC, g["oC"])
    # the synthetic overflow really wraps: 1000*255 = 255000 -> int16 -7144 -> clamp -1024
    assert g["oY"][0, 0, 0, 0, 0] == -1024
    gY, gC = O.dequantize(g["Y"], None, g["quant"])
    assert gC.shape == (2, 1, 1, 8, 8) and not gC.any()


def test_g3_conversion_matrices(golden):
    g = golden("g3_convmat.npz")
    for ls, m in [(8, 2), (4, 2), (2, 4), (8, 1)]:
        A = O.conversion_matrix(ls, m)
        assert A.dtype == np.float32
        np.testing.assert_allclose(A, g[f"A_{ls}_{m}"], rtol=0, atol=2e-6)
        np.testing.assert_allclose(O.conversion_matrix(ls, m, np.float64), g[f"A64_{ls}_{m}"], atol=1e-12)
        np.testing.assert_allclose(A @ A.T, np.eye(ls * m), atol=5e-6)
    np.testing.assert_allclose(O.basis_matrix(8), g["basis8"], atol=1e-6)


def test_g4_geometric_bit_exact(golden):
    g = golden("g4_geom.npz")
    Y, C = g["Y"], g["C"]
    assert np.array_equal(O.crop(Y, 2, 2, 4, 4), g["crop_Y_2_2_4_4"])
    assert np.array_equal(O.crop(C, 1, 1, 2, 2), g["crop_C_1_1_2_2"])
    assert np.array_equal(O.flip(Y), g["flipH_Y"]) and np.array_equal(O.flip(C), g["flipH_C"])
    assert np.array_equal(O.flip(Y, "vertical"), g["flipV_Y"])
    for r in (-1, 1, 2, 3, -3, 4):
        assert np.array_equal(O.rotate90(Y, r), g[f"rot{r}_Y"]), r
        assert np.array_equal(O.rotate90(C, r), g[f"rot{r}_C"]), r
    for mag in (2, -4):
        for d in ("H", "W"):
            assert np.array_equal(O.translate(Y, mag, d), g[f"trans{d}{mag}_Y"])
            assert np.array_equal(O.translate(C, mag // 2, d), g[f"trans{d}{mag}_C"])
    for (ch, cw) in [(0, 0), (2, 4), (4, 6)]:
        assert np.array_equal(O.cutout(Y, 2, ch, cw), g[f"cutout_{ch}_{cw}_Y"])
        assert np.array_equal(O.cutout(C, 1, ch // 2, cw // 2), g[f"cutout_{ch}_{cw}_C"])


def test_g5_resize_within_one_lsb_exact_off_ties(golden):
    g = golden("g5_resize.npz")
    for nm in ("Y", "C"):
        x = g[nm]
        hb = x.shape[1]
        for size in (hb * 2, hb, hb // 2):
            ref32 = g[f"{nm}_to{size}_f32"]
            raw64 = g[f"{nm}_to{size}_f64raw"]
            out = O.resize(x, size)
            assert out.dtype == np.int16 and out.shape == ref32.shape
            diff = np.abs(out.astype(np.int32) - ref32.astype(np.int32))
            assert diff.max() <= 1
            # exact wherever the fp64 pre-round value is not within 1e-3 of a .5 tie
            frac = np.abs(raw64 - np.floor(raw64) - 0.5)
            assert (diff[frac > 1e-3] == 0).all()
            raw = O.resize_raw(x, size)
            np.testing.assert_allclose(raw, raw64, atol=2e-3)


def test_g23_general_resize_factors(golden):
    """g23 (make_golden_r5_resize.py): utils/dct_ops.py:529-580 for grids that are not size / 2, size or 2 x size (20 -> 28 = x7 / 5,
    36 -> 28 = x7 / 9, a non-square 24 x 20, chroma 10 x 12 -> 14, 6 -> 4) and Resize_DCT(28) on a (Y, CbCr) pair (chroma size
    ceil(28 / 2), custom_transforms.py:505-507): the oracle's resize is the same up / down composition -- <= 1 LSB, exact off ties."""
    g = golden("g23_resize_general.npz")
    for nm in [str(s) for s in g["case_names"]]:
        x, size = g[nm + "_in"], int(g[nm + "_size"])
        out, ref, raw = O.resize(x, size), g[nm + "_f32"], g[nm + "_f64raw"].astype(np.float64)
        assert out.dtype == np.int16 and out.shape == ref.shape
        diff = np.abs(out.astype(np.int32) - ref.astype(np.int32))
        assert diff.max() <= 1
        frac = np.abs(raw - np.floor(raw) - 0.5)
        assert (diff[frac > 1e-3] == 0).all()
        np.testing.assert_allclose(O.resize_raw(x, size), raw, atol=4e-3)
    for inp, want, size in ((g["pair_Y"], g["pair_oY"], 28), (g["pair_C"], g["pair_oC"], 14)):
        d = np.abs(O.resize(inp, size).astype(np.int32) - want.astype(np.int32))
        assert d.max() <= 1 and (d > 0).mean() < 1e-3


def test_g6_photometric_and_dispatch(golden):
    g = golden("g6_photo.npz")
    Y, C = g["Y"], g["C"]
    names = [str(s) for s in g["mag_names"]]
    table = O.magnitude_table(11, (6, 6))
    for n, v in zip(names, g["mag_vals"]):
        m, _ = table[n]
        mine = float(m[3]) if np.ndim(m) > 0 else float(m)
        assert mine == v, (n, mine, v)
    m28, _ = O.magnitude_table(11, (28, 28))["TranslateX"]
    assert float(m28[3]) == float(g["mag_translate_28"])
    ops = ["AutoContrast", "Posterize", "SolarizeAdd", "Color", "Contrast", "Brightness", "MidfreqAug",
           "TranslateX", "TranslateY", "Rotate90", "AutoSaturation", "Grayscale", "Sharpness", "Identity"]
    for op in ops:
        for s in ("p", "n"):
            key = f"{op}_{s}_Y"
            if key not in g.files:
                continue
            mag = float(g[f"{op}_{s}_mag"])
            oy, oc = O.apply_op(Y, C, op, mag)
            assert np.array_equal(oy, g[key]), (op, s)
            assert np.array_equal(oc, g[f"{op}_{s}_C"]), (op, s)
    cut_mag = dict(zip(names, g["mag_vals"]))["Cutout"]
    for seed in (0, 1, 2):
        ch, cw = g[f"Cutout_s{seed}_center"]
        oy, oc = O.apply_op(Y, C, "Cutout", cut_mag, (int(ch), int(cw)))
        assert np.array_equal(oy, g[f"Cutout_s{seed}_Y"]) and np.array_equal(oc, g[f"Cutout_s{seed}_C"])
        _, oc = O.apply_op(Y, C, "ChromaDrop", 0.0, int(g[f"ChromaDrop_s{seed}_dropcb"]))
        assert np.array_equal(oc, g[f"ChromaDrop_s{seed}_C"])
    assert np.array_equal(O.autocontrast(g["AutoContrast_zero_in"]), g["AutoContrast_zero_out"])


def test_magnitude_quirks():
    # +3.75 -> +2 blocks, -3.75 -> -4 blocks; cutout round(1.8)=2
    m = 3.75
    assert int(m - (m % 2)) == 2 and int(-m - ((-m) % 2)) == -4
    t = O.magnitude_table(11, (28, 28))
    assert abs(float(t["TranslateX"][0][3]) - 3.75) < 1e-6
    assert round(float(t["Cutout"][0][3])) == 2
    assert int(t["Posterize"][0][3]) == 2 and int(float(t["SolarizeAdd"][0][3])) == 264


def test_g7_torange(golden):
    g = golden("g7_torange.npz")
    assert np.array_equal(O.to_range(g["x"]), g["out"])


def test_g8_get_params(golden):
    g = golden("g8_params.npz")
    n_ok = 0
    for row in g["rrc"]:
        size, H, W, seed, u, ri, rj, i, j, h, w, first_ok = row
        side = O.rrc_box_side(int(H), int(W), int(size), float(u))
        if first_ok:
            assert side == int(w) == int(h)
            assert O.rrc_params(int(H), int(W), int(size), float(u), int(ri), int(rj)) == (int(i), int(j), int(h), int(w))
            n_ok += 1
    assert n_ok > 150
    for H, W, i, j, h, w in g["rcc"]:
        assert O.rcc_params(int(H), int(W)) == (int(i), int(j), int(h), int(w))
    assert O.rcc_params(64, 64) == (4, 4, 56, 56)


def test_g9_subblock_features(golden):
    g = golden("g9_subblock.npz")
    y = torch.from_numpy(detfill.normalish((2, 1, 4, 6, 8, 8), 61))
    c = torch.from_numpy(detfill.normalish((2, 2, 2, 3, 8, 8), 62))
    feat = V.subblock_features(y, c).numpy()
    np.testing.assert_allclose(feat, g["feat"], rtol=0, atol=3e-6)
    # the chroma 128 features are a pure index shuffle: bit exact
    assert np.array_equal(feat[..., 256:], g["feat"][..., 256:])


def test_g10_sincos(golden):
    g = golden("g10_sincos.npz")
    assert np.array_equal(V.sincos_table(14, 14, 192).numpy(), g["t192"])
    assert np.array_equal(V.sincos_table(14, 14, 384).numpy()[[0, 1, 13, 14, 97, 195]], g["t384_rows"])


@pytest.mark.parametrize("tag,emb,heads,depth,B", [("ti_d2", 192, 3, 2, 2), ("ti_d12", 192, 3, 12, 4), ("s_d2", 384, 6, 2, 2)])
def test_g11_model(golden, tag, emb, heads, depth, B):
    g = golden("g11_model.npz")
    shapes = V.param_shapes(depth, emb, heads)
    assert [str(s) for s in g[tag + "_names"]] == list(shapes.keys())
    if depth == 12:
        assert len(shapes) == 152 and sum(int(np.prod(s)) for s in shapes.values()) == 5642728
    p = {k: torch.from_numpy(v).requires_grad_(True) for k, v in detfill.fill_state_dict(shapes, 1).items()}
    y = torch.from_numpy(detfill.normalish((B, 1, 28, 28, 8, 8), 71))
    c = torch.from_numpy(detfill.normalish((B, 2, 14, 14, 8, 8), 72))
    tgt = detfill.uniform((B, 1000), 73, 0.0, 1.0)
    tgt = torch.from_numpy(tgt / tgt.sum(1, keepdims=True))
    logits, inter = V.vit_forward(p, y, c, depth, heads, emb, return_inter=True)
    np.testing.assert_allclose(inter[0].detach()[:, ::49, ::16].numpy(), g[tag + "_x0_slice"], atol=2e-5)
    np.testing.assert_allclose(inter[1].detach()[:, ::49, ::16].numpy(), g[tag + "_x1_slice"], atol=2e-5)
    np.testing.assert_allclose(logits.detach().numpy(), g[tag + "_logits"], atol=2e-5)
    loss = V.soft_xent(logits, tgt)
    assert abs(loss.item() - float(g[tag + "_loss"])) < 1e-5
    loss.backward()
    gn = np.array([p[k].grad.double().norm().item() for k in shapes])
    np.testing.assert_allclose(gn, g[tag + "_gradnorms"], rtol=2e-4, atol=1e-7)
    for nm in ("patchembed.projection.0.weight", "encoder.0.0.fn.eb_mha.qkv.weight", "encoder.0.0.fn.eb_mha.qkv.bias",
               "encoder.1.1.fn.eb_ffb.3.weight", "encoder.0.0.fn.eb_lrnorm1.weight", "classhead.ch_linear2.bias"):
        np.testing.assert_allclose(p[nm].grad.reshape(-1)[::37].numpy(), g[tag + "_grad_" + nm], rtol=1e-3, atol=2e-7)


def test_g12_optimizer(golden):
    g = golden("g12_optim.npz")
    names = ["a.weight", "a.bias", "x_lrnorm.weight", "b.weight"]
    shapes = [(5, 7), (5,), (7,), (3, 5)]
    params = [detfill.uniform(s, 81 + i).astype(np.float32) for i, s in enumerate(shapes)]
    m = [np.zeros_like(p) for p in params]
    v = [np.zeros_like(p) for p in params]
    mask = [(".weight" in n) and ("lrnorm" not in n) for n in names]
    lrs = [3e-3, 1.5e-3, 2.5e-3]
    for it in range(3):
        grads = [detfill.uniform(shapes[i], 91 + 10 * it + i, -2.0, 2.0) for i in range(4)]
        tn = V.clip_adamw_wd_step(params, grads, m, v, it + 1, lrs[it], 3e-3, 1e-4, mask)
        assert abs(tn - float(g[f"norm{it + 1}"])) < 1e-4
        flat = np.concatenate([p.reshape(-1) for p in params])
        np.testing.assert_allclose(flat, g[f"p{it + 1}"], rtol=0, atol=2e-6)


def test_g13_mixup(golden):
    g = golden("g13_mixup.npz")
    lam = g["lam"]
    assert lam[0] >= lam[1] and abs(lam.sum() - 1) < 1e-6
    oh = torch.nn.functional.one_hot(torch.from_numpy(g["lab"]), 10).float()
    my, mc, mt = V.mixup(torch.from_numpy(g["y"]), torch.from_numpy(g["c"]), oh, float(lam[0]), float(lam[1]))
    np.testing.assert_allclose(my.numpy(), g["my"], atol=1e-6)
    np.testing.assert_allclose(mc.numpy(), g["mc"], atol=1e-6)
    np.testing.assert_allclose(mt.numpy(), g["mt"], atol=1e-6)


def test_g16_out_of_list_ops_invert_solarize_freqenhance(golden):
    """Invert / Solarize / FreqEnhance / Equalize through the reference's _apply_op_dct: bit exact."""
    g = golden("g16_ops2.npz")
    for k in range(int(g["ncases"])):
        oy, oc = O.apply_op(g["Y"], g["C"], str(g[f"case{k}_name"]), float(g[f"case{k}_mag"]))
        assert np.array_equal(oy, g[f"case{k}_Y"]) and np.array_equal(oc, g[f"case{k}_C"]), k


def test_g22_agrees_with_g20(golden):
    """g22 (make_golden_r5.py) re-ran the reference JPEG-Ti at B = 256 for more gradient slices: everything it shares with g20
    (make_golden_r3.py, same seeds) must be the same numbers -- the two files pin each other."""
    g20, g22 = golden("g20_fullsize.npz"), golden("g22_ti_b256_grads.npz")
    tag = "ti_d12_b256"
    assert abs(float(g20[tag + "_loss"]) - float(g22[tag + "_loss"])) < 1e-6
    np.testing.assert_allclose(g22[tag + "_gradnorms"], g20[tag + "_gradnorms"], rtol=1e-5)
    np.testing.assert_allclose(g22[tag + "_logits_every8"], g20[tag + "_logits"][:, ::8], atol=1e-5)
    names = [str(n) for n in g22[tag + "_slice_names"]]
    assert len(names) == 14
    shared = [n for n in names if tag + "_grad_" + n in g20.files]
    assert len(shared) >= 2
    for n in shared:
        np.testing.assert_allclose(g22[tag + "_grad_" + n], g20[tag + "_grad_" + n], rtol=1e-4, atol=1e-9)
