"""Row a3 on the PRODUCT side: `custom_transforms.RandomResizedCrop_DCT.get_params` and the vectorised
`FastParamSampler` (the sampler bench.py times) against golden G8 -- the (u, ri, rj) -> (i, j, h, w) table captured from
the reference's `RandomResizedCrop_DCT.get_params` (utils/custom_transforms.py:557-629) by tests/golden/make_golden.py,
plus the central-crop fallback rows of make_golden_r2.py.  CPU only (parameter sampling is host logic)."""
import numpy as np
import pytest
import torch

import rgb_no_more_amd as rg
from rgb_no_more_amd import custom_transforms as CT


def test_get_params_replays_reference_seeds(golden):
    """Same torch RNG stream as the reference: seed -> identical (i, j, h, w), including rows whose first attempt
    did not fit (odd grids: the 10-attempt loop draws again)."""
    g = golden("g8_params.npz")
    n = 0
    for size, H, W, seed, u, ri, rj, i, j, h, w, first_ok in g["rrc"]:
        rrc = CT.RandomResizedCrop_DCT(int(size), scale=(0.05, 1.0), ratio=(1, 1))
        torch.manual_seed(int(seed))
        assert rrc.get_params(int(H), int(W)) == (int(i), int(j), int(h), int(w)), (size, H, W, seed)
        n += 1
    assert n == 200


def test_fast_sampler_boxes_from_golden_draws(golden):
    """FastParamSampler's vectorised snapping, fed the golden draws: every first-attempt row must give the golden box;
    rows whose first attempt did not fit must be flagged for a re-draw."""
    g = golden("g8_params.npz")
    rows = g["rrc"]
    for size in (28, 32):
        t = CT.TrainTransform_DCT(size=size)
        fs = CT.FastParamSampler(t, seed=0)
        for H, W in sorted({(int(r[1]), int(r[2])) for r in rows if int(r[0]) == size}):
            sel = rows[(rows[:, 0] == size) & (rows[:, 1] == H) & (rows[:, 2] == W)]
            box = fs.boxes_from_draws(sel[:, 4], sel[:, 5].astype(np.int64), sel[:, 6].astype(np.int64), H, W)
            ok = sel[:, 11] == 1
            assert np.array_equal(box[ok], sel[ok][:, 7:11].astype(np.int64)), (size, H, W)
            # first_ok == 0: either the first box did not fit (re-draw) or the replayed offsets are of a later attempt;
            # the side must still be the reference's snapping of u whenever it fits
            sides = fs.sides_from_draws(sel[:, 4], H, W)
            fit = (sides <= H) & (sides <= W)
            assert np.all(box[~fit] == -1)
    assert (rows[:, 11] == 1).sum() > 150


def test_fallback_box_matches_reference(golden):
    g = golden("g8b_fallback.npz")
    for size, H, W, i, j, h, w in g["rows"]:
        rrc = CT.RandomResizedCrop_DCT(int(size), scale=(float(g["scale"][0]), float(g["scale"][1])), ratio=(1, 1))
        torch.manual_seed(0)
        assert rrc.get_params(int(H), int(W)) == (int(i), int(j), int(h), int(w)), (size, H, W)
        assert rrc.fallback_box(int(H), int(W)) == (int(i), int(j), int(h), int(w))


def test_fast_sampler_redraws_and_falls_back():
    """Grids where the snapped box often / never fits: every sampled box lies inside the grid, has an allowed side and
    even offsets; a grid where no box ever fits returns the reference's central crop."""
    t = CT.TrainTransform_DCT(size=28)
    fs = CT.FastParamSampler(t, seed=3)
    box = fs.sample_boxes(4000, 47, 63)
    fb = np.asarray(t.rrc.fallback_box(47, 63))         # (8, 2, 28, 56): P(10 misses) = P(side 56)^10, about 1 %
    is_fb = np.all(box == fb, axis=1)
    assert 0 < is_fb.sum() < 200
    i, j, h, w = box[~is_fb].T
    assert np.all(h == w) and np.all((i % 2 == 0) & (j % 2 == 0))
    assert np.all((i >= 0) & (j >= 0) & (i + h <= 47) & (j + w <= 63))
    assert set(np.unique(w)) <= {2, 4, 14, 28}          # 56 never fits a 47-row grid
    t2 = CT.TrainTransform_DCT(size=28, scale=(0.9, 1.0))
    fs2 = CT.FastParamSampler(t2, seed=3)
    b2 = fs2.sample_boxes(8, 20, 100)
    assert np.all(b2 == np.asarray(t2.rrc.fallback_box(20, 100)))


def test_fast_sampler_matches_scalar_distribution():
    """The two product samplers draw from the same distribution (chi-square on box sides and op ids)."""
    t = CT.TrainTransform_DCT(size=28)
    fs = CT.FastParamSampler(t, seed=11)
    packed, nops = fs.sample(20000, 64, 64)
    assert nops == 2
    sides, cnt = np.unique(packed["crop"][:, 2], return_counts=True)
    assert list(sides) == [14, 28, 56]
    p = cnt / cnt.sum()
    # analytic from w = round(64 sqrt(u)), u ~ U(0.05, 1): side 14 iff w <= 21, 28 iff w <= 41 (42 / 28 = 1.5 rounds to
    # even = 2): 6.6 % / 32.4 % / 61.0 %; 40 000 draws of the reference's get_params gave 6.8 / 32.0 / 61.2 %
    assert np.allclose(p, [0.0662, 0.3239, 0.6099], atol=0.012), p
    torch.manual_seed(5)
    ref = t.sample_params(3000, 64, 64)
    pr = np.bincount([{14: 0, 28: 1, 56: 2}[d["box"][2]] for d in ref], minlength=3) / 3000
    assert np.allclose(p, pr, atol=0.03)


def test_op_exclusion_rules_exact():
    """custom_transforms.py:1111-1119: after Grayscale no chroma op; after a chroma op no Grayscale; otherwise the full
    list.  Checked structurally (candidate tables) and on every sampled pair."""
    t = CT.TrainTransform_DCT(size=28)
    fs = CT.FastParamSampler(t, seed=2)
    names = fs.names
    chroma = {"Grayscale", "Color", "AutoSaturation", "ChromaDrop"}
    assert [names[k] for k in fs.cand_after_gray] == [n for n in names if n not in chroma]
    assert [names[k] for k in fs.cand_after_chroma] == [n for n in names if n != "Grayscale"]
    packed, _ = fs.sample(50000, 64, 64)
    inv = {v: k for k, v in CT.OPS.items()}
    a = np.array([inv[i] for i in packed["op"][:, 0]])
    b = np.array([inv[i] for i in packed["op"][:, 1]])
    gray_first = a == "Grayscale"
    assert gray_first.any() and not np.isin(b[gray_first], list(chroma)).any()
    chroma_first = np.isin(a, list(chroma - {"Grayscale"}))
    assert chroma_first.any() and not (b[chroma_first] == "Grayscale").any()
    # every allowed ordered pair occurs
    seen = set(zip(a.tolist(), b.tolist()))
    for x in names:
        allowed = ([n for n in names if n not in chroma] if x == "Grayscale" else
                   [n for n in names if n != "Grayscale"] if x in chroma else names)
        for y in allowed:
            assert (x, y) in seen, (x, y)
    # the scalar sampler obeys the same rules
    torch.manual_seed(0)
    for d in t.sample_params(4000, 64, 64):
        (n0, _, _), (n1, _, _) = d["ops"]
        assert not (n0 == "Grayscale" and n1 in chroma)
        assert not (n0 in chroma and n0 != "Grayscale" and n1 == "Grayscale")


def test_encoded_magnitudes_are_the_reference_quirks():
    """SURVEY Appendix A.6 on the product encoder (`encode_op`)."""
    t = CT.TrainTransform_DCT(size=28)
    meta = CT.magnitude_table(11, (28, 28))
    m = float(meta["TranslateX"][0][3])
    assert CT.encode_op("TranslateX", m, None, t.bank)[2] == 2
    assert CT.encode_op("TranslateX", -m, None, t.bank)[2] == -4
    assert CT.encode_op("Cutout", float(meta["Cutout"][0][3]), (4, 6), t.bank)[2:] == (2, 4, 6)
    assert CT.encode_op("Posterize", float(meta["Posterize"][0][3]), None, t.bank)[2:4] == (2, 511)
    assert CT.encode_op("SolarizeAdd", float(meta["SolarizeAdd"][0][3]), None, t.bank)[2] == 264
    with pytest.raises(ValueError):
        CT.encode_op("NoSuchOp", 0.0, None, t.bank)
    assert rg.custom_transforms is CT
