"""The data path bench.py times -- `FastParamSampler.sample -> apply_packed` on S-coef batches (bench.py:data_part) -- against
the numpy oracle at BASELINE config 2's full batch (256 images, the real mix of crop sides 14 / 28 / 56 and of op pairs).

Reference chain: datasets.py:286-293 (de-quantise, clamp), :354-361 (RandomResizedCrop_DCT, RandomFlip_DCT, RandAugment_dct,
ToRange), utils/custom_transforms.py:557-629, 1095-1127.

Three statements, every image of the batch:
  (1) the packed parameter array the sampler emits is byte-identical to `TrainTransform_DCT.pack` of the same logical
      (box, flip, [(op, magnitude, aux)]) parameters -- op ids, fmag, Cutout centre, ChromaDrop coin, filter-bank index;
  (2) crop + resize + flip + entry clamp (kernel 1): bit exact for side 28, <= 1 LSB for the fp32 resizes (side 14 / 56);
  (3) the op chain + ToRange (kernel 2) applied by the oracle to kernel 1's OWN int16 output equals the packed path's final
      output bit for bit -- so a 1-LSB resize difference cannot hide (or be blamed for) anything in the op chain -- and the
      end-to-end oracle output is bit exact wherever the resize agreed.
"""
import os
import sys

import numpy as np
import pytest
import torch

import rgb_no_more_amd as rg   # noqa: F401
from rgb_no_more_amd import custom_transforms as CT
from oracle import dct_np as O

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

pytestmark = pytest.mark.gpu
DEV = "cuda"
INV_OPS = {v: k for k, v in CT.OPS.items()}


def decode_packed(t, packed, nops):
    """AUG_DTYPE rows -> the logical parameters the reference would have drawn: op NAME from the id, magnitude from the
    reference's magnitude table at the transform's bin with the sign whose encoding matches the row."""
    meta = CT.magnitude_table(t.num_magnitude_bins, (t.size, t.size))
    out = []
    for row in packed:
        ops = []
        for s in range(nops):
            name = INV_OPS[int(row["op"][s])]
            mags, signed = meta[name]
            mag = float(mags[t.magnitude].item()) if mags.ndim > 0 else float(mags.item())
            aux = None
            if name == "Cutout":
                aux = (int(row["iarg1"][s]), int(row["iarg2"][s]))
            elif name == "ChromaDrop":
                aux = bool(row["iarg0"][s])
            got = None
            for sg in ((1.0, -1.0) if signed else (1.0,)):
                e = CT.encode_op(name, mag * sg, aux, t.bank, t.size)
                if (e[0], np.float32(e[1]), e[2]) == (int(row["op"][s]), np.float32(row["fmag"][s]), int(row["iarg0"][s])):
                    got = (name, mag * sg, aux)
                    break
            assert got is not None, (name, row)
            ops.append(got)
        out.append(dict(box=tuple(int(v) for v in row["crop"]), flip=bool(row["flip"]), ops=ops))
    return out


@pytest.mark.parametrize("size,grid", [(28, 64)])
def test_packed_bench_datapath_matches_oracle_at_full_batch(size, grid):
    import bench
    B = 256
    Yq, Cq, quant = bench.synth_coefficients(B, DEV, 1234)
    aug = CT.TrainTransform_DCT(size=size, out_dtype=torch.float32)
    sampler = CT.FastParamSampler(aug, seed=1234)
    for _ in range(3):                                   # a few draws in: not the first batch of the generator
        packed, nops = sampler.sample(B, grid, grid)
    assert nops == 2
    params = decode_packed(aug, packed, nops)

    # (1) the packed bytes are the encoding of the logical parameters
    arr, n2 = aug.pack(params)
    assert n2 == nops and bytes(arr) == np.ascontiguousarray(packed).tobytes()
    sides = packed["crop"][:, 2]
    assert set(np.unique(sides)) == {size // 2, size, 2 * size}, np.unique(sides)     # the real mix, not one branch
    names = {o[0] for p in params for o in p["ops"]}
    assert len(names) >= 12, names

    oy, oc = CT.apply_packed(aug, Yq, Cq, quant, packed, nops)
    aug.out_dtype = torch.bfloat16
    by, bc = CT.apply_packed(aug, Yq, Cq, quant, packed, nops)
    aug.out_dtype = torch.float32
    # kernel 1 alone: raw int16 after crop / resize / flip / entry clamp
    t16 = CT.TrainTransform_DCT(size=size, out_dtype=torch.int16)
    sy, sc = t16(Yq, Cq, quant, params=[dict(box=p["box"], flip=p["flip"], ops=[]) for p in params])
    torch.cuda.synchronize()
    assert torch.equal(by, oy.to(torch.bfloat16)) and torch.equal(bc, oc.to(torch.bfloat16))   # bench's dtype: same values, rounded
    oy, oc, sy, sc = oy.cpu().numpy(), oc.cpu().numpy(), sy.cpu().numpy(), sc.cpu().numpy()
    Yh, Ch, qh = Yq.cpu().numpy(), Cq.cpu().numpy(), quant.cpu().numpy()

    n_exact = n_lsb = 0
    worst = 0
    for b, p in enumerate(params):
        i, j, h, w = p["box"]
        Y, Cc = O.dequantize(Yh[b], Ch[b], qh[b])
        Y = O.resize(O.crop(Y, i, j, h, w), size)
        Cc = O.resize(O.crop(Cc, i // 2, j // 2, max(1, h // 2), max(1, w // 2)), size // 2)
        if p["flip"]:
            Y, Cc = O.flip(Y), O.flip(Cc)
        Y, Cc = np.clip(Y, O.CMIN, O.CMAX), np.clip(Cc, O.CMIN, O.CMAX)
        # (2) kernel 1 vs the oracle
        dy, dc = np.abs(sy[b].astype(np.int32) - Y), np.abs(sc[b].astype(np.int32) - Cc)
        if h == size:
            assert dy.max() == 0 and dc.max() == 0, (b, p["box"])
        else:
            assert dy.max() <= 1 and dc.max() <= 1, (b, p["box"], dy.max(), dc.max())
            assert (dy > 0).mean() < 0.06 and (dc > 0).mean() < 0.06, (b, p["box"])
        worst = max(worst, int(dy.max()), int(dc.max()))
        # (3) oracle op chain + ToRange on kernel 1's own output == the packed path's output, bit for bit
        y1, c1 = sy[b].astype(Y.dtype), sc[b].astype(Cc.dtype)
        for name, mag, aux in p["ops"]:
            y1, c1 = O.apply_op(y1, c1, name, mag, aux)
        assert np.array_equal(O.to_range(y1), oy[b]) and np.array_equal(O.to_range(c1), oc[b]), (b, p)
        # ... and the end-to-end oracle agrees exactly wherever the resize did
        if dy.max() == 0 and dc.max() == 0:
            ry, rc = O.train_transform(Yh[b], Ch[b], qh[b], p["box"], p["flip"], p["ops"], size)
            assert np.array_equal(ry, oy[b]) and np.array_equal(rc, oc[b]), (b, p)
            n_exact += 1
        else:
            n_lsb += 1
    print(f"B={B}: {n_exact} images bit exact end to end, {n_lsb} with 1-LSB resize ties (max |d| = {worst} LSB); "
          f"sides {dict(zip(*np.unique(sides, return_counts=True)))}")
    assert n_exact >= (sides == size).sum()
