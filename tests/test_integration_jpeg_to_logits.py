"""End-to-end chain on real files: 512x512 4:2:0 JPEG -> host reader (librgbnm_reader.so, libjpeg) -> HIP eval transform
(dequantise, centre crop, DCT-domain /2 resize, ToRange) -> HIP ViT (fp32 mode) -> logits, against the oracle chain on the
same files (numpy data path + torch fp32 model).  Tolerance: the north-star 1e-3 on logits; the resize may differ from
the oracle by one LSB on .5 ties (tests/test_augment.py), which moves the logits by ~1e-5."""
import numpy as np
import pytest
import torch

import rgb_no_more_amd as rg
from rgb_no_more_amd import detfill, custom_transforms as CT, dct_manip as dm
from oracle import dct_np as O
from oracle import vit_torch as V

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _write_jpegs(tmp_path, n):
    Image = pytest.importorskip("PIL.Image")
    paths = []
    for i in range(n):
        rng = np.random.default_rng(100 + i)
        small = rng.integers(0, 256, (32, 32, 3), dtype=np.uint8)
        img = Image.fromarray(small).resize((512, 512), Image.BICUBIC)
        arr = np.asarray(img).astype(np.float32) + rng.normal(0, 8, (512, 512, 3))
        p = tmp_path / f"s{i}.jpg"
        Image.fromarray(np.clip(arr, 0, 255).astype(np.uint8)).save(p, quality=90, subsampling=2)
        paths.append(str(p))
    return paths


def test_jpeg_files_to_logits_match_oracle_chain(tmp_path):
    B, emb, heads, depth = 3, 192, 3, 2
    paths = _write_jpegs(tmp_path, B)
    items = [dm.read_coefficients(p) for p in paths]
    for dim, quant, Y, C in items:
        assert tuple(Y.shape) == (1, 64, 64, 8, 8) and tuple(C.shape) == (2, 32, 32, 8, 8)
        assert dim.tolist() == [[512, 512], [256, 256], [256, 256]]
    Yq = torch.stack([it[2] for it in items]).to(DEV)
    Cq = torch.stack([it[3] for it in items]).to(DEV)
    quant = torch.stack([it[1] for it in items]).to(DEV)
    t = CT.EvalTransform_DCT()
    y, c = t(Yq, Cq, quant)
    m = rg.ViT(3, 16, emb, depth=depth, n_classes=1000, drop_p=0.0, device=DEV, num_heads=heads, head_size=64,
               pixel_space="DCT", ver=1, use_subblock=True)
    sd = detfill.fill_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, base_seed=7)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m.compute_dtype = torch.float32
    m.eval()
    with torch.no_grad():
        logits = m(y, c).float().cpu().numpy()
    # oracle chain on the same coefficients
    oy, oc = [], []
    for dim, q, Y, C in items:
        ry, rc = O.eval_transform(Y.numpy(), C.numpy(), q.numpy())
        oy.append(ry)
        oc.append(rc)
    p = {k: torch.from_numpy(v) for k, v in sd.items()}
    ref = V.vit_forward(p, torch.from_numpy(np.stack(oy)), torch.from_numpy(np.stack(oc)), depth, heads, emb).numpy()
    err = np.abs(logits - ref).max()
    print(f"JPEG -> logits: max |dlogit| = {err:.3e}")
    assert err <= 1e-3
    assert (logits.argmax(1) == ref.argmax(1)).all()
