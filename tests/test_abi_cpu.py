"""CPU-only checks of the boundary: the C-ABI library loads and exports every symbol include/rgbnm.h declares,
the drop-in module keeps the reference's state_dict surface, and the product path fails loudly (no fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import rgb_no_more_amd as rg
from rgb_no_more_amd import lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "rgbnm.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(rgbnm_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    syms = _declared_symbols()
    assert len(syms) >= 25
    dll = ctypes.CDLL(L.LIB_PATH)
    for s in syms:
        assert hasattr(dll, s), f"librgbnm.so does not export {s}"
    # and the python prototypes cover exactly the header
    assert sorted(L.PROTOTYPES) == syms
    assert L.lib().rgbnm_abi_version() == L.ABI_VERSION == 3
    assert b"workspace" in L.lib().rgbnm_strerror(-3)


def test_state_dict_surface_matches_reference(golden):
    g = golden("g11_model.npz")
    m = rg.ViT(3, 16, 192, depth=12, n_classes=1000, drop_p=0.0, num_heads=3, head_size=64, pixel_space="DCT", ver=1)
    assert list(m.state_dict().keys()) == [str(s) for s in g["ti_d12_names"]]
    assert len(m.state_dict()) == 152 and sum(p.numel() for p in m.parameters()) == 5642728
    ms = rg.ViT(3, 16, 384, depth=12, n_classes=1000, drop_p=0.0, num_heads=6, head_size=64, pixel_space="DCT", ver=1)
    assert sum(p.numel() for p in ms.parameters()) == 21975016
    # the weight-decay name filter of pipeline_utils.py:537 selects the Linear weights only
    sel = [n for n, _ in m.named_parameters() if (".weight" in n) and ("lrnorm" not in n)]
    assert len(sel) == 1 + 12 * 4 + 2
    # ver=2 (embed_type 2): the reference's PatchEmbedding_DCT_Separate_subblock keys and shapes (golden g14)
    g2 = golden("g14_model_v2.npz")
    m2 = rg.ViT(3, 16, 192, depth=2, n_classes=1000, drop_p=0.0, num_heads=3, head_size=64, pixel_space="DCT", ver=2)
    assert list(m2.state_dict().keys()) == [str(s) for s in g2["ti_d2_v2_names"]]
    assert [str(tuple(v.shape)) for v in m2.state_dict().values()] == [str(s) for s in g2["ti_d2_v2_shapes"]]
    # ver=2 without sub-block conversion: PatchEmbedding_DCT_Separate keys, LinearMix registered twice like the reference
    g3 = golden("g19_model_v2ns.npz")
    m3 = rg.ViT(3, 16, 192, depth=2, n_classes=1000, drop_p=0.0, num_heads=3, head_size=64, pixel_space="DCT", ver=2,
                use_subblock=False)
    assert list(m3.state_dict().keys()) == [str(s) for s in g3["ti_d2_v2ns_names"]]
    assert [str(tuple(v.shape)) for v in m3.state_dict().values()] == [str(s) for s in g3["ti_d2_v2ns_shapes"]]
    assert m3.state_dict()["patchembed.projection.1.weight"].data_ptr() == m3.state_dict()["patchembed.LinearMix.weight"].data_ptr()
    # ver=3 (embed_type 3): PatchEmbedding_DCT_Concat keys; 294 tokens
    g4 = golden("g18_model_v3.npz")
    m4 = rg.ViT(3, 16, 192, depth=2, n_classes=1000, drop_p=0.0, num_heads=3, head_size=64, pixel_space="DCT", ver=3)
    assert list(m4.state_dict().keys()) == [str(s) for s in g4["ti_d2_v3_names"]] and m4.n_tokens == 294
    with pytest.raises(NotImplementedError):
        rg.ViT(3, 16, 192, depth=1, drop_p=0.0, num_heads=3, pixel_space="DCT", ver=4)


def test_no_cpu_fallback():
    m = rg.ViT(3, 16, 192, depth=1, n_classes=16, drop_p=0.0, num_heads=3, head_size=64, pixel_space="DCT", ver=1)
    # any class count, as the reference's ctor (padded to 16-byte rows inside; parameters keep the real shapes)
    m10 = rg.ViT(3, 16, 192, depth=1, n_classes=10, drop_p=0.0, num_heads=3, head_size=64, pixel_space="DCT", ver=1)
    assert tuple(m10.state_dict()["classhead.ch_linear2.weight"].shape) == (10, 192)
    y = torch.zeros(1, 1, 28, 28, 8, 8)
    c = torch.zeros(1, 2, 14, 14, 8, 8)
    with pytest.raises(L.RgbnmError):
        m(y, c)
    with pytest.raises(NotImplementedError):
        rg.ViT(3, 16, 192, depth=1, pixel_space="RGB", drop_p=0.0, num_heads=3)
    with pytest.raises(L.RgbnmError):
        rg.cls_transforms.cross_entropy(torch.zeros(2, 10), torch.zeros(2, dtype=torch.int64))


def test_conversion_matrix_matches_reference_golden(golden):
    g = golden("g3_convmat.npz")
    for ls, mlt in [(8, 2), (4, 2), (2, 4), (8, 1)]:
        A = rg.dct_ops.generate_conversion_matrix(ls, mlt).numpy()
        np.testing.assert_allclose(A, g[f"A_{ls}_{mlt}"], atol=1e-6)
    t = rg.plainvit.sincos_table(14, 14, 192)
    assert np.array_equal(t.numpy(), golden("g10_sincos.npz")["t192"])
