"""Host reader (csrc/reader.c) vs golden vectors captured from the reference's own dct_manip.read_coefficients
(tests/golden/g1_reader.npz) and, where oracle/_ref is present, vs the compiled reference reader itself.  CPU only."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import rgb_no_more_amd as rg
from rgb_no_more_amd import dct_manip as dm

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = ["c64x64", "c48x80", "g40x56", "c37x53"]


def test_reader_exports_declared_symbols():
    txt = open(os.path.join(ROOT, "include", "rgbnm_reader.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    syms = sorted(set(re.findall(r"\b(rgbnm_[a-z0-9_]+)\s*\(", txt)))
    dll = ctypes.CDLL(dm.LIB_PATH)
    assert len(syms) == 7          # incl. rgbnm_read_coefficients_batch_crop (round 3)
    for s in syms:
        assert hasattr(dll, s), s
    assert dm.lib().rgbnm_reader_abi_version() == 1


@pytest.mark.parametrize("name", CASES)
def test_read_coefficients_matches_reference_golden(golden, tmp_path, name):
    g = golden("g1_reader.npz")
    data = g[name + "_jpeg"].tobytes()
    path = tmp_path / (name + ".jpg")
    path.write_bytes(data)
    for dim, quant, Y, CbCr in (dm.read_coefficients(str(path)), dm.read_coefficients_bytes(data)):
        assert dim.dtype == torch.int32 and quant.dtype == torch.int16 and Y.dtype == torch.int16
        assert np.array_equal(dim.numpy(), g[name + "_dim"])
        assert np.array_equal(quant.numpy(), g[name + "_quant"])
        assert np.array_equal(Y.numpy(), g[name + "_Y"])
        if name + "_CbCr" in g.files:
            assert np.array_equal(CbCr.numpy(), g[name + "_CbCr"])
        else:
            assert CbCr is None


def test_against_compiled_reference_reader(golden, tmp_path):
    from oracle import build_ref
    ref = build_ref.load_ref()
    if ref is None:
        pytest.skip("oracle/_ref not built in this checkout")
    g = golden("g1_reader.npz")
    for name in CASES:
        path = tmp_path / (name + ".jpg")
        path.write_bytes(g[name + "_jpeg"].tobytes())
        rd, rq, ry, rc = ref.read_coefficients(str(path))
        d, q, y, c = dm.read_coefficients(str(path))
        assert torch.equal(rd, d) and torch.equal(rq, q) and torch.equal(ry, y)
        assert (rc is None and c is None) or torch.equal(rc, c)


def test_error_behaviour(tmp_path):
    with pytest.raises(RuntimeError, match="Unable to open file for reading"):
        dm.read_coefficients(str(tmp_path / "missing.jpg"))
    bad = tmp_path / "bad.jpg"
    bad.write_bytes(b"this is not a jpeg file at all")
    with pytest.raises(dm.libjpeg_exception):
        dm.read_coefficients(str(bad))


def test_batch_reader_threads(golden, tmp_path):
    g = golden("g1_reader.npz")
    paths = []
    for i in range(6):
        p = tmp_path / f"a{i}.jpg"
        p.write_bytes(g["c64x64_jpeg"].tobytes())
        paths.append(str(p))
    Y, C, Q = dm.read_coefficients_batch(paths, threads=3, grid=(8, 8))
    assert Y.shape == (6, 1, 8, 8, 8, 8) and C.shape == (6, 2, 4, 4, 8, 8) and Q.shape == (6, 3, 8, 8)
    for i in range(6):
        assert np.array_equal(Y[i].numpy(), g["c64x64_Y"]) and np.array_equal(C[i].numpy(), g["c64x64_CbCr"])
        assert np.array_equal(Q[i].numpy(), g["c64x64_quant"])
    gp = tmp_path / "gray.jpg"
    gp.write_bytes(g["g40x56_jpeg"].tobytes())
    with pytest.raises(dm.libjpeg_exception):       # different grid -> shape error for that file
        dm.read_coefficients_batch(paths + [str(gp)], threads=2, grid=(8, 8))


def _patch_sof_sampling(data: bytes, comp_index: int, hv: int) -> bytes:
    """Rewrite the H/V sampling byte of one component in the SOF0 header of a baseline JPEG."""
    i = data.index(b"\xff\xc0")
    ncomp = data[i + 9]
    assert comp_index < ncomp
    off = i + 10 + 3 * comp_index + 1
    return data[:off] + bytes([hv]) + data[off + 1:]


def test_mismatched_chroma_grids_are_rejected_not_overrun(golden):
    """A 3-component file whose Cr sampling differs from Cb would be copied past the (2, Hc, Wc) CbCr tensor that is
    sized from Cb alone (the reference has the same flaw, dct_manip.cpp:112-117): the reader must refuse it."""
    g = golden("g1_reader.npz")
    data = g["c64x64_jpeg"].tobytes()
    dm.read_coefficients_bytes(data)                                    # sanity: the unpatched file reads
    bad = _patch_sof_sampling(data, 2, 0x21)                            # Cr: 2x1 instead of 1x1 -> twice Cb's width
    with pytest.raises(RuntimeError, match="different block grids"):
        dm.read_coefficients_bytes(bad)


def test_grayscale_in_batch_gets_zero_chroma(golden, tmp_path):
    g = golden("g1_reader.npz")
    gp = tmp_path / "gray.jpg"
    gp.write_bytes(g["g40x56_jpeg"].tobytes())
    Y, C, Q = dm.read_coefficients_batch([str(gp)] * 3, threads=64, grid=(5, 7))     # more threads than files
    assert np.array_equal(Y[2].numpy(), g["g40x56_Y"]) and not C.any() and (Q[:, 1:] == 1).all()
