"""Table GELU of the fused FeedForwardBlock forward (csrc/mlp_fused.hip; reference: nn.GELU inside FeedForwardBlock,
models/plainvit.py:487-488, on bf16 pre-activations under autocast).

The table is the library's own GELU arithmetic evaluated for all 65 536 bf16 inputs; the kernel reads a compact LDS image of
it and closed forms outside the image's window.  Checked here: (i) the closed forms reproduce the full table for EVERY finite
input except those whose u or u / 2 is a bf16 denormal; (ii) the table is the erf GELU to bf16 rounding; (iii) the fused
forward with the table gives the bits of the fused forward without it (and both equal the two-GEMM path, see
test_fastpath_model.py) -- also for pre-activations pushed far outside the window."""
import ctypes as C

import numpy as np
import pytest
import torch

import rgb_no_more_amd as rg
from rgb_no_more_amd import lib as L

pytestmark = pytest.mark.gpu


def _f(bits):
    return torch.from_numpy((np.asarray(bits, dtype=np.uint32).astype(np.int64) << 16).astype(np.int32)).view(torch.float32)


def test_table_window_and_closed_forms_cover_every_input():
    lib = L.lib()
    L.check(lib.rgbnm_gelu_table_init(L.stream()))
    win = (C.c_int * 16)()
    full = np.zeros(65536, dtype=np.uint32)
    L.check(lib.rgbnm_gelu_table_info(win, full.ctypes.data))
    valid, A0, P1, N1, ndw, gpn = list(win)[:6]
    assert valid == 1 and 4 * ndw <= 13328, (valid, ndw)
    assert 0x100 <= A0 < P1 <= N1 < 0x7F80
    A = np.arange(0x100, A0, dtype=np.uint32)                      # tiny: gelu = u / 2 (exponent - 1), gelu' = 0.5
    assert (full[A] == ((A - 0x80) | (0x3F00 << 16))).all()
    assert (full[0x8000 | A] == ((0x8000 | (A - 0x80)) | (0x3F00 << 16))).all()
    A = np.arange(P1, 0x7F80, dtype=np.uint32)                     # large positive: gelu = u, gelu' = 1
    assert (full[A] == (A | (0x3F80 << 16))).all()
    A = np.arange(N1, 0x7F80, dtype=np.uint32)                     # large negative: gelu = -0, gelu' constant
    assert (full[0x8000 | A] == (0x8000 | (gpn << 16))).all()
    assert (full[np.arange(0x100, 0x7F80)] & 0x8000 == 0).all() and (full[0x8000 | np.arange(0x100, 0x7F80)] & 0x8000 != 0).all()
    # the function itself: erf GELU to bf16 rounding where its value is not negligible
    u = _f(np.arange(65536))
    g, gp = _f(full & 0xFFFF), _f(full >> 16)
    sel = torch.isfinite(u) & (u.abs() > 1e-3) & (u > -3) & (u < 100)
    ref = torch.nn.functional.gelu(u.double())
    refp = 0.5 * (1 + torch.erf(u.double() / 2 ** 0.5)) + u.double() * torch.exp(-0.5 * u.double() ** 2) / (2 * np.pi) ** 0.5
    assert ((g.double() - ref).abs() / ref.abs())[sel].max().item() < 4.2e-3          # half a bf16 ulp = 3.9e-3
    assert (gp.double() - refp).abs()[sel].max().item() < 4.2e-3


@pytest.mark.parametrize("scale", [1.0, 40.0, 1e-4])
def test_fused_forward_with_table_equals_arithmetic(scale):
    """scale 40: most pre-activations beyond the window (|u| >> 16); 1e-4: most of them tiny."""
    import test_fastpath_model as T
    lib = L.lib()
    m, sd, y, c, tgt = T.build("ti_d2_b256" if "ti_d2_b256" in T.CASES else "ti_d12_b256", torch.bfloat16)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if "eb_ffb.0" in n:
                p.mul_(scale)
    m.eval()
    outs = []
    lib.rgbnm_set_option(b"fwd_chain", 0)      # the fused FeedForwardBlock kernel itself (the one-launch forward always uses the table)
    try:
        for opt in (1, 0):
            L.check(lib.rgbnm_set_option(b"gelu_table", opt))
            with torch.no_grad():
                outs.append(m(y, c).clone())
            torch.cuda.synchronize()
    finally:
        lib.rgbnm_set_option(b"gelu_table", 1)
        lib.rgbnm_set_option(b"fwd_chain", 1)
    assert torch.isfinite(outs[0]).all()
    assert torch.equal(outs[0], outs[1])
