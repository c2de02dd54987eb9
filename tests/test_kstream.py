"""Streaming row-panel GEMM (csrc/gemm_nt_kstream.hip, option nt_kstream; reference: nn.Linear [+ nn.GELU],
models/plainvit.py:467-491): the opt-in kernel gives the BITS of the row-panel kernels it stands in for (gemm_nt_kpipe.hip, which the
model-level tests pin against the reference) -- plain and fc1 + GELU epilogues (gelu and gelu', table and arithmetic form), full and
ragged last row panel, one and several k-tile sweeps per unit."""
import pytest
import torch

from rgb_no_more_amd import lib as L

pytestmark = pytest.mark.gpu


def _run(opt, epi, A, W, b, M, N, K, table=1):
    lib = L.lib()
    L.check(lib.rgbnm_set_option(b"nt_kstream", opt))
    L.check(lib.rgbnm_set_option(b"gelu_table", table))
    try:
        Cc = torch.full((M, N), 7.0, device="cuda", dtype=torch.bfloat16)
        C2 = torch.full((M, N), 7.0, device="cuda", dtype=torch.bfloat16)
        L.check(lib.rgbnm_gemm_nt(1, epi, A.data_ptr(), K, W.data_ptr(), K, Cc.data_ptr(), N, b.data_ptr(), None, 0,
                                  C2.data_ptr(), N, None, 0, M, N, K, 0, L.stream()))
        torch.cuda.synchronize()
        return Cc, C2
    finally:
        L.check(lib.rgbnm_set_option(b"nt_kstream", 0))
        L.check(lib.rgbnm_set_option(b"gelu_table", 1))


@pytest.mark.parametrize("M", [8960, 8300])                      # 40 full panels of 224 rows / a ragged last one
@pytest.mark.parametrize("N,K,epi", [(384, 384, 0), (1152, 384, 0), (384, 1536, 0), (1536, 384, 2)])
def test_streaming_kernel_gives_the_bits_of_the_row_panel_kernels(M, N, K, epi):
    L.check(L.lib().rgbnm_gelu_table_init(L.stream()))
    torch.manual_seed(M + N + K + epi)
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    W = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
    b = torch.randn(N, device="cuda")
    ref = _run(0, epi, A, W, b, M, N, K)
    got = _run(2, epi, A, W, b, M, N, K)
    assert torch.equal(ref[0], got[0])
    if epi == 2:
        assert torch.equal(ref[1], got[1])
        got_arith = _run(2, epi, A, W, b, M, N, K, table=0)      # the arithmetic GELU form of the same kernel
        assert torch.equal(ref[0], got_arith[0]) and torch.equal(ref[1], got_arith[1])
    else:
        assert (got[1] == 7.0).all()                             # C2 untouched by the plain epilogue
    assert torch.isfinite(got[0].float()).all()


@pytest.mark.parametrize("M", [50176, 8300, 65536])      # persistent 7-wave kernel (full / ragged panels), 8-wave kernel (resident image)
def test_row_panel_gelu_epilogue_table_form_gives_the_bits_of_the_arithmetic_form(M):
    """fc1 + GELU through the persistent row-panel kernel (gemm_nt_kpipe_body.inc): the table lookup of its staging pass (image
    reloaded behind every k-loop) against the erf arithmetic -- gelu and gelu' bit for bit."""
    L.check(L.lib().rgbnm_gelu_table_init(L.stream()))
    N, K = 1536, 384
    torch.manual_seed(M)
    A = (torch.randn(M, K, device="cuda") * 2.0).to(torch.bfloat16)      # pre-activations well outside +-4 too
    W = (torch.randn(N, K, device="cuda") * 0.08).to(torch.bfloat16)
    b = torch.randn(N, device="cuda")
    arith = _run(0, 2, A, W, b, M, N, K, table=0)
    tab = _run(0, 2, A, W, b, M, N, K, table=1)
    assert torch.equal(arith[0], tab[0]) and torch.equal(arith[1], tab[1])
    assert float(arith[0].float().abs().max()) > 4.0
