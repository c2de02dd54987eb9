"""Size-independent properties at BASELINE config 2's full sizes (per-GPU batch 256: 256 x 512x512 coefficient images,
50 176 tokens), where the CPU oracle would take minutes: involutions / round trips of the DCT-domain index ops (bit
exact), linearity of the GEMM family that carries the model, permutation equivariance of attention, and agreement of
every fast kernel family with the plain tile-per-workgroup kernels on the same full-size inputs."""
import numpy as np
import pytest
import torch

from rgb_no_more_amd import custom_transforms as CT, lib as L
from test_augment import synth
from test_hip_kernels import gemm_nt, relerr

pytestmark = pytest.mark.gpu
DEV = "cuda"
B = 256


def _aug(Y, Cc, quant, params):
    t = CT.TrainTransform_DCT(out_dtype=torch.float32)
    oy, oc = t(Y, Cc, quant, params=params)
    return oy, oc


def _inputs():
    Y, Cc, quant = synth(8, seed=3)
    rep = B // 8
    to = lambda a: torch.from_numpy(np.tile(a, (rep,) + (1,) * (a.ndim - 1))).to(DEV)   # noqa: E731
    return to(Y), to(Cc), to(quant)


def _params(ops, flip=False, box=(4, 6, 28, 28)):
    return [dict(box=box, flip=flip, ops=ops) for _ in range(B)]


def test_index_ops_are_involutions_at_batch_256():
    Y, Cc, quant = _inputs()
    base = _aug(Y, Cc, quant, _params([("Identity", 0.0, None), ("Identity", 0.0, None)]))
    # Rotate90(+1) then Rotate90(-1) is the identity, bit for bit (dct_ops.py:99-130)
    rr = _aug(Y, Cc, quant, _params([("Rotate90", 1.0, None), ("Rotate90", -1.0, None)]))
    assert torch.equal(rr[0], base[0]) and torch.equal(rr[1], base[1])
    # a flip is its own inverse: flipping the flipped crop == reversing blocks and negating odd columns again
    f = _aug(Y, Cc, quant, _params([("Identity", 0.0, None), ("Identity", 0.0, None)], flip=True))
    for a, b in zip(f, base):
        # compare in the integer coefficient domain (ToRange is y = (x + 1024) / 1020 - 1) to stay bit exact
        ia = torch.round((a.double() + 1.0) * 1020.0 - 1024.0)
        ib = torch.round((b.double() + 1.0) * 1020.0 - 1024.0)
        ia = ia.flip(3)                               # reverse the block columns again ...
        ia[..., 1::2] = -ia[..., 1::2]                # ... and re-negate the odd horizontal frequencies
        assert torch.equal(ia, ib)
    # translating by +2 blocks then -4... is not an inverse pair (reference quirk); Grayscale is idempotent instead
    g1 = _aug(Y, Cc, quant, _params([("Grayscale", 0.0, None), ("Identity", 0.0, None)]))
    g2 = _aug(Y, Cc, quant, _params([("Grayscale", 0.0, None), ("Grayscale", 0.0, None)]))
    assert torch.equal(g1[0], g2[0]) and torch.equal(g1[1], g2[1])
    assert torch.equal(g1[0], base[0])                # luma untouched, chroma at the value ToRange gives to 0
    assert float(g1[1].abs().max()) == pytest.approx(1024.0 / 1020.0 - 1.0, abs=1e-6)


def test_images_in_a_batch_are_independent_checksum_of_checksums():
    """The batch is 32 copies of 8 distinct images with identical parameters: every copy must produce the same bits
    (no cross-image leakage through LDS / workspace), and the checksum of per-image checksums must match 32 x."""
    Y, Cc, quant = _inputs()
    oy, oc = _aug(Y, Cc, quant, _params([("Contrast", 0.27, None), ("MidfreqAug", -0.27, None)], flip=True))
    oy8, oc8 = oy.view(B // 8, 8, -1), oc.view(B // 8, 8, -1)
    assert torch.equal(oy8, oy8[:1].expand_as(oy8)) and torch.equal(oc8, oc8[:1].expand_as(oc8))
    per = torch.round((oy.double() + 1.0) * 1020.0).view(B, -1).sum(1)
    assert per.sum().item() == per[:8].sum().item() * (B // 8)


@pytest.mark.parametrize("N,K,epi", [(576, 192, "none"), (768, 192, "none"), (192, 768, "res"), (192, 576, "none")])
def test_gemm_family_linearity_and_agreement_at_50176_tokens(N, K, epi):
    """C(a X1 + X2) = a C(X1) + C(X2) up to bf16 rounding, on the shapes of the model at B = 256 (wres / kpipe paths),
    and fast path == plain path bit for bit on the same inputs."""
    M, dt = 50176, torch.bfloat16
    g = torch.Generator(device=DEV)
    g.manual_seed(5)
    X1 = torch.randn(M, K, device=DEV, generator=g).to(dt)
    X2 = torch.randn(M, K, device=DEV, generator=g).to(dt)
    W = (torch.randn(N, K, device=DEV, generator=g) * 0.05).to(dt)
    R = torch.randn(M, N, device=DEV, generator=g).to(dt) if epi == "res" else None
    code = L.EPI_RES if epi == "res" else L.EPI_NONE
    lib = L.lib()
    c1 = gemm_nt(dt, L.EPI_NONE, X1, W)[0].float()
    c2 = gemm_nt(dt, L.EPI_NONE, X2, W)[0].float()
    c12 = gemm_nt(dt, L.EPI_NONE, (2.0 * X1.float() + X2.float()).to(dt), W)[0].float()
    assert relerr(c12, 2.0 * c1 + c2) < 1.5e-2            # three bf16 roundings of operands / results
    saved = {k: lib.rgbnm_get_option(k) for k in (b"nt_wres", b"nt_kpipe")}
    try:
        fast = gemm_nt(dt, code, X1, W, None, R=R)[0]
        for k in saved:
            lib.rgbnm_set_option(k, 0)
        plain = gemm_nt(dt, code, X1, W, None, R=R)[0]
    finally:
        for k, v in saved.items():
            lib.rgbnm_set_option(k, v)
    assert torch.equal(fast, plain)


def test_attention_is_equivariant_to_token_permutation_at_batch_256():
    """softmax(QK^T)V commutes with a permutation of the tokens (no positional term inside attention): permuting the
    tokens of qkv permutes the output rows, forward and backward, up to bf16 rounding of different summation orders."""
    Bn, N, H = 256, 196, 3
    I = H * 64
    g = torch.Generator(device=DEV)
    g.manual_seed(6)
    qkv = torch.randn(Bn, N, 3 * I, device=DEV, generator=g).bfloat16()
    dout = torch.randn(Bn, N, I, device=DEV, generator=g).bfloat16()
    perm = torch.randperm(N, device=DEV, generator=g)
    scale = 1.0 / (192 ** 0.5)
    lib = L.lib()

    def run(q, do):
        out = torch.empty(Bn, N, I, device=DEV, dtype=torch.bfloat16)
        lse = torch.empty(Bn * H * N, device=DEV)
        dq = torch.empty_like(q)
        L.check(lib.rgbnm_attention_fwd(1, q.data_ptr(), out.data_ptr(), lse.data_ptr(), Bn, N, H, scale, L.stream()))
        L.check(lib.rgbnm_attention_bwd(1, q.data_ptr(), out.data_ptr(), do.data_ptr(), lse.data_ptr(), dq.data_ptr(),
                                        Bn, N, H, scale, L.stream()))
        return out, dq

    o1, d1 = run(qkv, dout)
    o2, d2 = run(qkv[:, perm].contiguous(), dout[:, perm].contiguous())
    assert relerr(o2, o1[:, perm]) < 1e-2
    assert relerr(d2, d1[:, perm]) < 2e-2
    # rows of the attention output are convex combinations of V rows: bounded by the per-head extreme of V
    v = qkv[:, :, 2 * I:].float().view(Bn, N, H, 64)
    o = o1.float().view(Bn, N, H, 64)
    assert bool((o <= v.amax(1, keepdim=True) + 2e-2).all()) and bool((o >= v.amin(1, keepdim=True) - 2e-2).all())
