"""GPU parity: every C-ABI kernel of librgbnm.so vs the oracle (oracle/*.py, torch fp32 / numpy) on the same
seeded inputs.  fp32 ("strict") mode carries the tight tolerances; bf16 tolerances are stated per test."""
import ctypes as C
import math

import numpy as np
import pytest
import torch

import rgb_no_more_amd as rg
from rgb_no_more_amd import detfill, lib as L
from oracle import vit_torch as V
from oracle import dct_np as O

pytestmark = pytest.mark.gpu

DEV = "cuda"
DTS = [torch.float32, torch.bfloat16]


def dev(a, dt=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV).to(dt).contiguous()


def sync():
    torch.cuda.synchronize()


def tol(dt, f32, bf16):
    return f32 if dt == torch.float32 else bf16


def relerr(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def gemm_nt(dt, epi, A, W, bias=None, R=None, pos=None, period=0, c_f32=False):
    M, K = A.shape
    N = W.shape[0]
    Cc = torch.empty(M, N, device=DEV, dtype=torch.float32 if c_f32 else dt)
    C2 = torch.empty(M, N, device=DEV, dtype=dt) if epi == L.EPI_GELU else None
    L.check(L.lib().rgbnm_gemm_nt(L.dt_of(dt), epi, A.data_ptr(), K, W.data_ptr(), K, Cc.data_ptr(), N, L.ptr(bias),
                                  L.ptr(R), N, L.ptr(C2), N, L.ptr(pos), period, M, N, K, int(c_f32), L.stream()))
    return Cc, C2


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("M,N,K", [(256, 192, 192), (1568, 576, 192), (200, 1000, 192), (392, 192, 768),
                                   (130, 192, 384), (64, 192, 1000), (300, 768, 192)])
def test_gemm_nt_bias(dt, M, N, K):
    A = dev(detfill.normalish((M, K), 1), dt)
    W = dev(detfill.uniform((N, K), 2, -0.1, 0.1), dt)
    b = dev(detfill.uniform((N,), 3))
    out, _ = gemm_nt(dt, L.EPI_NONE, A, W, b)
    ref = A.float() @ W.float().T + b
    sync()
    e = relerr(out, ref)
    assert e < tol(dt, 2e-6, 4e-3), e
    out32, _ = gemm_nt(dt, L.EPI_NONE, A, W, b, c_f32=True)
    assert out32.dtype == torch.float32 and relerr(out32, ref) < tol(dt, 2e-6, 1e-5 if dt == torch.float32 else 2e-3)


def test_gemm_nt_transpose_detecting():
    # asymmetric operands, A = selector: catches row/col swaps of the accumulator layout
    M, N, K = 128, 192, 64
    A = torch.zeros(M, K, device=DEV)
    for i in range(M):
        A[i, (i * 7) % K] = 1.0
    W = dev(detfill.uniform((N, K), 5))
    out, _ = gemm_nt(torch.float32, L.EPI_NONE, A, W)
    ref = A @ W.T
    assert torch.equal(out, ref)


@pytest.mark.parametrize("dt", DTS)
def test_gemm_nt_epilogues(dt):
    M, N, K = 392, 192, 192
    A = dev(detfill.normalish((M, K), 11), dt)
    W = dev(detfill.uniform((N, K), 12, -0.1, 0.1), dt)
    b = dev(detfill.uniform((N,), 13))
    R = dev(detfill.normalish((M, N), 14), dt)
    base = A.float() @ W.float().T
    t_hi = tol(dt, 5e-6, 6e-3)
    out, _ = gemm_nt(dt, L.EPI_RES, A, W, b, R=R)
    assert relerr(out, base + b + R.float()) < t_hi
    out, dg = gemm_nt(dt, L.EPI_GELU, A, W, b)
    u = (base + b).to(dt).float().requires_grad_(True)      # pre-activation at the activation dtype's precision
    gref = torch.nn.functional.gelu(u)
    gref.sum().backward()
    assert relerr(out, gref) < tol(dt, 2e-6, 4e-3)
    assert relerr(dg, u.grad) < tol(dt, 2e-6, 4e-3)          # C2 = gelu'(u), consumed by EPI_DGELU
    pos = dev(detfill.uniform((196, N), 15))
    out, _ = gemm_nt(dt, L.EPI_POS, A, W, b, pos=pos, period=196)
    rows = torch.arange(M, device=DEV) % 196
    assert relerr(out, base + b + pos[rows]) < t_hi
    out, _ = gemm_nt(dt, L.EPI_DGELU, A, W, None, R=R)
    assert relerr(out, base * R.float()) < t_hi
    out, _ = gemm_nt(dt, L.EPI_TANH, A, W, b)
    assert relerr(out, torch.tanh(base + b)) < t_hi
    h = torch.tanh(R.float()).to(dt)
    out, _ = gemm_nt(dt, L.EPI_DTANH, A, W, None, R=h)
    assert relerr(out, base * (1 - h.float() ** 2)) < t_hi


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("M,No,Ki,heads", [(1568, 576, 192, 3), (392, 192, 768, 0), (392, 768, 192, 0),
                                           (100, 1000, 192, 0), (260, 192, 384, 0), (50176 // 8, 1152, 384, 6)])
def test_gemm_tn(dt, M, No, Ki, heads):
    dY = dev(detfill.normalish((M, No), 21), dt)
    X = dev(detfill.normalish((M, Ki), 22), dt)
    dW = torch.full((No, Ki), 7.0, device=DEV)
    db = torch.full((No,), 7.0, device=DEV)
    wsb = L.lib().rgbnm_gemm_tn_workspace(M, No, Ki)
    ws = torch.empty(wsb, device=DEV, dtype=torch.uint8)
    L.check(L.lib().rgbnm_gemm_tn(L.dt_of(dt), dY.data_ptr(), No, X.data_ptr(), Ki, dW.data_ptr(), db.data_ptr(), M, No,
                                  Ki, heads, 0, ws.data_ptr(), wsb, L.stream()))
    ref = dY.float().T @ X.float()
    rb = dY.float().sum(0)
    if heads:
        inner = heads * 64
        n = torch.arange(No, device=DEV)
        s3, rem = n // inner, n % inner
        dst = (rem // 64) * 192 + (rem % 64) * 3 + s3
        r2, b2 = torch.empty_like(ref), torch.empty_like(rb)
        r2[dst] = ref
        b2[dst] = rb
        ref, rb = r2, b2
    sync()
    assert relerr(dW, ref) < tol(dt, 3e-6, 1e-5), relerr(dW, ref)   # inputs are exact in both modes; fp32 accumulate
    assert relerr(db, rb) < 1e-5
    # accumulate
    L.check(L.lib().rgbnm_gemm_tn(L.dt_of(dt), dY.data_ptr(), No, X.data_ptr(), Ki, dW.data_ptr(), db.data_ptr(), M, No,
                                  Ki, heads, 1, ws.data_ptr(), wsb, L.stream()))
    assert relerr(dW, 2 * ref) < 1e-5


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("E", [192, 384])
def test_layernorm(dt, E):
    M = 1000
    x = dev(detfill.normalish((M, E), 31) * 2 + 0.5, dt)
    g = dev(1 + detfill.uniform((E,), 32, -0.2, 0.2))
    b = dev(detfill.uniform((E,), 33, -0.2, 0.2))
    y = torch.empty_like(x)
    mean = torch.empty(M, device=DEV)
    rstd = torch.empty(M, device=DEV)
    L.check(L.lib().rgbnm_layernorm_fwd(L.dt_of(dt), x.data_ptr(), g.data_ptr(), b.data_ptr(), y.data_ptr(),
                                        mean.data_ptr(), rstd.data_ptr(), M, E, 1e-5, L.stream()))
    xr = x.float().clone().requires_grad_(True)
    gr, br = g.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = torch.nn.functional.layer_norm(xr, (E,), gr, br, 1e-5)
    assert relerr(y, yr) < tol(dt, 2e-6, 3e-3)
    assert relerr(mean, xr.mean(1)) < 1e-5
    dy = dev(detfill.normalish((M, E), 34), dt)
    dres = dev(detfill.normalish((M, E), 35), dt)
    yr.backward(dy.float())
    dx = torch.empty_like(x)
    dg, dbt = torch.empty(E, device=DEV), torch.empty(E, device=DEV)
    wsb = L.lib().rgbnm_layernorm_bwd_workspace(M, E)
    ws = torch.empty(wsb, device=DEV, dtype=torch.uint8)
    L.check(L.lib().rgbnm_layernorm_bwd(L.dt_of(dt), dy.data_ptr(), x.data_ptr(), g.data_ptr(), mean.data_ptr(),
                                        rstd.data_ptr(), dres.data_ptr(), dx.data_ptr(), dg.data_ptr(), dbt.data_ptr(),
                                        M, E, 0, ws.data_ptr(), wsb, L.stream()))
    assert relerr(dx, xr.grad + dres.float()) < tol(dt, 3e-6, 4e-3)
    assert relerr(dg, gr.grad) < 1e-5 and relerr(dbt, br.grad) < 1e-5


def ref_attention(qkv, B, N, H, scale):
    I = H * 64
    q, k, v = [qkv[:, i * I:(i + 1) * I].reshape(B, N, H, 64).permute(0, 2, 1, 3) for i in range(3)]
    att = torch.softmax((q @ k.transpose(-1, -2)) * scale, dim=-1)
    out = (att @ v).permute(0, 2, 1, 3).reshape(B * N, I)
    lse = torch.logsumexp((q @ k.transpose(-1, -2)) * scale, dim=-1)
    return out, lse


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("B,N,H", [(3, 196, 3), (2, 196, 6), (2, 64, 3), (1, 100, 2),
                                   (2, 294, 3), (1, 294, 6), (1, 225, 2), (1, 320, 1)])   # > 224 tokens: embed_type 3 (10-tile kernels)
def test_attention_fwd_bwd(dt, B, N, H):
    I = H * 64
    scale = 1.0 / math.sqrt(H * 64)
    qkv = dev(detfill.normalish((B * N, 3 * I), 41) * 1.5, dt)
    out = torch.empty(B * N, I, device=DEV, dtype=dt)
    lse = torch.empty(B * H * N, device=DEV)
    L.check(L.lib().rgbnm_attention_fwd(L.dt_of(dt), qkv.data_ptr(), out.data_ptr(), lse.data_ptr(), B, N, H, scale,
                                        L.stream()))
    qr = qkv.float().clone().requires_grad_(True)
    oref, lref = ref_attention(qr, B, N, H, scale)
    sync()
    assert relerr(out, oref) < tol(dt, 3e-6, 6e-3), relerr(out, oref)
    assert relerr(lse.view(B, H, N), lref) < tol(dt, 2e-6, 2e-3)
    dout = dev(detfill.normalish((B * N, I), 42), dt)
    oref.backward(dout.float())
    dqkv = torch.full_like(qkv, float("nan"))
    L.check(L.lib().rgbnm_attention_bwd(L.dt_of(dt), qkv.data_ptr(), out.data_ptr(), dout.data_ptr(), lse.data_ptr(),
                                        dqkv.data_ptr(), B, N, H, scale, L.stream()))
    sync()
    assert torch.isfinite(dqkv.float()).all()
    for i, nm in enumerate("qkv"):
        e = relerr(dqkv[:, i * I:(i + 1) * I], qr.grad[:, i * I:(i + 1) * I])
        assert e < tol(dt, 1e-5, 1.5e-2), (nm, e)


def test_attention_spiked_scores():
    # one query/key pair with a huge score: softmax must stay finite and exact (max-subtraction path)
    B, N, H, dt = 1, 196, 3, torch.float32
    I = H * 64
    qkv = dev(detfill.normalish((B * N, 3 * I), 43) * 0.5)
    qkv[17, :64] *= 40
    qkv[101, I:I + 64] = qkv[17, :64]
    out = torch.empty(B * N, I, device=DEV)
    lse = torch.empty(B * H * N, device=DEV)
    scale = 1 / math.sqrt(192)
    L.check(L.lib().rgbnm_attention_fwd(0, qkv.data_ptr(), out.data_ptr(), lse.data_ptr(), B, N, H, scale, L.stream()))
    oref, _ = ref_attention(qkv, B, N, H, scale)
    assert torch.isfinite(out).all() and relerr(out, oref) < 1e-5


@pytest.mark.parametrize("odt", DTS)
def test_subblock_embed_golden(golden, odt):
    g = golden("g9_subblock.npz")
    y = dev(detfill.normalish((2, 1, 4, 6, 8, 8), 61))
    c = dev(detfill.normalish((2, 2, 2, 3, 8, 8), 62))
    A = rg.dct_ops.generate_conversion_matrix(8, 2).to(DEV).contiguous()
    np.testing.assert_allclose(A.cpu().numpy(), g["convY"], atol=2e-6)
    feat = torch.empty(2 * 2 * 3, 384, device=DEV, dtype=odt)
    L.check(L.lib().rgbnm_subblock_embed(0, L.dt_of(odt), y.data_ptr(), c.data_ptr(), A.data_ptr(), feat.data_ptr(), 2, 4,
                                         6, 0, L.stream()))
    ref = torch.from_numpy(g["feat"]).reshape(12, 384)
    if odt == torch.float32:
        # index shuffle bit-exact on the chroma half; A.X.A^T within fp32 round-off
        assert torch.equal(feat[:, 256:].cpu(), ref[:, 256:])
        assert (feat[:, :256].cpu() - ref[:, :256]).abs().max() < 5e-6
    else:
        assert torch.equal(feat[:, 256:].cpu(), ref[:, 256:].bfloat16())
        assert relerr(feat, ref) < 3e-3


@pytest.mark.parametrize("hard", [False, True])
def test_softxent(hard):
    B, Cn = 37, 1000
    z = dev(detfill.normalish((B, Cn), 51) * 3)
    if hard:
        t = torch.from_numpy(detfill.integers((B,), 52, 0, Cn - 1, np.int64)).to(DEV)
    else:
        tt = detfill.uniform((B, Cn), 53, 0, 1)
        t = dev(tt / tt.sum(1, keepdims=True))
    zr = z.clone().requires_grad_(True)
    lr = torch.nn.CrossEntropyLoss()(zr, t)
    lr.backward()
    zz = z.clone().requires_grad_(True)
    loss = rg.cls_transforms.cross_entropy(zz, t)
    loss.backward()
    assert abs(loss.item() - lr.item()) < 2e-6 * max(1, abs(lr.item()))
    assert relerr(zz.grad, zr.grad) < 1e-5


@pytest.mark.parametrize("hard", [False, True])
def test_softxent_one_launch_per_direction(hard):
    """rgbnm_softxent_loss / _grad (round 6) against the two-launch rgbnm_softxent: the loss the SAME BITS (the last workgroup sums
    the rows in mean_kernel's order), dlogits the same bits at gout = 1 and scaled by a device-side gout otherwise; run twice so
    that the self-resetting ticket is exercised."""
    B, Cn = 256, 1000
    z = dev(detfill.normalish((B, Cn), 51) * 3)
    if hard:
        t = torch.from_numpy(detfill.integers((B,), 52, 0, Cn - 1, np.int64)).to(DEV)
    else:
        tt = detfill.uniform((B, Cn), 53, 0, 1)
        t = dev(tt / tt.sum(1, keepdims=True))
    soft, hardp = (None, t.data_ptr()) if hard else (t.data_ptr(), None)
    rows0, loss0 = torch.empty(B, device=DEV), torch.empty(1, device=DEV)
    for dt in DTS:
        dl0 = torch.empty(B, Cn, device=DEV, dtype=dt)
        L.check(L.lib().rgbnm_softxent(L.dt_of(dt), z.data_ptr(), soft, hardp, rows0.data_ptr(), loss0.data_ptr(), dl0.data_ptr(),
                                       B, Cn, 1.0 / B, L.stream()))
        ticket = torch.zeros(1, device=DEV, dtype=torch.int32)
        for rep in range(2):
            rows, stat, loss = torch.empty(B, device=DEV), torch.empty(2 * B, device=DEV), torch.empty(1, device=DEV)
            L.check(L.lib().rgbnm_softxent_loss(z.data_ptr(), soft, hardp, rows.data_ptr(), stat.data_ptr(), loss.data_ptr(),
                                                ticket.data_ptr(), B, Cn, L.stream()))
            assert torch.equal(loss, loss0) and torch.equal(rows, rows0) and int(ticket.item()) == 0
        for gv in (1.0, 0.37, 1024.0):
            gout = torch.full((), gv, device=DEV)
            dl = torch.empty(B, Cn, device=DEV, dtype=dt)
            L.check(L.lib().rgbnm_softxent_grad(L.dt_of(dt), z.data_ptr(), soft, hardp, stat.data_ptr(), gout.data_ptr(),
                                                dl.data_ptr(), B, Cn, 1.0 / B, L.stream()))
            if gv == 1.0:
                assert torch.equal(dl, dl0)
                dl1 = torch.empty_like(dl)
                L.check(L.lib().rgbnm_softxent_grad(L.dt_of(dt), z.data_ptr(), soft, hardp, stat.data_ptr(), None, dl1.data_ptr(),
                                                    B, Cn, 1.0 / B, L.stream()))
                assert torch.equal(dl1, dl0)
            else:
                assert relerr(dl, dl0.float() * gv) < tol(dt, 1e-6, 4e-3)


def test_cross_entropy_scaled_backward():
    """autograd's output gradient (GradScaler's scale, train.py:159; a weighted sum of losses) reaches dlogits on the device."""
    B, Cn = 19, 1000
    z = dev(detfill.normalish((B, Cn), 54) * 2)
    t = torch.from_numpy(detfill.integers((B,), 55, 0, Cn - 1, np.int64)).to(DEV)
    zr = z.clone().requires_grad_(True)
    (torch.nn.CrossEntropyLoss()(zr, t) * 3.5).backward()
    zz = z.clone().requires_grad_(True)
    (rg.cls_transforms.cross_entropy(zz, t) * 3.5).backward()
    assert relerr(zz.grad, zr.grad) < 1e-5


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("B", [1, 2, 7])
def test_subblock_embed_mix_equals_mixup_then_subblock_embed(dt, B):
    """rgbnm_subblock_embed_mix (round 6): RandomMixup_DCT's roll-by-one applied while the sub-block kernel loads -- the same bits
    as rgbnm_mixup (input dtype -> input dtype) followed by rgbnm_subblock_embed, for the luma tiles and the chroma halves; B = 1
    mixes an image with itself, like torch.roll on a batch of one."""
    y = dev(detfill.normalish((B, 1, 28, 28, 8, 8), 61), dt)
    c = dev(detfill.normalish((B, 2, 14, 14, 8, 8), 62), dt)
    lam = dev(np.array([0.8125, 0.1875], dtype=np.float32) if B != 7 else np.array([0.61803, 0.38197], dtype=np.float32))
    A = rg.dct_ops.generate_conversion_matrix(8, 2).to(DEV).contiguous()
    my, mc = torch.empty_like(y), torch.empty_like(c)
    for t, o in ((y, my), (c, mc)):
        L.check(L.lib().rgbnm_mixup(L.dt_of(dt), L.dt_of(dt), t.data_ptr(), o.data_ptr(), lam.data_ptr(), B, t.numel() // B, L.stream()))
    for odt in DTS:
        want = torch.empty(B * 196, 384, device=DEV, dtype=odt)
        got = torch.full_like(want, 7.0)
        L.check(L.lib().rgbnm_subblock_embed(L.dt_of(dt), L.dt_of(odt), my.data_ptr(), mc.data_ptr(), A.data_ptr(), want.data_ptr(),
                                             B, 28, 28, 0, L.stream()))
        L.check(L.lib().rgbnm_subblock_embed_mix(L.dt_of(dt), L.dt_of(odt), y.data_ptr(), c.data_ptr(), lam.data_ptr(), A.data_ptr(),
                                                 got.data_ptr(), B, 28, 28, 0, L.stream()))
        assert torch.equal(got, want)
    # and the mixed values themselves against the reference expression (cls_transforms.py:165-176) in fp32
    ref = y.float() * lam[0] + y.float().roll(1, 0) * lam[1]
    assert (my.float() - ref).abs().max() <= (1e-6 if dt == torch.float32 else 2e-2)


def test_lazy_mixup_through_the_model():
    """RandomMixup_DCT(lazy=True): the model mixes while it loads; logits and every gradient the same bits as with the mixed tensors."""
    B = 6
    m = rg.ViT(3, 16, 192, depth=2, n_classes=1000, drop_p=0.0, device=DEV, num_heads=3, head_size=64, pixel_space="DCT", ver=1)
    m.compute_dtype = torch.bfloat16
    y = dev(detfill.normalish((B, 1, 28, 28, 8, 8), 63), torch.bfloat16)
    c = dev(detfill.normalish((B, 2, 14, 14, 8, 8), 64), torch.bfloat16)
    lab = torch.from_numpy(detfill.integers((B,), 65, 0, 998, np.int64)).to(DEV)
    lam = dev(np.array([0.7, 0.3], dtype=np.float32))
    mix = rg.cls_transforms.RandomMixup_DCT(1000, alpha=0.2)
    res = []
    for lazy in (False, True):
        mix.lazy = lazy
        (my, mc), mt = mix((y, c), lab, lam=lam)
        assert isinstance(my, rg.cls_transforms.LazyMixed) == lazy
        m.zero_grad()
        logits = m(my, mc)
        rg.cls_transforms.cross_entropy(logits, mt, grad_dtype=torch.bfloat16).backward()
        sync()
        res.append((logits.detach().clone(), mt.clone(), {n: p.grad.clone() for n, p in m.named_parameters()}))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    bad = [n for n in res[0][2] if not torch.equal(res[0][2][n], res[1][2][n])]
    assert not bad, bad[:4]
    # a consumer that is not the group patch embedding gets the mixed tensor
    mix.lazy = False
    (ey, _), _ = mix((y, c), lab, lam=lam)
    assert torch.equal(my.materialize(), ey)


def test_mixup_golden(golden):
    g = golden("g13_mixup.npz")
    mix = rg.cls_transforms.RandomMixup_DCT(10, alpha=0.2)
    lam = dev(g["lam"])
    (my, mc), mt = mix((dev(g["y"]), dev(g["c"])), torch.from_numpy(g["lab"]).to(DEV), lam=lam)
    np.testing.assert_allclose(my.cpu().numpy(), g["my"], atol=1e-6)
    np.testing.assert_allclose(mc.cpu().numpy(), g["mc"], atol=1e-6)
    np.testing.assert_allclose(mt.cpu().numpy(), g["mt"], atol=1e-6)
    # out=: the same results written into caller-owned (static) buffers, as bench.py's HIP-graph mode needs them
    oy, oc, ot = torch.empty_like(my), torch.empty_like(mc), torch.empty_like(mt)
    (my2, mc2), mt2 = mix((dev(g["y"]), dev(g["c"])), torch.from_numpy(g["lab"]).to(DEV), lam=lam, out=(oy, oc, ot))
    assert my2.data_ptr() == oy.data_ptr() and mt2.data_ptr() == ot.data_ptr()
    assert torch.equal(oy, my) and torch.equal(oc, mc) and torch.equal(ot, mt)
    with pytest.raises(ValueError):
        mix((dev(g["y"]), dev(g["c"])), torch.from_numpy(g["lab"]).to(DEV), lam=lam, out=(oy, oc, ot[:, :5]))
    lam2 = mix.sample_lambda(DEV)
    assert lam2[0] >= lam2[1] and abs(lam2.sum().item() - 1) < 1e-5
    # the draw is the reference's own expression on the CPU generator (cls_transforms.py:168): same seed, same lambda --
    # also across more draws than the pinned staging ring has slots
    torch.manual_seed(77)
    want = [torch._sample_dirichlet(torch.tensor([0.2, 0.2])).sort(descending=True)[0].float() for _ in range(40)]
    torch.manual_seed(77)
    got = [mix.sample_lambda(DEV).clone() for _ in range(40)]
    torch.cuda.synchronize()
    for w, g_ in zip(want, got):
        assert torch.equal(w, g_.cpu())


def test_clip_adamw_wd_golden(golden):
    g = golden("g12_optim.npz")
    names = ["a.weight", "a.bias", "x_lrnorm.weight", "b.weight"]
    shapes = [(5, 7), (5,), (7,), (3, 5)]
    offs, total = [], 0
    for s in shapes:
        offs.append(total)
        total += (int(np.prod(s)) + 255) // 256 * 256
    p = torch.zeros(total, device=DEV)
    gr, m, v = torch.zeros_like(p), torch.zeros_like(p), torch.zeros_like(p)
    flags = torch.zeros(total // 256, dtype=torch.uint8)
    for i, s in enumerate(shapes):
        n = int(np.prod(s))
        p[offs[i]:offs[i] + n] = dev(detfill.uniform(s, 81 + i)).reshape(-1)
        if (".weight" in names[i]) and ("lrnorm" not in names[i]):
            flags[offs[i] // 256] = 1
    flags = flags.to(DEV)
    ws = torch.empty(L.lib().rgbnm_clip_adamw_wd_workspace(), device=DEV, dtype=torch.uint8)
    norm = torch.zeros(1, device=DEV)
    lrs = [3e-3, 1.5e-3, 2.5e-3]
    for it in range(3):
        for i, s in enumerate(shapes):
            n = int(np.prod(s))
            gr[offs[i]:offs[i] + n] = dev(detfill.uniform(s, 91 + 10 * it + i, -2.0, 2.0)).reshape(-1)
        L.check(L.lib().rgbnm_clip_adamw_wd_step(p.data_ptr(), gr.data_ptr(), m.data_ptr(), v.data_ptr(),
                                                 flags.data_ptr(), total, lrs[it], 0.9, 0.999, 1e-8, it + 1,
                                                 (lrs[it] / 3e-3) * 1e-4, 1.0, norm.data_ptr(), ws.data_ptr(),
                                                 ws.numel(), L.stream()))
        flat = np.concatenate([p[offs[i]:offs[i] + int(np.prod(s))].cpu().numpy() for i, s in enumerate(shapes)])
        np.testing.assert_allclose(flat, g[f"p{it + 1}"], rtol=0, atol=3e-6)
        assert abs(norm.item() - float(g[f"norm{it + 1}"])) < 1e-4


# ---- both code paths of the runtime switches stay parity-green ------------------------------------------
@pytest.fixture
def option():
    saved = {}

    def set_(name, val):
        saved.setdefault(name, L.lib().rgbnm_get_option(name.encode()))
        L.check(L.lib().rgbnm_set_option(name.encode(), val))
    yield set_
    for k, v in saved.items():
        L.lib().rgbnm_set_option(k.encode(), v)


@pytest.mark.parametrize("staged", [0, 1])
def test_gemm_nt_bf16_epilogue_paths(option, staged):
    option("nt_staged", staged)
    test_gemm_nt_epilogues(torch.bfloat16)
    for shp in [(1568, 576, 192), (200, 1000, 192), (64, 192, 1000), (300, 768, 192)]:
        test_gemm_nt_bias(torch.bfloat16, *shp)


@pytest.mark.parametrize("small", [0, 1])
def test_gemm_nt_bf16_few_rows_path(option, small):
    """gemm_nt_small.hip (M <= 512: the head's GEMMs) and the tile-per-workgroup kernel on the head's shapes and on ragged
    ones, against torch fp32: bias, tanh, the (1 - h^2) product, bf16 and fp32 outputs, K with a half MFMA step (1000)."""
    option("nt_small", small)
    dt = torch.bfloat16
    for M, N, K in [(256, 192, 192), (256, 1000, 192), (256, 192, 1000), (77, 1000, 192), (33, 36, 72), (512, 192, 1000)]:
        A = dev(detfill.normalish((M, K), 31), dt)
        W = dev(detfill.uniform((N, K), 32, -0.1, 0.1), dt)
        b = dev(detfill.uniform((N,), 33))
        base = A.float() @ W.float().T
        out, _ = gemm_nt(dt, L.EPI_NONE, A, W, b)
        assert relerr(out, base + b) < 4e-3, (M, N, K)
        out32, _ = gemm_nt(dt, L.EPI_NONE, A, W, b, c_f32=True)
        assert out32.dtype == torch.float32 and relerr(out32, base + b) < 1e-5, (M, N, K)
        out, _ = gemm_nt(dt, L.EPI_TANH, A, W, b)
        assert relerr(out, torch.tanh((base + b).to(dt).float())) < 4e-3, (M, N, K)
        h = torch.tanh(dev(detfill.normalish((M, N), 34))).to(dt)
        out, _ = gemm_nt(dt, L.EPI_DTANH, A, W, None, R=h)
        assert relerr(out, base * (1 - h.float() ** 2)) < 6e-3, (M, N, K)
    # a selector operand catches row / column swaps of the accumulator layout
    M, N, K = 64, 96, 64
    A = torch.zeros(M, K, device=DEV)
    for i in range(M):
        A[i, (i * 7) % K] = 1.0
    W = dev(detfill.uniform((N, K), 35), dt)
    out, _ = gemm_nt(dt, L.EPI_NONE, A.to(dt), W)
    assert torch.equal(out.float(), (A @ W.float().T))


@pytest.mark.parametrize("tr", [0, 1])
def test_gemm_tn_bf16_fragment_paths(option, tr):
    option("tn_tr", tr)
    for shp in [(1568, 576, 192, 3), (392, 192, 768, 0), (100, 1000, 192, 0), (260, 192, 384, 0)]:
        test_gemm_tn(torch.bfloat16, *shp)


@pytest.mark.parametrize("pipe", [0, 1])
def test_gemm_tn_bf16_pipelined_path(option, pipe):
    """M % 64 == 0 shapes take the LDS-DMA pipelined kernel (tn_pipe=1); both paths must agree with the oracle."""
    option("tn_pipe", pipe)
    for shp in [(1024, 576, 192, 3), (64 * 49, 192, 768, 0), (64 * 49, 768, 192, 0), (256, 1000, 192, 0),
                (64 * 7, 192, 384, 0), (64 * 100, 1152, 384, 6), (50176, 768, 192, 0)]:
        test_gemm_tn(torch.bfloat16, *shp)


@pytest.mark.parametrize("v2", [0, 1])
def test_attention_bf16_both_generations(option, v2):
    option("attn_v2", v2)
    for shp in [(3, 196, 3), (2, 196, 6), (2, 64, 3), (1, 100, 2), (2, 33, 3)]:
        test_attention_fwd_bwd(torch.bfloat16, *shp)


@pytest.mark.parametrize("M,N", [(4096, 576), (64 * 67, 768), (6400, 192), (64 * 263, 96), (50176, 768)])
def test_gemm_nt_bf16_weight_resident_path(option, M, N):
    """K = 192, M % 64 == 0, N % 96 == 0 shapes take the persistent weight-resident kernel (nt_wres=1): same bits as
    the tile-per-workgroup kernel (identical accumulation and rounding order), and both agree with the fp32 product."""
    dt, K = torch.bfloat16, 192
    A = dev(detfill.normalish((M, K), 31), dt)
    W = dev(detfill.uniform((N, K), 32, -0.1, 0.1), dt)
    b = dev(detfill.uniform((N,), 33))
    R = dev(detfill.normalish((M, N), 34), dt)
    base = A.float() @ W.float().T
    outs = {}
    for wres in (0, 1):
        option("nt_wres", wres)
        outs[wres] = [gemm_nt(dt, L.EPI_NONE, A, W, b)[0], gemm_nt(dt, L.EPI_NONE, A, W, None)[0],
                      gemm_nt(dt, L.EPI_RES, A, W, b, R=R)[0], *gemm_nt(dt, L.EPI_GELU, A, W, b),
                      gemm_nt(dt, L.EPI_DGELU, A, W, None, R=R)[0]]
        sync()
    for x, y in zip(outs[0], outs[1]):
        assert torch.equal(x, y)
    assert relerr(outs[1][0], base + b) < 4e-3
    assert relerr(outs[1][2], base + b + R.float()) < 6e-3
    assert relerr(outs[1][5], base * R.float()) < 6e-3
    # repeated launches are deterministic (no race between the DMA ring and the staging tile)
    for _ in range(3):
        again = gemm_nt(dt, L.EPI_GELU, A, W, b)
        assert torch.equal(again[0], outs[1][3]) and torch.equal(again[1], outs[1][4])


@pytest.mark.parametrize("M,K", [(50176, 768), (50176, 576), (196 * 64, 768), (8192 + 40, 256), (224 * 300, 384)])
def test_gemm_nt_bf16_row_panel_path(option, M, K):
    """N = 192, K % 64 == 0, K >= 256 shapes take the row-panel kernel with the pipelined reduction (nt_kpipe=1):
    same bits as the tile-per-workgroup kernel (same k order, same rounding points), both close to the fp32 product."""
    dt, N = torch.bfloat16, 192
    A = dev(detfill.normalish((M, K), 41), dt)
    W = dev(detfill.uniform((N, K), 42, -0.1, 0.1), dt)
    b = dev(detfill.uniform((N,), 43))
    R = dev(detfill.normalish((M, N), 44), dt)
    base = A.float() @ W.float().T
    outs = {}
    for kp in (0, 1):
        option("nt_kpipe", kp)
        outs[kp] = [gemm_nt(dt, L.EPI_NONE, A, W, None)[0], gemm_nt(dt, L.EPI_NONE, A, W, b)[0],
                    gemm_nt(dt, L.EPI_RES, A, W, b, R=R)[0]]
        sync()
    for x, y in zip(outs[0], outs[1]):
        assert torch.equal(x, y)
    assert relerr(outs[1][1], base + b) < 4e-3
    assert relerr(outs[1][2], base + b + R.float()) < 6e-3
    for _ in range(3):
        assert torch.equal(gemm_nt(dt, L.EPI_RES, A, W, b, R=R)[0], outs[1][2])


@pytest.mark.parametrize("M,N,K", [(224 * 40 + 17, 384, 384), (50176, 1152, 384), (12544, 1536, 384), (9000, 384, 1536),
                                   (16384, 768, 768), (16384, 768, 3072), (65536, 384, 1536), (8192, 1152, 384)])
def test_gemm_nt_bf16_row_panel_column_tiles(option, M, N, K):
    """N a multiple of 192 (JPEG-S: 384 / 1152 / 1536 wide Linears): the row-panel kernel walks 224-row panels x
    192-column tiles, with the residual, GELU (+ GELU') and dGELU epilogues.  Same bits as the tile-per-workgroup kernel
    (same k order and rounding points), ragged last panel included.  Row counts that are multiples of 256 and not of 224 (the
    SwinV2-T stages) take the 8-wave / 256-row geometry (kp8)."""
    dt = torch.bfloat16
    A = dev(detfill.normalish((M, K), 51), dt)
    W = dev(detfill.uniform((N, K), 52, -0.1, 0.1), dt)
    b = dev(detfill.uniform((N,), 53))
    R = dev(detfill.normalish((M, N), 54), dt)
    outs = {}
    for kp in (0, 1):
        option("nt_kpipe", kp)
        g = gemm_nt(dt, L.EPI_GELU, A, W, b)
        outs[kp] = [gemm_nt(dt, L.EPI_NONE, A, W, b)[0], gemm_nt(dt, L.EPI_RES, A, W, b, R=R)[0], g[0], g[1],
                    gemm_nt(dt, L.EPI_DGELU, A, W, None, R=R)[0]]
        sync()
    for x, y in zip(outs[0], outs[1]):
        assert torch.equal(x, y)
    base = A.float() @ W.float().T
    assert relerr(outs[1][0], base + b) < 4e-3
    assert relerr(outs[1][4], base * R.float()) < 8e-3


@pytest.mark.parametrize("persist", [0, 1])
def test_attention_bf16_backward_schedules(option, persist):
    """attn_persist=1: persistent backward (several (image, head) pairs per workgroup, operands prefetched a phase
    ahead); 0: one workgroup per pair.  Both against the oracle, including > 256 pairs (grid wrap) and ragged N."""
    option("attn_persist", persist)
    for shp in [(3, 196, 3), (100, 196, 3), (2, 64, 3), (1, 100, 2), (2, 33, 3), (50, 196, 6)]:
        test_attention_fwd_bwd(torch.bfloat16, *shp)


def test_gemm_tn_bf16_large_and_ragged_shapes():
    """The pipelined weight-gradient kernel (128 x 192 tiles) on the shapes of the ViT / JPEG-S blocks, the head and ragged rows."""
    for shp in [(50176, 768, 192, 0), (50176, 192, 768, 0), (50176, 576, 192, 3), (64 * 49, 576, 192, 3),
                (256, 1000, 192, 0), (64 * 100, 1152, 384, 6), (64 * 30, 384, 384, 0)]:
        test_gemm_tn(torch.bfloat16, *shp)


@pytest.mark.parametrize("wide", [1, 0])
def test_gemm_tn_bf16_wide_tiles(option, wide):
    """Launches whose GEMMs all have No % 192 == 0 and Ki % 384 == 0 take the 192 x 384-tile kernel (tn_wide = 1, E = 384 and
    wider): the Linears of a JPEG-S block and of SwinV2's stages 3 / 4, with the qkv row permutation, bias sums, token splits
    (small M: many splits; one 32-token tile per split at M = 64 * 4) and accumulation -- against fp32 matmul, both settings."""
    option("tn_wide", wide)
    for shp in [(50176 // 4, 1152, 384, 6), (64 * 30, 384, 384, 0), (64 * 49, 1536, 384, 0), (64 * 49, 384, 1536, 0),
                (64 * 4, 192, 384, 0), (16384, 2304, 768, 12), (64 * 20, 768, 3072, 0), (64 * 9, 3072, 768, 0)]:
        test_gemm_tn(torch.bfloat16, *shp)


def test_lazy_mixup_target_has_the_bits_of_the_dense_one():
    """RandomMixup_DCT(lazy_target=True) hands cross_entropy the labels and lambda (cls_transforms.LazyTarget) instead of the dense
    [B, classes] target: loss, the gradient of the logits (fp32 and through a bf16 head edge's dtype) and the materialised target must
    be the bits of the dense path -- rgbnm_mixup_target followed by rgbnm_softxent_loss / _grad."""
    from rgb_no_more_amd import cls_transforms as CL
    B, Cn = 256, 1000
    lab = torch.from_numpy(detfill.integers((B,), 61, 0, Cn - 1, np.int64)).to(DEV)
    lab[7] = lab[6]                                        # a row whose partner has the same label: target mass lam0 + lam1 on one class
    z = dev(detfill.normalish((B, Cn), 62) * 3)
    y = dev(detfill.normalish((B, 1, 4, 4, 8, 8), 63))
    for lam_v in ((0.7, 0.3), (1.0, 0.0), (0.5, 0.5)):
        lam = torch.tensor(lam_v, device=DEV, dtype=torch.float32)
        mix = CL.RandomMixup_DCT(Cn, alpha=0.2)
        (yd,), td = mix((y,), lab, lam=lam)
        mix.lazy_target = True
        (yl,), tl = mix((y,), lab, lam=lam)
        assert isinstance(tl, CL.LazyTarget) and tuple(tl.shape) == (B, Cn) and torch.equal(yd, yl)
        assert torch.equal(tl.materialize(), td)
        for gd in (torch.float32, torch.bfloat16):
            res = []
            for tgt in (td, tl):
                zz = z.clone().requires_grad_(True)
                loss = CL.cross_entropy(zz, tgt, grad_dtype=gd)
                (loss * 3.0).backward()
                res.append((loss.detach().clone(), zz.grad.clone()))
            assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]), (lam_v, gd)
        ref = torch.nn.functional.cross_entropy(z, td)
        assert abs(float(res[1][0]) - float(ref)) < 1e-5
    with pytest.raises(ValueError):
        CL.cross_entropy(z[:, :10].contiguous(), tl)


def test_gemm_tn_wide_tiles_repeatable():
    """The 192 x 384-tile kernel hand-counts its DMA waits over tiles of 4 and 5 instructions per wave (nine per two tiles); a
    miscounted wait would read a tile before it has landed -- visible as run-to-run differences.  Thirty launches of a split and of
    an unsplit shape must give the same bits, next to other traffic on the device."""
    lib = L.lib()
    for (M, No, Ki) in [(64 * 49, 1536, 384), (50176 // 2, 384, 1536)]:
        dY = dev(detfill.normalish((M, No), 71), torch.bfloat16)
        X = dev(detfill.normalish((M, Ki), 72), torch.bfloat16)
        wsb = lib.rgbnm_gemm_tn_workspace(M, No, Ki)
        ws = torch.empty(wsb, device=DEV, dtype=torch.uint8)
        noise = torch.randn(64 << 20, device=DEV)
        first = None
        for rep in range(30):
            dW = torch.empty(No, Ki, device=DEV)
            db = torch.empty(No, device=DEV)
            noise.mul_(1.0001)                              # something else in flight in front of every launch
            L.check(lib.rgbnm_gemm_tn(L.dt_of(torch.bfloat16), dY.data_ptr(), No, X.data_ptr(), Ki, dW.data_ptr(), db.data_ptr(), M, No,
                                      Ki, 0, 0, ws.data_ptr(), wsb, L.stream()))
            if first is None:
                first = (dW.clone(), db.clone())
                ref = dY.float().T @ X.float()
                assert relerr(dW, ref) < 1e-5
            else:
                assert torch.equal(dW, first[0]) and torch.equal(db, first[1]), rep
