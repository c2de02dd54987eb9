"""SwinV2 DCT (BASELINE config 5) on the HIP kernels against the oracle (oracle/swin_torch.py) and
the reference goldens (tests/golden/g15_swin.npz, generated from /root/reference/models/swinv2.py).
fp32: logits within the north-star 1e-3 (measured ~1e-5); bf16: tolerance of the ViT bf16 test; index ops bit exact."""
import numpy as np
import pytest
import torch

import rgb_no_more_amd as rg
from rgb_no_more_amd import detfill, lib as L, swinv2 as SW
from oracle import swin_torch as S

DEV = "cuda"
CASES = {"sw3": (128, [2, 2, 2], [3, 6, 12], 2), "swt": (256, [2, 2, 6, 2], [3, 6, 12, 24], 1)}


def _model(tag, device):
    img, depths, heads, B = CASES[tag]
    m = rg.SwinTransformerV2(img_size=img, patch_size=4, embed_dim=96, depths=depths, num_heads=heads, window_size=8,
                             mlp_ratio=4.0, drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.0, qkv_bias=True,
                             ape=False, patch_norm=True, pretrained_window_sizes=[0] * len(depths), device=device,
                             pixel_space="dct")
    return m, img, depths, heads, B


def test_state_dict_surface_matches_reference(golden):
    g = golden("g15_swin.npz")
    for tag in CASES:
        m, *_ = _model(tag, "cpu")
        assert [n for n, _ in m.named_parameters()] == [str(s) for s in g[tag + "_names"]]
        assert [str(tuple(p.shape)) for _, p in m.named_parameters()] == [str(s) for s in g[tag + "_shapes"]]
        assert [n for n, _ in m.named_buffers()] == [str(s) for s in g[tag + "_buffers"]]
    with pytest.raises(NotImplementedError):
        rg.SwinTransformerV2(img_size=256, patch_size=4, window_size=7, pixel_space="dct")
    with pytest.raises(NotImplementedError):
        rg.SwinTransformerV2(img_size=256, patch_size=4, window_size=8, pixel_space="rgb")


def _load(m, tag, g):
    names = [str(n) for n in g[tag + "_names"]]
    img, depths, heads, B = CASES[tag]
    shapes = S.param_shapes(depths, heads)
    sd = S.fill_params({n: shapes[n] for n in names})
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    nb = img // 8
    y = torch.from_numpy(detfill.normalish((B, 1, nb, nb, 8, 8), 171)).to(DEV)
    c = torch.from_numpy(detfill.normalish((B, 2, nb // 2, nb // 2, 8, 8), 172)).to(DEV)
    tgt = detfill.uniform((B, 1000), 173, 0.0, 1.0)
    tgt = torch.from_numpy(tgt / tgt.sum(1, keepdims=True)).to(DEV)
    return names, y, c, tgt


@pytest.mark.gpu
def test_embed_decomposition_is_exact_index_work_plus_fp32_products():
    B, H = 3, 6
    y = torch.from_numpy(detfill.normalish((B, 1, H, H, 8, 8), 5))
    c = torch.from_numpy(detfill.normalish((B, 2, H // 2, H // 2, 8, 8), 6))
    ref = S.decompose_features(y, c).reshape(B * 4 * H * H, 24)
    Ay = torch.from_numpy(S.dct_np.conversion_matrix(4, 2)).float().to(DEV)
    Ac = torch.from_numpy(S.dct_np.conversion_matrix(2, 4)).float().to(DEV)
    out = torch.full((B * 4 * H * H, 24), float("nan"), device=DEV)
    yd, cd = y.to(DEV), c.to(DEV)
    L.check(L.lib().rgbnm_swin_embed(0, 0, yd.data_ptr(), cd.data_ptr(), Ay.data_ptr(), Ac.data_ptr(), out.data_ptr(), B, H,
                                     H, L.stream()))
    assert torch.isfinite(out).all()                      # every (token, feature) slot written exactly once
    np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), atol=2e-6)
    # a block holding a single non-zero coefficient lands in the tokens / features the einops string dictates
    y0 = torch.zeros(1, 1, 2, 2, 8, 8)
    y0[0, 0, 1, 0, 0, 0] = 1.0                            # DC of block (1, 0): spreads to the DC of its 2x2 sub-blocks
    r0 = S.decompose_features(y0, torch.zeros(1, 2, 1, 1, 8, 8)).reshape(16, 24)
    o0 = torch.empty(16, 24, device=DEV)
    z = torch.zeros(1, 2, 1, 1, 8, 8, device=DEV)
    y0d = y0.to(DEV)
    L.check(L.lib().rgbnm_swin_embed(0, 0, y0d.data_ptr(), z.data_ptr(), Ay.data_ptr(), Ac.data_ptr(), o0.data_ptr(), 1, 2, 2,
                                     L.stream()))
    np.testing.assert_allclose(o0.cpu().numpy(), r0.numpy(), atol=1e-6)
    assert (o0.abs() > 1e-3).sum().item() == (r0.abs() > 1e-3).sum().item()


@pytest.mark.gpu
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("E,N", [(96, 70), (192, 70), (384, 70), (768, 70), (100, 70), (96, 4099), (768, 1031)])
def test_layernorm_any_width_with_residual_and_sample_scale(dt, E, N):
    """Stage widths 96 / 192 / 384 / 768 take the lanes-per-row kernels, any other width (100) one wave per row; the long cases
    walk several row passes per workgroup with a ragged last one."""
    B = 3
    M = B * N
    x = torch.from_numpy(detfill.normalish((M, E), 11)).to(DEV).to(dt)
    res = torch.from_numpy(detfill.normalish((M, E), 12)).to(DEV).to(dt)
    g = torch.from_numpy(1.0 + detfill.uniform((E,), 13, -0.3, 0.3)).to(DEV).requires_grad_(True)
    b = torch.from_numpy(detfill.uniform((E,), 14, -0.3, 0.3)).to(DEV).requires_grad_(True)
    ss = torch.tensor([0.0, 1.25, 1.25], device=DEV)
    xg, rg_ = x.clone().requires_grad_(True), res.clone().requires_grad_(True)
    out = SW._LNFn.apply(xg, g, b, rg_, ss, N)
    xr, rr = x.float().clone().requires_grad_(True), res.float().clone().requires_grad_(True)
    gr, br = g.detach().clone().requires_grad_(True), b.detach().clone().requires_grad_(True)
    ref = rr + ss.repeat_interleave(N)[:, None] * torch.nn.functional.layer_norm(xr, (E,), gr, br, 1e-5)
    tol = 2e-5 if dt == torch.float32 else 3e-2
    assert (out.float() - ref).abs().max().item() < tol
    w = torch.from_numpy(detfill.normalish((M, E), 15)).to(DEV)
    (out.float() * w).sum().backward()
    (ref * w).sum().backward()
    rel = lambda a, bb: ((a.float() - bb).norm() / (bb.norm() + 1e-12)).item()   # noqa: E731
    assert rel(xg.grad, xr.grad) < (1e-5 if dt == torch.float32 else 2e-2)
    assert rel(rg_.grad, rr.grad) < (1e-6 if dt == torch.float32 else 1e-2)
    assert rel(g.grad, gr.grad) < (1e-5 if dt == torch.float32 else 2e-2)
    assert rel(b.grad, br.grad) < (1e-5 if dt == torch.float32 else 2e-2)


@pytest.mark.gpu
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("res,heads,shift", [(16, 3, 0), (16, 3, 4), (8, 6, 0), (32, 3, 4)])
def test_window_attention_forward_backward_vs_oracle(dt, res, heads, shift):
    """Cyclic shift + window partition + cosine attention + bias + mask + reverse, against the oracle's explicit
    roll / partition / softmax pipeline (oracle/swin_torch.py:window_attention) with identity qkv / proj Linears."""
    B, C_ = 2, heads * 32
    qkv = torch.from_numpy(detfill.normalish((B * res * res, 3 * C_), 21)).to(DEV).to(dt)
    bias = torch.from_numpy(detfill.uniform((heads, 64, 64), 22, 0.0, 16.0)).to(DEV).requires_grad_(True)
    scale = torch.from_numpy(detfill.uniform((heads,), 23, 5.0, 30.0)).to(DEV).requires_grad_(True)
    qg = qkv.clone().requires_grad_(True)
    out = SW._WinAttnFn.apply(qg, bias, scale, B, res, C_, heads, shift)
    # oracle: same math with torch ops on the CPU in fp32
    q32 = qkv.float().cpu().clone().requires_grad_(True)
    bias_r, scale_r = bias.detach().cpu().clone().requires_grad_(True), scale.detach().cpu().clone().requires_grad_(True)
    x = q32.reshape(B, res, res, 3 * C_)
    xs = torch.roll(x, (-shift, -shift), (1, 2)) if shift else x
    nw = res // 8
    xw = xs.reshape(B, nw, 8, nw, 8, 3 * C_).permute(0, 1, 3, 2, 4, 5).reshape(B * nw * nw, 64, 3, heads, 32)
    q, k, v = xw.permute(2, 0, 3, 1, 4)
    att = torch.nn.functional.normalize(q, dim=-1) @ torch.nn.functional.normalize(k, dim=-1).transpose(-2, -1)
    att = att * scale_r.view(1, heads, 1, 1) + bias_r.unsqueeze(0)
    if shift:
        m = S.shift_mask(res, 8, shift)
        att = (att.reshape(B, nw * nw, heads, 64, 64) + m[None, :, None]).reshape(-1, heads, 64, 64)
    o = (torch.softmax(att, -1) @ v).transpose(1, 2).reshape(B, nw, nw, 8, 8, C_).permute(0, 1, 3, 2, 4, 5)
    o = o.reshape(B, res, res, C_)
    ref = (torch.roll(o, (shift, shift), (1, 2)) if shift else o).reshape(B * res * res, C_)
    # bf16: q/|q| and k/|k| are rounded to bf16 before the MFMA (as under autocast) and the logit scale (up to 30 here)
    # multiplies that rounding: tolerance of a bf16 attention, not of the fp32 arithmetic around it
    tol = 2e-5 if dt == torch.float32 else 6e-2
    err = (out.float().cpu() - ref).abs().max().item()
    print(f"[win attn {dt} res={res} heads={heads} shift={shift}] max |d out| = {err:.3e}")
    assert err < tol
    w = torch.from_numpy(detfill.normalish((B * res * res, C_), 24))
    (out.float() * w.to(DEV)).sum().backward()
    (ref * w).sum().backward()
    rel = lambda a, bb: ((a.float().cpu() - bb).norm() / (bb.norm() + 1e-12)).item()   # noqa: E731
    t = 2e-5 if dt == torch.float32 else 3e-2
    print("   rel grad errors q/bias/scale:", rel(qg.grad, q32.grad), rel(bias.grad, bias_r.grad), rel(scale.grad, scale_r.grad))
    assert rel(qg.grad, q32.grad) < t
    assert rel(bias.grad, bias_r.grad) < t
    # d(scale) is a heavily cancelling sum over all logits; in bf16 it also sees the rounding of O inside D = dO.O
    assert rel(scale.grad, scale_r.grad) < (t if dt == torch.float32 else 1e-1)


@pytest.mark.gpu
def test_merge_gather_and_token_mean_are_exact():
    B, res, C_ = 2, 6, 8
    x = torch.arange(B * res * res * C_, dtype=torch.float32, device=DEV).reshape(B * res * res, C_).requires_grad_(True)
    out = SW._MergeFn.apply(x, B, res, C_)
    xr = x.detach().reshape(B, res, res, C_)
    ref = torch.cat([xr[:, 0::2, 0::2], xr[:, 1::2, 0::2], xr[:, 0::2, 1::2], xr[:, 1::2, 1::2]], -1).reshape(-1, 4 * C_)
    assert torch.equal(out, ref)
    out.backward(out.detach())
    assert torch.equal(x.grad, x.detach())                # scatter is the exact inverse of the gather
    t = torch.from_numpy(detfill.normalish((3 * 10, 16), 31)).to(DEV).requires_grad_(True)
    mo = SW._MeanFn.apply(t, 3, 10, 16)
    np.testing.assert_allclose(mo.detach().cpu().numpy(), t.detach().reshape(3, 10, 16).mean(1).cpu().numpy(), atol=1e-6)
    mo.sum().backward()
    np.testing.assert_allclose(t.grad.cpu().numpy(), np.full((30, 16), 0.1, np.float32), atol=1e-7)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["sw3", "swt"])
def test_swin_fp32_logits_and_grads_vs_reference_golden(golden, tag):
    g = golden("g15_swin.npz")
    m, img, depths, heads, B = _model(tag, DEV)
    names, y, c, tgt = _load(m, tag, g)
    m.train()
    m.compute_dtype = torch.float32
    logits = m(y, c)
    err = np.abs(logits.detach().cpu().numpy() - g[tag + "_logits"]).max()
    print(f"[{tag}] fp32 max |dlogit| = {err:.3e}")
    assert err <= 1e-3 and err <= 1e-4
    loss = rg.cls_transforms.cross_entropy(logits, tgt)
    assert abs(loss.item() - float(g[tag + "_loss"])) < 2e-5
    loss.backward()
    named = dict(m.named_parameters())
    gn = np.array([named[n].grad.double().norm().item() for n in names])
    np.testing.assert_allclose(gn, g[tag + "_gradnorms"], rtol=2e-3, atol=2e-7)


@pytest.mark.gpu
def test_swin_bf16_and_drop_path_training_step(golden):
    g = golden("g15_swin.npz")
    m, img, depths, heads, B = _model("sw3", DEV)
    names, y, c, tgt = _load(m, "sw3", g)
    m.train()
    m.compute_dtype = torch.bfloat16
    logits = m(y, c)
    err = np.abs(logits.detach().float().cpu().numpy() - g["sw3_logits"]).max()
    print(f"[sw3] bf16 max |dlogit| = {err:.3e}")
    assert err <= 6e-2
    rg.cls_transforms.cross_entropy(logits, tgt, grad_dtype=torch.bfloat16).backward()
    named = dict(m.named_parameters())
    gn = np.array([named[n].grad.double().norm().item() for n in names])
    rel = np.abs(gn - g["sw3_gradnorms"]) / (g["sw3_gradnorms"] + 1e-9)
    assert np.median(rel) < 3e-2
    # stochastic depth: with p > 0 in training mode some samples skip a branch entirely; eval mode is deterministic
    for ly in m.layers:
        for blk in ly.blocks:
            blk.drop_path_p = 0.5
    m.compute_dtype = torch.float32
    torch.manual_seed(0)
    a = m(y, c)
    b2 = m(y, c)
    assert not torch.equal(a, b2)
    m.eval()
    with torch.no_grad():
        e1, e2 = m(y, c), m(y, c)
    assert torch.equal(e1, e2)
    np.testing.assert_allclose(e1.cpu().numpy(), g["sw3_logits"], atol=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_swinv2t_at_batch_64_vs_reference_golden(golden, dt):
    """g20 (make_golden_r3.py): the reference SwinV2-T DCT itself at B = 64 -- every logit, the loss and the gradient norm of
    every parameter -- in both compute modes (bench.py --arch swinv2t runs the same check before timing)."""
    g = golden("g20_fullsize.npz")
    tag, B = "swt_b64", 64
    m, img, depths, heads, _ = _model("swt", DEV)
    names = [str(n) for n in g[tag + "_names"]]
    shapes = S.param_shapes(depths, heads)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in S.fill_params({n: shapes[n] for n in names}).items()}, strict=False)
    nb = img // 8
    y = torch.from_numpy(detfill.normalish((B, 1, nb, nb, 8, 8), 171)).to(DEV)
    c = torch.from_numpy(detfill.normalish((B, 2, nb // 2, nb // 2, 8, 8), 172)).to(DEV)
    tgt = detfill.uniform((B, 1000), 173, 0.0, 1.0)
    tgt = torch.from_numpy(tgt / tgt.sum(1, keepdims=True)).to(DEV)
    m.train()
    m.compute_dtype = dt
    logits = m(y, c)
    loss = rg.cls_transforms.cross_entropy(logits, tgt, grad_dtype=dt)
    loss.backward()
    torch.cuda.synchronize()
    err = np.abs(logits.detach().float().cpu().numpy() - g[tag + "_logits"]).max()
    named = dict(m.named_parameters())
    gn = np.array([named[n].grad.double().norm().item() for n in names])
    rel = np.abs(gn - g[tag + "_gradnorms"]) / (g[tag + "_gradnorms"] + 1e-9)
    print(f"[swt B=64 {dt}] max |dlogit| = {err:.3e}, loss {loss.item():.6f} vs {float(g[tag + '_loss']):.6f}, grad-norm rel "
          f"median {np.median(rel):.3e} max {rel.max():.3e}")
    if dt == torch.float32:
        assert err <= 1e-4
        assert abs(loss.item() - float(g[tag + "_loss"])) < 2e-5
        np.testing.assert_allclose(gn, g[tag + "_gradnorms"], rtol=2e-3, atol=2e-7)
    else:
        assert err <= 2.5e-2                                 # (measured 1.8e-2; the bar was 6e-2 until round 4, 3e-2 until round 6)
        assert abs(loss.item() - float(g[tag + "_loss"])) < 1e-3
        assert np.median(rel) < 1e-2                         # (measured 1.4e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_swinv2t_at_the_timed_batch_256_vs_reference_golden(golden, dt):
    """g21 (make_golden_r4.py): the reference SwinV2-T DCT at B = 256 -- the batch bench.py --arch swinv2t times, where the GEMM
    geometry is chosen from the row count -- every logit, the loss, every gradient norm and strided slices of twelve gradients."""
    g = golden("g21_b256.npz")
    tag, B = "swt_b256", 256
    m, img, depths, heads, _ = _model("swt", DEV)
    names = [str(n) for n in g[tag + "_names"]]
    shapes = S.param_shapes(depths, heads)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in S.fill_params({n: shapes[n] for n in names}).items()}, strict=False)
    nb = img // 8
    y = torch.from_numpy(detfill.normalish((B, 1, nb, nb, 8, 8), 171)).to(DEV)
    c = torch.from_numpy(detfill.normalish((B, 2, nb // 2, nb // 2, 8, 8), 172)).to(DEV)
    tgt = detfill.uniform((B, 1000), 173, 0.0, 1.0)
    tgt = torch.from_numpy(tgt / tgt.sum(1, keepdims=True)).to(DEV)
    m.train()
    m.compute_dtype = dt
    logits = m(y, c)
    loss = rg.cls_transforms.cross_entropy(logits, tgt, grad_dtype=dt)
    loss.backward()
    torch.cuda.synchronize()
    err = np.abs(logits.detach().float().cpu().numpy() - g[tag + "_logits"]).max()
    named = dict(m.named_parameters())
    gn = np.array([named[n].grad.double().norm().item() for n in names])
    rel = np.abs(gn - g[tag + "_gradnorms"]) / (g[tag + "_gradnorms"] + 1e-9)
    worst = 0.0
    for nm in [str(x) for x in g[tag + "_slice_names"]]:
        got = named[nm].grad.reshape(-1)[::37].double().cpu().numpy()
        want = g[tag + "_grad_" + nm].astype(np.float64)
        worst = max(worst, float(np.linalg.norm(got - want) / (np.linalg.norm(want) + 1e-30)))
    print(f"[swt B=256 {dt}] max |dlogit| = {err:.3e}, loss {loss.item():.6f} vs {float(g[tag + '_loss']):.6f}, grad-norm rel "
          f"median {np.median(rel):.3e} max {rel.max():.3e}, worst gradient slice rel {worst:.3e}")
    if dt == torch.float32:
        assert err <= 1e-4
        assert abs(loss.item() - float(g[tag + "_loss"])) < 2e-5
        np.testing.assert_allclose(gn, g[tag + "_gradnorms"], rtol=2e-3, atol=2e-7)
        assert worst < 2e-3
    else:
        assert err <= 2.5e-2                                 # all 256 x 1000 logits (measured 1.9e-2; 3e-2 until round 6)
        assert abs(loss.item() - float(g[tag + "_loss"])) < 1e-3
        assert np.median(rel) < 1e-2                         # (measured 1.5e-3)
        assert worst < 2.5e-2                                # per-tensor bf16 gradient bar (measured 1.6e-2)


@pytest.mark.gpu
def test_fused_optimizer_equals_train_py_objects_on_swin(golden):
    """pipeline_utils.py:535-537 / train.py:172-176: clip_grad_norm_(1) + torch AdamW(weight_decay=0) + the name-filtered WeightDecay,
    against the one-launch FusedClipAdamWWD that bench.py --arch swinv2t times (gradients gathered into the flat buffer first:
    SwinV2's autograd nodes return per-parameter tensors).  Same fp32 gradients in, three steps, parameters compared."""
    g = golden("g15_swin.npz")
    runs = []
    for fused in (False, True):
        m, *_ = _model("sw3", DEV)
        names, y, c, tgt = _load(m, "sw3", g)
        m.train()
        m.compute_dtype = torch.float32
        if fused:
            opt = rg.custom_optims.FusedClipAdamWWD(m, lr=1e-3, eps=1e-8, weight_decay=1e-4, max_norm=1.0)
            step = opt.step
        else:
            adamw = torch.optim.AdamW(m.parameters(), lr=1e-3, weight_decay=0, eps=1e-8)
            wdec = rg.custom_optims.WeightDecay([p for n, p in m.named_parameters() if (".weight" in n) and ("lrnorm" not in n)],
                                                lr=1e-3, weight_decay=1e-4)

            def step():
                torch.nn.utils.clip_grad_norm_(m.parameters(), max_norm=1)
                adamw.step()
                wdec.step()
        for _ in range(3):
            for p in m.parameters():
                p.grad = None
            rg.cls_transforms.cross_entropy(m(y, c), tgt).backward()
            step()
        runs.append({n: p.detach().clone() for n, p in m.named_parameters()})
    worst = max(((runs[0][n] - runs[1][n]).abs().max() / (runs[0][n].abs().max() + 1e-12)).item() for n in runs[0])
    print(f"fused vs train.py objects after 3 steps: worst relative parameter difference {worst:.2e}")
    assert worst < 1e-4        # Adam normalises by sqrt(v): a last-bit difference in the clip norm moves small-gradient entries by ~1e-5 of the largest weight


@pytest.mark.gpu
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_swin_step_is_bit_reproducible(golden, dt):
    """No atomics anywhere (window-attention d(bias), LayerNorm and weight-gradient partial sums go through the ordered reduction):
    the same inputs give the same bits, run after run -- a race in the wave-private LDS staging of the attention kernels or in a
    persistent launch's work split would show here."""
    g = golden("g15_swin.npz")
    m, *_ = _model("swt", DEV)
    names, y, c, tgt = _load(m, "swt", g)
    m.train()
    m.compute_dtype = dt
    runs = []
    for _ in range(3):
        for p in m.parameters():
            p.grad = None
        logits = m(y, c)
        rg.cls_transforms.cross_entropy(logits, tgt, grad_dtype=dt).backward()
        torch.cuda.synchronize()
        runs.append((logits.detach().clone(), {n: p.grad.detach().clone() for n, p in m.named_parameters()}))
    for k in (1, 2):
        assert torch.equal(runs[0][0], runs[k][0])
        bad = [n for n in runs[0][1] if not torch.equal(runs[0][1][n], runs[k][1][n])]
        assert not bad, bad[:5]


@pytest.mark.gpu
@pytest.mark.parametrize("B", [8, 64])
def test_backward_wide_weight_gradient_bracket_equals_per_linear_launches(B):
    """SwinTransformerV2.group_dw_backward (swinv2._DwBracket, VERDICT r4: weight-gradient GEMMs grouped across a stage's blocks):
    with the whole backward in ONE rgbnm_gemm_tn_group bracket every gradient must equal the per-Linear launches' -- same products,
    fp32 sums in another split of the token axis (rel 2e-5 of the tensor's scale; row-paired stage-1 layers and the fp32 CPB tables
    run the old way: same bits) -- the logits the same bits.  A second backward onto ATTACHED gradients (accumulation) must fall
    back to the old path and still be right; the bracket must be closed when backward() returns."""
    from rgb_no_more_amd import swinv2 as SW
    m, img, depths, heads, _ = _model("swt", DEV)
    nb = img // 8
    y = torch.from_numpy(detfill.normalish((B, 1, nb, nb, 8, 8), 271)).to(DEV)
    c = torch.from_numpy(detfill.normalish((B, 2, nb // 2, nb // 2, 8, 8), 272)).to(DEV)
    tgt = detfill.uniform((B, 1000), 273, 0.0, 1.0)
    tgt = torch.from_numpy(tgt / tgt.sum(1, keepdims=True)).to(DEV)
    m.eval()                    # no DropPath draws: both passes see the same network
    m.compute_dtype = torch.bfloat16

    def run(flag, zero=True):
        m.group_dw_backward = flag
        if zero:
            m.zero_grad(set_to_none=True)
        logits = m(y, c)
        rg.cls_transforms.cross_entropy(logits, tgt, grad_dtype=torch.bfloat16).backward()
        torch.cuda.synchronize()
        assert SW._ACTIVE[0] is None and not getattr(m, "_dw_bracket", SW._DwBracket()).active
        return logits.detach().clone(), {n: p.grad.clone() for n, p in m.named_parameters()}

    l0, g0 = run(False)
    l1, g1 = run(True)
    assert torch.equal(l0, l1)
    worst = ("", 0.0)
    for n in g0:
        scale = g0[n].abs().max().item() + 1e-30
        d = (g0[n] - g1[n]).abs().max().item() / scale
        if d > worst[1]:
            worst = (n, d)
    print(f"[dW bracket B={B}] worst gradient difference {worst[1]:.2e} of the tensor's scale ({worst[0]})")
    assert worst[1] < 2e-5, worst
    # accumulation onto attached gradients: the pass must not defer anything (AccumulateGrad adds as each node returns)
    _, g2 = run(True, zero=False)
    for n in g0:
        scale = g0[n].abs().max().item() + 1e-30
        assert (g2[n] - 2 * g1[n]).abs().max().item() / scale < 1e-4, n


@pytest.mark.gpu
def test_abandoned_weight_gradient_bracket_is_dropped_not_launched():
    """a backward pass that dies in a node which is not a Linear (here: the window attention) leaves the backward-wide
    bracket open with jobs queued on the AUTOGRAD WORKER's thread-local queue; the next forward runs on the caller's thread and frees
    the jobs' operands.  The stale jobs must never be launched (they would read freed memory and write into the next pass's
    gradients): the next step must give exactly the gradients of an undisturbed step, and so must a partial backward
    (torch.autograd.grad onto the logits' input side only) followed by a normal step."""
    from rgb_no_more_amd import swinv2 as SW
    B = 8
    m, img, depths, heads, _ = _model("swt", DEV)
    nb = img // 8
    y = torch.from_numpy(detfill.normalish((B, 1, nb, nb, 8, 8), 281)).to(DEV)
    c = torch.from_numpy(detfill.normalish((B, 2, nb // 2, nb // 2, 8, 8), 282)).to(DEV)
    tgt = detfill.uniform((B, 1000), 283, 0.0, 1.0)
    tgt = torch.from_numpy(tgt / tgt.sum(1, keepdims=True)).to(DEV)
    m.eval()
    m.compute_dtype = torch.bfloat16
    m.group_dw_backward = True

    def step():
        m.zero_grad(set_to_none=True)
        rg.cls_transforms.cross_entropy(m(y, c), tgt, grad_dtype=torch.bfloat16).backward()
        torch.cuda.synchronize()
        return {n: p.grad.clone() for n, p in m.named_parameters()}

    want = step()
    # (1) the window attention's backward raises in the middle of the pass
    orig = SW._WinAttnFn.backward
    calls = {"n": 0}

    def boom(ctx, *g):
        calls["n"] += 1
        if calls["n"] == 3:
            raise RuntimeError("injected failure in the window attention backward")
        return orig(ctx, *g)

    SW._WinAttnFn.backward = staticmethod(boom)
    try:
        m.zero_grad(set_to_none=True)
        loss = rg.cls_transforms.cross_entropy(m(y, c), tgt, grad_dtype=torch.bfloat16)
        with pytest.raises(RuntimeError, match="injected failure"):
            loss.backward()
    finally:
        SW._WinAttnFn.backward = staticmethod(orig)
    torch.cuda.synchronize()
    assert m._dw_bracket.active                      # left open by the pass that died ...
    got = step()                                     # ... found and abandoned by this forward; its jobs never run
    assert not m._dw_bracket.active and SW._ACTIVE[0] is None
    bad = [n for n in want if not torch.equal(want[n], got[n])]
    assert not bad, bad[:5]
    # (2) junk where the dead pass's operands used to be, then another step: still the same bits
    junk = [torch.full((1 << 22,), float("nan"), device=DEV) for _ in range(8)]
    del junk
    got = step()
    bad = [n for n in want if not torch.equal(want[n], got[n])]
    assert not bad, bad[:5]
    # (3) a ViT backward on the same worker thread afterwards (its internal groupings must not meet stale jobs)
    v = rg.ViT(3, 16, 192, depth=1, n_classes=1000, drop_p=0.0, device=DEV, num_heads=3, head_size=64, pixel_space="DCT", ver=1)
    v.compute_dtype = torch.bfloat16
    yv = torch.from_numpy(detfill.normalish((2, 1, 28, 28, 8, 8), 71)).to(DEV)
    cv = torch.from_numpy(detfill.normalish((2, 2, 14, 14, 8, 8), 72)).to(DEV)
    lv = torch.from_numpy(detfill.integers((2,), 74, 0, 998, np.int64)).to(DEV)
    rg.cls_transforms.cross_entropy(v(yv, cv), lv, grad_dtype=torch.bfloat16).backward()
    torch.cuda.synchronize()
    assert all(torch.isfinite(p.grad).all() for p in v.parameters())


@pytest.mark.gpu
def test_position_bias_kernels_vs_oracle():
    """rgbnm_swin_cpb_fwd / _bwd (round 6: the continuous position bias and logit scale of every block in two launches per direction)
    against the oracle's restatement of swinv2.py:158-168 in fp64 autograd: bias, scale, and the gradients of cpb_mlp's two Linears
    and of logit_scale (one of them beyond the clamp at ln 100, where the gradient must be zero), for blocks of 3, 6 and 24 heads in
    ONE call; run twice: the same bits."""
    from rgb_no_more_amd import swinv2 as SW
    from oracle import swin_torch as ST
    heads = [3, 6, 24]
    blocks = []
    for i, h in enumerate(heads):
        blk = SW.SwinTransformerBlock(32 * h, (16, 16), h, 8, 0, 0.0, device=DEV)
        with torch.no_grad():
            blk.attn.cpb_mlp[0].weight.copy_(torch.from_numpy(detfill.normalish((512, 2), 300 + i)).to(DEV))
            blk.attn.cpb_mlp[0].bias.copy_(torch.from_numpy(detfill.normalish((512,), 310 + i)).to(DEV) * 0.5)
            blk.attn.cpb_mlp[2].weight.copy_(torch.from_numpy(detfill.normalish((h, 512), 320 + i)).to(DEV) * 0.1)
            ls = detfill.uniform((h, 1, 1), 330 + i, 1.0, 5.0).astype(np.float32)     # some beyond ln 100 = 4.605
            blk.attn.logit_scale.copy_(torch.from_numpy(ls).to(DEV))
        blocks.append(blk)

    def run():
        for b in blocks:
            b.zero_grad()
        outs = SW.model_bias_and_scale(blocks)
        gs = []
        loss = 0
        for i, (bias, scale) in enumerate(outs):
            gb = torch.from_numpy(detfill.normalish(tuple(bias.shape), 340 + i)).to(DEV)
            gsc = torch.from_numpy(detfill.normalish(tuple(scale.shape), 350 + i)).to(DEV)
            gs.append((gb, gsc))
            loss = loss + (bias * gb).sum() + (scale * gsc).sum()
        loss.backward()
        torch.cuda.synchronize()
        return outs, gs, [[p.grad.clone() for p in (b.attn.cpb_mlp[0].weight, b.attn.cpb_mlp[0].bias, b.attn.cpb_mlp[2].weight,
                                                      b.attn.logit_scale)] for b in blocks]

    outs, gs, grads = run()
    outs2, _, grads2 = run()
    for (b1, s1), (b2, s2) in zip(outs, outs2):
        assert torch.equal(b1, b2) and torch.equal(s1, s2)
    for g1, g2 in zip(grads, grads2):
        assert all(torch.equal(a, b) for a, b in zip(g1, g2))
    for i, blk in enumerate(blocks):
        a = blk.attn
        p = {"x.attn.cpb_mlp.0.weight": a.cpb_mlp[0].weight.detach().double().cpu().requires_grad_(True),
             "x.attn.cpb_mlp.0.bias": a.cpb_mlp[0].bias.detach().double().cpu().requires_grad_(True),
             "x.attn.cpb_mlp.2.weight": a.cpb_mlp[2].weight.detach().double().cpu().requires_grad_(True)}
        ls = a.logit_scale.detach().double().cpu().requires_grad_(True)
        # oracle/swin_torch.py position_bias in fp64, on the ORACLE's own table and index (not the product's buffers)
        tab = ST.coords_table(8).double().reshape(-1, 2)
        hid = torch.relu(tab @ p["x.attn.cpb_mlp.0.weight"].T + p["x.attn.cpb_mlp.0.bias"])
        t = hid @ p["x.attn.cpb_mlp.2.weight"].T
        bias_ref = 16.0 * torch.sigmoid(t[ST.position_index(8).reshape(-1)].reshape(64, 64, -1).permute(2, 0, 1))
        scale_ref = torch.clamp(ls, max=float(np.log(1.0 / 0.01))).exp().view(-1)
        gb, gsc = gs[i]
        ((bias_ref * gb.double().cpu()).sum() + (scale_ref * gsc.double().cpu()).sum()).backward()
        bias, scale = outs[i]
        assert (bias.double().cpu() - bias_ref.detach()).abs().max() < 2e-5
        assert ((scale.double().cpu() - scale_ref.detach()).abs() / scale_ref.detach()).max() < 2e-6
        refs = [p["x.attn.cpb_mlp.0.weight"].grad, p["x.attn.cpb_mlp.0.bias"].grad, p["x.attn.cpb_mlp.2.weight"].grad, ls.grad]
        for got, want, nm in zip(grads[i], refs, ("dW1", "db1", "dW2", "dls")):
            err = (got.double().cpu().view(-1) - want.view(-1)).norm() / (want.norm() + 1e-30)
            assert err < 2e-5, (heads[i], nm, err.item())
        assert (ls.grad.view(-1)[ls.detach().view(-1) > np.log(100.0)] == 0).all()
        assert (grads[i][3].view(-1).cpu()[ls.detach().view(-1) > np.log(100.0)] == 0).all()


@pytest.mark.gpu
def test_graph_replay_with_backward_wide_brackets_equals_the_eager_pass():
    """group_dw_backward + hold_reductions (round 6: every split-sum reduction of the backward as one launch) inside a HIP-graph capture:
    the buffers of the capture differ from the eager passes', the job tables are uploaded by copy nodes of the graph from page-locked
    records of their own, partial sums must live until the bracket's launch (a freed d(scale) partial buffer showed as ONE differing
    logit_scale gradient in two of five bench runs): three replays, every gradient the bits of the eager pass."""
    B = 32
    ws = torch.cuda.Stream()
    with torch.cuda.stream(ws):
        m, img, depths, heads, _ = _model("swt", DEV)
        nb = img // 8
        y = torch.from_numpy(detfill.normalish((B, 1, nb, nb, 8, 8), 291)).to(DEV).bfloat16()
        c = torch.from_numpy(detfill.normalish((B, 2, nb // 2, nb // 2, 8, 8), 292)).to(DEV).bfloat16()
        tgt = detfill.uniform((B, 1000), 293, 0.0, 1.0)
        tgt = torch.from_numpy(tgt / tgt.sum(1, keepdims=True)).to(DEV)
        m.eval()
        m.compute_dtype = torch.bfloat16
        m.group_dw_backward = m.hold_reductions = True

        def part():
            rg.cls_transforms.cross_entropy(m(y, c), tgt, grad_dtype=torch.bfloat16).backward()

        m.zero_grad(set_to_none=True)
        part()
        torch.cuda.synchronize()
        ref = {n: p.grad.clone() for n, p in m.named_parameters()}
        for _ in range(2):
            m.zero_grad(set_to_none=True)
            part()
        m.zero_grad(set_to_none=True)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=ws):
            part()
        for r in range(3):
            junk = torch.full((1 << 24,), float("nan"), device=DEV)      # whatever the allocator hands out between replays
            del junk
            g.replay()
            torch.cuda.synchronize()
            bad = [n for n, p in m.named_parameters() if not torch.equal(ref[n], p.grad)]
            assert not bad, (r, bad[:5])
        del g
