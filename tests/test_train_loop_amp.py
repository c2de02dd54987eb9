"""The reference's UNCHANGED training step driving the HIP model (train.py:146-176), AMP branch included.

BASELINE's bf16 configs reach the model only through `cfg.TRAIN.AMP`:
    autocast(bf16) -> torch CrossEntropyLoss -> GradScaler(1.6, 0.625, 600).scale(loss).backward() -> unscale_(optimizer)
    -> clip_grad_norm_(1) -> scaler.step(AdamW) -> scaler.step(WeightDecay) -> scaler.update() -> clip_gradscaler
with warm-up LR copied into the weight decayer (pipeline_utils.py:90-103, 399-409, 535-541).  Everything below except the
model class and `WeightDecay` is stock torch, exactly the objects train.py builds.  The oracle runs the same three steps in
fp32 without a scaler (loss scaling is a power of two: mathematically neutral).
"""
import copy

import numpy as np
import pytest
import torch

import rgb_no_more_amd as rg
from rgb_no_more_amd import detfill
from oracle import vit_torch as V

pytestmark = pytest.mark.gpu
DEV = "cuda"
LR, WD, WARMUP = 1e-3, 0.05, 5


def build(B=8, depth=2):
    m = rg.ViT(3, 16, 192, depth=depth, n_classes=1000, drop_p=0.0, device=DEV, num_heads=3, head_size=64,
               pixel_space="DCT", ver=1, use_subblock=True)
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    sd = detfill.fill_state_dict(shapes, base_seed=1)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    y = torch.from_numpy(detfill.normalish((B, 1, 28, 28, 8, 8), 71)).to(DEV)
    c = torch.from_numpy(detfill.normalish((B, 2, 14, 14, 8, 8), 72)).to(DEV)
    t = detfill.uniform((B, 1000), 73, 0.0, 1.0)
    tgt = torch.from_numpy(t / t.sum(1, keepdims=True)).to(DEV)
    return m, sd, y, c, tgt


def clip_gradscaler(gradscaler, scale_max=2 ** 18, scale_min=2 ** (-4)):
    """pipeline_utils.py:399-409 semantics."""
    if gradscaler._scale > scale_max:
        gradscaler._scale = torch.tensor(scale_max).to(gradscaler._scale)
    if gradscaler._scale < scale_min:
        gradscaler._scale = torch.tensor(scale_min).to(gradscaler._scale)


def reference_objects(m):
    """pipeline_utils.py:535-541."""
    criterion = torch.nn.CrossEntropyLoss()
    optimizer = torch.optim.AdamW(m.parameters(), lr=LR, weight_decay=0, eps=1e-8)
    weight_decayer = rg.custom_optims.WeightDecay(
        [p for n, p in m.named_parameters() if (".weight" in n) and ("lrnorm" not in n)], lr=LR, weight_decay=WD)
    gradscaler = torch.amp.GradScaler("cuda", growth_factor=1.6, backoff_factor=0.625, growth_interval=600)
    return criterion, optimizer, weight_decayer, gradscaler


def train_py_step(m, y, c, tgt, criterion, optimizer, weight_decayer, gradscaler, itr, amp=True):
    """train.py:146-172, verbatim structure."""
    optimizer.zero_grad()
    weight_decayer.zero_grad()
    if itr < WARMUP:
        for g in optimizer.param_groups:
            g["lr"] = LR * (itr + 1) / WARMUP
        for g in weight_decayer.param_groups:
            g["lr"] = optimizer.param_groups[0]["lr"]
    with torch.autocast("cuda", enabled=amp, dtype=torch.bfloat16):
        outputs = m(y, c)
        loss = criterion(outputs, tgt)
    if amp:
        gradscaler.scale(loss).backward()
        gradscaler.unscale_(optimizer)
        torch.nn.utils.clip_grad_norm_(m.parameters(), max_norm=1)
        gradscaler.step(optimizer)
        gradscaler.step(weight_decayer)
        gradscaler.update()
        clip_gradscaler(gradscaler)
    else:
        loss.backward()
        torch.nn.utils.clip_grad_norm_(m.parameters(), max_norm=1)
        optimizer.step()
        weight_decayer.step()
    return loss.item()


def oracle_steps(sd, y, c, tgt, nsteps, depth=2):
    names = list(sd.keys())
    p = {k: torch.from_numpy(v.copy()).requires_grad_(True) for k, v in sd.items()}
    mm = [np.zeros_like(sd[k]) for k in names]
    vv = [np.zeros_like(sd[k]) for k in names]
    mask = [(".weight" in n) and ("lrnorm" not in n) for n in names]
    losses = []
    for step in range(1, nsteps + 1):
        lr = LR * step / WARMUP if step - 1 < WARMUP else LR
        for k in names:
            p[k].grad = None
        lo = V.soft_xent(V.vit_forward(p, y, c, depth, 3, 192), tgt)
        lo.backward()
        losses.append(lo.item())
        V.clip_adamw_wd_step([p[k].detach().numpy() for k in names], [p[k].grad.numpy() for k in names], mm, vv, step,
                             lr, LR, WD, mask)
    return losses, {k: p[k].detach().numpy() for k in names}


@pytest.mark.parametrize("amp", [True, False])
def test_unchanged_train_py_step_tracks_oracle(amp):
    m, sd, y, c, tgt = build()
    criterion, optimizer, weight_decayer, gradscaler = reference_objects(m)
    m.train()
    losses = [train_py_step(m, y, c, tgt, criterion, optimizer, weight_decayer, gradscaler, i, amp) for i in range(3)]
    ol, ow = oracle_steps(sd, y.cpu(), c.cpu(), tgt.cpu(), 3)
    got = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
    diffs = np.concatenate([np.abs(got[k] - ow[k]).reshape(-1) for k in ow])
    print(f"amp={amp}: losses {losses} oracle {ol}; |w - w_oracle| median {np.median(diffs):.3e} max {diffs.max():.3e}; "
          f"scale {gradscaler.get_scale() if amp else None}")
    tol = 2e-2 if amp else 3e-5
    for a, b in zip(losses, ol):
        assert abs(a - b) < tol, (losses, ol)
    if amp:
        assert gradscaler.get_scale() == 65536.0          # no inf/nan step was skipped
        assert np.median(diffs) < 5e-5 and diffs.max() < 6.5e-3      # <= the 3 sign-like Adam steps taken (sum lr = 1.2e-3 * ...)
    else:
        assert np.median(diffs) < 1e-6 and diffs.max() < 2e-3


def test_amp_inf_step_is_skipped_and_scale_backs_off():
    """A non-finite gradient must make GradScaler skip BOTH optimizers and back the scale off (x0.625), and the
    model must keep training afterwards -- i.e. the flat gradient views behave like ordinary .grad tensors."""
    m, sd, y, c, tgt = build()
    criterion, optimizer, weight_decayer, gradscaler = reference_objects(m)
    m.train()
    train_py_step(m, y, c, tgt, criterion, optimizer, weight_decayer, gradscaler, 0)
    before = {k: v.detach().clone() for k, v in m.state_dict().items()}
    ybad = y.clone()
    ybad[0, 0, 0, 0, 0, 0] = float("inf")
    train_py_step(m, ybad, c, tgt, criterion, optimizer, weight_decayer, gradscaler, 1)
    assert gradscaler.get_scale() == 65536.0 * 0.625
    for k, v in m.state_dict().items():
        assert torch.equal(v, before[k]), k
    l2 = train_py_step(m, y, c, tgt, criterion, optimizer, weight_decayer, gradscaler, 2)
    assert np.isfinite(l2)
    assert any(not torch.equal(v, before[k]) for k, v in m.state_dict().items())


def _fused_steps(m, opt, y, c, tgt, n, cdt=torch.float32):
    m.compute_dtype = cdt
    out = []
    for _ in range(n):
        opt.zero_grad(set_to_none=True)
        loss = rg.cls_transforms.cross_entropy(m(y, c), tgt, grad_dtype=cdt)
        loss.backward()
        opt.step()
        out.append(loss.item())
    return out


def test_fused_optimizer_checkpoint_resume_is_bit_exact():
    """save_ckpt / load_checkpoint (train.py:195, pipeline_utils.py:490-580): model + optimizer state saved after 2 steps
    and restored into fresh objects must continue exactly like the uninterrupted run."""
    m, sd, y, c, tgt = build()
    opt = rg.custom_optims.FusedClipAdamWWD(m, lr=LR, eps=1e-8, weight_decay=WD, max_norm=1.0)
    _fused_steps(m, opt, y, c, tgt, 2)
    ck_model = copy.deepcopy(m.state_dict())
    ck_opt = copy.deepcopy(opt.state_dict())
    assert len(ck_opt["state"]) == len(list(m.parameters()))
    st0 = ck_opt["state"][0]
    assert set(st0) == {"step", "exp_avg", "exp_avg_sq"} and float(st0["step"]) == 2.0
    assert float(st0["exp_avg"].abs().sum()) > 0
    cont = _fused_steps(m, opt, y, c, tgt, 2)
    want = {k: v.detach().clone() for k, v in m.state_dict().items()}

    m2, _, _, _, _ = build()
    m2.load_state_dict(ck_model)
    opt2 = rg.custom_optims.FusedClipAdamWWD(m2, lr=LR, eps=1e-8, weight_decay=WD, max_norm=1.0)
    opt2.load_state_dict(ck_opt)
    resumed = _fused_steps(m2, opt2, y, c, tgt, 2)
    assert resumed == cont
    for k, v in m2.state_dict().items():
        assert torch.equal(v, want[k]), k
    # without the optimizer state the run diverges (this is what silently happened before)
    m3, _, _, _, _ = build()
    m3.load_state_dict(ck_model)
    opt3 = rg.custom_optims.FusedClipAdamWWD(m3, lr=LR, eps=1e-8, weight_decay=WD, max_norm=1.0)
    assert _fused_steps(m3, opt3, y, c, tgt, 2) != cont


def test_fused_optimizer_loads_a_torch_adamw_checkpoint():
    """The state layout is torch.optim.AdamW's: a checkpoint written by the reference's optimizer resumes on the fused one."""
    m, sd, y, c, tgt = build()
    criterion, optimizer, weight_decayer, gradscaler = reference_objects(m)
    m.compute_dtype = torch.float32
    for i in range(2):
        train_py_step(m, y, c, tgt, criterion, optimizer, weight_decayer, gradscaler, WARMUP + i, amp=False)
    ck_model, ck_opt = copy.deepcopy(m.state_dict()), copy.deepcopy(optimizer.state_dict())
    train_py_step(m, y, c, tgt, criterion, optimizer, weight_decayer, gradscaler, WARMUP + 2, amp=False)
    want = {k: v.detach().clone() for k, v in m.state_dict().items()}

    m2, _, _, _, _ = build()
    m2.load_state_dict(ck_model)
    opt2 = rg.custom_optims.FusedClipAdamWWD(m2, lr=LR, eps=1e-8, weight_decay=WD, max_norm=1.0)
    opt2.load_state_dict(ck_opt)
    assert opt2._step == 2
    _fused_steps(m2, opt2, y, c, tgt, 1)
    worst = max((m2.state_dict()[k] - want[k]).abs().max().item() for k in want)
    print(f"torch AdamW checkpoint -> fused optimizer, one more step: max |dw| = {worst:.3e}")
    assert worst < 2e-6
