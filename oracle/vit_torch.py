"""ORACLE (test infrastructure, never shipped in the product path).

Plain-PyTorch fp32 CPU restatement of the reference's JPEG-ViT forward (models/plainvit.py, SURVEY.md
section 8a rows a13-a20) and of the train-step tail (a22).  Functional: parameters come in a dict keyed
exactly like the reference `state_dict()`, so the golden logits generated from the
reference (tests/golden/g11_model.npz) pin it.  Backward is obtained with torch.autograd on this
restatement (floating-point path => a torch fp32 reference is the prescribed oracle).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import math
import numpy as np
import torch
import torch.nn.functional as F

from . import dct_np


def conv_matrix(patch_size, dtype=torch.float32):
    """plainvit.py:19-48 patch2subblock: 16 -> A(8,2) [16x16]; 8 -> None."""
    if patch_size > 8:
        return torch.from_numpy(dct_np.conversion_matrix(8, patch_size // 8)).to(dtype)
    if patch_size == 8:
        return None
    return torch.from_numpy(dct_np.conversion_matrix(patch_size, 8 // patch_size)).to(dtype)


def subblock_features(y, cbcr, patch_size=16):
    """plainvit.py:200-216 (PatchEmbedding_DCT_Group.forward up to the projection), patch_size>=16:
    'b c (h pdh)(w pdw) p1 p2 -> b c h w (pdh p1)(pdw p2)', A.X.A^T, collapse 'b c h w i j -> b h w (c i j)',
    cat(Y, CbCr)."""
    B, _, H, W, _, _ = y.shape
    pd = patch_size // 8
    A = conv_matrix(patch_size, y.dtype)
    t = y.reshape(B, 1, H // pd, pd, W // pd, pd, 8, 8).permute(0, 1, 2, 4, 3, 6, 5, 7)
    t = t.reshape(B, 1, H // pd, W // pd, pd * 8, pd * 8)
    t = A @ t @ A.T
    yf = t.permute(0, 2, 3, 1, 4, 5).reshape(B, H // pd, W // pd, -1)
    pc = patch_size // 2
    assert pc == 8, "oracle covers patch_size 16 (chroma patch 8 => identity)"
    cf = cbcr.permute(0, 2, 3, 1, 4, 5).reshape(B, cbcr.shape[2], cbcr.shape[3], -1)
    return torch.cat([yf, cf], dim=3)


def sincos_table(h, w, e, dtype=torch.float32):
    """plainvit.py:97-116: cat(sin(w f), cos(w f), sin(h f), cos(h f)), f = exp(-k ln(1e4)/(e/4-1))."""
    hg, wg = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
    fr = torch.log(torch.tensor(10000, dtype=torch.int32)) / (e // 4 - 1)
    fr = torch.exp(-torch.arange(e // 4, dtype=dtype) * fr)
    ph = torch.einsum("p,f->pf", hg.flatten().to(dtype), fr)
    pw = torch.einsum("p,f->pf", wg.flatten().to(dtype), fr)
    return torch.cat((pw.sin(), pw.cos(), ph.sin(), ph.cos()), dim=-1)  # (h*w, e)


def patch_embed(p, y, cbcr, patch_size=16):
    feat = subblock_features(y, cbcr, patch_size)                       # (B, h, w, 384)
    B, h, w, _ = feat.shape
    if "patchembed.linearMix.weight" in p:
        # ver=2, use_subblock: PatchEmbedding_DCT_Separate_subblock (plainvit.py:280-352).  Y tile (256) and CbCr
        # blocks (2 x 64; chroma patch 8 => identity sub-block matrix) are projected separately, concatenated,
        # GELU'd ("mistakenly not added" before the concat in the reference: it acts on the concatenation),
        # mixed by a residual Linear, then the sin-cos table is added.
        fy = F.linear(feat[..., :256], p["patchembed.projection_Y.1.weight"], p["patchembed.projection_Y.1.bias"])
        fc = F.linear(feat[..., 256:], p["patchembed.projection_C.1.weight"], p["patchembed.projection_C.1.bias"])
        g = F.gelu(torch.cat([fy, fc], dim=3))
        x = F.linear(g, p["patchembed.linearMix.weight"], p["patchembed.linearMix.bias"]) + g
        e = x.shape[-1]
        x = x + sincos_table(h, w, e, x.dtype).view(1, h, w, e)
        return x.reshape(B, h * w, e)
    x = F.linear(feat, p["patchembed.projection.0.weight"], p["patchembed.projection.0.bias"])
    e = x.shape[-1]
    x = x + sincos_table(h, w, e, x.dtype).view(1, h, w, e)
    return x.reshape(B, h * w, e)


def attention(p, pre, x, num_heads, emb_size):
    """plainvit.py:445-464: interleaved '(h d qkv)' split; softmax(QK^T / sqrt(emb_size)); merge '(h d)'."""
    B, N, _ = x.shape
    qkv = F.linear(x, p[pre + "qkv.weight"], p[pre + "qkv.bias"])
    qkv = qkv.reshape(B, N, num_heads, -1, 3).permute(4, 0, 2, 1, 3)     # qkv b h n d
    q, k, v = qkv[0], qkv[1], qkv[2]
    energy = q @ k.transpose(-1, -2)
    att = torch.softmax(energy / (emb_size ** 0.5), dim=-1)
    out = (att @ v).permute(0, 2, 1, 3).reshape(B, N, -1)
    return F.linear(out, p[pre + "projection.weight"], p[pre + "projection.bias"])


def encoder_block(p, i, x, num_heads, emb_size):
    """plainvit.py:493-529: x += MHA(LN1(x)); x += FFB(LN2(x)); LN eps 1e-5; exact erf GELU."""
    a = f"encoder.{i}.0.fn."
    b = f"encoder.{i}.1.fn."
    h = F.layer_norm(x, (emb_size,), p[a + "eb_lrnorm1.weight"], p[a + "eb_lrnorm1.bias"], 1e-5)
    x = x + attention(p, a + "eb_mha.", h, num_heads, emb_size)
    h = F.layer_norm(x, (emb_size,), p[b + "eb_lrnorm2.weight"], p[b + "eb_lrnorm2.bias"], 1e-5)
    h = F.linear(h, p[b + "eb_ffb.0.weight"], p[b + "eb_ffb.0.bias"])
    h = F.gelu(h)
    h = F.linear(h, p[b + "eb_ffb.3.weight"], p[b + "eb_ffb.3.bias"])
    return x + h


def class_head(p, x, emb_size):
    """plainvit.py:542-557: LN -> token mean -> Linear -> tanh -> Linear."""
    h = F.layer_norm(x, (emb_size,), p["classhead.ch_lrnorm.weight"], p["classhead.ch_lrnorm.bias"], 1e-5)
    h = h.mean(dim=1)
    h = torch.tanh(F.linear(h, p["classhead.ch_linear1.weight"], p["classhead.ch_linear1.bias"]))
    return F.linear(h, p["classhead.ch_linear2.weight"], p["classhead.ch_linear2.bias"])


def vit_forward(p, y, cbcr, depth, num_heads, emb_size, patch_size=16, return_inter=False):
    """plainvit.py:601-611 with ver=1 (PatchEmbedding_DCT_Group)."""
    x0 = patch_embed(p, y, cbcr, patch_size)
    x = x0
    inter = [x0]
    for i in range(depth):
        x = encoder_block(p, i, x, num_heads, emb_size)
        inter.append(x)
    logits = class_head(p, x, emb_size)
    return (logits, inter) if return_inter else logits


def param_shapes(depth=12, emb=192, heads=3, n_classes=1000, patch_size=16, ver=1):
    """The reference state_dict keys/shapes (152 tensors for depth 12, ver=1; 156 for ver=2)."""
    inner = heads * 64
    fin = patch_size ** 2 + 2 * (patch_size // 2) ** 2
    if ver == 1:
        s = {"patchembed.projection.0.weight": (emb, fin), "patchembed.projection.0.bias": (emb,)}
    else:
        ey, ec = emb // 6 * 4, emb // 6 * 2
        s = {"patchembed.projection_Y.1.weight": (ey, patch_size ** 2), "patchembed.projection_Y.1.bias": (ey,),
             "patchembed.projection_C.1.weight": (ec, 2 * (patch_size // 2) ** 2),
             "patchembed.projection_C.1.bias": (ec,),
             "patchembed.linearMix.weight": (emb, emb), "patchembed.linearMix.bias": (emb,)}
    for i in range(depth):
        a, b = f"encoder.{i}.0.fn.", f"encoder.{i}.1.fn."
        s[a + "eb_lrnorm1.weight"] = (emb,)
        s[a + "eb_lrnorm1.bias"] = (emb,)
        s[a + "eb_mha.qkv.weight"] = (3 * inner, emb)
        s[a + "eb_mha.qkv.bias"] = (3 * inner,)
        s[a + "eb_mha.projection.weight"] = (emb, inner)
        s[a + "eb_mha.projection.bias"] = (emb,)
        s[b + "eb_lrnorm2.weight"] = (emb,)
        s[b + "eb_lrnorm2.bias"] = (emb,)
        s[b + "eb_ffb.0.weight"] = (4 * emb, emb)
        s[b + "eb_ffb.0.bias"] = (4 * emb,)
        s[b + "eb_ffb.3.weight"] = (emb, 4 * emb)
        s[b + "eb_ffb.3.bias"] = (emb,)
    s["classhead.ch_lrnorm.weight"] = (emb,)
    s["classhead.ch_lrnorm.bias"] = (emb,)
    s["classhead.ch_linear1.weight"] = (emb, emb)
    s["classhead.ch_linear1.bias"] = (emb,)
    s["classhead.ch_linear2.weight"] = (n_classes, emb)
    s["classhead.ch_linear2.bias"] = (n_classes,)
    return s


# ------------------------------------------------------------------------------------ a22 tail
def mixup(y, cbcr, target_onehot, lam0, lam1):
    """cls_transforms.py:163-176: roll-by-1 pairs; x*lam0 + roll(x)*lam1 (lam sorted descending)."""
    my = y * lam0 + y.roll(1, 0) * lam1
    mc = cbcr * lam0 + cbcr.roll(1, 0) * lam1
    mt = target_onehot * lam0 + target_onehot.roll(1, 0) * lam1
    return my, mc, mt


def soft_xent(logits, target):
    """torch.nn.CrossEntropyLoss() with probability targets (pipeline_utils.py:535), mean over batch."""
    return -(target * torch.log_softmax(logits, dim=-1)).sum(-1).mean()


def clip_adamw_wd_step(params, grads, m, v, step, lr, base_lr, wd, wd_mask, max_norm=1.0,
                       beta1=0.9, beta2=0.999, eps=1e-8):
    """train.py:163-165/170-172: clip_grad_norm_(max_norm) -> AdamW(weight_decay=0) -> WeightDecay
    (custom_optims.py:37-42: p -= (lr/base_lr)*wd*p on tensors with wd_mask).  In-place on float64/32
    numpy arrays; `step` is the 1-based Adam step count.  Returns the pre-clip total norm."""
    total = math.sqrt(sum(float((g.astype(np.float64) ** 2).sum()) for g in grads))
    coef = min(1.0, max_norm / (total + 1e-6))
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    for p, g, mm, vv, dec in zip(params, grads, m, v, wd_mask):
        g = g * np.float32(coef)
        mm[...] = beta1 * mm + (1 - beta1) * g
        vv[...] = beta2 * vv + (1 - beta2) * g * g
        denom = np.sqrt(vv) / math.sqrt(bc2) + eps
        p[...] = p - (lr / bc1) * (mm / denom)
        if dec:
            p[...] = p - ((lr / base_lr) * wd) * p
    return total
