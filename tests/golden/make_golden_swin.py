#!/usr/bin/env python
"""G15: the reference SwinV2 DCT model (models/swinv2.py, `--domain DCT`, patch 4, window 8) on detfill weights and
inputs.  Survey container only (imports /root/reference with the make_golden.py stubs + a timm.models.layers stub).
Two configurations: a 3-stage model on 128x128 inputs (depths 2,2,2: plain + shifted windows, two patch mergings, all
windows 8x8) and SwinV2-T itself (depths 2,2,6,2 on 256x256, batch 1)."""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import make_golden as mg  # noqa: E402
from rgb_no_more_amd import detfill  # noqa: E402
from oracle import swin_torch as S  # noqa: E402


def _timm_stub():
    timm = types.ModuleType("timm")
    tm = types.ModuleType("timm.models")
    tl = types.ModuleType("timm.models.layers")

    class DropPath(torch.nn.Module):            # drop_path_rate = 0 in the goldens: never instantiated with p > 0
        def __init__(self, p=0.0):
            super().__init__()
            self.p = p

        def forward(self, x):
            assert self.p == 0.0 or not self.training
            return x

    tl.DropPath = DropPath
    tl.to_2tuple = lambda v: (v, v) if not isinstance(v, (tuple, list)) else tuple(v)
    tl.trunc_normal_ = torch.nn.init.trunc_normal_
    timm.models, tm.layers = tm, tl
    sys.modules.update({"timm": timm, "timm.models": tm, "timm.models.layers": tl})


def main():
    torch.set_num_threads(8)
    mg._stub_modules()
    _timm_stub()
    sys.path.insert(0, mg.REF)
    import models.swinv2 as sw
    T = torch.from_numpy
    out = {}
    for tag, img, depths, heads, B in (("sw3", 128, [2, 2, 2], [3, 6, 12], 2), ("swt", 256, [2, 2, 6, 2], [3, 6, 12, 24], 1)):
        model = sw.SwinTransformerV2(img_size=img, patch_size=4, embed_dim=96, depths=depths, num_heads=heads,
                                     window_size=8, mlp_ratio=4.0, drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.0,
                                     qkv_bias=True, ape=False, patch_norm=True,
                                     pretrained_window_sizes=[0] * len(depths), device="cpu", pixel_space="dct")
        names = [n for n, _ in model.named_parameters()]
        shapes = {n: tuple(p.shape) for n, p in model.named_parameters()}
        sd = S.fill_params(shapes, base_seed=3)       # LayerNorm scales ~1, logit_scale ~ln 10
        model.load_state_dict({k: T(v) for k, v in sd.items()}, strict=False)
        model.train()
        nb = img // 8
        y = detfill.normalish((B, 1, nb, nb, 8, 8), 171)
        c = detfill.normalish((B, 2, nb // 2, nb // 2, 8, 8), 172)
        ty, tc = T(y), T(c)
        x0 = model.patch_embed(ty, tc)
        x1 = model.layers[0].blocks[0](x0.clone())
        x2 = model.layers[0].blocks[1](x1.clone())
        logits = model(ty, tc)
        tgt = detfill.uniform((B, 1000), 173, 0.0, 1.0)
        tgt = T(tgt / tgt.sum(1, keepdims=True))
        loss = torch.nn.CrossEntropyLoss()(logits, tgt)
        loss.backward()
        out[tag + "_names"] = np.array(names)
        out[tag + "_shapes"] = np.array([str(shapes[n]) for n in names])
        out[tag + "_buffers"] = np.array([n for n, _ in model.named_buffers()])
        out[tag + "_x0_slice"] = x0.detach()[:, ::97, ::7].numpy()
        out[tag + "_x1_slice"] = x1.detach()[:, ::97, ::7].numpy()
        out[tag + "_x2_slice"] = x2.detach()[:, ::97, ::7].numpy()
        out[tag + "_logits"] = logits.detach().numpy()
        out[tag + "_loss"] = np.float64(loss.item())
        out[tag + "_gradnorms"] = np.array([p.grad.double().norm().item() for _, p in model.named_parameters()])
        print("G15", tag, "loss", loss.item(), "params", len(names), "logit absmax", logits.abs().max().item())
    np.savez_compressed(os.path.join(HERE, "g15_swin.npz"), **out)


if __name__ == "__main__":
    main()
