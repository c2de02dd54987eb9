#!/usr/bin/env python
"""G14: the reference ViT with ver=2 (embed_type 2, PatchEmbedding_DCT_Separate_subblock, models/plainvit.py:280-352;
train.py's default embed type) on detfill weights / inputs.  Run in the survey container only (imports
/root/reference with the same stubs as make_golden.py); commits inputs-by-seed + outputs as g14_model_v2.npz."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import make_golden as mg  # noqa: E402
from rgb_no_more_amd import detfill  # noqa: E402


def main():
    torch.set_num_threads(4)
    mg._stub_modules()
    sys.path.insert(0, mg.REF)
    import models.plainvit as pvit
    T = torch.from_numpy
    out = {}
    for tag, emb, heads, depth, B in (("ti_d2_v2", 192, 3, 2, 2), ("s_d2_v2", 384, 6, 2, 2)):
        model = pvit.ViT(3, 16, emb, depth=depth, n_classes=1000, drop_p=0.0, device="cpu", num_heads=heads,
                         head_size=64, pixel_space="DCT", ver=2, use_subblock=True)
        shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
        sd = detfill.fill_state_dict(shapes, base_seed=1)
        model.load_state_dict({k: T(v) for k, v in sd.items()})
        model.train()
        y = detfill.normalish((B, 1, 28, 28, 8, 8), 71)
        c = detfill.normalish((B, 2, 14, 14, 8, 8), 72)
        tgt = detfill.uniform((B, 1000), 73, 0.0, 1.0)
        tgt = tgt / tgt.sum(1, keepdims=True)
        ty, tc, tt = T(y), T(c), T(tgt)
        x0 = model.patchembed(ty, tc)
        logits = model(ty, tc)
        loss = torch.nn.CrossEntropyLoss()(logits, tt)
        loss.backward()
        out[tag + "_names"] = np.array(list(shapes.keys()))
        out[tag + "_shapes"] = np.array([str(v) for v in shapes.values()])
        out[tag + "_x0_slice"] = x0.detach()[:, ::49, ::16].numpy()
        out[tag + "_logits"] = logits.detach().numpy()
        out[tag + "_loss"] = np.float64(loss.item())
        out[tag + "_gradnorms"] = np.array([p.grad.double().norm().item() for _, p in model.named_parameters()])
        named = dict(model.named_parameters())
        for nm in ("patchembed.projection_Y.1.weight", "patchembed.projection_C.1.bias", "patchembed.linearMix.weight"):
            out[tag + "_grad_" + nm] = named[nm].grad.reshape(-1)[::37].numpy().copy()
        print("G14", tag, "loss", loss.item(), "keys", len(shapes))
    np.savez_compressed(os.path.join(HERE, "g14_model_v2.npz"), **out)


if __name__ == "__main__":
    main()
