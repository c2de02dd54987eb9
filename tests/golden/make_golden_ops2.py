#!/usr/bin/env python
"""G16: the out-of-list RandAugment ops Invert / Solarize / FreqEnhance / Equalize through the reference's own dispatcher
(utils/custom_transforms.py:_apply_op_dct, with its per-op clamp) on seeded int16 coefficients.  Survey container only."""
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
    mg._stub_modules()
    sys.path.insert(0, mg.REF)
    import utils.custom_transforms as ctrans
    Y = detfill.integers((1, 28, 28, 8, 8), 201, -1024, 1016, np.int16)
    C = detfill.integers((2, 14, 14, 8, 8), 202, -1024, 1016, np.int16)
    Y[0, :, :, 0, 0] = detfill.integers((28, 28), 203, -900, 900, np.int16)
    Y[0, 3, 4] = -1024                                   # inverting -1024 overflows the clamp range: 1024 -> 1016
    out = {"Y": Y, "C": C}
    cases = [("Invert", 0.0), ("Solarize", 327.2), ("Solarize", -163.6), ("Solarize", 818.0), ("FreqEnhance", 0.27),
             ("FreqEnhance", -0.27), ("FreqEnhance", 0.9), ("Equalize", 0.0)]
    for k, (name, mag) in enumerate(cases):
        oy, oc = ctrans._apply_op_dct([torch.from_numpy(Y.copy()), torch.from_numpy(C.copy())], name, mag, None,
                                      [None, None], [None, None])
        out[f"case{k}_name"] = np.array(name)
        out[f"case{k}_mag"] = np.float64(mag)
        out[f"case{k}_Y"] = oy.numpy()
        out[f"case{k}_C"] = oc.numpy()
    out["ncases"] = np.int64(len(cases))
    np.savez_compressed(os.path.join(HERE, "g16_ops2.npz"), **out)
    print("G16 done", len(cases))


if __name__ == "__main__":
    main()
