"""CPU: the SwinV2 DCT oracle (oracle/swin_torch.py) against golden g15, generated from the reference model
(tests/golden/make_golden_swin.py): parameter surface, activations after the embedding / a plain-window block / a
shifted-window block, logits, loss and every gradient norm (torch autograd through the restatement)."""
import numpy as np
import pytest
import torch

from oracle import swin_torch as S
from rgb_no_more_amd import detfill

CASES = {"sw3": (128, [2, 2, 2], [3, 6, 12], 2), "swt": (256, [2, 2, 6, 2], [3, 6, 12, 24], 1)}


@pytest.mark.parametrize("tag", ["sw3", "swt"])
def test_swin_oracle_matches_reference_golden(golden, tag):
    g = golden("g15_swin.npz")
    img, depths, heads, B = CASES[tag]
    names = [str(n) for n in g[tag + "_names"]]
    shapes = S.param_shapes(depths, heads)
    assert sorted(shapes) == sorted(names)
    assert all(str(shapes[n]) == s for n, s in zip(names, g[tag + "_shapes"]))
    sd = S.fill_params({n: shapes[n] for n in names})
    p = {k: torch.from_numpy(v).requires_grad_(True) for k, v in sd.items()}
    nb = img // 8
    y = torch.from_numpy(detfill.normalish((B, 1, nb, nb, 8, 8), 171))
    c = torch.from_numpy(detfill.normalish((B, 2, nb // 2, nb // 2, 8, 8), 172))
    logits, inter = S.swin_forward(p, y, c, depths, heads, return_inter=True)
    for i in range(3):
        np.testing.assert_allclose(inter[i][:, ::97, ::7].detach().numpy(), g[f"{tag}_x{i}_slice"], atol=2e-5)
    np.testing.assert_allclose(logits.detach().numpy(), g[tag + "_logits"], atol=2e-5)
    tgt = detfill.uniform((B, 1000), 173, 0.0, 1.0)
    tgt = torch.from_numpy(tgt / tgt.sum(1, keepdims=True))
    loss = -(tgt * torch.log_softmax(logits, 1)).sum(1).mean()
    assert abs(loss.item() - float(g[tag + "_loss"])) < 1e-5
    if tag == "sw3":
        loss.backward()
        gn = np.array([p[n].grad.double().norm().item() for n in names])
        np.testing.assert_allclose(gn, g[tag + "_gradnorms"], rtol=2e-3, atol=1e-7)
