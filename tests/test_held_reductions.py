"""ViT.defer_grad_reduction: the encoder blocks' gradient reductions held during the backward and run as ONE launch in front of
the patch embedding's backward (rgbnm.h rgbnm_reduce_hold_*) -- same summation order, so every gradient keeps its bits."""
import importlib

import pytest
import torch

rg = importlib.import_module("rgb-no-more_amd")
L = rg.lib

pytestmark = pytest.mark.gpu


def _grads(m, y, c, tgt, cdt):
    m.zero_grad(set_to_none=True)
    logits = m(y, c)
    rg.cls_transforms.cross_entropy(logits, tgt, grad_dtype=cdt).backward()
    torch.cuda.synchronize()
    return {n: p.grad.detach().clone() for n, p in m.named_parameters()}, logits.detach().clone()


@pytest.mark.parametrize("tag,cdt", [("ti_d2_b64", torch.bfloat16), ("ti_d12_b64", torch.bfloat16), ("ti_d2_b64", torch.float32),
                                     ("s_d2_b64", torch.bfloat16)])
def test_held_reductions_keep_every_bit(tag, cdt):
    import test_fastpath_model as T
    m, sd, y, c, tgt = T.build(tag, cdt)
    m.train()
    assert m.defer_grad_reduction is False
    base, lb = _grads(m, y, c, tgt, cdt)
    m.defer_grad_reduction = True
    for rep in range(3):                                   # first held pass allocates per-block workspaces and uploads the job table
        held, lh = _grads(m, y, c, tgt, cdt)
        assert torch.equal(lb, lh)
        for n in base:
            assert torch.equal(base[n], held[n]), (n, rep)
    m.defer_grad_reduction = False
    again, _ = _grads(m, y, c, tgt, cdt)
    for n in base:
        assert torch.equal(base[n], again[n]), n


def test_not_held_when_something_reads_block_gradients_early():
    import test_fastpath_model as T
    m, sd, y, c, tgt = T.build("ti_d2_b64", torch.bfloat16)
    m.train()
    m.defer_grad_reduction = True
    for n, p in m.named_parameters():                      # no patch-embedding backward node: nothing would close the bracket
        if n.startswith("patchembed."):
            p.requires_grad_(False)
    m.zero_grad(set_to_none=True)
    logits = m(y, c)
    st = logits.grad_fn.st
    rg.cls_transforms.cross_entropy(logits, tgt, grad_dtype=torch.bfloat16).backward()
    torch.cuda.synchronize()
    assert st.holding is False and st.arena.ws_blk is None
    assert all(torch.isfinite(p.grad).all() for p in m.parameters() if p.grad is not None)
