"""ViT.defer_grad_reduction: the encoder blocks' gradient reductions held during the backward and run as ONE launch in front of
the patch embedding's backward (rgbnm.h rgbnm_reduce_hold_*) -- same summation order, so every gradient keeps its bits."""
import importlib

import pytest
import torch

rg = importlib.import_module("rgb-no-more_amd")
L = rg.lib

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def per_operation_kernels():
    """This file is about the per-operation kernels: the one-launch encoder forward / backward (tests/test_chain_*.py) is off."""
    lib = L.lib()
    lib.rgbnm_set_option(b"fwd_chain", 0)
    lib.rgbnm_set_option(b"bwd_chain", 0)
    yield
    lib.rgbnm_set_option(b"fwd_chain", 1)
    lib.rgbnm_set_option(b"bwd_chain", 1)


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


def test_rebuilt_model_uploads_its_own_job_table():
    """A model (and its arena) that dies hands its device addresses back to the caching allocator; the next model can get the
    same addresses for its gradient buffer, workspaces and job table.  The table must be uploaded again (the record of what a
    device table holds lives and dies with that table), not skipped because pointer and jobs look familiar."""
    import gc
    import test_fastpath_model as T
    for rep in range(3):
        m, sd, y, c, tgt = T.build("ti_d2_b64", torch.bfloat16)
        m.train()
        base, lb = _grads(m, y, c, tgt, torch.bfloat16)
        m.defer_grad_reduction = True
        held, lh = _grads(m, y, c, tgt, torch.bfloat16)
        held2, _ = _grads(m, y, c, tgt, torch.bfloat16)
        for n in base:
            assert torch.equal(base[n], held[n]) and torch.equal(base[n], held2[n]), (n, rep)
        del m, base, held, held2
        gc.collect()                                        # no empty_cache(): the next build recycles the freed blocks


def test_two_holding_models_in_one_backward_pass():
    """One bracket per host thread: the second model's head closes the first model's bracket (its held jobs run, its remaining
    blocks reduce at once) instead of dropping the jobs collected so far."""
    import test_fastpath_model as T
    m1, _, y, c, tgt = T.build("ti_d2_b64", torch.bfloat16)
    m2, _, _, _, _ = T.build("ti_d2_b64", torch.bfloat16)
    m1.train(), m2.train()

    def both():
        m1.zero_grad(set_to_none=True), m2.zero_grad(set_to_none=True)
        loss = (rg.cls_transforms.cross_entropy(m1(y, c), tgt, grad_dtype=torch.bfloat16) +
                rg.cls_transforms.cross_entropy(m2(y, c), tgt, grad_dtype=torch.bfloat16))
        loss.backward()
        torch.cuda.synchronize()
        return [{n: p.grad.detach().clone() for n, p in m.named_parameters()} for m in (m1, m2)]

    base = both()
    m1.defer_grad_reduction = m2.defer_grad_reduction = True
    for rep in range(2):
        held = both()
        for b, h in zip(base, held):
            for n in b:
                assert torch.equal(b[n], h[n]), (n, rep)


def test_flat_sync_refuses_a_frozen_patch_embedding():
    import torch.distributed as dist
    import test_fastpath_model as T
    if not dist.is_initialized():
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:29631", rank=0, world_size=1)
    try:
        m, _, y, c, tgt = T.build("ti_d2_b64", torch.bfloat16)
        m.train()
        rg.parallel.FlatGradSync(m, broadcast=False)
        for n, p in m.named_parameters():
            if n.startswith("patchembed."):
                p.requires_grad_(False)
        with pytest.raises(RuntimeError, match="patch-embedding"):
            m(y, c)
    finally:
        dist.destroy_process_group()
