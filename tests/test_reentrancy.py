"""The C-ABI library is re-entrant (include/rgbnm.h conventions; SURVEY.md 8b "Threading: re-entrant; all work enqueued on
the passed stream"): two models driven from two host threads on two HIP streams must produce bit-identical logits and
gradients to the same work run sequentially.  Before round 2 the deferred dW / reduction queues were process-global, so
a concurrent call could interleave into, or flush, another call's queue (silently wrong gradients)."""
import threading

import numpy as np
import pytest
import torch

import rgb_no_more_amd as rg
from rgb_no_more_amd import detfill

pytestmark = pytest.mark.gpu
DEV = "cuda"


def make(seed, B, depth=2):
    m = rg.ViT(3, 16, 192, depth=depth, n_classes=1000, drop_p=0.0, device=DEV, num_heads=3, head_size=64,
               pixel_space="DCT", ver=1, use_subblock=True)
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    m.load_state_dict({k: torch.from_numpy(v) for k, v in detfill.fill_state_dict(shapes, base_seed=seed).items()})
    m.compute_dtype = torch.bfloat16
    y = torch.from_numpy(detfill.normalish((B, 1, 28, 28, 8, 8), 70 + seed)).to(DEV)
    c = torch.from_numpy(detfill.normalish((B, 2, 14, 14, 8, 8), 80 + seed)).to(DEV)
    lab = torch.from_numpy(detfill.integers((B,), 90 + seed, 0, 998, np.int64)).to(DEV)
    return m, y, c, lab


def step(m, y, c, lab):
    m.zero_grad(set_to_none=True)
    logits = m(y, c)
    rg.cls_transforms.cross_entropy(logits, lab, grad_dtype=torch.bfloat16).backward()
    return logits.detach().clone(), torch.cat([p.grad.reshape(-1) for p in m.parameters()]).clone()


@pytest.mark.parametrize("B", [48, 4])      # 48: grouped dW launches + deferred reductions + fused LN epilogues; 4: generic
def test_two_threads_two_streams_bitwise_equal_sequential(B):
    jobs = [make(1, B), make(2, B)]
    want = [step(*j) for j in jobs]
    torch.cuda.synchronize()
    results = [[None] * 6 for _ in jobs]
    errors = []
    barrier = threading.Barrier(len(jobs))

    def worker(k):
        try:
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                for it in range(6):
                    barrier.wait()                       # start every iteration together: maximal interleaving
                    results[k][it] = step(*jobs[k])
                s.synchronize()
        except Exception as e:  # noqa: BLE001
            errors.append(e)
            barrier.abort()

    th = [threading.Thread(target=worker, args=(k,)) for k in range(len(jobs))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
    torch.cuda.synchronize()
    for k in range(len(jobs)):
        for it in range(6):
            lg, gr = results[k][it]
            assert torch.equal(lg, want[k][0]), (k, it, "logits")
            assert torch.equal(gr, want[k][1]), (k, it, "gradients")


def test_train_and_eval_model_interleaved_on_side_stream():
    """An eval forward of a second model on a side stream while the first model trains (train.py + eval.py in one
    process) leaves the training gradients untouched."""
    m, y, c, lab = make(1, 48)
    e, ye, ce, _ = make(3, 32)
    want_l, want_g = step(m, y, c, lab)
    with torch.no_grad():
        want_e = e(ye, ce).clone()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    stop = threading.Event()
    got_e = []

    def evaluator():
        with torch.cuda.stream(side), torch.no_grad():
            while not stop.is_set():
                got_e.append(e(ye, ce).clone())
            side.synchronize()

    t = threading.Thread(target=evaluator)
    t.start()
    try:
        for _ in range(8):
            lg, gr = step(m, y, c, lab)
            assert torch.equal(lg, want_l) and torch.equal(gr, want_g)
    finally:
        stop.set()
        t.join()
    torch.cuda.synchronize()
    assert got_e and all(torch.equal(g, want_e) for g in got_e)
