"""parallel.DeferredFlatExchange (VERDICT r4 item 4 ii): the all-reduce of step i issued behind its backward, the NEXT step's data
stage queued while it is in flight, optimizer step i behind the wait -- against the blocking schedule (data stage, forward /
backward, all-reduce, optimizer; reference loop: train.py:145-176 under DDP, train.py:137).  Two ranks over gloo on CPU: the
parameters after every step, the reduced gradient every optimizer step saw and the losses must be the SAME BITS under both
schedules, on both ranks; the order in which data stage and optimizer step were queued must be the swapped one."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


N, STEPS = 4096 + 37, 6


def _data(rank, step):
    g = torch.Generator().manual_seed(1000 * step + rank)
    return torch.randn(N, generator=g), torch.randn(N, generator=g)


def _run_schedule(rank, deferred):
    """A flat parameter / gradient pair with a rank- and step-dependent quadratic loss and a momentum optimizer."""
    from rgb_no_more_amd.parallel import DeferredFlatExchange
    p = torch.linspace(-1, 1, N).clone()
    gflat = torch.zeros(N)
    mom = torch.zeros(N)
    log, seen = [], []

    def data_stage(i):
        log.append(("data", i))
        return _data(rank, i)

    def fwd_bwd(batch):
        x, t = batch
        r = p * x - t
        gflat.copy_(r * x)                      # d/dp of 0.5 * ||p x - t||^2
        return 0.5 * float((r * r).sum())

    def opt_step():
        log.append(("opt", len(seen)))
        seen.append(gflat.clone())
        mom.mul_(0.9).add_(gflat)
        p.add_(mom, alpha=-0.05)

    losses, params = [], []
    if deferred:
        ex = DeferredFlatExchange(lambda: gflat)
        for i in range(STEPS):
            batch = data_stage(i)
            if ex.finish(opt_step):
                params.append(p.clone())
            losses.append(fwd_bwd(batch))
            ex.issue()
            assert ex.pending
            with pytest.raises(RuntimeError):   # one exchange per step
                ex.issue()
        assert ex.finish(opt_step) and not ex.pending and not ex.finish(opt_step)
        params.append(p.clone())
        assert ex.collectives == STEPS
    else:
        for i in range(STEPS):
            batch = data_stage(i)
            losses.append(fwd_bwd(batch))
            dist.all_reduce(gflat, op=dist.ReduceOp.SUM)
            gflat.mul_(1.0 / dist.get_world_size())
            opt_step()
            params.append(p.clone())
    return losses, params, seen, log


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        a = _run_schedule(rank, deferred=False)
        b = _run_schedule(rank, deferred=True)
        ok = (a[0] == b[0] and len(a[1]) == len(b[1]) == STEPS and all(torch.equal(x, y) for x, y in zip(a[1], b[1]))
              and len(a[2]) == len(b[2]) == STEPS and all(torch.equal(x, y) for x, y in zip(a[2], b[2])))
        # the reduced gradient really is the mean over both ranks
        x0, t0 = _data(0, 0)
        x1, t1 = _data(1, 0)
        p0 = torch.linspace(-1, 1, N)
        want = 0.5 * ((p0 * x0 - t0) * x0 + (p0 * x1 - t1) * x1)
        ok = ok and torch.allclose(b[2][0], want, rtol=0, atol=1e-6)
        # queueing order: blocking = data_i, opt_i; deferred = data_0, data_1, opt_0, data_2, opt_1, ...
        want_log = [("data", 0)] + [e for i in range(1, STEPS) for e in (("data", i), ("opt", i - 1))] + [("opt", STEPS - 1)]
        ok = ok and b[3] == want_log and a[3] == [e for i in range(STEPS) for e in (("data", i), ("opt", i))]
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_deferred_exchange_equals_the_blocking_schedule_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p_ in procs:
        p_.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p_ in procs:
        p_.join(timeout=60)
    assert res == [(0, True), (1, True)], res
