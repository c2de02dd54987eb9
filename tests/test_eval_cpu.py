"""evaluate_model (mirror of the reference eval.py:8-51): accuracy = correct / total over all ranks, loss = mean of the
per-batch means, averaged over ranks.  Model-agnostic host glue, so a tiny torch module stands in on CPU."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from rgb_no_more_amd.eval import evaluate_model


class Tiny(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.lin = torch.nn.Linear(8, 5)
        with torch.no_grad():
            self.lin.weight.copy_(torch.arange(40, dtype=torch.float32).reshape(5, 8).sin())
            self.lin.bias.zero_()

    def forward(self, y, cbcr):
        return self.lin(y.reshape(y.shape[0], -1)[:, :8] + cbcr.reshape(cbcr.shape[0], -1)[:, :8])


def _batches(seed, n):
    g = torch.Generator().manual_seed(seed)
    out = []
    for _ in range(n):
        y = torch.randn(6, 1, 2, 2, 2, 2, generator=g)
        c = torch.randn(6, 2, 1, 1, 2, 2, generator=g).repeat(1, 1, 1, 2, 1, 1)
        out.append(((y, c), torch.randint(0, 5, (6,), generator=g)))
    return out


def _expected(batches, model):
    ce = torch.nn.CrossEntropyLoss()
    corr = tot = 0
    losses = []
    for (y, c), lab in batches:
        o = model(y, c)
        corr += int((o.argmax(1) == lab).sum())
        tot += lab.numel()
        losses.append(float(ce(o, lab)))
    return corr, tot, sum(losses) / len(losses)


def test_single_process():
    m = Tiny()
    m.train()
    b = _batches(1, 3)
    acc, loss = evaluate_model(m, b, device="cpu")
    corr, tot, el = _expected(b, m)
    assert abs(acc - corr / tot) < 1e-12 and abs(loss - el) < 1e-6
    assert m.training                      # mode restored
    # flat (Y, CbCr, labels) tuples are accepted too
    acc2, _ = evaluate_model(m, [(y, c, lab) for (y, c), lab in b], device="cpu")
    assert acc2 == acc


def _worker(rank, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=2)
    m = Tiny()
    q.put((rank,) + evaluate_model(m, _batches(10 + rank, 2 + rank), device="cpu"))
    dist.destroy_process_group()


def test_two_ranks_gloo():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    m = Tiny()
    c0, t0, l0 = _expected(_batches(10, 2), m)
    c1, t1, l1 = _expected(_batches(11, 3), m)
    for _, acc, loss in res:
        assert abs(acc - (c0 + c1) / (t0 + t1)) < 1e-12
        assert abs(loss - (l0 + l1) / 2) < 1e-6
