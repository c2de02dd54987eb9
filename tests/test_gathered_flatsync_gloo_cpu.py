"""parallel.GatheredFlatGradSync -- the flat, overlapped gradient exchange for models whose backward hands autograd SEPARATE
gradient tensors (SwinV2: bench.py --arch swinv2t at N > 1; reference wrap: torch DDP, train.py:137) -- on CPU, world_size 2 over
gloo, with a toy FlatParamModule of plain torch layers: buckets cover every parameter once, every rank ends with the average of
the local gradients in ONE flat buffer (optimizer zero-copy path), backward() returning means the gradients are final, unused
parameters are exchanged as zeros by wait(), a second backward without zero_grad is refused."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch import nn

from rgb_no_more_amd.flatparams import FlatParamModule
from rgb_no_more_amd.parallel import GatheredFlatGradSync


class Toy(FlatParamModule):
    def __init__(self):
        super().__init__()
        self.a = nn.Linear(7, 300)          # 2100 + 300 elements: segments are 256-aligned, so slices contain padding
        self.b_lrnorm = nn.LayerNorm(300)
        self.c = nn.Linear(300, 5)
        self.unused = nn.Linear(3, 3)       # never reached by forward: exchanged as zeros by wait()

    def forward(self, x, use_all=True):
        self._ensure_flat()
        if self._grad_sync is not None and torch.is_grad_enabled():
            self._grad_sync.begin_step()
        return self.c(self.b_lrnorm(torch.tanh(self.a(x))))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(3)
    m = Toy()
    ref = Toy()
    ref.load_state_dict(m.state_dict())
    with torch.no_grad():
        if rank == 1:
            for p in m.parameters():
                p.add_(0.5)                 # the constructor must broadcast rank 0's parameters
    for bucket_bytes in (64, 4096, 1 << 30):
        if m._grad_sync is not None:
            m._grad_sync.detach()
        sync = GatheredFlatGradSync(m, bucket_bytes=bucket_bytes)
        assert torch.equal(torch.cat([p.detach().reshape(-1) for p in m.parameters()]),
                           torch.cat([p.detach().reshape(-1) for p in ref.parameters()]))
        # buckets: every parameter exactly once, slices inside the flat buffer, disjoint
        names = [n for b in sync._buckets for n in b["names"]]
        assert sorted(names) == sorted(m._named) and len(names) == len(set(names))
        spans = sorted((b["lo"], b["hi"]) for b in sync._buckets)
        assert all(spans[i][1] <= spans[i + 1][0] for i in range(len(spans) - 1)) and spans[-1][1] <= m._gflat.numel()
        if bucket_bytes == 1 << 30:
            assert len(sync._buckets) == 1
        for step in range(2):
            m.zero_grad(set_to_none=True)
            torch.manual_seed(100 + 10 * step + rank)          # per-rank shard
            x = torch.randn(6, 7)
            before = sync.collectives
            m(x).square().mean().backward()
            # the bucket holding `unused` cannot complete during the backward: the end-of-backward callback flushes it, so the
            # gradients are final when backward() returns -- no wait() needed (what GradScaler.unscale_ /
            # clip_grad_norm_ between backward and step, train.py:159-166, rely on)
            assert sync.collectives - before == len(sync._buckets)
            assert not sync._handles
            assert m.flat_grad_base() == m._gflat.data_ptr()   # one flat buffer: the fused optimizer's zero-copy path
            g = torch.cat([p.grad.reshape(-1) for p in m.parameters()])
            ref.zero_grad(set_to_none=True)
            ref._ensure_flat()
            ref(x).square().mean().backward()
            local = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in ref.parameters()])
            ls = [torch.zeros_like(local) for _ in range(world)]
            dist.all_gather(ls, local)
            assert torch.allclose(g, (ls[0] + ls[1]) / 2, atol=1e-6)
            gs = [torch.zeros_like(g) for _ in range(world)]
            dist.all_gather(gs, g)
            assert torch.equal(gs[0], gs[1])
            assert float(m.unused.weight.grad.abs().max()) == 0.0
        # a second backward without zero_grad(set_to_none=True) is refused loudly
        try:
            m(x).square().mean().backward()
            raise AssertionError("accepted a second backward over attached gradients")
        except RuntimeError as e:
            assert "accumulation" in str(e)
    # a model whose every parameter gets a gradient: backward() alone leaves final gradients (no wait())
    m.zero_grad(set_to_none=True)
    sync.detach()
    del m.unused
    m._flat = None
    m._flatten()
    sync = GatheredFlatGradSync(m, bucket_bytes=2048)
    torch.manual_seed(500 + rank)
    x = torch.randn(6, 7)
    m(x).square().mean().backward()
    assert not sync._handles and all(b["done"] for b in sync._buckets)
    g = torch.cat([p.grad.reshape(-1) for p in m.parameters()])
    gs = [torch.zeros_like(g) for _ in range(world)]
    dist.all_gather(gs, g)
    assert torch.equal(gs[0], gs[1]) and float(g.abs().max()) > 0
    dist.barrier()
    dist.destroy_process_group()


def test_gathered_flat_exchange_world2_gloo():
    mp.spawn(_worker, args=(2, _free_port()), nprocs=2, join=True)
