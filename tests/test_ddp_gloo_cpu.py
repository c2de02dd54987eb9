"""N > 1 path on CPU: world_size 2 over gloo (no GPU).  The data-parallel path of this framework is
  per-rank shard of the batch (seed + rank)  ->  HIP forward/backward producing gradient VIEWS of one flat buffer
  ->  torch DDP bucketed all-reduce (RCCL on the GPU box, gloo here)  ->  fused flat optimizer on every rank.
What is HIP-free in that chain is exercised here with a toy module that uses the very same FlatParamModule machinery
as rgb_no_more_amd.plainvit.ViT: gradient views survive DDP's in-place all-reduce (optimizer zero-copy path), ranks
end with identical averaged gradients, and the max-over-ranks timing reduction of bench.py."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch import nn

from rgb_no_more_amd.flatparams import FlatParamModule


class _Fn(torch.autograd.Function):
    """y = (x @ W^T + b) * g + s with ALL parameter gradients written into views of the module's flat gradient
    buffer (exactly what the HIP backward kernels do)."""

    @staticmethod
    def forward(ctx, x, m, gbuf, w, b, g, s):
        z = x @ w.t() + b
        ctx.m, ctx.gbuf = m, gbuf
        ctx.save_for_backward(x, w, z, g)
        return z * g + s

    @staticmethod
    def backward(ctx, dy):
        x, w, z, g = ctx.saved_tensors
        m, gbuf = ctx.m, ctx.gbuf
        gw, gb = m._gview(gbuf, "lin.weight"), m._gview(gbuf, "lin.bias")
        gg, gs = m._gview(gbuf, "x_lrnorm.weight"), m._gview(gbuf, "x_lrnorm.bias")
        dz = dy * g
        gw.copy_(dz.t() @ x)
        gb.copy_(dz.sum(0))
        gg.copy_((dy * z).sum(0))
        gs.copy_(dy.sum(0))
        if m._grad_sync is not None:     # same hook as the HIP autograd nodes (plainvit.py)
            m._grad_sync.ready(gbuf, ["x_lrnorm.weight", "x_lrnorm.bias"])
            m._grad_sync.ready(gbuf, ["lin.weight", "lin.bias"], last=True)
        return dz @ w, None, None, gw, gb, gg, gs


class Toy(FlatParamModule):
    def __init__(self):
        super().__init__()
        self.lin = nn.Linear(5, 3)
        self.x_lrnorm = nn.LayerNorm(3)      # only a parameter holder (scale / shift), like the holders in plainvit.ViT

    def forward(self, x):
        self._ensure_flat()
        if self._grad_sync is not None and torch.is_grad_enabled():
            self._grad_sync.begin_step()     # same hook as plainvit.ViT.forward
        n = self._named
        return _Fn.apply(x, self, self._grad_buffer(), n["lin.weight"], n["lin.bias"], n["x_lrnorm.weight"],
                         n["x_lrnorm.bias"])


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(7)                       # same init on every rank (DDP would broadcast rank 0 anyway)
    m = Toy()
    m._ensure_flat()
    assert all(p.data_ptr() == m._flat.data_ptr() + m._offs[n] * 4 for n, p in m.named_parameters())
    assert m._wd_flags.tolist() == [1, 0, 0, 0]          # name filter: only lin.weight decays
    ddp = nn.parallel.DistributedDataParallel(m, bucket_cap_mb=1)
    torch.manual_seed(1234 + rank)                       # per-rank data shard (train.py:119)
    x = torch.randn(4, 5)
    y = ddp(x)
    y.square().mean().backward()
    # gradients are still views of ONE flat buffer after DDP's all-reduce -> the fused optimizer takes its zero-copy path
    base = m.flat_grad_base()
    assert base is not None and base == m._gflat.data_ptr()
    g = torch.cat([p.grad.reshape(-1) for p in m.parameters()])
    gathered = [torch.zeros_like(g) for _ in range(world)]
    dist.all_gather(gathered, g)
    assert torch.equal(gathered[0], gathered[1])         # every rank holds the same averaged gradient
    # reference: average of the per-rank local gradients
    m2 = Toy()
    m2.load_state_dict(m.state_dict())
    m2._ensure_flat()
    m2(x).square().mean().backward()
    local = torch.cat([p.grad.reshape(-1) for p in m2.parameters()])
    ls = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(ls, local)
    assert torch.allclose(g, (ls[0] + ls[1]) / 2, atol=1e-6)
    # a second backward without zero_grad must accumulate, not clobber (fresh buffer because .grad aliases _gflat)
    ddp(x).square().mean().backward()
    g2 = torch.cat([p.grad.reshape(-1) for p in m.parameters()])
    assert torch.allclose(g2, 2 * g, atol=1e-5)
    # ---- the zero-copy exchange used by bench.py at N > 1: parallel.FlatGradSync all-reduces slices of the flat
    # gradient buffer in place, launched from the autograd node (here: the toy node, like plainvit's nodes)
    from rgb_no_more_amd.parallel import FlatGradSync
    m3 = Toy()
    m3.load_state_dict(m.state_dict())
    with torch.no_grad():
        if rank == 1:                                    # diverge rank 1 on purpose: the ctor must broadcast rank 0
            for p in m3.parameters():
                p.add_(1.0)
    sync = FlatGradSync(m3, bucket_bytes=16)             # tiny buckets: several collectives per backward
    assert torch.equal(torch.cat([p.detach().reshape(-1) for p in m3.parameters()]),
                       torch.cat([p.detach().reshape(-1) for p in m.parameters()]))
    m3(x).square().mean().backward()
    # contract: when backward() returns the gradients are final (the last autograd node waited for every collective),
    # so train.py's gradscaler.unscale_ / clip_grad_norm_ (train.py:160-163) see REDUCED gradients -- no sync.wait() here
    assert not sync._handles and sync._pending is None
    assert sync.collectives >= 2
    g3 = torch.cat([p.grad.reshape(-1) for p in m3.parameters()])
    assert torch.allclose(g3, g, atol=1e-6)              # same averaged gradient as torch DDP
    assert m3.flat_grad_base() == m3._gflat.data_ptr()   # still one flat buffer: optimizer zero-copy path
    sync.wait()                                          # idempotent (the fused optimizer still calls it)
    assert torch.allclose(torch.cat([p.grad.reshape(-1) for p in m3.parameters()]), g, atol=1e-6)
    # gradient accumulation (a second backward while .grad is still attached) is refused loudly, not silently unsynced
    try:
        m3(x).square().mean().backward()
        raise AssertionError("FlatGradSync accepted a second backward without zero_grad")
    except RuntimeError as e:
        assert "accumulation" in str(e)
    # a backward that died half-way leaves a collective in flight: the next forward drains it and the step is right
    m3.zero_grad(set_to_none=True)
    sync.ready(m3._gflat, ["x_lrnorm.weight", "x_lrnorm.bias"])
    sync.flush()
    assert sync._handles
    m3(x).square().mean().backward()
    assert not sync._handles
    g4 = torch.cat([p.grad.reshape(-1) for p in m3.parameters()])
    assert torch.allclose(g4, g, atol=1e-6)
    # bench.py re-times the step under different slice sizes before the timed region (overlapped slices vs one all-reduce
    # after the backward) by changing bucket_elems between steps: same gradients, and the huge bucket collapses the step to
    # at most a few collectives (ranges separated by more than alignment padding are not merged)
    before = sync.collectives
    sync.bucket_elems = 1 << 60
    m3.zero_grad(set_to_none=True)
    m3(x).square().mean().backward()
    assert not sync._handles and sync._pending is None and 1 <= sync.collectives - before <= 3
    assert torch.allclose(torch.cat([p.grad.reshape(-1) for p in m3.parameters()]), g, atol=1e-6)
    sync.detach()
    # bench.py's timing reduction: MAX over ranks
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert t.item() == float(world)
    dist.barrier()
    if rank == 0:
        out.put("ok")
    dist.destroy_process_group()


def test_ddp_world2_gloo_flat_gradients():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0, f"worker exited with {p.exitcode}"
    assert q.get(timeout=5) == "ok"
