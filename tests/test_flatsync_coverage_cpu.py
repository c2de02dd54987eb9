"""parallel.FlatGradSync must exchange EVERY gradient element exactly once per step, whatever the model layout: the slices
it all-reduces are derived from the order in which the autograd nodes report finished parameter groups (head, blocks
depth-1 .. 0, patch embedding) and from the flat-buffer offsets.  Checked here on CPU for JPEG-Ti / JPEG-S and the three
patch-embedding variants by recording the slices of a single-process gloo group (no GPU: only naming + offsets matter)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist

import rgb_no_more_amd as rg
from rgb_no_more_amd.flatparams import FlatParamModule
from rgb_no_more_amd.parallel import FlatGradSync


@pytest.fixture(scope="module")
def gloo():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=0, world_size=1)
    yield
    dist.destroy_process_group()


CASES = [dict(emb=192, heads=3, depth=12, ver=1), dict(emb=384, heads=6, depth=12, ver=1), dict(emb=192, heads=3, depth=3, ver=2),
         dict(emb=192, heads=3, depth=2, ver=2, use_subblock=False), dict(emb=384, heads=6, depth=2, ver=3)]


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"E{c['emb']}d{c['depth']}v{c['ver']}{'' if c.get('use_subblock', True) else 'ns'}")
@pytest.mark.parametrize("bucket", [4 << 20, 1 << 16, 1 << 30])
def test_every_gradient_element_is_exchanged_exactly_once(gloo, monkeypatch, case, bucket):
    m = rg.ViT(3, 16, case["emb"], depth=case["depth"], n_classes=1000, drop_p=0.0, num_heads=case["heads"], head_size=64,
               pixel_space="DCT", ver=case["ver"], use_subblock=case.get("use_subblock", True))
    FlatParamModule._pack_parameters(m)          # the flat layout is pure torch; ViT._flatten (shadows etc.) needs a GPU
    sync = FlatGradSync(m, bucket_bytes=bucket)
    spans = []
    real = dist.all_reduce

    def rec(t, op=None, group=None, async_op=False):
        off = (t.data_ptr() - m._gflat.data_ptr()) // 4
        spans.append((off, off + t.numel()))
        return real(t, op=op, group=group, async_op=async_op)

    monkeypatch.setattr(dist, "all_reduce", rec)
    order = m.grad_ready_order()
    assert order[-1][1] and not any(last for _, last in order[:-1])
    reported = [n for names, _ in order for n in names]
    assert sorted(reported) == sorted(n for n, _ in m.named_parameters())        # every parameter is reported, once
    for names, last in order:
        sync.ready(m._gflat, names, last=last)
    assert not sync._handles and sync._pending is None                           # the last group flushed and waited
    spans.sort()
    covered = torch.zeros(m._total, dtype=torch.int32)
    for lo, hi in spans:
        covered[lo:hi] += 1
    assert int(covered.max()) == 1, "a gradient element was all-reduced twice"
    for n, p in m.named_parameters():
        o = m._offs[n]
        assert bool((covered[o:o + p.numel()] == 1).all()), n
    # what is not covered is alignment padding only (< 256 elements behind a tensor)
    assert int((covered == 0).sum()) <= 255 * len(m._offs)
    if bucket >= 1 << 30:
        assert len(spans) <= 3          # one huge bucket: the step collapses to (almost) one collective
    if bucket == 4 << 20 and case["depth"] == 12 and case["emb"] == 192:
        assert 4 <= len(spans) <= 8     # 22.6 MB of JPEG-Ti gradients in ~4 MB slices
    sync.detach()
