"""Data-parallel gradient exchange on the FLAT gradient buffer (pure data parallelism, one all-reduce of
all gradients per step, overlapped with backward; reference: torch DDP, train.py:137).

torch DDP works on this model unchanged (leaf parameters with ordinary .grad; tests/test_ddp_gloo_cpu.py) but copies
every gradient into its buckets and back: 2 x 152 small kernels per step, visible next to a 6 ms step.  The HIP
backward already writes all gradients into ONE fp32 buffer in reverse parameter order (head, block 11 .. 0, patch
embedding), so `FlatGradSync` all-reduces contiguous slices of that buffer in place (RCCL over xGMI via
torch.distributed "nccl"; gloo on CPU in the tests), launched from the autograd nodes as soon as a slice is final:
the exchange of block i overlaps the backward of blocks i-1 .. 0.  No copies, ~7 collectives of >= 4 MB per step.

    sync = FlatGradSync(model)            # broadcasts rank 0's parameters, hooks the model
    loss.backward()                       # returns with every collective ordered before later work on the stream
    opt.step()                            # any consumer: GradScaler.unscale_, clip_grad_norm_, AdamW, FusedClipAdamWWD

Ordering contract: the LAST autograd node of the model (patch embedding) flushes the final slice and makes the backward
stream wait for every outstanding collective, so once `backward()` has returned the gradients are final in stream order
-- `gradscaler.unscale_` / `clip_grad_norm_` / the inf check of train.py:159-166 see reduced gradients, exactly as with DDP.
The overlap is untouched (nothing but the optimizer runs after the last node).  Not supported, and refused loudly:
gradient accumulation over several backward passes (the second pass writes a fresh buffer that autograd accumulates
into .grad while a collective would still own it) -- use torch DDP for that.  A backward that died half-way leaves
collectives in flight; they are drained at the next forward (`begin_step`).  A frozen patch embedding (no last node) is
refused at the forward.
"""
import os

import torch
import torch.distributed as dist

_DEBUG = bool(os.environ.get("RGBNM_FLATSYNC_DEBUG"))


class FlatGradSync:
    def __init__(self, module, process_group=None, bucket_bytes=4 << 20, broadcast=True):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.module, self.pg = module, process_group
        self.world = dist.get_world_size(process_group)
        self.bucket_elems = max(1, bucket_bytes // 4)
        module._ensure_flat()
        if broadcast:
            dist.broadcast(module._flat, src=0, group=process_group)     # train.py relies on DDP doing this
        backend = dist.get_backend(process_group)
        self._avg = backend == "nccl"         # RCCL has ReduceOp.AVG; gloo sums and we scale
        self._handles, self._pending, self._gbuf = [], None, None
        self.collectives = 0
        module._grad_sync = self

    # ---- called from the autograd nodes (plainvit.py) right after a backward kernel group has written `names`
    def begin_step(self):
        """Called by the model at the start of a forward that will be differentiated: nothing of an earlier backward
        may still be in flight when its kernels start rewriting the flat gradient buffer."""
        m = self.module
        if not all(m._named[n].requires_grad for n in getattr(m, "_pe_names", ())):
            # the patch embedding's backward node is the one that flushes the last slice and orders the collectives before
            # backward() returns; with frozen patch-embedding parameters that node does not exist
            raise RuntimeError("FlatGradSync needs trainable patch-embedding parameters (its backward node closes the "
                               "exchange); for a frozen patch embedding use torch DDP")
        if self._handles or self._pending is not None:
            self._pending = None
            for h, _seg in self._handles:
                h.wait()                      # stale gradients: ordered, not rescaled -- they are about to be overwritten
            self._handles.clear()
        self._gbuf = None

    def ready(self, gbuf, names, last=False):
        m = self.module
        if gbuf.data_ptr() != m._gflat.data_ptr():
            raise RuntimeError("FlatGradSync: this backward writes a side gradient buffer, i.e. gradients of an earlier "
                               "backward are still attached (gradient accumulation / zero_grad(set_to_none=False)); "
                               "the flat exchange supports one backward per step -- use torch DDP for accumulation")
        lo = min(m._offs[n] for n in names)
        hi = max(m._offs[n] + _numel(m._shapes[n]) for n in names)
        if self._gbuf is not None and self._gbuf.data_ptr() != gbuf.data_ptr():
            self.flush()
        self._gbuf = gbuf
        if self._pending is not None and (hi == self._pending[0] or lo == self._pending[1] or
                                          _overlap_or_gap_is_padding(m, (lo, hi), self._pending)):
            self._pending = (min(lo, self._pending[0]), max(hi, self._pending[1]))
        else:
            self.flush()
            self._pending = (lo, hi)
        if last:
            self.wait()                       # gradients are final in stream order when backward() returns
        elif self._pending[1] - self._pending[0] >= self.bucket_elems:
            self.flush()

    def flush(self):
        if self._pending is None:
            return
        lo, hi = self._pending
        self._pending = None
        seg = self._gbuf[lo:hi]
        if _DEBUG:
            import sys
            print(f"[flatsync r{dist.get_rank(self.pg)}] all_reduce #{self.collectives} [{lo}, {hi}) bucket {self.bucket_elems}", file=sys.stderr, flush=True)
        op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
        h = dist.all_reduce(seg, op=op, group=self.pg, async_op=True)
        self._handles.append((h, seg))
        self.collectives += 1

    def wait(self):
        """Make the current stream wait for every outstanding collective (call before reading gradients)."""
        self.flush()
        for h, seg in self._handles:
            h.wait()
            if not self._avg:
                seg.mul_(1.0 / self.world)
        self._handles.clear()
        self._gbuf = None

    def detach(self):
        self.wait()
        self.module._grad_sync = None


def _numel(shape):
    n = 1
    for s in shape:
        n *= s
    return n


def _overlap_or_gap_is_padding(m, a, b):
    """Segments are 256-element aligned: two ranges separated only by alignment padding are merged (the padding is
    zero on every rank)."""
    lo, hi = (a, b) if a[0] <= b[0] else (b, a)
    gap = hi[0] - lo[1]
    return 0 <= gap < 256
