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


class DeferredFlatExchange:
    """ONE all-reduce (average) of a whole flat gradient buffer per step, ISSUED right behind the backward and WAITED FOR only in
    front of the optimizer -- with the NEXT step's data stage queued in between (reference loop: train.py:145-176; there the
    sample / augment / mixup of batch i + 1 is the DataLoader's business and also independent of optimizer step i).

    Why: with the one-launch encoder kernels the backward of JPEG-Ti is two big launches, so there is little left to overlap an
    exchange with INSIDE the backward; but sampling, DCT augment, mixup (and, eagerly, the sub-block embed) of the next batch read
    neither the weights nor the gradients.  The collective runs on the process group's own stream (torch: async_op = True), the
    compute stream goes on with the data stage and waits for the collective only where the optimizer needs the gradients:

        ex = DeferredFlatExchange(lambda: model._gflat)
        for batch in ...:
            x = data_stage(batch)          # queued while the previous step's all-reduce is in flight
            ex.finish(optimizer_step)      # wait for it, then optimizer step of the PREVIOUS backward (no-op in the first step)
            forward_backward(x)            # writes the flat gradient buffer
            ex.issue()                     # all-reduce of this step's gradients starts
        ex.finish(optimizer_step)          # drain

    Same arithmetic in the same order as the blocking schedule (data stage, forward / backward, all-reduce, optimizer): only the
    QUEUEING order of two independent things -- next data stage, this optimizer step -- is swapped, so parameters and gradients
    are bit-identical to it (tests/test_deferred_exchange_gloo_cpu.py, world 2 on gloo)."""

    def __init__(self, flat_grad, process_group=None):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self._flat_grad = flat_grad if callable(flat_grad) else (lambda: flat_grad)
        self.pg = process_group
        self.world = dist.get_world_size(process_group)
        self._avg = dist.get_backend(process_group) == "nccl"         # RCCL has ReduceOp.AVG; gloo sums and we scale
        self._handle, self._buf = None, None
        self.collectives = 0

    @property
    def pending(self):
        return self._handle is not None

    def issue(self):
        if self._handle is not None:
            raise RuntimeError("DeferredFlatExchange: the previous step's exchange has not been finished (one backward per step)")
        self._buf = self._flat_grad()
        op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
        self._handle = dist.all_reduce(self._buf, op=op, group=self.pg, async_op=True)
        self.collectives += 1

    def finish(self, optimizer_step=None):
        """Make the current stream wait for the outstanding all-reduce (if any), then run optimizer_step().  Returns whether there
        was one: the first step of a run has no gradients yet and skips the optimizer."""
        if self._handle is None:
            return False
        self._handle.wait()
        if not self._avg:
            self._buf.mul_(1.0 / self.world)
        self._handle, self._buf = None, None
        if optimizer_step is not None:
            optimizer_step()
        return True


class GatheredFlatGradSync:
    """The same exchange for a FlatParamModule whose backward hands autograd SEPARATE gradient tensors (SwinV2: 221 of them,
    models/swinv2.py:578-711; reference wrap: torch DDP, train.py:137).

    Parameters are cut into buckets of consecutive flat segments (>= bucket_bytes, walked in REVERSE registration order = the
    order the backward finishes them).  A post-accumulate hook per parameter counts arrivals; when a bucket is complete its
    gradients are gathered into the bucket's slice of the model's flat gradient buffer with ONE multi-tensor copy (the very copy
    the fused optimizer would do at the end of the step), every .grad is re-pointed at its flat view -- so the optimizer takes
    its zero-copy path and clip / unscale see reduced values -- and the slice is all-reduced in place, asynchronously, while the
    backward of the earlier stages still runs.  torch DDP does the same work as 2 x 221 bucket copies around its collectives.

    The bucket that completes last also orders the stream behind every collective, so when `backward()` returns the gradients
    are final in stream order (GradScaler.unscale_ / clip_grad_norm_ of train.py:159-166 see reduced values, as with DDP).
    `wait()` (called by FusedClipAdamWWD.step) additionally flushes buckets a backward left incomplete -- parameters without a
    gradient this step are exchanged as zeros.  One backward per step and zero_grad() between steps, as FlatGradSync."""

    def __init__(self, module, process_group=None, bucket_bytes=16 << 20, broadcast=True):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.module, self.pg = module, process_group
        self.world = dist.get_world_size(process_group)
        module._ensure_flat()
        if broadcast:
            dist.broadcast(module._flat, src=0, group=process_group)
        self._avg = dist.get_backend(process_group) == "nccl"
        self.bucket_elems = 1          # (read by ViT-side code that asks whether gradients are consumed during the backward)
        names = list(module._named.keys())
        self._buckets, cur = [], []
        lim = max(1, bucket_bytes // 4)
        for n in reversed(names):
            cur.append(n)
            lo = module._offs[cur[-1]]
            hi = module._offs[cur[0]] + _numel(module._shapes[cur[0]])
            if hi - lo >= lim:
                self._buckets.append({"names": cur, "lo": lo, "hi": hi})
                cur = []
        if cur:
            self._buckets.append({"names": cur, "lo": module._offs[cur[-1]], "hi": module._offs[cur[0]] + _numel(module._shapes[cur[0]])})
        self._of = {}
        for b in self._buckets:
            b["left"] = len(b["names"])
            b["done"] = False
            for n in b["names"]:
                self._of[n] = b
        self._handles = []
        self.collectives = 0
        self._cb_queued = False
        self._hooks = [p.register_post_accumulate_grad_hook(self._make_hook(n)) for n, p in module._named.items()]
        module._grad_sync = self

    def _make_hook(self, name):
        def hook(param):
            if self.module._grad_sync is not self:      # switched off (bench.py's self-check runs one plain backward)
                return
            b = self._of[name]
            if b["done"]:
                raise RuntimeError("GatheredFlatGradSync: a second backward reached an exchanged bucket (gradient accumulation "
                                   "is not supported by the flat exchange; use torch DDP)")
            if not self._cb_queued:
                # gradients are final whenever backward() returns, also when a parameter got no gradient in this pass and its bucket
                # never completed: the engine calls wait() at the end of the pass, as DDP's reducer finalises there.  All
                # ranks must leave the same parameters unused (the buckets' collectives are issued in completion order).
                self._cb_queued = True
                torch.autograd.Variable._execution_engine.queue_callback(self._end_of_backward)
            b["left"] -= 1
            if b["left"] == 0:
                self._flush(b)
        return hook

    def _end_of_backward(self):
        self._cb_queued = False
        if self.module._grad_sync is self:
            self.wait()

    def begin_step(self):
        """At the start of a differentiated forward: nothing of an earlier backward may be in flight, counts start over."""
        self._cb_queued = False
        for h, _ in self._handles:
            h.wait()
        self._handles.clear()
        if any(p.grad is not None for p in self.module._named.values()):
            raise RuntimeError("GatheredFlatGradSync: gradients of an earlier backward are still attached (gradient accumulation / "
                               "zero_grad(set_to_none=False)); the flat exchange supports one backward per step after "
                               "zero_grad(set_to_none=True) -- use torch DDP for accumulation")
        for b in self._buckets:
            b["left"], b["done"] = len(b["names"]), False

    def _flush(self, b):
        m = self.module
        g = m._gflat
        views, grads = [], []
        for n in b["names"]:
            p = m._named[n]
            v = m._gview(g, n)
            if p.grad is None:
                v.zero_()
            elif p.grad.data_ptr() != v.data_ptr():
                views.append(v)
                grads.append(p.grad)
            p.grad = v
        if views:
            torch._foreach_copy_(views, grads)
        b["done"] = True
        seg = g[b["lo"]:b["hi"]]
        h = dist.all_reduce(seg, op=dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM, group=self.pg, async_op=True)
        self._handles.append((h, seg))
        self.collectives += 1
        if all(x["done"] for x in self._buckets):
            self._finish()          # the last bucket of the pass: when backward() returns the gradients are final in stream order

    def _finish(self):
        for h, seg in self._handles:
            h.wait()
            if not self._avg:
                seg.mul_(1.0 / self.world)
        self._handles.clear()

    def wait(self):
        """Flush what the backward left incomplete and order the stream behind every collective (no-op after a complete pass)."""
        if any(b["done"] for b in self._buckets) or any(b["left"] != len(b["names"]) for b in self._buckets):
            for b in self._buckets:
                if not b["done"]:
                    self._flush(b)
        self._finish()

    def detach(self):
        self.wait()
        for h in self._hooks:
            h.remove()
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
