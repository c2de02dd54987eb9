"""Mirror of the reference `utils/cls_transforms.py::RandomMixup_DCT` (:100-193) on device tensors, plus the
soft-label cross entropy used on its output (pipeline_utils.py:535)."""
import time
from typing import Tuple

import torch
from torch import Tensor

from . import lib as L


class LazyMixed:
    """A batch item RandomMixup_DCT(lazy=True) has NOT mixed yet: the un-mixed tensor plus the device lambda.  rgb-no-more_amd's
    ViT applies the roll-by-one mix while its sub-block kernel loads the values (rgbnm_subblock_embed_mix: the same bits as the
    mixed tensor, which then never exists in memory); any other consumer calls .materialize().  An explicit opt-in for loops that
    hand the batch straight to the model, as train.py / pipeline_utils.unpack_data do."""
    __slots__ = ("tensor", "lam")

    def __init__(self, tensor, lam):
        self.tensor, self.lam = tensor, lam

    shape = property(lambda self: self.tensor.shape)
    dtype = property(lambda self: self.tensor.dtype)
    device = property(lambda self: self.tensor.device)
    is_cuda = property(lambda self: self.tensor.is_cuda)

    def dim(self):
        return self.tensor.dim()

    def is_contiguous(self):
        return self.tensor.is_contiguous()

    def materialize(self, out=None):
        t = self.tensor
        o = torch.empty_like(t) if out is None else out
        B = t.shape[0]
        L.check(L.lib().rgbnm_mixup(L.dt_of(t.dtype), L.dt_of(o.dtype), t.data_ptr(), o.data_ptr(), self.lam.data_ptr(), B,
                                    t.numel() // B, L.stream()), "mixup")
        return o


class LazyTarget:
    """The target RandomMixup_DCT would have built -- one-hot labels, rolled by one, mixed with lambda -- as (labels, lambda): a
    `cross_entropy` of this package evaluates target[b][c] = (labels[b] == c) * lam[0] + (labels[b-1] == c) * lam[1] where it uses
    it (rgbnm_softxent_loss_mix / _grad_mix: the same bits), so the dense [B, num_classes] tensor and the launch that writes it
    do not exist.  .materialize() gives the dense tensor to anybody else."""

    def __init__(self, labels, lam, num_classes):
        self.labels, self.lam, self.num_classes = labels, lam, num_classes

    @property
    def shape(self):
        return torch.Size((self.labels.shape[0], self.num_classes))

    @property
    def device(self):
        return self.labels.device

    dtype = torch.float32

    def materialize(self, out=None):
        B = self.labels.shape[0]
        tgt = torch.empty(B, self.num_classes, device=self.labels.device, dtype=torch.float32) if out is None else out
        L.check(L.lib().rgbnm_mixup_target(self.labels.data_ptr(), tgt.data_ptr(), self.lam.data_ptr(), B, self.num_classes,
                                           L.stream()), "mixup_target")
        return tgt


class RandomMixup_DCT(torch.nn.Module):
    """Roll-by-one batch mixup of (Y, CbCr) and labels; lambda ~ Dirichlet(alpha, alpha) sorted descending
    (cls_transforms.py:135-182).  As in the reference (:168) lambda is drawn on the HOST from torch's CPU generator -- the same
    random stream for the same seed -- and reaches the kernels through a small ring of pinned slots with one asynchronous copy:
    no host sync in the step, and none of the half-dozen tiny device kernels a device-side Dirichlet + sort costs."""
    _SLOTS = 16

    def __init__(self, num_classes: int, alpha: float = 1.0, inplace: bool = False) -> None:
        super().__init__()
        if num_classes < 1:
            raise ValueError(f"Please provide a valid positive value for the num_classes. Got num_classes={num_classes}")
        if alpha <= 0:
            raise ValueError("Alpha param can't be zero.")
        self.num_classes, self.alpha, self.inplace = num_classes, alpha, inplace
        self.out_dtype = None   # None: keep the input dtype
        # lazy = True: the batch items come back as LazyMixed (un-mixed tensor + lambda) for a rgb-no-more_amd model to mix while
        # it loads them -- same bits, two launches and one round trip of the batch less per step; the target is mixed at once
        self.lazy = False
        # lazy_target = True: the target comes back as LazyTarget (labels + lambda) for this package's cross_entropy to mix where it
        # reads it -- same bits, one launch and the dense [B, num_classes] tensor less per step
        self.lazy_target = False

    def draw_lambda(self):
        """The reference's own draw on the CPU generator (cls_transforms.py:168): (lambda, 1 - lambda) sorted descending, fp32."""
        lam, _ = torch._sample_dirichlet(torch.tensor([self.alpha, self.alpha])).sort(descending=True)
        return lam.to(torch.float32).contiguous()

    def sample_lambda(self, device, out=None):
        """out: a device tensor (2 floats) to receive the draw (a static buffer of a captured HIP graph) instead of a ring slot."""
        lam = self.draw_lambda()
        device = torch.device(device)
        if device.type != "cuda":
            return lam.contiguous()
        ring = self.__dict__.setdefault("_ring", {})
        st = ring.get(device)
        if st is None:
            st = ring[device] = {"host": torch.empty(self._SLOTS, 2, dtype=torch.float32).pin_memory(),
                                 "dev": torch.empty(self._SLOTS, 2, dtype=torch.float32, device=device),
                                 "ev": [None] * self._SLOTS, "i": 0}
        i = st["i"]
        st["i"] = (i + 1) % self._SLOTS
        if st["ev"][i] is not None:
            t0 = time.perf_counter()
            st["ev"][i].synchronize()          # the copy that last used this pinned slot (16 steps ago) has long completed
            L.HOST_WAIT["sec"] += time.perf_counter() - t0
        st["host"][i].copy_(lam)
        dst = st["dev"][i] if out is None else out
        dst.copy_(st["host"][i], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(device))
        st["ev"][i] = ev
        return dst

    def forward(self, batch, target: Tensor, lam: Tensor = None, out=None) -> Tuple[Tensor, Tensor]:
        """out: optional (tensors for the mixed batch items ..., tensor for the mixed target) to write into (static buffers of a
        captured HIP graph; None entries for items that come back as LazyMixed); default: fresh tensors."""
        if target.ndim != 1:
            raise ValueError(f"Target ndim should be 1. Got {target.ndim}")
        if target.dtype != torch.int64:
            raise TypeError(f"Target dtype should be torch.int64. Got {target.dtype}")
        single = not isinstance(batch, (tuple, list))
        items = [batch] if single else list(batch)
        L.require_cuda(*items, target)
        if lam is None:
            lam = self.sample_lambda(target.device)
        outs = []
        for k, t in enumerate(items):
            od = self.out_dtype or t.dtype
            if self.lazy and od == t.dtype and (out is None or out[k] is None):
                outs.append(LazyMixed(t, lam))
                continue
            o = torch.empty(t.shape, device=t.device, dtype=od) if out is None else out[k]
            if o.shape != t.shape or o.dtype != od or not o.is_contiguous():
                raise ValueError("out tensors must match the batch items in shape and output dtype")
            B = t.shape[0]
            L.check(L.lib().rgbnm_mixup(L.dt_of(t.dtype), L.dt_of(od), t.data_ptr(), o.data_ptr(), lam.data_ptr(), B,
                                        t.numel() // B, L.stream()), "mixup")
            outs.append(o)
        if self.lazy_target and (out is None or out[-1] is None):
            return (outs[0] if single else tuple(outs)), LazyTarget(target, lam, self.num_classes)
        tgt = torch.empty(target.shape[0], self.num_classes, device=target.device, dtype=torch.float32) if out is None else out[-1]
        if tgt.shape != (target.shape[0], self.num_classes) or tgt.dtype != torch.float32 or not tgt.is_contiguous():
            raise ValueError("the out tensor of the target must be float32 [B, num_classes]")
        L.check(L.lib().rgbnm_mixup_target(target.data_ptr(), tgt.data_ptr(), lam.data_ptr(), target.shape[0],
                                           self.num_classes, L.stream()), "mixup_target")
        return (outs[0] if single else tuple(outs)), tgt


_TICKETS = {}       # (device index, raw stream) -> one zeroed 32-bit word for rgbnm_softxent_loss (left zero by every launch)


def _ticket(device):
    key = (device.index, L.stream())
    t = _TICKETS.get(key)
    if t is None:
        t = torch.zeros(1, device=device, dtype=torch.int32)
        if not torch.cuda.is_current_stream_capturing():      # (memory of a graph's private pool is not kept beyond the capture)
            _TICKETS[key] = t
    return t


class _SoftXent(torch.autograd.Function):
    """loss = CrossEntropyLoss()(logits, target) as one launch; its backward as one launch that reads autograd's output gradient on
    the device.  `edge` is the tensor the gradient flows back through: the logits themselves (gradient in their dtype), or the
    compute-dtype twin a rgb-no-more_amd head hands out next to its fp32 logits (plainvit._HeadFn) -- the bf16 head backward then
    gets its bf16 dlogits without the fp32 round trip autograd's dtype check would force on a gradient of fp32 logits."""

    @staticmethod
    def forward(ctx, edge, logits, target, mixlam=None):
        B, Cn = logits.shape
        hard = target.dtype == torch.int64
        rows = torch.empty(B, device=logits.device, dtype=torch.float32)
        stat = torch.empty(2 * B, device=logits.device, dtype=torch.float32)
        loss = torch.empty(1, device=logits.device, dtype=torch.float32)
        if mixlam is not None:          # LazyTarget: labels + lambda, the mixed target built where it is read
            L.check(L.lib().rgbnm_softxent_loss_mix(logits.data_ptr(), target.data_ptr(), mixlam.data_ptr(), rows.data_ptr(),
                                                    stat.data_ptr(), loss.data_ptr(), _ticket(logits.device).data_ptr(), B, Cn,
                                                    L.stream()), "softxent_loss_mix")
            ctx.save_for_backward(logits, target, stat, mixlam)
        else:
            L.check(L.lib().rgbnm_softxent_loss(logits.data_ptr(), None if hard else target.data_ptr(),
                                                target.data_ptr() if hard else None, rows.data_ptr(), stat.data_ptr(),
                                                loss.data_ptr(), _ticket(logits.device).data_ptr(), B, Cn, L.stream()), "softxent_loss")
            ctx.save_for_backward(logits, target, stat)
        ctx.dl_dtype = edge.dtype
        return loss[0]

    @staticmethod
    def backward(ctx, gout):
        logits, target, stat = ctx.saved_tensors[:3]
        mixlam = ctx.saved_tensors[3] if len(ctx.saved_tensors) > 3 else None
        B, Cn = logits.shape
        hard = target.dtype == torch.int64
        dl = torch.empty(B, Cn, device=logits.device, dtype=ctx.dl_dtype)
        if gout.dtype != torch.float32 or not gout.is_cuda:
            gout = gout.to(device=logits.device, dtype=torch.float32)
        if mixlam is not None:
            L.check(L.lib().rgbnm_softxent_grad_mix(L.dt_of(ctx.dl_dtype), logits.data_ptr(), target.data_ptr(), mixlam.data_ptr(),
                                                    stat.data_ptr(), gout.data_ptr(), dl.data_ptr(), B, Cn, 1.0 / B, L.stream()),
                    "softxent_grad_mix")
        else:
            L.check(L.lib().rgbnm_softxent_grad(L.dt_of(ctx.dl_dtype), logits.data_ptr(), None if hard else target.data_ptr(),
                                                target.data_ptr() if hard else None, stat.data_ptr(), gout.data_ptr(), dl.data_ptr(),
                                                B, Cn, 1.0 / B, L.stream()), "softxent_grad")
        return dl, None, None, None


def cross_entropy(logits: Tensor, target: Tensor, grad_dtype=torch.float32) -> Tensor:
    """torch.nn.CrossEntropyLoss()(logits, target) for class-index (int64 [B]) or probability ([B,C] fp32)
    targets, mean reduction: one HIP launch forward, one backward.  grad_dtype: the dtype the model's head wants its dlogits
    in; honoured when `logits` come from a rgb-no-more_amd head that handed out a gradient edge of that dtype (fp32 logits keep an
    fp32 gradient otherwise, as autograd demands)."""
    mixlam = None
    if isinstance(target, LazyTarget):          # RandomMixup_DCT(lazy_target=True): labels + lambda
        if target.num_classes != logits.shape[1]:
            raise ValueError(f"LazyTarget for {target.num_classes} classes, logits have {logits.shape[1]}")
        target, mixlam = target.labels, target.lam
        L.require_cuda(mixlam)
    L.require_cuda(logits, target)
    edge = getattr(logits, "_rgbnm_grad_edge", None)
    if logits.dtype != torch.float32:
        logits, edge = logits.float(), None
    if target.dtype != torch.int64:
        target = target.float().contiguous()
    logits = logits.contiguous()
    if edge is None or edge.dtype != grad_dtype or edge.shape != logits.shape or not edge.requires_grad:
        edge = logits
    return _SoftXent.apply(edge, logits.detach(), target, mixlam)
