"""Flat fp32 parameter / gradient storage shared by the HIP model and the fused optimizer.

Every nn.Parameter of the module becomes a view into ONE fp32 buffer (segments aligned to 256 elements), gradients
produced by the backward kernels are views into ONE flat gradient buffer.  Effects:
  * the optimizer tail (clip + AdamW + WeightDecay) is a single pass over flat memory (rgbnm_clip_adamw_wd_step);
  * DDP still sees ordinary leaf parameters with ordinary .grad tensors, so its bucketed RCCL all-reduce works
    unchanged (train.py:137) -- the reducer copies grads into its buckets and back *in place*, which keeps the aliasing.
Pure torch (no HIP): unit-tested on CPU under gloo with world_size 2 (tests/test_ddp_gloo_cpu.py).
"""
import torch
from torch import nn

SEG = 256


def align(n, a=SEG):
    return (n + a - 1) // a * a


class FlatParamModule(nn.Module):
    """Mixin: call `_pack_parameters()` once parameters sit on their final device."""

    _flat = None
    _grad_sync = None      # parallel.FlatGradSync: overlapped all-reduce of the flat gradient buffer

    def _pack_parameters(self):
        params = list(self.named_parameters())
        dev = params[0][1].device
        offs, total = {}, 0
        for n, p in params:
            offs[n] = total
            total += align(p.numel())
        flat = torch.zeros(total, device=dev, dtype=torch.float32)
        for n, p in params:
            seg = flat[offs[n]:offs[n] + p.numel()].view(p.shape)
            seg.copy_(p.data)
            p.data = seg
        self._flat, self._offs, self._total = flat, offs, total
        self._shapes = {n: tuple(p.shape) for n, p in params}
        self._gflat = torch.zeros(total, device=dev, dtype=torch.float32)
        # weight-decay mask per 256-element chunk, following the reference's NAME filter (pipeline_utils.py:537)
        flags = torch.zeros(total // SEG, dtype=torch.uint8)
        for n, p in params:
            if (".weight" in n) and ("lrnorm" not in n):
                flags[offs[n] // SEG:(offs[n] + align(p.numel())) // SEG] = 1
        self._wd_flags = flags.to(dev)
        self._named = dict(params)
        self._probe_params = [params[0], params[len(params) // 2], params[-1]]
        return params

    def _flatten(self):          # subclasses extend
        self._pack_parameters()

    def _ensure_flat(self):
        ok = self._flat is not None
        if ok:
            base = self._flat.data_ptr()
            for n, p in self._probe_params:
                if p.data_ptr() != base + self._offs[n] * 4:
                    ok = False
                    break
        if not ok:
            self._flatten()

    def _gview(self, gbuf, name):
        n = 1
        for s in self._shapes[name]:
            n *= s
        return gbuf[self._offs[name]:self._offs[name] + n].view(self._shapes[name])

    def _grad_buffer(self):
        """flat fp32 buffer the backward writes into; a fresh one if live .grad tensors still alias it (gradient
        accumulation across several backward passes must not be clobbered)."""
        base, end = self._gflat.data_ptr(), self._gflat.data_ptr() + self._gflat.numel() * 4
        for p in self._named.values():                 # (the cached name -> parameter map: no module-tree walk per step)
            if p.grad is not None and base <= p.grad.data_ptr() < end:
                return torch.zeros_like(self._gflat)
        return self._gflat

    def flat_grad_base(self):
        """data_ptr of a flat buffer that holds EVERY current .grad at its segment offset, or None (optimizer fast path)."""
        first_name, first = next(iter(self._named.items()))
        g0 = first.grad
        if g0 is None:
            return None
        base = g0.data_ptr() - self._offs[first_name] * 4
        for n, p in self._named.items():
            if p.grad is None or p.grad.data_ptr() != base + self._offs[n] * 4 or p.grad.dtype != torch.float32:
                return None
        return base
