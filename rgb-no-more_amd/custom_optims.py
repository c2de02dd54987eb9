"""Mirror of the reference `utils/custom_optims.py` (WeightDecay) plus the fused train-step tail.

* `WeightDecay`      -- same semantics as custom_optims.py:3-42: p -= (lr/base_lr) * weight_decay * p.
* `FusedClipAdamWWD` -- clip_grad_norm_(max_norm) + torch.optim.AdamW(weight_decay=0) + WeightDecay
  (train.py:163-165 / :170-172) as ONE pass over the model's flat fp32 buffers (rgbnm_clip_adamw_wd_step).
  The decayed set follows the reference's name filter (".weight" in name and "lrnorm" not in name,
  pipeline_utils.py:537).  `param_groups[0]['lr']` is honoured, so torch LR schedulers drive it unchanged.
  `state_dict()` / `load_state_dict()` carry the Adam moments and the step count in torch.optim.AdamW's own layout
  (state[i] = {step, exp_avg, exp_avg_sq} per parameter), so train.py's checkpoint / resume (train.py:195,
  pipeline_utils.py:490-580) works, and a checkpoint written by the reference's AdamW loads here (and vice versa).
"""
import torch
from torch.optim.optimizer import Optimizer

from . import lib as L


class WeightDecay(Optimizer):
    """Additive, schedule-relative weight decay (reference: utils/custom_optims.py:3-42)."""

    def __init__(self, params, lr: float = 1e-3, weight_decay: float = 0.0):
        if weight_decay < 0.0:
            raise ValueError("Invalid weight_decay value: {}".format(weight_decay))
        super().__init__(params, dict(lr=lr, base_lr=lr, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self):
        for group in self.param_groups:
            f = -((group["lr"] / group["base_lr"]) * group["weight_decay"])
            ps = [p for p in group["params"]]
            if ps:
                torch._foreach_add_(ps, ps, alpha=f)


class FusedClipAdamWWD(Optimizer):
    def __init__(self, model, lr=3e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, max_norm=1.0):
        m = model.module if hasattr(model, "module") else model
        super().__init__(list(m.parameters()), dict(lr=lr, base_lr=lr, betas=betas, eps=eps,
                                                    weight_decay=weight_decay, max_norm=max_norm))
        self._m = m
        self._step = 0
        self._state_ready = False
        self.last_norm = None

    def _ensure(self):
        m = self._m
        m._ensure_flat()
        if not self._state_ready or self._exp_avg.numel() != m._flat.numel() or self._exp_avg.device != m._flat.device:
            self._exp_avg = torch.zeros_like(m._flat)
            self._exp_avg_sq = torch.zeros_like(m._flat)
            self._ws = torch.empty(L.lib().rgbnm_clip_adamw_wd_workspace(), device=m._flat.device, dtype=torch.uint8)
            self._norm = torch.zeros(1, device=m._flat.device, dtype=torch.float32)
            self._gather = None
            self._state_ready = True
            self._publish_state()

    def _publish_state(self):
        """Expose the flat moment buffers as torch.optim.AdamW-style per-parameter state (views, always current)."""
        m = self._m
        self.state.clear()
        for n, p in m._named.items():
            self.state[p] = {"step": torch.tensor(float(self._step)), "exp_avg": m._gview(self._exp_avg, n),
                             "exp_avg_sq": m._gview(self._exp_avg_sq, n)}

    def state_dict(self):
        self._ensure()
        for st in self.state.values():
            st["step"] = torch.tensor(float(self._step))
        return super().state_dict()

    def load_state_dict(self, state_dict):
        """Restore moments + step (written by this class or by torch.optim.AdamW over the same parameter list)."""
        self._ensure()
        own = [dict((k, v) for k, v in g.items() if k != "params") for g in self.param_groups]
        super().load_state_dict(state_dict)          # validates groups / sizes, casts to the parameter devices
        for g, o in zip(self.param_groups, own):
            if "base_lr" not in g:                   # a torch.optim.AdamW checkpoint: its weight_decay (0, the decay lives
                g["weight_decay"] = o["weight_decay"]     # in the reference's separate WeightDecay optimizer) is not ours
            for k, v in o.items():
                g.setdefault(k, v)                   # base_lr, max_norm, ... keep this optimizer's values
        m = self._m
        steps = set()
        loaded = dict(self.state)
        for n, p in m._named.items():
            st = loaded.get(p)
            if not st:                               # parameter without state (never stepped): zero moments
                m._gview(self._exp_avg, n).zero_()
                m._gview(self._exp_avg_sq, n).zero_()
                continue
            m._gview(self._exp_avg, n).copy_(st["exp_avg"])
            m._gview(self._exp_avg_sq, n).copy_(st["exp_avg_sq"])
            steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise ValueError(f"FusedClipAdamWWD keeps ONE step count for all parameters; the state has {sorted(steps)}")
        self._step = steps.pop() if steps else 0
        self._publish_state()

    def _flat_grads(self):
        """The flat fp32 gradient buffer: zero-copy when every .grad is the view the backward kernels wrote (the
        normal case, also after DDP's in-place all-reduce), otherwise gathered into a staging buffer."""
        m = self._m
        base = m.flat_grad_base()
        if base is not None:
            return base
        if self._gather is None:
            self._gather = torch.zeros_like(m._flat)
        views, grads = [], []
        for n, p in m._named.items():
            if p.grad is None:
                raise L.RgbnmError(f"parameter {n} has no gradient")
            views.append(m._gview(self._gather, n))
            grads.append(p.grad)
        torch._foreach_copy_(views, grads)
        return self._gather.data_ptr()

    @torch.no_grad()
    def step(self):
        self._ensure()
        g = self.param_groups[0]
        m = self._m
        self._step += 1
        if m._grad_sync is not None:       # overlapped flat all-reduce (parallel.FlatGradSync): gradients are final after this
            m._grad_sync.wait()
        gptr = self._flat_grads()
        L.check(L.lib().rgbnm_clip_adamw_wd_step(
            m._flat.data_ptr(), gptr, self._exp_avg.data_ptr(), self._exp_avg_sq.data_ptr(),
            m._wd_flags.data_ptr(), m._flat.numel(), g["lr"], g["betas"][0], g["betas"][1], g["eps"], self._step,
            (g["lr"] / g["base_lr"]) * g["weight_decay"], g["max_norm"] if g["max_norm"] else 0.0,
            self._norm.data_ptr(), self._ws.data_ptr(), self._ws.numel(), L.stream()), "clip_adamw_wd_step")
        self.last_norm = self._norm
