"""Host mirror of the reference `models/swinv2.py` for `--domain DCT` (BASELINE config 5):
`SwinTransformerV2(img_size=256, patch_size=4, embed_dim=96, depths, num_heads, window_size=8, ..., pixel_space='dct')`
with the reference's constructor signature, parameter / buffer names (state_dict-compatible) and `forward(y, cbcr)`.

Compute runs on the C-ABI kernels of librgbnm.so: `rgbnm_swin_embed` (8x8 -> 4x4 / 2x2 sub-block decomposition),
`rgbnm_gemm_nt/tn` (every Linear, GELU fused), `rgbnm_ln_generic_*` (res-post-norm LayerNorm + DropPath scale),
`rgbnm_window_attention_*` (cosine attention, position bias, shift mask; roll / window partition are index arithmetic
inside the kernel), `rgbnm_merge_gather`, `rgbnm_token_mean`.  torch is used for autograd bookkeeping, for the
parameter-only continuous-position-bias MLP (15x15 table -> [heads,64,64]; 0.1 MFLOP) and for per-step weight casts.
Round 1: correct and complete for training; window attention on the MFMA pipe (csrc/swin_attn.hip), the Linears still
on the generic GEMM tiles.
"""
import ctypes as C
import itertools
import threading
import warnings
import math

import numpy as np
import torch
from torch import nn

from . import dct_ops as dops
from . import lib as L
from .flatparams import FlatParamModule, align

WS = 8


# ------------------------------------------------------------------ autograd nodes over the C ABI
class _HoldBracket:
    """The split-sum reductions of a WHOLE backward pass as one launch (rgbnm.h rgbnm_reduce_hold_*; ViT.defer_grad_reduction's
    counterpart for SwinTransformerV2.group_dw_backward): opened by the head's backward node, closed by the patch embedding's.  While
    it is open every reduction of the pass (weight-gradient partials, LayerNorm parameter sums, window attention's d(bias) /
    d(scale)) is only recorded; each call therefore gets scratch of its own (_ws hands out fresh tensors, kept alive here), and
    whoever must READ a reduced value before the end -- the position-bias backward -- flushes first.  post: work that needs reduced
    values and can wait for the end (the row-paired layers' fold of their 2N x 2K product)."""

    def __init__(self):
        self.active, self.keep, self.post = False, [], []
        self.table = self.table_host = self.ftable = self.ftable_host = None
        self._tabs = {}

    def begin(self, dev):
        # one (device table, page-locked host record) pair per device AND per launch mode: a HIP-graph capture carries the upload of
        # its table as a copy node that reads the host record at every replay, so eager passes in between (other buffers, another
        # table) must not write that record
        if torch.cuda.is_current_stream_capturing():         # every capture its own pair, alive as long as this model
            spare = self._tabs.get(("spare", dev)) or []
            if not spare:
                return False                                 # (page-locked memory cannot be allocated inside a capture: reserve() first)
            pair = spare.pop()
            self._tabs.setdefault("captured", []).append(pair)
        else:
            self.reserve(dev)
            pair = self._tabs[dev]
        (self.ftable, self.ftable_host), (self.table, self.table_host) = pair
        if L.lib().rgbnm_reduce_hold_begin() != 0:      # left open by a pass that died on this thread
            L.lib().rgbnm_reduce_hold_cancel()
            L.check(L.lib().rgbnm_reduce_hold_begin(), "reduce_hold_begin")
        self.active, self.keep, self.post = True, [], []
        return True

    def reserve(self, dev):
        """Tables for this device's eager passes and two spare pairs for captures (called from eager code: forward())."""
        if torch.cuda.is_current_stream_capturing():
            return
        n = L.lib().rgbnm_reduce_hold_table_bytes()
        # a set = TWO (device table, host record) pairs: one for the mid-pass flush, one for the end -- inside a capture both uploads are
        # copy nodes that read their host record at every replay, so the two must not share one
        one = lambda: (torch.zeros(n, device=dev, dtype=torch.uint8), torch.zeros(n, dtype=torch.uint8).pin_memory())  # noqa: E731
        mk = lambda: (one(), one())  # noqa: E731
        if dev not in self._tabs:
            self._tabs[dev] = mk()
        spare = self._tabs.setdefault(("spare", dev), [])
        while len(spare) < 2:
            spare.append(mk())

    def flush(self):
        """Run what has been recorded so far and keep recording (somebody needs reduced values now)."""
        if self.active:
            L.check(L.lib().rgbnm_reduce_hold_end(self.ftable.data_ptr(), self.ftable_host.data_ptr(), self.ftable.numel(), L.stream()),
                    "reduce_hold_end")
            post, self.post = self.post, []
            for f in post:
                f()
            L.check(L.lib().rgbnm_reduce_hold_begin(), "reduce_hold_begin")

    def end(self):
        if self.active:
            self.active = False
            rc = L.lib().rgbnm_reduce_hold_end(self.table.data_ptr(), self.table_host.data_ptr(), self.table.numel(), L.stream())
            post, self.post, self.keep = self.post, [], []
            L.check(rc, "reduce_hold_end")
            for f in post:
                f()

    def cancel(self):
        if self.active:
            self.active = False
            L.lib().rgbnm_reduce_hold_cancel()
            self.keep, self.post = [], []


_HOLD = [None]          # the open _HoldBracket of the backward pass that is running (one autograd thread runs its nodes in turn)


def _ws(dev, nbytes, slot=0):
    """Scratch per (device, slot): calls whose partial sums must coexist (grouped dW GEMMs) take different slots.  Inside a held
    backward (_HoldBracket) every call gets a region of its own: its partial sums live until the bracket's one reduction launch."""
    hb = _HOLD[0]
    if hb is not None and hb.active:
        t = torch.empty(nbytes, device=dev, dtype=torch.uint8)
        hb.keep.append(t)
        return t
    t = _ws.cache.get((dev, slot))
    if t is None or t.numel() < nbytes:
        t = torch.empty(nbytes, device=dev, dtype=torch.uint8)
        _ws.cache[(dev, slot)] = t
    return t


_ws.cache = {}


def _gemm_nt(epi, A, W, bias=None, R=None, want_c2=False):
    M, K = A.shape
    N = W.shape[0]
    out = torch.empty(M, N, device=A.device, dtype=A.dtype)
    c2 = torch.empty_like(out) if want_c2 else None
    L.check(L.lib().rgbnm_gemm_nt(L.dt_of(A.dtype), epi, A.data_ptr(), K, W.data_ptr(), K, out.data_ptr(), N, L.ptr(bias),
                                  L.ptr(R), N, L.ptr(c2), N, None, 0, M, N, K, 0, L.stream()), "gemm_nt")
    return out, c2


class _DwBracket:
    """ONE rgbnm_gemm_tn_group bracket around a whole backward pass (grouped weight-gradient GEMMs across a stage's
    blocks, reference: every nn.Linear's weight gradient is its own GEMM under autograd, models/swinv2.py:70-199).

    Every Linear of a stage has the same row count, so with the bracket open the library queues their weight-gradient GEMMs and
    runs them as launches of up to 256 output tiles (rgbnm.h rgbnm_gemm_tn_group_begin_n): at stages 3 / 4 that is three blocks /
    most of a block per launch with NO token split -- the kernel writes dW / db itself, no fp32 partial sums and no reduction
    launch -- where the per-Linear launches split the tokens 5 - 128 ways.  Opened by the backward of the classification head
    (the first node of the pass), closed by the patch embedding's (the last); until then the queued operands and workspaces are
    kept alive here and the dW / db tensors handed to autograd hold nothing -- which nobody reads before the pass is over
    (AccumulateGrad adopts them; an exchange that copies gradients DURING the pass, parallel.GatheredFlatGradSync, switches the
    bracket off).  Row-paired layers (stage 1) finish their gradients with tensor arithmetic on the spot: what is queued runs
    first, they run the old way."""

    _ids = itertools.count(1)

    def __init__(self):
        self.active = False
        self.keep = []
        self.id = 0
        self.thread = None

    def begin(self):
        if not self.active:
            # a NAMED bracket: the queue is thread_local to the autograd worker that runs the backward nodes; should the pass never
            # reach its last node, the next forward -- on another thread -- can only name it (abandon)
            self.id = next(self._ids)
            self.thread = threading.get_ident()
            L.lib().rgbnm_gemm_tn_group_begin_id(48, self.id)
            self.active, self.keep = True, []

    def end(self, hold=None):
        """hold: the pass's open _HoldBracket -- the queued jobs' reductions are recorded there and READ the jobs' partial sums when it
        closes, so it closes before their workspaces are let go."""
        if self.active:
            self.active = False
            rc = L.lib().rgbnm_gemm_tn_group_end(L.stream())
            try:
                if hold is not None:
                    hold.end()
            finally:
                self.keep = []
            L.check(rc, "gemm_tn_group_end")
        elif hold is not None:
            hold.end()

    def abandon(self):
        """Called where a bracket is found still open AFTER its backward pass is over (a node other than a Linear raised, or a
        partial backward never reached the patch embedding): the queued jobs must never run -- their operands die with `keep`.  On
        the thread that opened it the queue is dropped at once, from any other thread the library is told its name and the owning
        thread drops it the next time it touches its queue (end() on the caller's thread used to do nothing there,
        and the stale jobs ran behind the next pass's)."""
        if self.active:
            self.active = False
            L.lib().rgbnm_gemm_tn_group_abort(self.id)
            self.keep = []

    def pause(self):
        L.check(L.lib().rgbnm_gemm_tn_group_end(L.stream()), "gemm_tn_group_end")

    def resume(self):
        L.lib().rgbnm_gemm_tn_group_begin_id(48, self.id)


_ACTIVE = [None]        # the bracket of the backward pass that is running (its nodes run one after the other on one autograd thread)


def _tn_max_splits(No, Ki):
    """Token slices a single grouped launch can give this GEMM: 256 / its own tiles -- 192 x 384 tiles where the library uses them
    (csrc/gemm_tn_pipe.hip, option tn_wide), else 128 x 192."""
    t = ((No + 127) // 128) * (Ki // 192)
    if No % 192 == 0 and Ki % 384 == 0:
        t = min(t, (No // 192) * (Ki // 384))
    return max(1, 256 // t)


def _gemm_tn(dY, X, want_bias, slot=0, keep=None):
    M, No = dY.shape
    Ki = X.shape[1]
    dW = torch.empty(No, Ki, device=dY.device, dtype=torch.float32)
    db = torch.empty(No, device=dY.device, dtype=torch.float32) if want_bias else None
    if keep is None:
        hb = _HOLD[0]
        if hb is not None and hb.active and Ki % 192 == 0:
            # held: scratch of its own; a single launch splits the tokens at most 256 / its own 128 x 192 tiles ways
            wsb = L.lib().rgbnm_gemm_tn_workspace_splits(No, Ki, _tn_max_splits(No, Ki))
        else:
            wsb = L.lib().rgbnm_gemm_tn_workspace(M, No, Ki)
        ws = _ws(dY.device, wsb, slot)
    else:                               # queued: the call's own workspace and its operands live until the bracket closes; a grouped
        # launch splits the token axis 256 / (tiles of the group) ways, i.e. no further than 256 / this job's own tiles: room for
        # that many slices is all the job can use (the worst-case size pinned ~13 GB of scratch per backward)
        if Ki % 192 == 0:
            wsb = L.lib().rgbnm_gemm_tn_workspace_splits(No, Ki, _tn_max_splits(No, Ki))
        else:
            wsb = L.lib().rgbnm_gemm_tn_workspace(M, No, Ki)
        ws = torch.empty(wsb, device=dY.device, dtype=torch.uint8)
        keep.append((dY, X, ws))
    L.check(L.lib().rgbnm_gemm_tn(L.dt_of(dY.dtype), dY.data_ptr(), No, X.data_ptr(), Ki, dW.data_ptr(), L.ptr(db), M, No,
                                  Ki, 0, 0, ws.data_ptr(), ws.numel(), L.stream()), "gemm_tn")
    return dW, db


def _paired(sh, W):
    """Row pairing: the shadow is diag(W, W) (rgbnm_linear_desc.pair).  x [M,K] viewed as [M/2,2K] times diag(W,W)^T IS
    y [M,N] viewed as [M/2,2N], so the 96-wide Linears of SwinV2-T's first stage (N or K = 96 / 288: no multiple of the 192-column
    tiles) run on the kernels tuned for 192-wide rows; the zero blocks cost MFMA work these HBM-bound GEMMs have to spare."""
    return sh[0].shape[0] == 2 * W.shape[0]


def _nt(epi, x, Wsh, pair, bias=None, R=None, want_c2=False, bias_prep=None):
    """bias_prep: the layer's bias operand as rgbnm_prep_weights laid it out this step (fp32, [N]; [2N] for a row-paired layer)."""
    if bias is not None and bias_prep is not None:
        bias = bias_prep
    if not pair:
        return _gemm_nt(epi, x, Wsh, bias, R, want_c2)
    M = x.shape[0]
    if M % 2:
        raise ValueError("row pairing needs an even number of rows")
    N2 = Wsh.shape[0]
    if bias is not None and bias.numel() != N2:
        bias = torch.cat((bias, bias))
    y, c2 = _gemm_nt(epi, x.view(M // 2, -1), Wsh, bias,
                     None if R is None else R.view(M // 2, N2), want_c2)
    return y.view(M, N2 // 2), None if c2 is None else c2.view(M, N2 // 2)


def _tn_issue(dy, x, want_bias, pair, slot=0):
    """Launch (or, inside a rgbnm_gemm_tn_group bracket, queue) one weight-gradient GEMM; _tn_finish turns what it returns into
    (dW, db) once the results exist."""
    br = _ACTIVE[0]
    if not pair:
        queued = br is not None and br.active and dy.dtype == torch.bfloat16
        return _gemm_tn(dy, x, want_bias, slot, br.keep if queued else None) + (0, 0)
    M, N, K = dy.shape[0], dy.shape[1], x.shape[1]
    return _gemm_tn(dy.view(M // 2, 2 * N), x.view(M // 2, 2 * K), want_bias, slot) + (N, K)   # [[e.e, e.o], [o.e, o.o]] row parities


def _alias(t):
    """A second tensor over t's memory that is NOT a view of t (no reference to t's TensorImpl)."""
    return torch.empty(0, device=t.device, dtype=t.dtype).set_(t.untyped_storage(), t.storage_offset(), t.shape, t.stride())


def _tn_finish(t):
    dW2, db2, N, K = t
    if N == 0:
        return dW2, db2
    hb = _HOLD[0]
    if hb is not None and hb.active:
        # the product's reduction is held: the fold of its diagonal blocks waits for the bracket's end; the gradients handed to
        # autograd now are filled then (nobody reads a gradient before backward() returns: the bracket's contract)
        dW = torch.empty(N, K, device=dW2.device, dtype=dW2.dtype)
        db = None if db2 is None else torch.empty(N, device=dW2.device, dtype=dW2.dtype)
        # (the deferred fold writes through ALIASES of the storages: a reference to dW / db themselves would keep AccumulateGrad from
        # adopting them -- it would copy their still unwritten contents into .grad instead)
        dWa = _alias(dW)
        hb.post.append(lambda: torch.add(dW2[:N, :K], dW2[N:, K:], out=dWa))
        if db is not None:
            dba = _alias(db)
            hb.post.append(lambda: torch.add(db2[:N], db2[N:], out=dba))
        return dW, db
    return dW2[:N, :K] + dW2[N:, K:], None if db2 is None else db2[:N] + db2[N:]


def _tn(dy, x, want_bias, pair):
    br = _ACTIVE[0]
    if pair and br is not None and br.active:       # finished with tensor arithmetic at once: not a job for the open bracket
        br.pause()
        try:
            return _tn_finish(_tn_issue(dy, x, want_bias, pair))
        finally:
            br.resume()
    return _tn_finish(_tn_issue(dy, x, want_bias, pair))


def _fbias(b):
    return None if b is None else b.detach().float().contiguous()


class _QkvBiasFn(torch.autograd.Function):
    """The qkv Linear's bias vector q_bias | 0 | v_bias (swinv2.py:150-152) WITHOUT building it: rgbnm_prep_weights has written it
    this step (rgbnm_linear_desc.bias_mode 2); this node only routes the gradient of the vector back to its two parameters."""

    @staticmethod
    def forward(ctx, q_bias, v_bias, prepared):
        ctx.c = q_bias.numel()
        return prepared[:3 * ctx.c].detach()

    @staticmethod
    def backward(ctx, g):
        c = ctx.c
        return g[:c], g[2 * c:3 * c], None


class _LinearFn(torch.autograd.Function):
    """y = x W^T + b  (x [M,K] in the compute dtype, W fp32 master [N,K], b fp32 or None).  sh = (W, W^T) in the
    compute dtype from the per-step shadow buffer (rgbnm_prep_weights): no per-layer cast / transpose kernels."""

    @staticmethod
    def forward(ctx, x, W, b, sh, fork=False, role=None):
        """fork: also hand x back as a second output for the block's shortcut -- its gradient then arrives HERE and rides the
        dX GEMM's residual epilogue instead of costing autograd an [M,C] add kernel at the fork (24 per SwinV2-T step).
        role: ("open", bracket) for the classification head, ("close", bracket) for the patch embedding (_DwBracket)."""
        pair = _paired(sh, W)
        y, _ = _nt(L.EPI_NONE, x, sh[0], pair, _fbias(b), bias_prep=sh[2] if len(sh) > 2 else None)
        ctx.save_for_backward(x)
        ctx.sh, ctx.has_b, ctx.pair, ctx.role = sh, b is not None, pair, role
        return (y, x) if fork else y

    @staticmethod
    def backward(ctx, dy, dxs=None):
        (x,) = ctx.saved_tensors
        dy = dy.contiguous()
        role = ctx.role
        if role is not None and role[0] == "open" and all(p.grad is None for p in role[2]):
            # (gradients still attached -- accumulation over several passes -- would be ADDED TO by AccumulateGrad as each node
            # returns, i.e. before a queued GEMM has run: such a pass runs the old way)
            role[1].begin()
            _ACTIVE[0] = role[1]
            if len(role) > 3 and role[3] is not None and role[3].begin(dy.device):   # ... and the pass's reductions as one launch
                _HOLD[0] = role[3]
        try:
            dW, db = _tn(dy, x, ctx.has_b, ctx.pair)
        except BaseException:
            hb, _HOLD[0] = _HOLD[0], None
            if _ACTIVE[0] is not None:
                br, _ACTIVE[0] = _ACTIVE[0], None
                br.end(hb)
            elif hb is not None:
                hb.end()
            raise
        if role is not None and role[0] == "close" and _ACTIVE[0] is role[1]:    # the last node: what is still queued runs now
            _ACTIVE[0] = None
            hb, _HOLD[0] = _HOLD[0], None
            role[1].end(hb)
        if dxs is None:
            dx, _ = _nt(L.EPI_NONE, dy, ctx.sh[1], ctx.pair)
        else:
            dx, _ = _nt(L.EPI_RES, dy, ctx.sh[1], ctx.pair, None, R=dxs.contiguous())
        return dx, dW, db, None, None, None


class _MlpFn(torch.autograd.Function):
    """fc2(gelu(fc1(x))) with the GELU (and its derivative) fused into the fc1 epilogue and the derivative product
    fused into fc2's dX GEMM (swinv2.py:19-35)."""

    @staticmethod
    def forward(ctx, x, W1, b1, W2, b2, sh1, sh2, fork=False):
        p1, p2 = _paired(sh1, W1), _paired(sh2, W2)
        g, gp = _nt(L.EPI_GELU, x, sh1[0], p1, _fbias(b1), want_c2=True, bias_prep=sh1[2] if len(sh1) > 2 else None)
        y, _ = _nt(L.EPI_NONE, g, sh2[0], p2, _fbias(b2), bias_prep=sh2[2] if len(sh2) > 2 else None)
        ctx.save_for_backward(x, g, gp)
        ctx.sh1, ctx.sh2, ctx.p1, ctx.p2 = sh1, sh2, p1, p2
        return (y, x) if fork else y                   # fork: see _LinearFn

    @staticmethod
    def backward(ctx, dy, dxs=None):
        x, g, gp = ctx.saved_tensors
        dy = dy.contiguous()
        du, _ = _nt(L.EPI_DGELU, dy, ctx.sh2[1], ctx.p2, None, R=gp)
        # both weight gradients in one launch (rgbnm.h: rgbnm_gemm_tn_group_*): half the split count, one reduction
        grouped = dy.dtype == torch.bfloat16
        br = _ACTIVE[0]
        if br is not None and br.active and grouped and not (ctx.p1 or ctx.p2):
            # inside the backward-wide bracket (_DwBracket): both GEMMs join the queue of their stage
            t2 = _tn_issue(dy, g, True, False, 0)
            t1 = _tn_issue(du, x, True, False, 1)
        else:
            if br is not None and br.active:
                br.pause()                     # row-paired (stage 1) or fp32: what is queued runs first
            if grouped:
                L.lib().rgbnm_gemm_tn_group_begin()
            try:
                t2 = _tn_issue(dy, g, True, ctx.p2, 0)
                t1 = _tn_issue(du, x, True, ctx.p1, 1)
            finally:
                if grouped:
                    L.check(L.lib().rgbnm_gemm_tn_group_end(L.stream()), "gemm_tn_group_end")
                if br is not None and br.active:
                    br.resume()
        (dW2, db2), (dW1, db1) = _tn_finish(t2), _tn_finish(t1)
        if dxs is None:
            dx, _ = _nt(L.EPI_NONE, du, ctx.sh1[1], ctx.p1)
        else:
            dx, _ = _nt(L.EPI_RES, du, ctx.sh1[1], ctx.p1, None, R=dxs.contiguous())
        return dx, dW1, db1, dW2, db2, None, None, None


class _LNFn(torch.autograd.Function):
    """y = [res +] [s_b *] LayerNorm(x) over the last dim (any width up to 768)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, res, sscale, rows_per_sample):
        M, E = x.shape
        y = torch.empty_like(x)
        mean = torch.empty(M, device=x.device, dtype=torch.float32)
        rstd = torch.empty_like(mean)
        g, b = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
        L.check(L.lib().rgbnm_ln_generic_fwd(L.dt_of(x.dtype), x.data_ptr(), g.data_ptr(), b.data_ptr(), L.ptr(res),
                                             L.ptr(sscale), rows_per_sample, y.data_ptr(), mean.data_ptr(),
                                             rstd.data_ptr(), M, E, 1e-5, L.stream()), "ln_generic_fwd")
        ctx.save_for_backward(x, g, mean, rstd, sscale)
        ctx.rps, ctx.has_res = rows_per_sample, res is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, g, mean, rstd, sscale = ctx.saved_tensors
        M, E = x.shape
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        dg = torch.empty(E, device=x.device, dtype=torch.float32)
        db = torch.empty_like(dg)
        wsb = L.lib().rgbnm_ln_generic_bwd_workspace(M, E)
        ws = _ws(x.device, wsb)
        L.check(L.lib().rgbnm_ln_generic_bwd(L.dt_of(x.dtype), dy.data_ptr(), x.data_ptr(), g.data_ptr(), mean.data_ptr(),
                                             rstd.data_ptr(), L.ptr(sscale), ctx.rps, dx.data_ptr(), dg.data_ptr(),
                                             db.data_ptr(), M, E, 0, ws.data_ptr(), ws.numel(), L.stream()),
                "ln_generic_bwd")
        return dx, dg, db, (dy if ctx.has_res else None), None, None


class _WinAttnFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, bias, scale, B, res, C_, heads, shift):
        out = torch.empty(qkv.shape[0], C_, device=qkv.device, dtype=qkv.dtype)
        nw = (res // WS) ** 2
        lse = torch.empty(B * nw * heads * 64, device=qkv.device, dtype=torch.float32)
        bias_c, scale_c = bias.detach().float().contiguous(), scale.detach().float().contiguous()
        L.check(L.lib().rgbnm_window_attention_fwd(L.dt_of(qkv.dtype), qkv.data_ptr(), bias_c.data_ptr(),
                                                   scale_c.data_ptr(), out.data_ptr(), lse.data_ptr(), B, res, C_, heads,
                                                   shift, L.stream()), "window_attention_fwd")
        ctx.save_for_backward(qkv, out, bias_c, scale_c, lse)
        ctx.geo = (B, res, C_, heads, shift)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, out, bias_c, scale_c, lse = ctx.saved_tensors
        B, res, C_, heads, shift = ctx.geo
        dout = dout.contiguous()
        dqkv = torch.empty_like(qkv)
        dbias = torch.empty_like(bias_c)
        wsb = L.lib().rgbnm_window_attention_bwd_workspace(B, res, heads)
        ws = _ws(qkv.device, wsb)
        nw = (res // WS) ** 2
        dsp = torch.empty(B * nw * heads, device=qkv.device, dtype=torch.float32)
        dscale = torch.empty(heads, device=qkv.device, dtype=torch.float32)
        hb = _HOLD[0]
        if hb is not None and hb.active:
            hb.keep.append(dsp)         # partial sums of a held reduction: alive until the bracket's launch
        L.check(L.lib().rgbnm_window_attention_bwd(L.dt_of(qkv.dtype), qkv.data_ptr(), out.data_ptr(), dout.data_ptr(),
                                                   bias_c.data_ptr(), dscale.data_ptr(), scale_c.data_ptr(),
                                                   lse.data_ptr(), dqkv.data_ptr(), dbias.data_ptr(), dsp.data_ptr(), B,
                                                   res, C_, heads, shift, ws.data_ptr(), ws.numel(), L.stream()),
                "window_attention_bwd")
        return dqkv, dbias, dscale, None, None, None, None, None


class _MergeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, B, res, C_):
        out = torch.empty(B * (res // 2) ** 2, 4 * C_, device=x.device, dtype=x.dtype)
        L.check(L.lib().rgbnm_merge_gather(L.dt_of(x.dtype), x.data_ptr(), out.data_ptr(), B, res, C_, 0, L.stream()),
                "merge_gather")
        ctx.geo = (B, res, C_)
        return out

    @staticmethod
    def backward(ctx, dy):
        B, res, C_ = ctx.geo
        dy = dy.contiguous()
        dx = torch.empty(B * res * res, C_, device=dy.device, dtype=dy.dtype)
        L.check(L.lib().rgbnm_merge_gather(L.dt_of(dy.dtype), dy.data_ptr(), dx.data_ptr(), B, res, C_, 1, L.stream()),
                "merge_scatter")
        return dx, None, None, None


class _MeanFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, B, N, C_):
        out = torch.empty(B, C_, device=x.device, dtype=x.dtype)
        L.check(L.lib().rgbnm_token_mean(L.dt_of(x.dtype), x.data_ptr(), out.data_ptr(), B, N, C_, 0, L.stream()),
                "token_mean")
        ctx.geo = (B, N, C_)
        return out

    @staticmethod
    def backward(ctx, dy):
        B, N, C_ = ctx.geo
        dy = dy.contiguous()
        dx = torch.empty(B * N, C_, device=dy.device, dtype=dy.dtype)
        L.check(L.lib().rgbnm_token_mean(L.dt_of(dy.dtype), dy.data_ptr(), dx.data_ptr(), B, N, C_, 1, L.stream()),
                "token_mean_bwd")
        return dx, None, None, None


# ------------------------------------------------------------------ parameter holders (reference names)
class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features, **kw):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features, **kw)
        self.fc2 = nn.Linear(hidden_features, in_features, **kw)


class WindowAttention(nn.Module):
    """swinv2.py:70-182: parameters + the derived constant buffers; compute in SwinTransformerBlock.run."""

    def __init__(self, dim, window_size, num_heads, **kw):
        super().__init__()
        ws = window_size
        self.dim, self.window_size, self.num_heads = dim, (ws, ws), num_heads
        self.logit_scale = nn.Parameter(torch.log(10 * torch.ones((num_heads, 1, 1), **kw)))
        self.cpb_mlp = nn.Sequential(nn.Linear(2, 512, bias=True, **kw), nn.ReLU(inplace=True),
                                     nn.Linear(512, num_heads, bias=False, **kw))
        r = torch.arange(-(ws - 1), ws, dtype=torch.float32)
        tab = torch.stack(torch.meshgrid(r, r, indexing="ij"), dim=-1).unsqueeze(0) / (ws - 1) * 8
        tab = torch.sign(tab) * torch.log2(torch.abs(tab) + 1.0) / np.log2(8)
        self.register_buffer("relative_coords_table", tab.to(kw.get("device", "cpu")))
        c = torch.stack(torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij")).flatten(1)
        rel = (c[:, :, None] - c[:, None, :]).permute(1, 2, 0).contiguous() + (ws - 1)
        self.register_buffer("relative_position_index", (rel[:, :, 0] * (2 * ws - 1) + rel[:, :, 1]).to(kw.get("device", "cpu")))
        self.qkv = nn.Linear(dim, dim * 3, bias=False, **kw)
        self.q_bias = nn.Parameter(torch.zeros(dim, **kw))
        self.v_bias = nn.Parameter(torch.zeros(dim, **kw))
        self.proj = nn.Linear(dim, dim, **kw)


def inverse_index(idx, n_entries):
    """[n_entries, 64] int32: the positions i with idx[i] == entry (ascending), padded with idx.numel() -- what the backward of the
    position-bias gather sums over, in a fixed order (neither an atomicAdd scatter nor a sort-based index_put)."""
    i = idx.detach().cpu().view(-1)
    cnt = torch.bincount(i, minlength=n_entries)
    assert int(cnt.max()) <= 64
    inv = torch.full((n_entries, 64), i.numel(), dtype=torch.int32)
    order = torch.argsort(i, stable=True)
    pos = 0
    for e in range(n_entries):
        c = int(cnt[e])
        inv[e, :c] = order[pos:pos + c].to(torch.int32)
        pos += c
    return inv.to(idx.device)


class _CpbFn(torch.autograd.Function):
    """Position bias [heads,64,64] and logit scale [heads] of EVERY WindowAttention of the model (swinv2.py:158-168) as two launches
    forward and two backward (rgbnm.h rgbnm_swin_cpb_fwd / _bwd) -- the parameter-only work the reference runs as ~20 tiny kernels
    per block and direction.  Inputs: cpb_mlp[0].weight, cpb_mlp[0].bias, cpb_mlp[2].weight, logit_scale of each block, in block
    order; outputs: (bias_0, scale_0, bias_1, scale_1, ...)."""

    @staticmethod
    def forward(ctx, consts, *params):
        coords, index, inv, heads = consts
        nb = len(heads)
        dev = coords.device
        ps = [p.detach().float().contiguous() for p in params]
        outs = []
        blocks = (L.CpbBlock * nb)()
        for i, h in enumerate(heads):
            bias = torch.empty(h, 64, 64, device=dev, dtype=torch.float32)
            scale = torch.empty(h, device=dev, dtype=torch.float32)
            w1, b1, w2, ls = ps[4 * i:4 * i + 4]
            blocks[i] = L.CpbBlock(w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), ls.data_ptr(), bias.data_ptr(), scale.data_ptr(),
                                   None, None, None, None, None, None, h, 0)
            outs += [bias, scale]
        table = torch.empty(L.lib().rgbnm_swin_cpb_table_elems(nb), device=dev, dtype=torch.float32)
        L.check(L.lib().rgbnm_swin_cpb_fwd(blocks, nb, coords.data_ptr(), index.data_ptr(), table.data_ptr(), L.stream()), "swin_cpb_fwd")
        ctx.save_for_backward(table, *ps)
        ctx.consts = consts
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        coords, index, inv, heads = ctx.consts
        table, *ps = ctx.saved_tensors
        hb = _HOLD[0]
        if hb is not None and hb.active:
            hb.flush()              # d(bias) / d(scale) of the blocks are held reductions: this node reads them now
        nb = len(heads)
        dev = coords.device
        blocks = (L.CpbBlock * nb)()
        out, keep = [None], []
        for i, h in enumerate(heads):
            w1, b1, w2, ls = ps[4 * i:4 * i + 4]
            dbias, dscale = grads[2 * i].contiguous().float(), grads[2 * i + 1].contiguous().float()
            dw1, db1, dw2, dls = torch.empty_like(w1), torch.empty_like(b1), torch.empty_like(w2), torch.empty_like(ls)
            keep += [dbias, dscale]
            blocks[i] = L.CpbBlock(w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), ls.data_ptr(), None, None, dbias.data_ptr(),
                                   dscale.data_ptr(), dw1.data_ptr(), db1.data_ptr(), dw2.data_ptr(), dls.data_ptr(), h, 0)
            out += [dw1, db1, dw2, dls]
        dtable = torch.empty_like(table)
        L.check(L.lib().rgbnm_swin_cpb_bwd(blocks, nb, coords.data_ptr(), inv.data_ptr(), table.data_ptr(), dtable.data_ptr(),
                                           L.stream()), "swin_cpb_bwd")
        return tuple(out)


def model_bias_and_scale(blocks):
    """[(bias [heads,64,64], scale [heads])] for a list of SwinTransformerBlocks (the whole model: one _CpbFn node)."""
    a0 = blocks[0].attn
    consts = a0.__dict__.get("_cpb_consts")
    dev = a0.relative_coords_table.device
    if consts is None or consts[0].device != dev:
        idx = a0.relative_position_index.view(-1)
        consts = a0.__dict__["_cpb_consts"] = (a0.relative_coords_table.detach().reshape(-1, 2).float().contiguous(),
                                               idx.to(torch.int32).contiguous(), inverse_index(idx, 225).contiguous())
    for b in blocks:
        if b.attn.window_size != (WS, WS):
            raise NotImplementedError("continuous position bias kernels cover 8 x 8 windows")
    heads = tuple(b.attn.num_heads for b in blocks)
    params = []
    for b in blocks:
        params += [b.attn.cpb_mlp[0].weight, b.attn.cpb_mlp[0].bias, b.attn.cpb_mlp[2].weight, b.attn.logit_scale.view(-1)]
    res = _CpbFn.apply(consts + (heads,), *params)
    return [(res[2 * i], res[2 * i + 1]) for i in range(len(blocks))]


class SwinTransformerBlock(nn.Module):
    def __init__(self, dim, input_resolution, num_heads, window_size, shift_size, drop_path, **kw):
        super().__init__()
        self.dim, self.input_resolution, self.num_heads = dim, input_resolution, num_heads
        self.window_size, self.shift_size = window_size, shift_size
        if min(input_resolution) <= window_size:
            self.shift_size, self.window_size = 0, min(input_resolution)
        if self.window_size != WS or input_resolution[0] != input_resolution[1]:
            raise NotImplementedError("HIP window attention covers 8x8 windows on square grids (config 5: window 8)")
        self.norm1 = nn.LayerNorm(dim, **kw)
        self.attn = WindowAttention(dim, self.window_size, num_heads, **kw)
        self.drop_path_p = float(drop_path)
        self.norm2 = nn.LayerNorm(dim, **kw)
        self.mlp = Mlp(dim, dim * 4, **kw)
        if self.shift_size > 0:
            H = W = input_resolution[0]
            img = torch.zeros((1, H, W, 1))
            cnt = 0
            for hs in (slice(0, -WS), slice(-WS, -self.shift_size), slice(-self.shift_size, None)):
                for wsl in (slice(0, -WS), slice(-WS, -self.shift_size), slice(-self.shift_size, None)):
                    img[:, hs, wsl, :] = cnt
                    cnt += 1
            mw = img.view(1, H // WS, WS, W // WS, WS, 1).permute(0, 1, 3, 2, 4, 5).reshape(-1, WS * WS)
            am = mw.unsqueeze(1) - mw.unsqueeze(2)
            am = am.masked_fill(am != 0, float(-100.0)).masked_fill(am == 0, float(0.0))
            self.register_buffer("attn_mask", am.to(kw.get("device", "cpu")))     # kept for state_dict parity; the
        else:                                                                       # kernel derives it from coordinates
            self.register_buffer("attn_mask", None)

    def _drop_scale(self, B, dev):
        if not self.training or self.drop_path_p == 0.0:
            return None
        keep = 1.0 - self.drop_path_p                   # timm DropPath: per-sample Bernoulli(keep) / keep
        return (torch.rand(B, device=dev) < keep).float() / keep

    def run(self, x, B, sh, pre, bias_scale=None, drop=None):
        """bias_scale: (bias, scale) of this block from model_bias_and_scale (None: computed here, for this block alone); drop: the block's two DropPath scale vectors [2,B]
        (drawn for the whole model at once) or None."""
        res, C_ = self.input_resolution[0], self.dim
        a = self.attn
        bias, scale = model_bias_and_scale([self])[0] if bias_scale is None else bias_scale
        ds1 = self._drop_scale(B, x.device) if drop is None else drop[0]
        ds2 = self._drop_scale(B, x.device) if drop is None else drop[1]
        shq = sh[pre + "attn.qkv"]
        if len(shq) > 2 and shq[2] is not None:
            qb = _QkvBiasFn.apply(a.q_bias, a.v_bias, shq[2])       # q_bias | 0 | v_bias as the prep launch wrote it: no cat, no fill
        else:
            qb = torch.cat((a.q_bias, torch.zeros_like(a.v_bias, requires_grad=False), a.v_bias))
        qkv, xs = _LinearFn.apply(x, a.qkv.weight, qb, shq, True)      # xs = x: the shortcut (fork)
        o = _WinAttnFn.apply(qkv, bias, scale, B, res, C_, self.num_heads, self.shift_size)
        o = _LinearFn.apply(o, a.proj.weight, a.proj.bias, sh[pre + "attn.proj"])
        x = _LNFn.apply(o, self.norm1.weight, self.norm1.bias, xs, ds1, res * res)
        h, xs = _MlpFn.apply(x, self.mlp.fc1.weight, self.mlp.fc1.bias, self.mlp.fc2.weight, self.mlp.fc2.bias,
                             sh[pre + "mlp.fc1"], sh[pre + "mlp.fc2"], True)
        return _LNFn.apply(h, self.norm2.weight, self.norm2.bias, xs, ds2, res * res)


class PatchMerging(nn.Module):
    def __init__(self, input_resolution, dim, **kw):
        super().__init__()
        self.input_resolution, self.dim = input_resolution, dim
        self.reduction = nn.Linear(4 * dim, 2 * dim, bias=False, **kw)
        self.norm = nn.LayerNorm(2 * dim, **kw)

    def run(self, x, B, sh, pre):
        m = _MergeFn.apply(x, B, self.input_resolution[0], self.dim)
        m = _LinearFn.apply(m, self.reduction.weight, None, sh[pre + "reduction"])
        return _LNFn.apply(m, self.norm.weight, self.norm.bias, None, None, 1)


class BasicLayer(nn.Module):
    def __init__(self, dim, input_resolution, depth, num_heads, window_size, drop_path, downsample, **kw):
        super().__init__()
        self.blocks = nn.ModuleList([
            SwinTransformerBlock(dim, input_resolution, num_heads, window_size, 0 if i % 2 == 0 else window_size // 2,
                                 drop_path[i], **kw) for i in range(depth)])
        self.downsample = PatchMerging(input_resolution, dim, **kw) if downsample else None


class PatchEmbedding_DCT_Group(nn.Module):
    def __init__(self, img_size, emb_size, **kw):
        super().__init__()
        self.patches_resolution = [img_size // 4, img_size // 4]
        self.num_patches = self.patches_resolution[0] * self.patches_resolution[1]
        self.projection = nn.Sequential(nn.Linear(16 + 2 * 4, emb_size, **kw))
        self.norm = nn.LayerNorm(emb_size, **kw)
        self.conv_Y = dops.generate_conversion_matrix(4, 2, scale=True, dtype=torch.float32)
        self.conv_C = dops.generate_conversion_matrix(2, 4, scale=True, dtype=torch.float32)


class SwinTransformerV2(FlatParamModule):
    def __init__(self, img_size=224, patch_size=4, in_chans=3, num_classes=1000, embed_dim=96, depths=(2, 2, 6, 2),
                 num_heads=(3, 6, 12, 24), window_size=7, mlp_ratio=4., qkv_bias=True, drop_rate=0., attn_drop_rate=0.,
                 drop_path_rate=0.1, norm_layer=nn.LayerNorm, ape=False, patch_norm=True, use_checkpoint=False,
                 pretrained_window_sizes=(0, 0, 0, 0), device="cpu", pixel_space="rgb", **kwargs):
        super().__init__()
        if str(pixel_space).lower() != "dct":
            raise NotImplementedError("rgb-no-more_amd implements the --domain DCT path only")
        if patch_size != 4 or window_size != WS or mlp_ratio != 4.0 or not qkv_bias or ape or not patch_norm or \
                drop_rate or attn_drop_rate or any(pretrained_window_sizes) or norm_layer is not nn.LayerNorm:
            raise NotImplementedError("HIP SwinV2 covers the config-5 shape: patch 4, window 8, mlp_ratio 4, qkv_bias, "
                                      "no ape, patch_norm, no dropout, no pretrained window size")
        if any(embed_dim * 2 ** i != 32 * h for i, h in enumerate(num_heads)):
            raise NotImplementedError("head_dim must be 32 in every stage (SwinV2-T/S/B: embed_dim = 32 * heads[0])")
        if img_size % 8 or (img_size // 4) % (WS * 2 ** (len(depths) - 1)):
            raise NotImplementedError("every stage must keep a grid that is a multiple of the 8x8 window")
        kw = dict(device=device, dtype=torch.float32)
        self.num_classes, self.num_layers, self.embed_dim = num_classes, len(depths), embed_dim
        self.num_features = int(embed_dim * 2 ** (self.num_layers - 1))
        self.pixel_space = "dct"
        self.patch_embed = PatchEmbedding_DCT_Group(img_size, embed_dim, **kw)
        self.patches_resolution = self.patch_embed.patches_resolution
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, sum(depths))]
        self.layers = nn.ModuleList()
        for i in range(self.num_layers):
            r = self.patches_resolution[0] // (2 ** i)
            self.layers.append(BasicLayer(int(embed_dim * 2 ** i), (r, r), depths[i], num_heads[i], window_size,
                                          dpr[sum(depths[:i]):sum(depths[:i + 1])], i < self.num_layers - 1, **kw))
        self.norm = nn.LayerNorm(self.num_features, **kw)
        self.head = nn.Linear(self.num_features, num_classes, **kw)
        self.apply(self._init_weights)
        for ly in self.layers:                      # res-post-norm init (swinv2.py:450-455)
            for blk in ly.blocks:
                for n in (blk.norm1, blk.norm2):
                    nn.init.constant_(n.bias, 0)
                    nn.init.constant_(n.weight, 0)
        self.compute_dtype = None                   # None: follow autocast; or torch.float32 / torch.bfloat16
        # one weight-gradient bracket around the backward pass (_DwBracket).  Default off, like ViT.defer_grad_reduction: the
        # weight gradients then exist only when backward() has returned, which torch DDP's reducer hooks do not wait for
        self.group_dw_backward = False
        self.hold_reductions = True                 # with group_dw_backward: the pass's split-sum reductions as ONE launch too
        self._conv = None
        self._table_devs = set()                    # devices whose GELU table this model has made sure of (forward, lazily)

    @staticmethod
    def _init_weights(m):
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def no_weight_decay(self):
        return {"absolute_pos_embed"}

    def no_weight_decay_keywords(self):
        return {"cpb_mlp", "logit_scale", "relative_position_bias_table"}

    def forward(self, y, cbcr=None):
        if cbcr is None:
            raise ValueError("DCT path needs both Y and CbCr tensors")
        from .cls_transforms import LazyMixed          # a batch RandomMixup_DCT(lazy=True) left un-mixed: mixed here, the ordinary way
        y = y.materialize() if isinstance(y, LazyMixed) else y
        cbcr = cbcr.materialize() if isinstance(cbcr, LazyMixed) else cbcr
        L.require_cuda(y, cbcr)
        if y.device.index not in self._table_devs:
            # set-up call, once per device the model actually runs on (it synchronises, so it must not fall into a graph capture:
            # run one eager forward first, as bench.py does): the GELU table of the fc1 + GELU epilogues (rgbnm.h
            # rgbnm_gelu_table_init; without it those epilogues use the arithmetic form -- the same bits, more instructions)
            if not torch.cuda.is_current_stream_capturing():
                with torch.cuda.device(y.device):
                    L.check(L.lib().rgbnm_gelu_table_init(L.stream()), "gelu_table_init")
                self._table_devs.add(y.device.index)
            elif not getattr(self, "_warned_table", False):
                self._warned_table = True
                warnings.warn("rgb-no-more_amd: first SwinTransformerV2 forward on this device is being captured into a graph: the "
                              "GELU table cannot be set up here, the fc1 + GELU epilogues run their arithmetic form (same bits, "
                              "slower); run one eager forward before capturing", RuntimeWarning, stacklevel=2)
        B, _, Hb, Wb, _, _ = y.shape
        res = self.patches_resolution[0]
        if y.dim() != 6 or (2 * Hb, 2 * Wb) != (res, res) or tuple(cbcr.shape) != (B, 2, Hb // 2, Wb // 2, 8, 8):
            raise ValueError(f"expected Y (B,1,{res // 2},{res // 2},8,8) and CbCr (B,2,{res // 4},{res // 4},8,8)")
        cdt = self.compute_dtype
        if cdt is None:
            cdt = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else torch.float32
        if cdt not in (torch.float32, torch.bfloat16):
            raise NotImplementedError(f"compute dtype {cdt}")
        dev = y.device
        if self._conv is None or self._conv[0].device != dev:
            self._conv = (self.patch_embed.conv_Y.to(dev).contiguous(), self.patch_embed.conv_C.to(dev).contiguous())
        y, cbcr = y.contiguous(), cbcr.contiguous()
        if y.dtype not in (torch.float32, torch.bfloat16) or cbcr.dtype != y.dtype:
            raise TypeError("Y and CbCr must be fp32 or bf16 and share a dtype")
        feat = torch.empty(B * res * res, 24, device=dev, dtype=cdt)
        L.check(L.lib().rgbnm_swin_embed(L.dt_of(y.dtype), L.dt_of(cdt), y.data_ptr(), cbcr.data_ptr(),
                                         self._conv[0].data_ptr(), self._conv[1].data_ptr(), feat.data_ptr(), B, Hb, Wb,
                                         L.stream()), "swin_embed")
        sh = self._prep(cdt)
        if self._grad_sync is not None and torch.is_grad_enabled():
            self._grad_sync.begin_step()          # parallel.GatheredFlatGradSync: bucket counts start over
        pe = self.patch_embed
        # the weight-gradient GEMMs of the whole backward in one bracket (_DwBracket): only when nobody reads a gradient before the
        # pass is over, and when its first and last nodes will both run
        br = None
        if (self.group_dw_backward and torch.is_grad_enabled() and cdt == torch.bfloat16 and self._grad_sync is None
                and self.head.weight.requires_grad and pe.projection[0].weight.requires_grad):
            br = self.__dict__.setdefault("_dw_bracket", _DwBracket())
            hb = self.__dict__.setdefault("_hold_bracket", _HoldBracket())
            if br.active:                       # a backward pass that never reached the patch embedding
                _ACTIVE[0] = None
                br.abandon()
            if hb.active:                       # (its recorded reductions are dropped when the owning thread opens the next bracket)
                _HOLD[0] = None
                hb.active, hb.keep, hb.post = False, [], []
            hb.reserve(dev)
        x = _LinearFn.apply(feat, pe.projection[0].weight, pe.projection[0].bias, sh["patch_embed.projection.0"], False,
                            None if br is None else ("close", br))
        x = _LNFn.apply(x, pe.norm.weight, pe.norm.bias, None, None, 1)
        drops = self._drop_scales(B, dev)
        # position bias / logit scale of all blocks: ONE autograd node, two launches per direction (_CpbFn)
        cpb = model_bias_and_scale([blk for ly in self.layers for blk in ly.blocks])
        k = 0
        for li, ly in enumerate(self.layers):
            for bi, blk in enumerate(ly.blocks):
                x = blk.run(x, B, sh, f"layers.{li}.blocks.{bi}.", cpb[k], None if drops is None else drops[k])
                k += 1
            if ly.downsample is not None:
                x = ly.downsample.run(x, B, sh, f"layers.{li}.downsample.")
                res //= 2
        x = _LNFn.apply(x, self.norm.weight, self.norm.bias, None, None, 1)
        x = _MeanFn.apply(x, B, res * res, self.num_features)
        return _LinearFn.apply(x, self.head.weight, self.head.bias, sh["head"], False,
                               None if br is None else ("open", br, [p for p in self.parameters() if p.requires_grad],
                                                        self._hold_bracket if self.hold_reductions else None))

    def _drop_scales(self, B, dev):
        """timm DropPath (per-sample Bernoulli(keep) / keep) for every residual branch of the model from ONE uniform draw:
        [blocks][2][B] fp32, rows of ones where a block's rate is 0; None in eval mode or when no block drops."""
        ps = [blk.drop_path_p for ly in self.layers for blk in ly.blocks]
        if not self.training or not any(ps):
            return None
        key = (tuple(ps), str(dev))
        if getattr(self, "_keep_key", None) != key:
            self._keep_key, self._keep = key, torch.tensor([1.0 - p for p in ps], device=dev).view(-1, 1, 1)
        return (torch.rand(len(ps), 2, B, device=dev) < self._keep).float() / self._keep

    # ---------------------------------------------------------------- flat masters + per-step operand shadows
    def _flatten(self):
        """Pack every parameter into one fp32 buffer and describe the Linear layers for rgbnm_prep_weights: one launch
        per forward writes W and W^T of all 53 Linears in the compute dtype (instead of ~220 cast / transpose kernels)."""
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise L.RgbnmError("model parameters must live on a HIP device (no CPU fallback)")
        self._pack_parameters()
        lin = [n[:-len(".weight")] for n, p in self.named_parameters()
               if n.endswith(".weight") and p.dim() == 2 and "cpb_mlp" not in n]
        descs = (L.LinearDesc * len(lin))()
        self._sh_off, so, bo = {}, 0, 0
        for k, name in enumerate(lin):
            Nn, Kk = self._shapes[name + ".weight"]
            # row pairing (see _paired): widths that are multiples of 96 but not both of 192 -- the first stage of SwinV2-T
            pair = int(Nn % 96 == 0 and Kk % 96 == 0 and (Nn % 192 != 0 or Kk % 192 != 0) and max(Nn, Kk) <= 384)
            nel = Nn * Kk * (4 if pair else 1)
            ws, wst = so, so + align(nel)
            so = wst + align(nel)
            # the bias operand of the GEMM epilogue, prepared by the same launch (fp32; written twice for a row-paired layer; the
            # qkv Linear's is q_bias | 0 | v_bias, swinv2.py:150-152) -- instead of a torch.cat (+ a zeros fill) per layer and step
            mode, b_off, b2_off = 0, 0, 0
            if name.endswith("attn.qkv"):
                pre = name[:-len("qkv")]
                mode, b_off, b2_off = 2, self._offs[pre + "q_bias"], self._offs[pre + "v_bias"]
            elif name + ".bias" in self._offs:
                mode, b_off = 1, self._offs[name + ".bias"]
            bp = bo
            if mode:
                bo += align(Nn * (2 if pair else 1))
            descs[k] = L.LinearDesc(self._offs[name + ".weight"], b_off, ws, wst, bp, Nn, Kk, 0, 0, 0, pair, 0, 0, mode, 0, b2_off)
            self._sh_off[name] = (ws, wst, Nn * (2 if pair else 1), Kk * (2 if pair else 1), bp if mode else -1)
        self._ndesc, self._sh_total = len(lin), so
        self._descs_dev = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).to(dev)
        self._bias_prep = torch.zeros(max(bo, 1), device=dev, dtype=torch.float32)
        self._shadow, self._sh_views = {}, {}

    def _prep(self, cdtype):
        self._ensure_flat()
        if cdtype not in self._shadow:
            buf = torch.zeros(self._sh_total, device=self._flat.device, dtype=cdtype)
            self._shadow[cdtype] = buf
            self._sh_views[cdtype] = {n: (buf[ws:ws + Nn * Kk].view(Nn, Kk), buf[wst:wst + Nn * Kk].view(Kk, Nn),
                                          None if bp < 0 else self._bias_prep[bp:bp + Nn])
                                      for n, (ws, wst, Nn, Kk, bp) in self._sh_off.items()}
        L.check(L.lib().rgbnm_prep_weights(L.dt_of(cdtype), self._descs_dev.data_ptr(), self._ndesc,
                                           self._flat.data_ptr(), self._shadow[cdtype].data_ptr(), self._bias_prep.data_ptr(),
                                           L.stream()), "prep_weights")
        return self._sh_views[cdtype]
