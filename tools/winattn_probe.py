#!/usr/bin/env python
"""Window attention forward / backward at the four SwinV2-T stage shapes of BASELINE config 5 (B = 256, 256 x 256 input, window 8):
time per launch, algorithmic bytes (qkv in, out / d(out) in, d(qkv) out) and the HBM rate they correspond to.
usage: python tools/winattn_probe.py [B] [out.json | -] [option=value ...]"""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rgb_no_more_amd import lib as L

DEV = "cuda"


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    lib = L.lib()
    for kv in sys.argv[3:]:                      # library options, e.g. win_xcd=0
        k, v = kv.split("=")
        L.check(lib.rgbnm_set_option(k.encode(), int(v)))
    dt = torch.bfloat16
    rows = []
    tot_f = tot_b = 0.0
    for res, C, heads, nblk in ((64, 96, 3, 2), (32, 192, 6, 2), (16, 384, 12, 6), (8, 768, 24, 2)):
        M = B * res * res
        qkv = torch.randn(M, 3 * C, device=DEV).to(dt)
        bias = torch.randn(heads, 64, 64, device=DEV) * 0.5
        bias_t = bias.transpose(1, 2).contiguous()
        scale = torch.full((heads,), 10.0, device=DEV)
        out = torch.empty(M, C, device=DEV, dtype=dt)
        nwin = B * (res // 8) ** 2
        lse = torch.empty(nwin * heads * 64, device=DEV)
        dout = torch.randn(M, C, device=DEV).to(dt)
        dqkv = torch.empty_like(qkv)
        dbias = torch.empty_like(bias)
        dsp = torch.empty(nwin * heads, device=DEV)
        wsb = lib.rgbnm_window_attention_bwd_workspace(B, res, heads)
        ws = torch.empty(wsb, device=DEV, dtype=torch.uint8)
        for shift in (0, 4):
            if res == 8 and shift:
                continue
            f = lambda: L.check(lib.rgbnm_window_attention_fwd(1, qkv.data_ptr(), bias.data_ptr(), scale.data_ptr(), out.data_ptr(),  # noqa: E731
                                                               lse.data_ptr(), B, res, C, heads, shift, L.stream()))
            g = lambda: L.check(lib.rgbnm_window_attention_bwd(1, qkv.data_ptr(), out.data_ptr(), dout.data_ptr(), bias.data_ptr(),  # noqa: E731
                                                               None, scale.data_ptr(), lse.data_ptr(), dqkv.data_ptr(),
                                                               dbias.data_ptr(), dsp.data_ptr(), B, res, C, heads, shift,
                                                               ws.data_ptr(), wsb, L.stream()))
            tf, tb = timeit(f), timeit(g)
            bf, bb = M * C * 2 * 4 / 1e6, M * C * 2 * 8 / 1e6
            rows.append(dict(res=res, C=C, heads=heads, shift=shift, fwd_us=round(tf, 1), bwd_us=round(tb, 1), fwd_MB=round(bf, 1),
                             bwd_MB=round(bb, 1), fwd_TBps=round(bf / tf, 2), bwd_TBps=round(bb / tb, 2)))
            print(rows[-1], flush=True)
            n = nblk / 2 if res != 8 else nblk       # blocks alternate shift 0 / 4; the last stage (one window) never shifts
            tot_f += tf * n
            tot_b += tb * n
    print("per step (12 blocks): forward %.2f ms, backward %.2f ms" % (tot_f / 1e3, tot_b / 1e3))
    if len(sys.argv) > 2 and sys.argv[2] != "-":
        json.dump(dict(B=B, rows=rows, fwd_ms_per_step=tot_f / 1e3, bwd_ms_per_step=tot_b / 1e3), open(sys.argv[2], "w"), indent=1)


if __name__ == "__main__":
    main()
