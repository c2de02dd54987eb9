#!/usr/bin/env python
"""What does a collective's channel cost a 256-workgroups-on-256-CUs kernel?  (VERDICT r4 item 4 i)

RCCL runs one workgroup per channel; while it is resident it holds LDS and wave slots on a CU.  The chain kernels launch exactly one
160 KB-LDS workgroup per image: with B = 256 images and 256 CUs, a CU that cannot take its workgroup pushes one image into a
SECOND ROUND.  This probe keeps k stand-in workgroups (rgbnm_calib_occupy_log: 256 threads, `lds` bytes of LDS each, sleeping or
streaming from L2) resident on a side stream while the JPEG-Ti forward + loss + backward (B = 256, chain kernels) runs on the main
stream, and prints per (k, lds, mode): the step time, the time of the two chain launches and of the grouped weight-gradient launch
(library trace events), and -- from the stand-ins' own residency log -- how many of them were resident for the whole timed region and
on how many distinct CUs.  lds = 64 KB: a chain workgroup (160 KB) does not fit next to it (the RCCL-holds-the-CU case); lds = 0: it
does (pure issue-slot / bandwidth sharing).

usage: cu_steal_probe.py [out.json] [B] [opt=val ...]
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import rgb_no_more_amd as rg
from rgb_no_more_amd import lib as L
from test_chain_fwd import build

out_path = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].isdigit() and "=" not in sys.argv[1] else None
rest = [a for a in sys.argv[1:] if a != out_path]
B = int(rest[0]) if rest and rest[0].isdigit() else 256
lib = L.lib()
for o in rest:
    if "=" in o:
        k, v = o.split("=")
        L.check(lib.rgbnm_set_option(k.encode(), int(v)), k)

m, y, c, tgt = build(12, B)
side = torch.cuda.Stream()
buf = torch.zeros(64 << 20, dtype=torch.uint8, device="cuda")
sink = torch.zeros(4, dtype=torch.int32, device="cuda")
stop = torch.zeros(1, dtype=torch.int32).pin_memory()
N_TIMED = 12
KS = tuple(int(v) for v in os.environ.get("STEAL_KS", "1,2,4,8,16,32").split(","))


def one_step():
    m.zero_grad(set_to_none=False)
    logits = m(y, c)
    rg.cls_transforms.cross_entropy(logits, tgt, grad_dtype=torch.bfloat16).backward()


def timed(n=N_TIMED):
    """ms per step over n steps + the mean launch time of the kernels the library traces, by name"""
    for _ in range(3):
        one_step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(n):
        one_step()
    e1.record()
    e1.synchronize()
    wall = time.perf_counter() - t0
    return e0.elapsed_time(e1) / n, wall


def chain_times(n=4):
    """forward / backward chain launch and the weight-gradient launch timed one by one with events (synchronous, so nothing overlaps)"""
    res = {}
    fw, bw = [], []
    for _ in range(n):
        m.zero_grad(set_to_none=False)
        torch.cuda.synchronize()
        a, b2, c2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        a.record()
        logits = m(y, c)
        b2.record()
        loss = rg.cls_transforms.cross_entropy(logits, tgt, grad_dtype=torch.bfloat16)
        torch.cuda.synchronize()
        d0, d1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        d0.record()
        loss.backward()
        d1.record()
        d1.synchronize()
        fw.append(a.elapsed_time(b2))
        bw.append(d0.elapsed_time(d1))
    res["forward_ms"] = round(sorted(fw)[len(fw) // 2], 4)
    res["backward_ms"] = round(sorted(bw)[len(bw) // 2], 4)
    return res


m.train()
for _ in range(20):
    one_step()
torch.cuda.synchronize()
rows = []
base, _ = timed()
base_parts = chain_times()
print(f"no stand-in: {base:.3f} ms per forward + backward (B = {B}); alone: {base_parts}")
rows.append({"k": 0, "lds": 0, "mode": "-", "ms": round(base, 4), **base_parts})
for mode, mname in ((0, "sleep"), (1, "stream")):
    for lds in (65536, 0):
        for k in KS:
            stop[0] = 0
            log = torch.zeros(3 * k, dtype=torch.int64, device="cuda")
            slice_bytes = (buf.numel() // k) // 4096 * 4096
            # safety net: ends by itself after ~4e9 ticks (a few seconds) should the host flag not be seen
            L.check(lib.rgbnm_calib_occupy_log(buf.data_ptr(), slice_bytes, k, lds, 4_000_000_000, mode, stop.data_ptr(), sink.data_ptr(),
                                               log.data_ptr(), side.cuda_stream), "occupy")
            time.sleep(0.01)                                   # the stand-ins are resident before the first step is queued
            t, wall = timed()
            parts = chain_times()
            stop[0] = 1
            side.synchronize()
            lg = log.cpu().view(k, 3)
            dur_ms = (lg[:, 1] - lg[:, 0]).double() / 1e5          # 100 MHz ticks -> ms
            cus = len({(int(v) >> 32, int(v) & 0xFF00) for v in lg[:, 2].tolist()})     # (XCC, SE | SH | CU) of HW_ID bits 8..15
            xccs = len({int(v) >> 32 for v in lg[:, 2].tolist()})
            resident = int((dur_ms >= 1e3 * wall).sum())
            print(f"{mname:6s} lds {lds:6d} k {k:3d}: {t:.3f} ms (+{t - base:+.3f})  fwd {parts['forward_ms']:.3f} bwd {parts['backward_ms']:.3f}"
                  f"  stand-ins resident throughout: {resident}/{k} (min {dur_ms.min():.0f} ms) on {cus} distinct CUs, {xccs} XCDs")
            rows.append({"k": k, "lds": lds, "mode": mname, "ms": round(t, 4), **parts, "resident_throughout": resident,
                         "standin_min_ms": round(float(dur_ms.min()), 1), "distinct_cus": cus, "xcds": xccs})
res = {"what": "JPEG-Ti forward + loss + backward (chain kernels, eager launches), B = %d, with k resident stand-in workgroups on a side stream; "
               "forward_ms / backward_ms: each half timed alone with events around it" % B,
       "base_ms": round(base, 4), "rows": rows}
if out_path:
    json.dump(res, open(out_path, "w"), indent=1)
print(json.dumps(res))
