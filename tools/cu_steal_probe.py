#!/usr/bin/env python
"""What does a collective's channel cost a 256-workgroups-on-256-CUs kernel?  (VERDICT r4 item 4 i)

RCCL runs one workgroup per channel; while it is resident it holds LDS and wave slots on a CU.  The chain kernels launch exactly one
156 - 160 KB-LDS workgroup per image: with B = 256 images and 256 CUs, a CU that cannot take its workgroup pushes one image into a
SECOND ROUND.  This probe keeps k stand-in workgroups (rgbnm_calib_occupy: 256 threads, `lds` bytes of LDS each, sleeping or
streaming from L2) resident on a side stream while the JPEG-Ti train step (forward + loss + backward, B = 256, chain kernels) runs
on the main stream, and prints the step time per (k, lds, mode).  lds = 64 KB: the chain workgroup does not fit next to it (the
RCCL-holds-the-CU case); lds = 0: it does (pure issue-slot / bandwidth sharing).

usage: cu_steal_probe.py [out.json] [B] [opt=val ...]
"""
import json
import os
import sys

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


def one_step():
    m.zero_grad(set_to_none=False)
    logits = m(y, c)
    rg.cls_transforms.cross_entropy(logits, tgt, grad_dtype=torch.bfloat16).backward()


def timed(n=12):
    for _ in range(3):
        one_step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        one_step()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / n


m.train()
for _ in range(20):
    one_step()
torch.cuda.synchronize()
rows = []
base = timed()
print(f"no stand-in: {base:.3f} ms per forward + backward (B = {B})")
rows.append({"k": 0, "lds": 0, "mode": "-", "ms": round(base, 4)})
for mode, mname in ((0, "sleep"), (1, "stream")):
    for lds in (65536, 0):
        for k in (1, 2, 4, 8, 16, 32):
            stop[0] = 0
            slice_bytes = (buf.numel() // k) // 4096 * 4096
            # safety net: ends by itself after ~1.5e9 ticks (about a second) should the host flag not be seen
            L.check(lib.rgbnm_calib_occupy(buf.data_ptr(), slice_bytes, k, lds, 1_500_000_000, mode, stop.data_ptr(), sink.data_ptr(),
                                           side.cuda_stream), "occupy")
            t = timed()
            stop[0] = 1
            side.synchronize()
            print(f"{mname:6s} lds {lds:6d} k {k:3d}: {t:.3f} ms  (+{t - base:+.3f})")
            rows.append({"k": k, "lds": lds, "mode": mname, "ms": round(t, 4)})
res = {"what": "JPEG-Ti forward + loss + backward (chain kernels, eager launches), B = %d, with k resident stand-in workgroups on a side stream" % B,
       "base_ms": round(base, 4), "rows": rows}
if out_path:
    json.dump(res, open(out_path, "w"), indent=1)
print(json.dumps(res))
