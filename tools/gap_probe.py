#!/usr/bin/env python
"""Where the GPU idles inside a bench step: reads a rocprofv3 --kernel-trace CSV (columns Kernel_Name, Start_Timestamp,
End_Timestamp), orders the dispatches by start time and sums the gaps (next start - latest end so far) by the pair of kernels
around them.  usage: gap_probe.py <dir-or-csv> [out.json] [min_gap_us]"""
import csv
import glob
import json
import os
import sys


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    for k in ("vit_chain_bwd", "vit_chain_fwd", "gemm_tn_pipe", "reduce_table", "dct_resize", "dct_randaug", "subblock_embed", "adamw",
              "chain_gather", "prep_weights", "mixup_target", "mixup", "pool_bwd", "pool_fwd", "sqnorm", "softxent", "gather_bias",
              "mean_kernel", "gemm_nt_small", "gemm_nt_kernel", "copyBuffer", "fillBuffer"):
        if k in n:
            return k
    return n[:40]


def main(path, out=None, min_gap_us=1.0):
    if os.path.isdir(path):
        path = sorted(glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True))[0]
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    # the steady part: from the 20th chain backward to the last one
    bw = [i for i, r in enumerate(rows) if r[2] == "vit_chain_bwd"]
    if len(bw) < 30:
        print("not enough steps in the trace", len(bw))
        return
    lo, hi = bw[19], bw[-2]
    steps = len([i for i in bw if lo <= i < hi])
    span = rows[hi][0] - rows[lo][0]
    busy = 0
    gaps = {}
    end = rows[lo][1]
    busy += rows[lo][1] - rows[lo][0]
    for i in range(lo + 1, hi):
        s, e, n = rows[i]
        g = s - end
        if g > 0:
            key = f"{rows[i - 1][2]} -> {n}"
            a = gaps.setdefault(key, [0, 0.0, 0])
            a[0] += 1
            a[1] += g / 1e3
            if g / 1e3 >= min_gap_us:
                a[2] += 1
        busy += max(0, e - max(s, end))
        end = max(end, e)
    res = {"trace": os.path.basename(path), "steps": steps, "span_ms_per_step": span / steps / 1e6, "busy_ms_per_step": busy / steps / 1e6,
           "idle_ms_per_step": (span - busy) / steps / 1e6,
           "gaps_us_per_step": {k: {"count_per_step": round(v[0] / steps, 2), "us_per_step": round(v[1] / steps, 2)}
                                for k, v in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:25]}}
    print(json.dumps(res, indent=1))
    if out:
        json.dump(res, open(out, "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 and sys.argv[2] != "-" else None, float(sys.argv[3]) if len(sys.argv) > 3 else 1.0)
