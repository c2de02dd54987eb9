#!/usr/bin/env python
"""The launches of ONE steady-state bench step in start order, from a rocprofv3 --kernel-trace CSV: start offset, duration and the
idle gap in front of each (next start - latest end so far), averaged over the steady steps.  A step is cut at the optimizer kernel
(adamw).  usage: step_order.py <dir-or-csv> [out.json]"""
import csv
import glob
import json
import os
import re
import sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::|void |at::native::", "", n)
    m = re.match(r"_ZN\d+_GLOBAL__N_1(\d+)", n)
    if m:
        k = int(m.group(1))
        i = m.end()
        return n[i:i + k]
    m = re.match(r"_ZN5aug28(\d+)", n)
    if m:
        k = int(m.group(1))
        return n[m.end():m.end() + k]
    return n.split("(")[0][:60]


def main(path, out=None):
    if os.path.isdir(path):
        path = sorted(glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True))[0]
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    cuts = [i for i, r in enumerate(rows) if r[2].startswith("adamw")]
    if len(cuts) < 8:
        print("not enough steps", len(cuts))
        return
    steps = []
    for a, b in zip(cuts[3:-2], cuts[4:-1]):
        steps.append(rows[a + 1:b + 1])
    # the most common launch sequence
    seqs = {}
    for s in steps:
        seqs.setdefault(tuple(r[2] for r in s), []).append(s)
    seq, group = max(seqs.items(), key=lambda kv: len(kv[1]))
    n = len(group)
    res = []
    tot_d = tot_g = 0.0
    for k, name in enumerate(seq):
        d = sum(s[k][1] - s[k][0] for s in group) / n / 1e3
        g = 0.0
        if k:
            g = sum(max(0, s[k][0] - max(r[1] for r in s[:k])) for s in group) / n / 1e3
        tot_d += d
        tot_g += g
        res.append({"i": k, "kernel": name, "us": round(d, 1), "gap_before_us": round(g, 1)})
    span = sum(s[-1][1] - s[0][0] for s in group) / n / 1e3
    print(f"{n} of {len(steps)} steps share this sequence of {len(seq)} launches; kernel sum {tot_d:.1f} us, gaps {tot_g:.1f} us, span {span:.1f} us")
    for r in res:
        print(f"{r['i']:3d} {r['kernel']:60s} {r['us']:9.1f} us   gap {r['gap_before_us']:7.1f}")
    if out:
        json.dump({"steps": n, "launches": len(seq), "kernel_sum_us": tot_d, "gaps_us": tot_g, "span_us": span, "sequence": res},
                  open(out, "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
