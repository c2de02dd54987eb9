#!/usr/bin/env python
"""Per-kernel means of arbitrary rocprofv3 --pmc passes.  usage: tools/pmc_kernels.py <out.json> <dir> [<dir> ...]
Each <dir> holds one pass (counter_collection.csv + kernel_trace.csv); kernels are keyed by their (shortened) name and every
counter is averaged per dispatch, so passes with different counters merge into one row per kernel."""
import glob
import json
import os
import re
import sys

import pandas as pd


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"\(.*$", "", n)
    return n.strip()


def main(out, dirs):
    rows = {}
    for d in dirs:
        cc = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
        if not cc:
            print("no counters under", d)
            continue
        df = pd.concat([pd.read_csv(f) for f in cc])
        df["k"] = df.Kernel_Name.map(short)
        per = df.groupby(["k", "Dispatch_Id", "Counter_Name"]).Counter_Value.sum().reset_index()
        mean = per.groupby(["k", "Counter_Name"]).Counter_Value.mean()
        cnt = per.groupby("k").Dispatch_Id.nunique()
        for (k, c), v in mean.items():
            rows.setdefault(k, {})[c] = float(v)
        for k, v in cnt.items():
            rows.setdefault(k, {})["dispatches"] = int(v)
        if kt:
            t = pd.concat([pd.read_csv(f) for f in kt])
            t["k"] = t.Kernel_Name.map(short)
            t["dur"] = t.End_Timestamp - t.Start_Timestamp
            for k, v in t.groupby("k").dur.mean().items():
                rows.setdefault(k, {})["dur_us_under_pmc"] = float(v) / 1e3
    json.dump(rows, open(out, "w"), indent=1, sort_keys=True)
    return rows


if __name__ == "__main__":
    r = main(sys.argv[1], sys.argv[2:])
    for k in sorted(r, key=lambda k: -r[k].get("dur_us_under_pmc", 0) * r[k].get("dispatches", 0))[:14]:
        print(k[:60], {c: (round(v, 1) if isinstance(v, float) else v) for c, v in r[k].items()})
