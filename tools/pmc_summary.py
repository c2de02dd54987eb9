#!/usr/bin/env python
"""Aggregate rocprofv3 --pmc csv output (counter_collection.csv) into mean counter values per kernel(+grid).
usage: tools/pmc_summary.py <dir> [out.txt]"""
import glob
import os
import re
import sys

import pandas as pd


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    m = re.search(r"(gemm_nt_kernel|gemm_tn_kernel|attn_\w+_kernel|ln_\w+_kernel|reduce_partials_kernel|pool_\w+|subblock_\w+|softxent\w*|adamw\w*|prep_\w+)", n)
    tag = m.group(1) if m else n[:40]
    t = re.search(r"I(DF16b|f)(Li\d+E)*(Lb\d)?", n)
    return tag + ("<" + t.group(0) + ">" if t else "")


def main(d, out=None):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        print("no counter_collection.csv under", d)
        return
    df = pd.concat([pd.read_csv(f) for f in files])
    df["k"] = df["Kernel_Name"].map(short) + " g=" + df["Grid_Size"].astype(str)
    piv = df.pivot_table(index="k", columns="Counter_Name", values="Counter_Value", aggfunc="mean")
    piv["calls"] = df.groupby("k")["Dispatch_Id"].nunique()
    pd.set_option("display.width", 250, "display.max_columns", 30, "display.max_rows", 200, "display.float_format", lambda v: f"{v:,.0f}")
    txt = piv.sort_values(piv.columns[0], ascending=False).to_string()
    print(txt)
    if out:
        open(out, "w").write(txt + "\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
