#!/usr/bin/env python
"""MFMA utilisation from hardware counters (north_star: "MFMA utilisation from rocprof").

Input: one `rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES
GRBM_GUI_ACTIVE --kernel-trace` pass of bench.py.  Per kernel class and for the whole step:
  mfma_flops      = SQ_INSTS_VALU_MFMA_MOPS_BF16 * 512        (the counter ticks once per 512 bf16 FLOPs: a 32x32x16
                    MFMA = 32768 FLOP = 64 MOPS; calibrated by the gemm rows, whose MFMA count is known exactly)
  mfma_busy_frac  = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * 1024 SIMDs ... reported raw and as a ratio to
                    GRBM_GUI_ACTIVE * 4 SIMD * 256 CU)
  mfma_util       = mfma_flops / (kernel wall time * 2.5 PFLOP/s)
usage: tools/pmc_mfma.py <dir> <out.json>"""
import glob
import json
import os
import re
import sys

import pandas as pd


def _measured_at():
    """The tree these counters were measured on: hash of the kernel sources (rgb_no_more_amd.lib.source_hash) -- bench.py compares
    it with the tree it runs from and marks file-sourced figures `stale` when they differ; tools/collect_profiles.py adds HEAD."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    try:
        from rgb_no_more_amd import lib as L
        return {"csrc_sha16": L.source_hash()}
    except Exception as e:       # noqa: BLE001
        return {"csrc_sha16": None, "error": str(e)}


PEAK = 2.5e15
CLASSES = (("chain_fwd", r"vit_chain_fwd_kernel"), ("chain_bwd", r"vit_chain_bwd_kernel"), ("mlp_fused", r"mlp_fwd_kernel|mlp_bwd_kernel"), ("gemm_nt", r"gemm_nt_"), ("gemm_tn", r"gemm_tn_"), ("attn_fwd", r"attn[23]_fwd"), ("attn_bwd", r"attn[23]_bwd"),
           ("layernorm", r"ln_|layernorm|pool_"), ("augment", r"dct_"), ("embed", r"subblock|embed"),
           ("tail", r"adamw|sqnorm|softxent|mixup|prep_|reduce_"))


def main(d, out):
    cc = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    if not cc:
        print("no counter_collection.csv under", d)
        return
    df = pd.concat([pd.read_csv(f) for f in cc])
    piv = df.pivot_table(index=["Dispatch_Id", "Kernel_Name"], columns="Counter_Name", values="Counter_Value",
                         aggfunc="sum").reset_index()
    dur = None
    if kt:
        k = pd.concat([pd.read_csv(f) for f in kt])
        k["dur_ns"] = k["End_Timestamp"] - k["Start_Timestamp"]
        dur = k[["Dispatch_Id", "dur_ns"]]
        piv = piv.merge(dur, on="Dispatch_Id", how="left")
    res = {"note": __doc__.split("usage")[0].strip().splitlines()[0], "classes": {}}
    cols = [c for c in piv.columns if c not in ("Dispatch_Id", "Kernel_Name", "dur_ns")]

    def summarise(sel, name):
        if len(sel) == 0:
            return
        r = {"dispatches": int(len(sel))}
        for c in cols:
            r[c] = float(sel[c].sum())
        if "dur_ns" in sel:
            r["kernel_time_ms"] = float(sel["dur_ns"].sum()) / 1e6
        mops = r.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0)
        r["mfma_flops"] = mops * 512.0
        if r.get("kernel_time_ms"):
            r["mfma_util_vs_2.5PF"] = round(r["mfma_flops"] / (r["kernel_time_ms"] * 1e-3) / PEAK, 4)
        if r.get("GRBM_GUI_ACTIVE") and "SQ_VALU_MFMA_BUSY_CYCLES" in r:
            r["mfma_busy_over_gui_active_x1024simd"] = round(r["SQ_VALU_MFMA_BUSY_CYCLES"] / (r["GRBM_GUI_ACTIVE"] * 1024.0), 4)
        res["classes"][name] = r

    used = pd.Series(False, index=piv.index)
    for name, pat in CLASSES:
        m = piv.Kernel_Name.str.contains(pat, regex=True) & ~used
        used |= m
        summarise(piv[m], name)
    summarise(piv[~used], "other")
    summarise(piv, "whole_run")
    res["measured_at"] = _measured_at()
    json.dump(res, open(out, "w"), indent=1)
    for k, v in res["classes"].items():
        print(f"{k:10s} n={v['dispatches']:5d} t={v.get('kernel_time_ms', 0):9.3f} ms  mfma_flops={v['mfma_flops']:.3e} "
              f"util={v.get('mfma_util_vs_2.5PF')} busy={v.get('mfma_busy_over_gui_active_x1024simd')}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
