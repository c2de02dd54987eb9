#!/usr/bin/env python
"""Build profiles/pmc_traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py.
HBM bytes per launch = 2 * FETCH_SIZE KB (gfx950: FETCH_SIZE tallies 128-B requests as 64 B for wide coalesced reads,
MI355X_MICROARCH.md section HBM; checked here: the dGELU GEMM must read >= 96 MB and reports 51.8 MB) + WRITE_SIZE KB."""
import glob
import json
import os
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


def load(d):
    fs = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    return pd.concat([pd.read_csv(f) for f in fs])


def main(fetch_dir, write_dir, out):
    f, w = load(fetch_dir), load(write_dir)
    res = {}
    for key, pat in (("chain_fwd", "vit_chain_fwd_kernel"), ("chain_bwd", "vit_chain_bwd_kernel"), ("gemm_nt", "gemm_nt_|mlp_fwd_kernel|mlp_bwd_kernel"), ("gemm_tn", "gemm_tn_pipe_kernel"), ("attn_fwd", "attn[23]_fwd_kernel"),
                     ("attn_bwd", "attn[23]_bwd_kernel"), ("dct_resize", "dct_resize_kernel"), ("dct_randaug", "dct_randaug_kernel"),
                     ("subblock_embed", "subblock_embed_kernel"), ("reduce_table", "reduce_table_kernel"),
                     ("adamw", "adamw_kernel"), ("sqnorm", "sqnorm_kernel"), ("patch_gemm", "gemm_nt_kernel"), ("prep_weights", "prep_weights_kernel"), ("window_attn", "win_attn|window_attention")):
        ff = f[f.Kernel_Name.str.contains(pat) & (f.Counter_Name == "FETCH_SIZE")]
        ww = w[w.Kernel_Name.str.contains(pat) & (w.Counter_Name == "WRITE_SIZE")]
        if len(ff) == 0 or len(ww) == 0:
            continue
        fetch = 2.0 * 1024.0 * ff.Counter_Value.mean()
        write = 1024.0 * ww.Counter_Value.mean()
        res[key + "_bytes_per_launch"] = round(fetch + write)
        res[key + "_fetch_bytes_per_launch_corrected_x2"] = round(fetch)
        res[key + "_write_bytes_per_launch"] = round(write)
        res[key + "_launches_sampled"] = int(ff.Dispatch_Id.nunique())
    # the whole step: every dispatch of the pass (kernels of this library and torch's) over the number of optimizer launches
    nf = f[(f.Counter_Name == "FETCH_SIZE") & f.Kernel_Name.str.contains("adamw_kernel")].Dispatch_Id.nunique()
    nw = w[(w.Counter_Name == "WRITE_SIZE") & w.Kernel_Name.str.contains("adamw_kernel")].Dispatch_Id.nunique()
    if nf and nw:
        fetch = 2.0 * 1024.0 * f[f.Counter_Name == "FETCH_SIZE"].Counter_Value.sum() / nf
        write = 1024.0 * w[w.Counter_Name == "WRITE_SIZE"].Counter_Value.sum() / nw
        res["whole_step_bytes"] = round(fetch + write)
        res["whole_step_fetch_bytes_corrected_x2"] = round(fetch)
        res["whole_step_write_bytes"] = round(write)
        res["whole_step_steps_sampled"] = int(min(nf, nw))
    res["note"] = ("mean over all launches of the kernel class in `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` "
                   "passes of `python bench.py --steps 3 --warmup 1 --prewarm-sec 0 --no-cpu-baseline --no-trace`; "
                   "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 half-count of wide coalesced reads)")
    res["measured_at"] = _measured_at()
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3])
