#!/usr/bin/env python
"""Copy the summaries of one `tools/gpu.sh pass <tag>` / `archs <tag>` run from gpurun_out/ (scratch) into profiles/ (tracked).

    python tools/collect_profiles.py <tag> <name>          e.g.  collect_profiles.py final r03_final
    python tools/collect_profiles.py <tag>/vits <name>     for the per-arch directories of `archs`

Writes profiles/<name>_{bench_line.json,kernel_stats.csv,pmc_mfma.json,pmc_traffic.json} (those that exist) and refreshes
profiles/{pmc_traffic,pmc_mfma,calibration}[_<arch>].json, the files bench.py quotes its file-sourced fields from."""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _head():
    import subprocess
    try:
        return subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None
    except Exception:       # noqa: BLE001
        return None


def _stamp_head(path):
    """measured_at.head = the commit checked out when the summary was copied (the GPU box has no .git; the source hash the box
    computed is what bench.py checks, HEAD is for the reader)."""
    try:
        d = json.load(open(path))
        if isinstance(d, dict):
            d.setdefault("measured_at", {})["head"] = _head()
            json.dump(d, open(path, "w"), indent=1)
    except Exception:       # noqa: BLE001
        pass


def main(tag, name):
    src = os.path.join(ROOT, "gpurun_out", tag)
    dst = os.path.join(ROOT, "profiles")
    arch = "vitti"
    for f in ("bench_line.json", "bench_line_fp32.json", "kernel_stats.csv", "step_order.txt", "pmc_mfma.json", "pmc_traffic.json",
              "calibration.json", "pipeline_jpeg_fed.json"):
        p = os.path.join(src, f)
        if os.path.exists(p) and os.path.getsize(p) > 0:
            shutil.copy(p, os.path.join(dst, f"{name}_{f}"))
            print("profiles/" + f"{name}_{f}")
            if f == "bench_line.json":
                try:
                    arch = {"JPEG-Ti": "vitti", "JPEG-S": "vits", "SwinV2-T": "swinv2t"}[json.load(open(p))["metric"].split()[1]]
                except Exception:       # noqa: BLE001
                    pass
    for stem in ("pmc_traffic", "pmc_mfma", "calibration"):        # the files bench.py quotes (labelled file-sourced on its line)
        t = os.path.join(src, stem + ".json")
        if os.path.exists(t) and os.path.getsize(t) > 0:
            out = f"{stem}.json" if arch == "vitti" else f"{stem}_{arch}.json"
            shutil.copy(t, os.path.join(dst, out))
            _stamp_head(os.path.join(dst, out))
            print("profiles/" + out)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
