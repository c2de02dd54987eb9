#!/bin/bash
# Rebuild BOTH one-launch encoder kernels per flag set on the box and print their time under rocprofv3 kernel stats (same box, so the
# variants compare):  gpurun -- 'bash tools/chain_variants.sh "-DX_PRIO=1" "-DX_PRIO=2" ...'   ("base" always runs first and last)
export TMPDIR=/tmp
OUT=gpurun_out/chain_variants; mkdir -p $OUT
one() {
  local V="$1"
  # a flag set prefixed with "fwd:" / "bwd:" rebuilds only that kernel with the flags (the other one as committed)
  local only=both
  case "$V" in fwd:*) only=fwd; V="${V#fwd:}";; bwd:*) only=bwd; V="${V#bwd:}";; esac
  touch rgb-no-more_amd/csrc/vit_chain.hip rgb-no-more_amd/csrc/vit_chain_bwd.hip
  if [ $only != both ]; then
    python rgb-no-more_amd/build.py > $OUT/build.log 2>&1            # both as committed first
    [ $only = fwd ] && touch rgb-no-more_amd/csrc/vit_chain.hip || touch rgb-no-more_amd/csrc/vit_chain_bwd.hip
  fi
  if [ "$V" = base ]; then python rgb-no-more_amd/build.py > $OUT/build.log 2>&1; else RGBNM_HIPCC_FLAGS="$V" python rgb-no-more_amd/build.py > $OUT/build.log 2>&1; fi
  if [ $? -ne 0 ]; then echo "$V: BUILD FAILED"; grep -m3 error $OUT/build.log; return; fi
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt --output-format csv -- python bench.py --steps 40 --warmup 4 --prewarm-sec 1 --no-cpu-baseline --no-trace --no-parity-check > $OUT/kt.log 2>&1
  local f=$(find $OUT/kt -name "*kernel_stats.csv" | head -1)
  python - "$f" "$V" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
import re
g = {re.search(r"vit_chain_\w+_kernel", r["Name"]).group(0): float(r["AverageNs"]) / 1e3 for r in rows if "vit_chain" in r["Name"]}
tot = sum(float(r["TotalDurationNs"]) for r in rows); steps = [int(r["Calls"]) for r in rows if "vit_chain_fwd" in r["Name"]][0]
print(f"{sys.argv[2]:50s} fwd {g.get('vit_chain_fwd_kernel', 0):8.1f} us  bwd {g.get('vit_chain_bwd_kernel', 0):8.1f} us  kernel sum / step {tot / steps / 1e3:8.1f} us")
PY
  rm -rf $OUT/kt
}
one base
for V in "$@"; do one "$V"; done
one base
