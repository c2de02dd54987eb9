#!/bin/bash
# A/B of builds of the two one-launch encoder kernels on ONE box:  gpurun -- 'ROUNDS=3 bash tools/chain_variants.sh "-DX_PRIO=1" "fwd:-DX_NOPIPE" ...'
# Every flag set (plus "base" = as committed) is built first; then ROUNDS rounds run all builds in turn (interleaved: the box's clock drifts
# by 1 - 2 % over a minute, so single runs of different builds do not compare) under rocprofv3 kernel stats; prints the mean kernel times.
# A flag set prefixed with "fwd:" / "bwd:" rebuilds only that kernel with the flags.
export TMPDIR=/tmp
OUT=gpurun_out/chain_variants; mkdir -p $OUT; rm -f $OUT/times.txt
LIB=rgb-no-more_amd/librgbnm.so
build() {   # build <index> <flags>
  local V="$2" only=both
  case "$V" in fwd:*) only=fwd; V="${V#fwd:}";; bwd:*) only=bwd; V="${V#bwd:}";; tn:*) only=tn; V="${V#tn:}";; esac      # tn: = gemm_tn_pipe.hip
  touch rgb-no-more_amd/csrc/vit_chain.hip rgb-no-more_amd/csrc/vit_chain_bwd.hip rgb-no-more_amd/csrc/gemm_tn_pipe.hip
  if [ $only != both ]; then
    python rgb-no-more_amd/build.py > $OUT/build.log 2>&1
    case $only in fwd) touch rgb-no-more_amd/csrc/vit_chain.hip;; bwd) touch rgb-no-more_amd/csrc/vit_chain_bwd.hip;; tn) touch rgb-no-more_amd/csrc/gemm_tn_pipe.hip;; esac
  fi
  if [ "$V" = base ]; then python rgb-no-more_amd/build.py > $OUT/build.log 2>&1; else RGBNM_HIPCC_FLAGS="$V" python rgb-no-more_amd/build.py > $OUT/build.log 2>&1; fi
  if [ $? -ne 0 ]; then echo "$2: BUILD FAILED"; grep -m3 error $OUT/build.log; return 1; fi
  cp $LIB /tmp/variant_$1.so
}
NAMES=(base "$@")
for i in "${!NAMES[@]}"; do build $i "${NAMES[$i]}" || exit 1; done
for r in $(seq 1 ${ROUNDS:-3}); do
  for i in "${!NAMES[@]}"; do
    cp /tmp/variant_$i.so $LIB
    timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt --output-format csv -- python bench.py --steps 40 --warmup 4 --prewarm-sec 1 --no-cpu-baseline --no-trace --no-parity-check > $OUT/kt.log 2>&1
    f=$(find $OUT/kt -name "*kernel_stats.csv" | head -1)
    python - "$f" "$i" >> $OUT/times.txt <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
g = {re.search(r"vit_chain_\w+_kernel", r["Name"]).group(0): float(r["AverageNs"]) / 1e3 for r in rows if "vit_chain" in r["Name"]}
tn = sum(float(r["TotalDurationNs"]) for r in rows if "gemm_tn_pipe" in r["Name"])
tot = sum(float(r["TotalDurationNs"]) for r in rows); steps = [int(r["Calls"]) for r in rows if "vit_chain_fwd" in r["Name"]][0]
print(sys.argv[2], g.get("vit_chain_fwd_kernel", 0), g.get("vit_chain_bwd_kernel", 0), tot / steps / 1e3, tn / steps / 1e3)
PY
    rm -rf $OUT/kt
  done
done
cp /tmp/variant_0.so $LIB
python - $OUT/times.txt "${NAMES[@]}" <<'PY'
import sys
names = sys.argv[2:]
acc = {}
for ln in open(sys.argv[1]):
    i, f, b, t, n = ln.split()
    acc.setdefault(int(i), []).append((float(f), float(b), float(t), float(n)))
for i, n in enumerate(names):
    v = acc.get(i, [])
    if not v:
        continue
    m = [sum(x[k] for x in v) / len(v) for k in range(4)]
    sp = [max(x[k] for x in v) - min(x[k] for x in v) for k in range(4)]
    print(f"{n:40s} fwd {m[0]:7.1f} (+-{sp[0] / 2:4.1f})  bwd {m[1]:7.1f} (+-{sp[1] / 2:4.1f})  dW {m[3]:6.1f} (+-{sp[3] / 2:4.1f})  kernel sum / step {m[2]:7.1f} us  [{len(v)} runs]")
PY
