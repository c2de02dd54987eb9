#!/bin/bash
# time one kernel under compile-time variants: bash tools/gpu_variants.sh <file.hip> <kernel-name-pattern> V1 V2 ...
F=$1; PAT=$2; shift; shift
OUT=gpurun_out/variants; mkdir -p $OUT
export TMPDIR=/tmp
for V in base "$@"; do
  touch rgb-no-more_amd/csrc/$F
  if [ "$V" = "base" ]; then python rgb-no-more_amd/build.py > $OUT/build_$V.log 2>&1; else RGBNM_HIPCC_FLAGS="-D$V" python rgb-no-more_amd/build.py > $OUT/build_$V.log 2>&1; fi
  rm -rf $OUT/kt
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt --output-format csv -- python bench.py --steps 16 --warmup 4 --prewarm-sec 1 --no-cpu-baseline --no-trace > $OUT/kt_$V.log 2>&1
  f=$(find $OUT/kt -name "*kernel_stats.csv" | head -1)
  echo "$V: $(grep "$PAT" $f | cut -d, -f1-4 | cut -c1-120)"
done
rm -rf $OUT/kt
touch rgb-no-more_amd/csrc/$F; python rgb-no-more_amd/build.py > /dev/null 2>&1
