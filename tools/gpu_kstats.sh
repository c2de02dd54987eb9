#!/bin/bash
# rocprofv3 kernel stats of the bench step (optionally with options): bash tools/gpu_kstats.sh <tag> [--opt a=b ...]
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt --output-format csv -- python bench.py --steps 24 --warmup 4 --prewarm-sec 1 --no-cpu-baseline --no-trace "$@" > $OUT/kt.log 2>&1
f=$(find $OUT/kt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/kernel_stats.csv && head -14 $OUT/kernel_stats.csv | cut -c1-150
rm -rf $OUT/kt
