#!/bin/bash
# One GPU-box pass: parity tests, bench line, rocprofv3 kernel stats, PMC passes (MFMA counters, HBM traffic).
# usage (from the repo root on the GPU box): bash tools/gpu_pass.sh <tag> [tests|notests]
TAG=${1:-pass}
MODE=${2:-tests}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
if [ "$MODE" = "tests" ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -40 > $OUT/pytest.log
  tail -3 $OUT/pytest.log
fi
timeout 300 python bench.py --steps 100 --warmup 10 > $OUT/bench_line.json 2> $OUT/bench.err
cat $OUT/bench_line.json
BENCH="python bench.py --steps 24 --warmup 4 --prewarm-sec 1 --no-cpu-baseline --no-trace"
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt --output-format csv -- $BENCH > $OUT/kt.log 2>&1
f=$(find $OUT/kt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/kernel_stats.csv && head -25 $OUT/kernel_stats.csv
PMCB="python bench.py --steps 3 --warmup 1 --prewarm-sec 0 --no-cpu-baseline --no-trace --no-graph"   # eager: the capture passes of graph mode would add model-only dispatches to the counters
timeout 900 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $OUT/pmc_mfma -o p --output-format csv -- $PMCB > $OUT/pmc_mfma.log 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o p --output-format csv -- $PMCB > $OUT/pmc_fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o p --output-format csv -- $PMCB > $OUT/pmc_write.log 2>&1
python tools/pmc_traffic.py $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_traffic.json > /dev/null 2>&1
python tools/pmc_mfma.py $OUT/pmc_mfma $OUT/pmc_mfma.json 2>&1 | tail -30
# keep only the small summaries (gpurun_out merge limit)
find $OUT -name "*.csv" -size +3M -delete
find $OUT -name "*.db" -delete
du -sh $OUT
