#!/bin/bash
# One entry point for everything that runs on the GPU box (from the repo root):  gpurun -- 'bash tools/gpu.sh <cmd> ...'
#
#   pass <tag> [notests]            THE command that regenerates profiles/: all GPU tests, the bench line, rocprofv3 kernel stats,
#                                   PMC passes (MFMA counters, FETCH / WRITE traffic) -> gpurun_out/<tag>/ ; then, in the container,
#                                   `python tools/collect_profiles.py <tag> rNN_name` copies the summaries into profiles/
#   archs <tag>                     the same for --arch vits and --arch swinv2t (bench line, kernel stats, traffic)
#   tests <pytest args>             selected GPU tests (default: all), output tail
#   bench <n> [bench args]          n runs of bench.py (value, ms/step) -- box-to-box variation is +-3 %, so:
#   abopt <opt> <v0> <v1> [n]       interleaved A/B of one library option on the bench step (same box)
#   abso <base.so> [n] [bench args] interleaved A/B of two BUILDS: the in-tree librgbnm.so against <base.so> (built before an edit;
#                                   scratch copies live under tools/*.so, git-ignored)
#   abmany <n> "<bench args>" a.so b.so ...   the same over several prebuilt libraries
#   abkern <n> "<bench args>" "<regex>" a.so b.so ...   the same under rocprofv3: mean launch time of the matching kernels in the step
#   kstats <tag> [bench args]       rocprofv3 kernel stats of the bench step -> gpurun_out/<tag>/kernel_stats.csv
#   variants <file.hip> <kernel-pattern> <-DFLAG ...>   rebuild ONE csrc file per flag on the box and print that kernel's time
#   stalls <tag> [bench args]       SQ activity / wait / LDS-conflict counters per kernel (three --pmc passes)
#   stress [n]                      bit-identity loop of the fused kernels + a 3000-step run (races show as a differing bit / NaN)
#
# Stand-alone probes (run them directly on the box):  tools/nt384_probe.py E [out|-] M [opt=val]   activation GEMMs of a block vs the library GEMM
#   tools/tn_probe.py E M [opt=val]          weight-gradient GEMMs of a block      tools/winattn_probe.py [B] [out|-] [opt=val]   window attention per stage
#   tools/winattn_prof.py [shift] [fwd]      cycle stamps inside the window-attention kernels (build with RGBNM_HIPCC_FLAGS=-DWIN_PROF[=2])
#   tools/aug_prof.py                        per-wave cycle stamps of the two augment kernels on the bench's data stage (build with RGBNM_HIPCC_FLAGS=-DAUG_PROF)
export TMPDIR=/tmp
CMD=$1; shift
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; print('$1', d['value'], d['ms_per_step'], r.get('avg_launch_us'), (d.get('parity_check') or {}).get('max_abs_dlogit'), 'host', d.get('host_ms_per_step'), 'blocked', d.get('host_blocked_on_rings_ms_per_step'), d.get('host_cpu'))"; }
kstats() {   # kstats <outdir> <bench args...>
  local OUT=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt --output-format csv -- python bench.py --steps 24 --warmup 4 --prewarm-sec 1 --no-cpu-baseline --no-trace "$@" > $OUT/kt.log 2>&1
  local f=$(find $OUT/kt -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/kernel_stats.csv && head -${KSTAT_LINES:-22} $OUT/kernel_stats.csv | cut -d, -f1-4 | cut -c1-150
  python tools/step_order.py $OUT/kt $OUT/step_order.json > $OUT/step_order.txt 2>&1; head -1 $OUT/step_order.txt    # the launches of one step, in order
  rm -rf $OUT/kt
}
traffic() {  # traffic <outdir> <bench args...>: FETCH_SIZE / WRITE_SIZE in separate passes (eager: a graph capture would add dispatches)
  local OUT=$1; shift
  local B="python bench.py --steps 3 --warmup 1 --prewarm-sec 0 --no-cpu-baseline --no-trace --no-graph --no-parity-check $@"
  timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o p --output-format csv -- $B > $OUT/pmc_fetch.log 2>&1
  timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o p --output-format csv -- $B > $OUT/pmc_write.log 2>&1
  python tools/pmc_traffic.py $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_traffic.json 2>&1 | tail -12
}
cleanup() { find $1 -name "*.csv" -size +3M -delete; find $1 -name "*.db" -delete; du -sh $1; }
case $CMD in
pass)
  TAG=${1:-pass}; OUT=gpurun_out/$TAG; mkdir -p $OUT
  if [ "$2" != "notests" ]; then
    timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -40 > $OUT/pytest.log; tail -3 $OUT/pytest.log
  fi
  timeout 200 python tools/calib.py > $OUT/calibration.json 2>/dev/null; cat $OUT/calibration.json
  timeout 400 python bench.py --steps 100 --warmup 10 > $OUT/bench_line.json 2> $OUT/bench.err; cat $OUT/bench_line.json
  timeout 600 python bench.py --dtype fp32 --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_line_fp32.json 2>> $OUT/bench.err      # the mode that carries the 1e-3 tolerance
  timeout 600 python tools/pipeline_bench.py > $OUT/pipeline_jpeg_fed.json 2>> $OUT/bench.err; tail -c 400 $OUT/pipeline_jpeg_fed.json   # JPEG files -> host decode -> H2D -> step (never `value`)
  kstats $OUT
  timeout 900 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $OUT/pmc_mfma -o p --output-format csv -- python bench.py --steps 3 --warmup 1 --prewarm-sec 0 --no-cpu-baseline --no-trace --no-graph --no-parity-check > $OUT/pmc_mfma.log 2>&1
  python tools/pmc_mfma.py $OUT/pmc_mfma $OUT/pmc_mfma.json 2>&1 | tail -30
  traffic $OUT
  cleanup $OUT ;;
archs)
  TAG=${1:-archs}
  for A in vits swinv2t; do
    OUT=gpurun_out/$TAG/$A; mkdir -p $OUT
    timeout 900 python bench.py --arch $A --steps 30 --warmup 5 --cpu-baseline-images 32 > $OUT/bench_line.json 2> $OUT/bench.err; cat $OUT/bench_line.json; tail -2 $OUT/bench.err
    KSTAT_LINES=16 kstats $OUT --arch $A --steps 8
    traffic $OUT --arch $A
    cleanup $OUT
  done ;;
tests)
  timeout ${TEST_TIMEOUT:-2400} python -m pytest "${@:-tests}" -m gpu -q -s 2>&1 | grep -v "^$" | tail -${TAIL:-60} ;;
bench)
  N=$1; shift
  for i in $(seq 1 $N); do timeout 400 python bench.py --steps ${STEPS:-80} --warmup 10 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | line run$i; done ;;
abopt)
  OPT=$1; V0=$2; V1=$3; N=${4:-3}
  for r in $(seq 1 $N); do for v in $V0 $V1; do
    timeout 400 python bench.py --steps ${STEPS:-80} --warmup 10 --no-cpu-baseline --opt $OPT=$v 2>/dev/null | tail -1 | line "$OPT=$v"
  done; done ;;
abso)
  BASE=$1; N=${2:-2}; shift; shift
  cp rgb-no-more_amd/librgbnm.so /tmp/new.so
  for r in $(seq 1 $N); do for v in new base; do
    if [ $v = base ]; then cp $BASE rgb-no-more_amd/librgbnm.so; else cp /tmp/new.so rgb-no-more_amd/librgbnm.so; fi
    timeout 400 python bench.py --steps ${STEPS:-80} --warmup 10 --no-cpu-baseline --no-parity-check "$@" 2>/dev/null | tail -1 | line $v
  done; done
  cp /tmp/new.so rgb-no-more_amd/librgbnm.so ;;
abmany)     # abmany <n> "<bench args>" a.so b.so ...: interleaved rounds over several prebuilt libraries (built in the container, scratch_so/)
  N=$1; ARGS=$2; shift; shift
  cp rgb-no-more_amd/librgbnm.so /tmp/keep.so
  for r in $(seq 1 $N); do for so in "$@"; do
    cp $so rgb-no-more_amd/librgbnm.so
    timeout 400 python bench.py --steps ${STEPS:-80} --warmup 10 --no-cpu-baseline --no-parity-check $ARGS 2>/dev/null | tail -1 | line $(basename $so .so) | cut -d" " -f1-4
  done; done
  cp /tmp/keep.so rgb-no-more_amd/librgbnm.so ;;
abkern)     # abkern <n> "<bench args>" "<kernel regex>" a.so b.so ...: interleaved rounds under rocprofv3; mean launch time of the matching kernels IN THE STEP
  N=$1; ARGS=$2; PAT=$3; shift; shift; shift
  cp rgb-no-more_amd/librgbnm.so /tmp/keep.so; OUT=gpurun_out/abkern; mkdir -p $OUT
  for r in $(seq 1 $N); do for so in "$@"; do
    cp $so rgb-no-more_amd/librgbnm.so
    timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt --output-format csv -- python bench.py --steps ${STEPS:-10} --warmup 4 --prewarm-sec 1 --no-cpu-baseline --no-trace --no-parity-check $ARGS > $OUT/kt.log 2>&1
    python - "$(find $OUT/kt -name "*kernel_stats.csv" | head -1)" "$PAT" "$(basename $so .so)" <<'PY'
import csv, re, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if re.search(sys.argv[2], r["Name"])]
print(sys.argv[3], "  ".join(f"{re.sub(r'.*::', '', r['Name'].split('(')[0])[:34]} {float(r['AverageNs']) / 1e3:.1f}" for r in rows))
PY
    rm -rf $OUT/kt
  done; done
  cp /tmp/keep.so rgb-no-more_amd/librgbnm.so ;;
kstats)
  TAG=$1; shift; OUT=gpurun_out/$TAG; mkdir -p $OUT; kstats $OUT "$@" ;;
variants)
  F=$1; PAT=$2; shift; shift
  OUT=gpurun_out/variants; mkdir -p $OUT
  for V in base "$@"; do
    touch rgb-no-more_amd/csrc/$F
    if [ "$V" = base ]; then python rgb-no-more_amd/build.py > $OUT/build.log 2>&1; else RGBNM_HIPCC_FLAGS="$V" python rgb-no-more_amd/build.py > $OUT/build.log 2>&1; fi
    if [ $? -ne 0 ]; then echo "$V: BUILD FAILED"; grep -m3 "error" $OUT/build.log; continue; fi
    KSTAT_LINES=40 kstats $OUT $VARIANT_BENCH_ARGS | grep "$PAT" | sed "s/^/$V: /"
  done
  touch rgb-no-more_amd/csrc/$F; python rgb-no-more_amd/build.py > /dev/null 2>&1 ;;
stalls)
  TAG=${1:-stalls}; shift; OUT=gpurun_out/$TAG; mkdir -p $OUT
  B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-parity-check --no-graph $@"
  P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_ANY"
  P2="SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_MFMA"
  P3="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE SQ_WAVES"
  i=0
  for P in "$P1" "$P2" "$P3"; do
    i=$((i+1)); timeout 600 rocprofv3 --pmc $P --kernel-trace -d $OUT/p$i -o p --output-format csv -- $B > $OUT/p$i.log 2>&1; tail -2 $OUT/p$i.log
  done
  python tools/pmc_kernels.py $OUT/stalls.json $OUT/p1 $OUT/p2 $OUT/p3; rm -rf $OUT/p1 $OUT/p2 $OUT/p3 ;;
stress)
  N=${1:-20}; fail=0
  for i in $(seq 1 $N); do
    timeout 600 python -m pytest tests/test_fastpath_model.py -m gpu -x -q -k "fused_mlp" 2>&1 | tail -1 | grep -q "passed" || { echo "iteration $i FAILED"; fail=1; break; }
  done
  echo "bit-identity loop: $N iterations, fail=$fail"
  timeout 600 python bench.py --steps 3000 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | line long ;;
*) echo "unknown command $CMD"; exit 2 ;;
esac
