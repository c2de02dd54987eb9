#!/bin/bash
# bash tools/gpu_stalls.sh <tag> [bench opts]: SQ activity / wait counters per kernel (three passes) -> gpurun_out/<tag>/stalls.json
TAG=${1:-stalls}; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-parity-check $@"
rocprofv3 --list-avail > $OUT/avail.txt 2>&1
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_ANY"
P2="SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_MFMA"
P3="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE SQ_WAVES"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $P --kernel-trace -d $OUT/p$i -o p --output-format csv -- $B > $OUT/p$i.log 2>&1
  tail -2 $OUT/p$i.log
done
python tools/pmc_kernels.py $OUT/stalls.json $OUT/p1 $OUT/p2 $OUT/p3
rm -rf $OUT/p1 $OUT/p2 $OUT/p3
