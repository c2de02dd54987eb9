#!/bin/bash
# kernel-variant probes on the GPU box: rebuild ONE csrc file with a -D switch and time the block's GEMM shapes
OUT=gpurun_out/probe; mkdir -p $OUT
python tools/nt_probe.py base 2>&1 | tee $OUT/base.txt
for V in KP_NOW KP_NOA; do
  touch rgb-no-more_amd/csrc/gemm_nt_kpipe.hip
  RGBNM_HIPCC_FLAGS=-D$V python rgb-no-more_amd/build.py > $OUT/build_$V.log 2>&1 || { tail -5 $OUT/build_$V.log; continue; }
  python tools/nt_probe.py $V 2>&1 | grep kpipe | tee $OUT/$V.txt
done
touch rgb-no-more_amd/csrc/gemm_nt_kpipe.hip; python rgb-no-more_amd/build.py > /dev/null 2>&1
timeout 900 python -m pytest tests/test_train_loop_amp.py tests/test_vit_model.py tests/test_reentrancy.py -m gpu -x -q 2>&1 | tail -15
