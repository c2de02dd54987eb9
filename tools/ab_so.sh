#!/bin/bash
# Same-box A/B of two builds of librgbnm.so (box-to-box variation of the bench is +-3 %, larger than most kernel changes):
#   here:   cp rgb-no-more_amd/librgbnm.so tools/librgbnm_base.so ; <edit> ; python -c "import __graft_entry__ as g; g.build()"
#   on GPU: bash tools/ab_so.sh ["pytest -k expression"]
# tools/librgbnm_base.so is scratch: git-ignored, delete it afterwards.
K=${1:-fused_mlp}   # BENCH_ARGS="--arch vits" STEPS=30 select another config
run() { for i in 1 2; do python bench.py --steps ${STEPS:-80} --warmup 10 --no-cpu-baseline --no-parity-check $BENCH_ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'])"; done; }
cp rgb-no-more_amd/librgbnm.so /tmp/new.so
python -m pytest ${TESTS:-tests/test_fastpath_model.py tests/test_vit_model.py tests/test_hip_kernels.py} -m gpu -x -q -k "$K" 2>&1 | tail -2
run new
cp tools/librgbnm_base.so rgb-no-more_amd/librgbnm.so; run base
cp /tmp/new.so rgb-no-more_amd/librgbnm.so; run new
cp tools/librgbnm_base.so rgb-no-more_amd/librgbnm.so; run base
cp /tmp/new.so rgb-no-more_amd/librgbnm.so
