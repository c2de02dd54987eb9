#!/bin/bash
# A/B of two prebuilt libraries on one box: current build ("new") vs tools/tmp/librgbnm_base.so ("base")
run() { for i in 1 2; do python bench.py --steps 80 --warmup 10 --no-cpu-baseline --no-parity-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'])"; done; }
cp rgb-no-more_amd/librgbnm.so /tmp/new.so
python -m pytest tests/test_fastpath_model.py -m gpu -x -q -k "fused_mlp" 2>&1 | tail -2
run new
cp tools/tmp/librgbnm_base.so rgb-no-more_amd/librgbnm.so; run base
cp /tmp/new.so rgb-no-more_amd/librgbnm.so; run new
cp tools/tmp/librgbnm_base.so rgb-no-more_amd/librgbnm.so; run base
