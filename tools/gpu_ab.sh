#!/bin/bash
# parity of the changed kernels, then interleaved A/B of one option on the bench step: bash tools/gpu_ab.sh <opt> [pytest -k expr]
OPT=$1; KEXPR=${2:-"fused_mlp or row_panel or live_oracle"}
timeout 900 python -m pytest tests/test_fastpath_model.py tests/test_hip_kernels.py -m gpu -x -q -k "$KEXPR" 2>&1 | tail -4
for r in 1 2 3; do
  for v in 0 1; do
    timeout 300 python bench.py --steps 80 --warmup 10 --no-cpu-baseline --no-parity-check --opt $OPT=$v 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$OPT=$v', d['value'], d['ms_per_step'])"
  done
done
