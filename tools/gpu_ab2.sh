#!/bin/bash
# bash tools/gpu_ab2.sh <opt> <v0> <v1> [pytest -k expr]
OPT=$1; V0=$2; V1=$3; KEXPR=${4:-"fused_mlp"}
timeout 900 python -m pytest tests/test_fastpath_model.py -m gpu -x -q -k "$KEXPR" 2>&1 | tail -4
for r in 1 2 3; do
  for v in $V0 $V1; do
    timeout 300 python bench.py --steps 80 --warmup 10 --no-cpu-baseline --no-parity-check --opt $OPT=$v 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$OPT=$v', d['value'], d['ms_per_step'])"
  done
done
