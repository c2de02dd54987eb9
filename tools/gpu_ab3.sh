#!/bin/bash
# bash tools/gpu_ab3.sh <opt> <v0> <v1>: the fused-MLP bitwise tests under option value v1, then an A/B of the bench
OPT=$1; V0=$2; V1=$3
timeout 900 python -m pytest tests/test_fastpath_model.py -m gpu -x -q -k "fused_mlp" 2>&1 | tail -3
for r in 1 2 3 4 5; do
  for v in $V0 $V1; do
    timeout 300 python bench.py --steps 80 --warmup 10 --no-cpu-baseline --opt $OPT=$v 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$OPT=$v', d['value'], d['ms_per_step'], d['parity_check']['max_abs_dlogit'])"
  done
done
