#!/bin/bash
# HIP graph replay vs eager launches of the bench step
for r in 1 2 3 4 5 6; do
  for f in "--no-graph" ""; do
    timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-parity-check $f $@ 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('graph' if '$f'=='' else 'eager', d['ms_per_step'], end='  ')"
  done
  echo
done
