#!/bin/bash
# held gradient reductions: tests, then A/B of the bench with and without --no-defer-reduce
timeout 900 python -m pytest tests/test_held_reductions.py -m gpu -x -q 2>&1 | tail -5
for r in 1 2 3 4; do
  for f in "--no-defer-reduce" ""; do
    timeout 300 python bench.py --steps 80 --warmup 10 --no-cpu-baseline $f 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('held' if '$f'=='' else 'plain', d['value'], d['ms_per_step'], d['parity_check']['max_abs_dlogit'], d['parity_check']['gradnorm_rel_err_median'])"
  done
done
