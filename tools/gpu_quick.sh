#!/bin/bash
# quick GPU check of a kernel change: the fast-path parity tests, then A/B of one option on the bench step
OUT=gpurun_out/quick; mkdir -p $OUT
timeout 900 python -m pytest tests/test_fastpath_model.py -m gpu -x -q -s 2>&1 | grep -v "^$" | tail -40
for o in "$@"; do
  timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --opt $o 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$o', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'] if d['roofline'] else None)"
done
