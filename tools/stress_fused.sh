#!/bin/bash
# Repeat the bit-identity tests of the fused MLP kernels (LDS-flag hand-over to the DMA wave) and a long bench run: a race would
# show up as a differing bit or a non-finite loss.
N=${1:-20}
fail=0
for i in $(seq 1 $N); do
  timeout 600 python -m pytest tests/test_fastpath_model.py -m gpu -x -q -k "fused_mlp" 2>&1 | tail -1 | grep -q "2 passed" || { echo "iteration $i FAILED"; fail=1; break; }
done
echo "bit-identity loop: $N iterations, fail=$fail"
timeout 600 python bench.py --steps 3000 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('long run', d['value'], d['ms_per_step'], 'loss', d['config']['loss'], d['parity_check']['ok'])"
