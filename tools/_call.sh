export TMPDIR=/tmp
OUT=gpurun_out/r06_h; mkdir -p $OUT
timeout 3000 python -m pytest tests/test_held_reductions.py tests/test_fastpath_model.py tests/test_vit_model.py -m gpu -x -q -s 2>&1 | grep -E "worst|passed|failed|Error|assert" | tail -12 | tee $OUT/pytest.txt
for g in 1 0 1 0; do RGBNM_VIT_DWALL=$g timeout 600 python bench.py --arch vits --steps 20 --warmup 4 --no-cpu-baseline --no-trace 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('vits dwall=$g', d['value'], d['ms_per_step'], d['parity_check']['max_abs_dlogit'])" | tee -a $OUT/bench.txt; done
