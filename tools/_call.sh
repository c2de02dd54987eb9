export TMPDIR=/tmp
OUT=gpurun_out/r06_f; mkdir -p $OUT
timeout 3000 python -m pytest tests/test_swin.py tests/test_oracle_swin.py tests/test_bench_n2_gpu.py -m gpu -x -q 2>&1 | tail -4 | tee $OUT/pytest.txt
for h in 1 1 1 1 1 1 1 1; do RGBNM_SWIN_HOLD=$h timeout 600 python bench.py --arch swinv2t --steps 6 --warmup 1 --prewarm-sec 0.2 --no-cpu-baseline --no-trace 2>/tmp/err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('hold=$h', d['value'], d['ms_per_step'], d['config']['launch'][:10])"; grep -o "does not reproduce.*" /tmp/err.txt | cut -c1-600; done
