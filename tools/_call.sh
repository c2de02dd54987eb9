export TMPDIR=/tmp
OUT=gpurun_out/r06_d; mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $OUT/pytest.txt
timeout 400 python bench.py --steps 100 --warmup 10 > $OUT/bench_line.json 2> $OUT/bench.err; tail -c 600 $OUT/bench.err; python -c "
import json; d=json.load(open('$OUT/bench_line.json')); print(d['value'], d['ms_per_step']); r=d['roofline']; print({k:r[k] for k in ('bound','achieved','peak','frac','traffic','avg_launch_us')}); print(r['hbm']); print(r['traffic_source']); print([ (x['kernel'][:24], x['bound'], x['frac'], x['us_per_step']) for x in d['roofline_kernels']]); print(d['whole_step']); print(d['cpu_baseline'].get('reference_itself'))"
timeout 600 python bench.py --dtype fp32 --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_line_fp32.json 2> $OUT/bench_fp32.err; tail -c 300 $OUT/bench_fp32.err; python -c "
import json; d=json.load(open('$OUT/bench_line_fp32.json')); print('fp32', d['value'], d['ms_per_step'], d['parity_check'])"
