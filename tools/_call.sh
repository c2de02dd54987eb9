export TMPDIR=/tmp
OUT=gpurun_out/r06_h; mkdir -p $OUT
timeout 3000 python -m pytest tests/test_hip_kernels.py tests/test_vit_model.py tests/test_chain_fwd.py tests/test_chain_bwd.py -m gpu -x -q 2>&1 | tail -4 | tee $OUT/pytest.txt
bash tools/gpu.sh bench 2 2>&1 | cut -c1-60 | tee $OUT/bench.txt
timeout 600 rocprofv3 --kernel-trace -d $OUT/kt -o kt --output-format csv -- python bench.py --steps 24 --warmup 4 --prewarm-sec 1 --no-cpu-baseline --no-trace --no-parity-check > $OUT/kt.log 2>&1
python tools/step_order.py $OUT/kt $OUT/step_order.json 2>&1 | tee $OUT/step_order.txt | grep -E "share|pool|subblock"
rm -rf $OUT/kt
