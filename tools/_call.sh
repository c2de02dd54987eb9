export TMPDIR=/tmp
OUT=gpurun_out/r06_e; mkdir -p $OUT
timeout 3000 python -m pytest tests -m gpu -x -q --deselect tests/test_bench_n2_gpu.py::test_bench_two_ranks_flat_exchange_with_calibration 2>&1 | tail -8 | tee $OUT/pytest.txt
