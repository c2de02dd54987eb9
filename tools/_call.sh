export TMPDIR=/tmp
OUT=gpurun_out/r06_a; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_chain_fwd.py tests/test_vit_model.py tests/test_chain_bwd.py tests/test_held_reductions.py tests/test_fastpath_model.py tests/test_reentrancy.py tests/test_train_loop_amp.py -m gpu -x -q 2>&1 | tail -15 | tee $OUT/pytest.txt
bash tools/gpu.sh bench 3 2>&1 | tee $OUT/bench.txt
