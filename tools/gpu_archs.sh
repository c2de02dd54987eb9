#!/bin/bash
OUT=gpurun_out/archs; mkdir -p $OUT
timeout 600 python -m pytest tests/test_loader_gpu.py tests/test_train_loop_amp.py -m gpu -x -q 2>&1 | tail -4
timeout 400 python bench.py --steps 60 --warmup 10 > $OUT/vitti.json 2> $OUT/vitti.err; tail -c 2500 $OUT/vitti.json; tail -3 $OUT/vitti.err
timeout 600 python bench.py --arch vits --steps 30 --warmup 5 > $OUT/vits.json 2> $OUT/vits.err; tail -c 1800 $OUT/vits.json; tail -3 $OUT/vits.err
for B in 256 128; do
  timeout 900 python bench.py --arch swinv2t --batch $B --steps 10 --warmup 3 --cpu-baseline-images 32 > $OUT/swin_b$B.json 2> $OUT/swin_b$B.err && break
done
tail -c 1800 $OUT/swin_b$B.json; tail -3 $OUT/swin_b$B.err
timeout 600 python tools/pipeline_bench.py --steps 40 --warmup 6 > $OUT/pipeline.json 2> $OUT/pipeline.err; cat $OUT/pipeline.json; tail -2 $OUT/pipeline.err
