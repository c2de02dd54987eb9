export TMPDIR=/tmp
OUT=gpurun_out/r06_c; mkdir -p $OUT
for r in 1 2 3; do
bash tools/gpu.sh bench 1 2>&1 | cut -c1-40 | sed 's/^/trace   /' | tee -a $OUT/bench2.txt
bash tools/gpu.sh bench 1 --no-trace 2>&1 | cut -c1-40 | sed 's/^/notrace /' | tee -a $OUT/bench2.txt
bash tools/gpu.sh bench 1 --no-trace --no-graph 2>&1 | cut -c1-40 | sed 's/^/nograph /' | tee -a $OUT/bench2.txt
done
