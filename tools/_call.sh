export TMPDIR=/tmp
bash tools/gpu.sh stalls r06_stalls 2>&1 | tail -5
bash tools/gpu.sh stalls r06_stalls_vits --arch vits 2>&1 | tail -3
bash tools/gpu.sh stalls r06_stalls_swin --arch swinv2t 2>&1 | tail -3
bash tools/gpu.sh stress 6 2>&1 | tail -3
