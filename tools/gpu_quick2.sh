#!/bin/bash
timeout 600 python -m pytest tests/test_fastpath_model.py -m gpu -x -q -k "fused_mlp or live_oracle" 2>&1 | tail -3
bash tools/gpu_kstats.sh ks2 "$@" | cut -c1-140
