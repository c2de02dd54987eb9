#!/bin/bash
# run selected GPU tests: bash tools/gpu_t.sh <pytest args...>
timeout 1500 python -m pytest "$@" -m gpu -q -s 2>&1 | grep -v "^$" | tail -60
