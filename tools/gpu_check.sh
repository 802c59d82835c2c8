#!/bin/bash
# One GPU-box visit: parity tests through the C ABI, smoke, a bench line.  Usage: tools/gpu_check.sh [scale]
set -x
SCALE=${1:-1.0}
mkdir -p gpurun_out
python __graft_entry__.py > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 | tee gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3 | tee gpurun_out/smoke.log
timeout 900 python bench.py --steps 3 --warmup 1 --scale $SCALE 2>&1 | tail -5 | tee gpurun_out/bench.log
