#!/bin/bash
# long soak: volume (rare near-tie paths of the 32-bit window pass) at the bench parameters and at other (k, w), then the fuzz soak
TAG=${1:-r02bk}
mkdir -p gpurun_out/$TAG
python __graft_entry__.py > gpurun_out/$TAG/build.log 2>&1 || { tail -20 gpurun_out/$TAG/build.log; exit 1; }
run() { timeout 1500 python tests/gpu_volume_soak.py "$@" 2>&1 | tail -1 | tee -a gpurun_out/$TAG/volume.log; }
run C3 100 2.0e9 1000
run C5 60 2.0e9 2000
run C3 10 2.0e9 3000 20 16
run C3 10 2.0e9 3100 32 1000
run C3 10 2.0e9 3200 40 64
run C3 10 2.0e9 3300 15 100
run C3 10 2.0e9 3400 100 255
run C5 10 2.0e9 3500 24 33
timeout 900 python tests/gpu_soak.py 600 9000 2>&1 | tail -2 | tee gpurun_out/$TAG/fuzz.log
