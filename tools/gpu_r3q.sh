#!/bin/bash
# round 3: the gpu tests and the sketch-side soaks on the threshold window pass as the default
TAG=${1:-r03q}
O=gpurun_out/$TAG; mkdir -p $O
python __graft_entry__.py > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee $O/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2 | tee $O/smoke.log
timeout 1500 python tests/gpu_volume_soak.py C3 10 2e9 700 > $O/volume_soak_c3.log 2>&1; tail -1 $O/volume_soak_c3.log
timeout 1500 python tests/gpu_map_soak.py C3 6 2e9 400 > $O/map_soak_c3.log 2>&1; tail -2 $O/map_soak_c3.log
timeout 900 python tests/gpu_soak.py 240 > $O/soak_fuzz.log 2>&1; tail -2 $O/soak_fuzz.log
