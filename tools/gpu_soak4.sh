#!/bin/bash
# rounds 4-6: soaks of the window pass (sketch_wave_kernel): the pytest volume tests, the randomised threshold-pass soak, volume at the
# bench parameters
TAG=${1:-r04s}
mkdir -p gpurun_out/$TAG
python __graft_entry__.py > gpurun_out/$TAG/build.log 2>&1 || { tail -20 gpurun_out/$TAG/build.log; exit 1; }
timeout 1200 python -m pytest tests/test_gpu_soak.py -m gpu -x -q --durations=5 2>&1 | tail -12 | tee gpurun_out/$TAG/pytest_soak.log
timeout 900 python tests/gpu_thresh_soak.py ${THRESH_S:-300} 7000 2>&1 | tail -3 | tee gpurun_out/$TAG/thresh_soak.log
timeout 1500 python tests/gpu_volume_soak.py C3 ${C3_BATCHES:-10} 1.5e9 600 2>&1 | tail -2 | tee gpurun_out/$TAG/volume_C3.log
timeout 1500 python tests/gpu_volume_soak.py C5 ${C5_BATCHES:-6} 1.5e9 620 2>&1 | tail -2 | tee gpurun_out/$TAG/volume_C5.log
timeout 900 python tests/gpu_volume_soak.py C3 ${W500_BATCHES:-4} 1.5e9 640 32 500 2>&1 | tail -2 | tee gpurun_out/$TAG/volume_C3_w500.log
timeout 900 python tests/gpu_map_soak.py C3 ${MAP_BATCHES:-6} 1e9 660 0 2>&1 | tail -2 | tee gpurun_out/$TAG/map_C3_for_map.log
timeout 900 python tests/gpu_soak.py 2>&1 | tail -2 | tee gpurun_out/$TAG/gpu_soak.log
timeout 900 python tests/gpu_small_soak.py ${SMALL_S:-240} 9500 2>&1 | tail -2 | tee gpurun_out/$TAG/small_soak.log
