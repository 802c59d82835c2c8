#!/bin/bash
# round 3 soaks: the mapping half at volume (HiFi and ONT), the sketch at volume on the round-3 runtime, random configurations
TAG=${1:-r03g}
O=gpurun_out/$TAG; mkdir -p $O
python __graft_entry__.py > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout 1500 python tests/gpu_map_soak.py C5 12 1e9 300 > $O/map_soak_c5.log 2>&1; tail -2 $O/map_soak_c5.log
timeout 1500 python tests/gpu_map_soak.py C3 6 2e9 400 > $O/map_soak_c3.log 2>&1; tail -2 $O/map_soak_c3.log
NTL_PIPELINE=0 timeout 900 python tests/gpu_map_soak.py C5 3 1e9 500 > $O/map_soak_c5_serial.log 2>&1; tail -1 $O/map_soak_c5_serial.log
timeout 1500 python tests/gpu_volume_soak.py C3 10 2e9 700 > $O/volume_soak_c3.log 2>&1; tail -1 $O/volume_soak_c3.log
timeout 1500 python tests/gpu_volume_soak.py C5 6 2e9 800 > $O/volume_soak_c5.log 2>&1; tail -1 $O/volume_soak_c5.log
NTL_SKETCH_LANES=1 timeout 1500 python tests/gpu_volume_soak.py C3 5 2e9 900 > $O/volume_soak_c3_lanes.log 2>&1; tail -1 $O/volume_soak_c3_lanes.log
timeout 900 python tests/gpu_soak.py 240 > $O/soak_fuzz.log 2>&1; tail -2 $O/soak_fuzz.log
