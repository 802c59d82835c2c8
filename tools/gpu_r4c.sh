#!/bin/bash
# round 4: the C3 bench line for window-kernel shapes (NTL_SKETCH_WAVE) x grid sizes (NTL_SKW_WGS_PER_CU)
TAG=${1:-r04h}
mkdir -p gpurun_out/$TAG
CFGS=${CFGS:-8:1 4:2 4:3 4:4 4:5 4:6 8:2}
for cfg in $CFGS; do
v=${cfg%%:*}; wg=${cfg##*:}
NTL_SKETCH_WAVE=$v NTL_SKW_WGS_PER_CU=$wg python bench.py --no-e2e --no-cpu-baseline --no-others --steps 6 --serial-steps 1 > gpurun_out/$TAG/bench_c3_$v-$wg.json 2> gpurun_out/$TAG/bench_c3_$v-$wg.err
python - <<PY
import json
d=json.load(open("gpurun_out/$TAG/bench_c3_$v-$wg.json"))
print("wave", $v, "wgs/cu", $wg, d["value"], d["ms_per_step"], " serial", d["config"]["serial_pass"]["ms_per_step"], "window alone", d["config"]["serial_pass"]["stage_ms_per_step"]["sketch_mask"], " pipelined", d["config"]["stage_ms_per_step"])
PY
done
