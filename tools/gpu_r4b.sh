#!/bin/bash
# round 4: sketch stage alone + parity subset + the C3 bench line (pipelined and kernels alone) for window-kernel grid sizes
TAG=${1:-r04g}
mkdir -p gpurun_out/$TAG
python tools/sketch_bench.py | tee gpurun_out/$TAG/sk_c3.json
python tools/sketch_bench.py --w 100 --k 24 --read-len 20000 --bases 3.9e9 | tee gpurun_out/$TAG/sk_c5.json
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sketch or fast or fuzz or edge or golden or thresh" 2>&1 | tail -2
for wg in ${WGS:-4 3 2}; do
NTL_SKW_WGS_PER_CU=$wg python bench.py --no-e2e --no-cpu-baseline --no-others --steps 8 > gpurun_out/$TAG/bench_c3_wg$wg.json 2> gpurun_out/$TAG/bench_c3_wg$wg.err
tail -2 gpurun_out/$TAG/bench_c3_wg$wg.err
python - <<PY
import json
d=json.load(open("gpurun_out/$TAG/bench_c3_wg$wg.json"))
print("wgs/cu", $wg, d["value"], d["ms_per_step"])
print(" pipelined", d["config"]["stage_ms_per_step"])
print(" serial", d["config"]["serial_pass"]["ms_per_step"], d["config"]["serial_pass"]["stage_ms_per_step"])
PY
done
