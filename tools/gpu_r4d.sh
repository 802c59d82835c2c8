#!/bin/bash
# round 4: map/lookup parity subset + the C3 and C5 bench lines (pipelined + kernels alone)
TAG=${1:-r04t}
mkdir -p gpurun_out/$TAG
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "probe or tags or scenario or fixture or full or baseline or fuzz_mapping or text" 2>&1 | tail -3
for W in ${WLS:-C3 C5}; do
python bench.py --no-e2e --no-cpu-baseline --no-others --workload $W --steps ${STEPS:-4} --serial-steps 1 > gpurun_out/$TAG/bench_$W.json 2> gpurun_out/$TAG/bench_$W.err
python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/$TAG/bench_$W.json") if l.startswith("{")][-1])
print("$W", d["value"], d["ms_per_step"], " serial", d["config"]["serial_pass"]["ms_per_step"], d["config"]["serial_pass"]["stage_ms_per_step"], " pipelined", d["config"]["stage_ms_per_step"])
PY
done
