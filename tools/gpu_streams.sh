#!/bin/bash
# the C3 bench line with one and with two device contexts taking the sub-batches in turn (NTL_BENCH_STREAMS)
TAG=${1:-r02bi}
mkdir -p gpurun_out/$TAG
python __graft_entry__.py > gpurun_out/$TAG/build.log 2>&1 || { tail -20 gpurun_out/$TAG/build.log; exit 1; }
for ns in 2 1 2; do
NTL_BENCH_STREAMS=$ns timeout 900 python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-e2e > gpurun_out/$TAG/bench_s$ns.json 2> gpurun_out/$TAG/bench_s$ns.err
python - <<PY
import json
d = json.loads(open("gpurun_out/$TAG/bench_s$ns.json").read().strip().splitlines()[-1])
print("streams $ns", d["value"], d["ms_per_step"], d["config"]["stage_ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["kernel_Gbases_per_s"])
PY
done
