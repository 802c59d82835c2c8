#!/bin/bash
# bench lines of the other two workloads (C5: HiFi 180 Gbases k24 w100 --sensitive; C2: 0.5 Gbases on 50 Mbp) for DESIGN section 5
TAG=${1:-r02bh}
mkdir -p gpurun_out/$TAG
python __graft_entry__.py > gpurun_out/$TAG/build.log 2>&1 || { tail -20 gpurun_out/$TAG/build.log; exit 1; }
timeout 1500 python bench.py --workload C5 --steps 3 --warmup 1 --no-cpu-baseline --no-e2e > gpurun_out/$TAG/bench_C5.json 2> gpurun_out/$TAG/bench_C5.err
timeout 900 python bench.py --workload C2 --steps 50 --warmup 5 --no-cpu-baseline --no-e2e > gpurun_out/$TAG/bench_C2.json 2> gpurun_out/$TAG/bench_C2.err
for w in C5 C2; do python - <<PY
import json
d = json.loads(open("gpurun_out/$TAG/bench_$w.json").read().strip().splitlines()[-1])
print("$w", d["value"], d["ms_per_step"], d["config"]["stage_ms_per_step"], d["roofline"].get("kernel_Gbases_per_s"))
PY
tail -2 gpurun_out/$TAG/bench_$w.err
done
