#!/bin/bash
# round 3, last visit: soaks on the final kernels + the default bench line (traffic.json of r03ab applies)
TAG=${1:-r03ac}
O=gpurun_out/$TAG; mkdir -p $O
python __graft_entry__.py > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout 1500 python tests/gpu_volume_soak.py C3 10 2e9 700 > $O/volume_soak_c3.log 2>&1; tail -1 $O/volume_soak_c3.log
timeout 1500 python tests/gpu_volume_soak.py C5 6 2e9 800 > $O/volume_soak_c5.log 2>&1; tail -1 $O/volume_soak_c5.log
timeout 1500 python tests/gpu_map_soak.py C5 6 1e9 300 > $O/map_soak_c5.log 2>&1; tail -1 $O/map_soak_c5.log
timeout 900 python tests/gpu_soak.py 300 > $O/soak_fuzz.log 2>&1; tail -1 $O/soak_fuzz.log
timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err
tail -3 $O/bench.err
python - $O/bench.json <<'PY'
import json,sys
j=[json.loads(l) for l in open(sys.argv[1]) if l.startswith('{"metric')][-1]
print("C3", j["value"], j["ms_per_step"], "serial", j["config"]["serial_pass"]["ms_per_step"], "traffic", j["roofline"]["traffic"], "valu", (j["roofline"]["valu"] or {}))
for k,v in j["other_workloads"].items(): print(k, v.get("value"), v.get("ms_per_step"), v.get("error"))
c=j["cpu_baseline"]; print("cpu", c["value"], c["cores"])
e=j["end_to_end"]; print("e2e", e["value"], e["seconds"], e["runs_s"], e["compressed_inputs"])
PY
