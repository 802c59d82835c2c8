#!/bin/bash
# tests + the default bench line, no profiling passes
TAG=${1:-r03k}
O=gpurun_out/$TAG; mkdir -p $O
python __graft_entry__.py > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee $O/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2 | tee $O/smoke.log
timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err
tail -3 $O/bench.err
python - $O/bench.json <<'PY'
import json,sys
j=[json.loads(l) for l in open(sys.argv[1]) if l.startswith('{"metric')][-1]
print("C3", j["value"], j["ms_per_step"], "serial", j["config"]["serial_pass"]["ms_per_step"], "traffic", j["roofline"]["traffic"], "valu", (j["roofline"]["valu"] or {}).get("frac"))
for k,v in j["other_workloads"].items(): print(k, v.get("value"), v.get("ms_per_step"), v.get("error"))
c=j["cpu_baseline"]; print("cpu", c["value"], c["cores"], c["cpus_visible"], c["cpu_quota_cores"], c["threads_used"], c["read_sketch_scaling_Mbases_per_s"], c["if_it_scaled"]["value"], c["reference_faithful"]["value"])
e=j["end_to_end"]; print("e2e", e["value"], e["seconds"], e["first_run_cold"], e["compressed_inputs"], e["reader"], e["t_contig_stage"], e["t_wait_for_ingest"], e["t_handover"], e["t_write"], e["host_cpu"])
PY
