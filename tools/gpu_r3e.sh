#!/bin/bash
TAG=${1:-r03e}
O=gpurun_out/$TAG; mkdir -p $O
python __graft_entry__.py > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
run() { name=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 900 python bench.py --no-cpu-baseline --no-e2e --no-others "$@" > $O/bench_$name.json 2> $O/bench_$name.err
  python - $O/bench_$name.json $name <<'PY'
import json,sys
try:
    j=[json.loads(l) for l in open(sys.argv[1]) if l.startswith('{"metric')][-1]; c=j["config"]
    print(sys.argv[2], j["value"], "Gbases/s", j["ms_per_step"], "ms", c["stage_ms_per_step"], "SERIAL", c.get("serial_pass",{}).get("ms_per_step"), c.get("serial_pass",{}).get("stage_ms_per_step"))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
  tail -2 $O/bench_$name.err | cut -c1-300; }
run c5 NTL_POOL_TRACE=1 -- --workload C5 --steps 3 --warmup 1
echo "mallocs $(grep -c hipMalloc $O/bench_c5.err) frees $(grep -c hipFree $O/bench_c5.err)"
run c2 X=1 -- --workload C2 --steps 50 --warmup 3
run c3 X=1 -- --steps 6 --warmup 1
timeout 1200 python tools/e2e_diag.py --bases 16e9 > $O/e2e_diag.jsonl 2> $O/e2e_diag.err
python - $O/e2e_diag.jsonl <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if not l.startswith('{"Gbases'): continue
    j=json.loads(l); print(j["Gbases_per_s"], j["seconds"], j["env"], j["batch_bases"], "contigs", j["t_contigs"], "ingest-wait", j["t_ingest"], "device", j["t_device"], "handover", j["t_handover"], "write", j["t_write"], "reader", j["reader"])
PY
tail -3 $O/e2e_diag.err
