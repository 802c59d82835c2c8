#!/bin/bash
# round 3, third visit: A/B of the window pass that only walks lanes whose minimum can change (NTL_SKETCH_LANES), the map size
# classes, the hit-fraction feedback; then the end-to-end diagnostics.
TAG=${1:-r03c}
O=gpurun_out/$TAG; mkdir -p $O
python __graft_entry__.py > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $O/pytest_gpu.log
run() { # name, env..., -- args
  name=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 900 python bench.py --no-cpu-baseline --no-e2e --no-others "$@" > $O/bench_$name.json 2> $O/bench_$name.err
  python - $O/bench_$name.json $name <<'PY'
import json,sys
try:
    j=[json.loads(l) for l in open(sys.argv[1]) if l.startswith('{"metric')][-1]; c=j["config"]
    print(sys.argv[2], j["value"], "Gbases/s", j["ms_per_step"], "ms", c["stage_ms_per_step"], "SERIAL", c.get("serial_pass",{}).get("ms_per_step"), c.get("serial_pass",{}).get("stage_ms_per_step"))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
  tail -2 $O/bench_$name.err
}
run c3_lanes1 X=1 -- --steps 8 --warmup 1
run c3_lanes0 NTL_SKETCH_LANES=0 -- --steps 8 --warmup 1
run c5_lanes1 X=1 -- --workload C5 --steps 3 --warmup 1
run c5_lanes0 NTL_SKETCH_LANES=0 -- --workload C5 --steps 3 --warmup 1
run c2_lanes1 X=1 -- --workload C2 --steps 50 --warmup 3
run c2_lanes0 NTL_SKETCH_LANES=0 -- --workload C2 --steps 50 --warmup 3
timeout 1200 python tools/e2e_diag.py --bases 16e9 --forms > $O/e2e_diag.jsonl 2> $O/e2e_diag.err
cat $O/e2e_diag.jsonl | cut -c1-900; tail -3 $O/e2e_diag.err
ls $O
