#!/bin/bash
# round 3: emit kernel on a a resident-size grid walking the tiles (NTL_EMIT_GRID) beside the window kernel, C3 / C5
TAG=${1:-r03u}
O=gpurun_out/$TAG; mkdir -p $O
python __graft_entry__.py > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
run() { name=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 900 python bench.py --no-cpu-baseline --no-e2e --no-others "$@" > $O/bench_$name.json 2> $O/bench_$name.err
  python - $O/bench_$name.json <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{"metric'):
        j=json.loads(l); sp=j["config"].get("serial_pass",{}); print(sys.argv[1], j["value"], j["ms_per_step"], "serial", sp.get("ms_per_step"), "emit alone", sp["stage_ms_per_step"]["sketch_emit"], "pipelined spans", j["config"]["stage_ms_per_step"]["sketch_mask"], j["config"]["stage_ms_per_step"]["sketch_emit"])
PY
tail -2 $O/bench_$name.err
}
for g in 0 4096 2048 1536 1024 512; do run c3_g$g NTL_EMIT_GRID=$g -- --steps 6 --warmup 1; done
for g in 0 4096 2048 1024; do run c5_g$g NTL_EMIT_GRID=$g -- --workload C5 --steps 3 --warmup 1; done
