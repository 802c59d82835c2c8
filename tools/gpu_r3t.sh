#!/bin/bash
# round 3: the window stream on a subset of every XCD's CUs (NTL_WSTREAM_CU_OFF), C3 / C5 pipelined step
TAG=${1:-r03t}
O=gpurun_out/$TAG; mkdir -p $O
python __graft_entry__.py > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
run() { name=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 900 python bench.py --no-cpu-baseline --no-e2e --no-others --serial-steps 0 "$@" > $O/bench_$name.json 2> $O/bench_$name.err
  python - $O/bench_$name.json <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{"metric'):
        j=json.loads(l); print(sys.argv[1], j["value"], j["ms_per_step"], "window ms/launch (pipelined)", j["roofline"]["in_timed_region"]["avg_launch_ms"], j["config"]["stage_ms_per_step"], j["config"]["read_minimizers_per_step"])
PY
tail -2 $O/bench_$name.err
}
for d in 0 2 4 6 8 12; do run c3_off$d NTL_WSTREAM_CU_OFF=$d -- --steps 6 --warmup 1; done
for d in 0 4 8 12; do run c5_off$d NTL_WSTREAM_CU_OFF=$d -- --workload C5 --steps 3 --warmup 1; done
