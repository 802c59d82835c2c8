#!/bin/bash
# round 3: threshold pass with the block-minima fallback: gpu tests, candidates-per-window sweep on C3 / C5 / C2
TAG=${1:-r03s}
O=gpurun_out/$TAG; mkdir -p $O
python __graft_entry__.py > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee $O/pytest_gpu.log
run() { name=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 900 python bench.py --no-cpu-baseline --no-e2e --no-others "$@" > $O/bench_$name.json 2> $O/bench_$name.err
  python - $O/bench_$name.json <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{"metric'):
        j=json.loads(l); sp=j["config"].get("serial_pass",{}); print(sys.argv[1], j["value"], j["ms_per_step"], "serial", sp.get("ms_per_step"), "window ms/launch", j["roofline"]["avg_launch_ms"], "mask", sp["stage_ms_per_step"]["sketch_mask"], "redo", sp["stage_ms_per_step"]["sketch_redo"], j["config"]["read_minimizers_per_step"])
PY
}
for c in 10 9 8 7; do run c3_t$c NTL_SKETCH_THRESH=$c -- --steps 6 --warmup 1; done
for c in 0 10 9 8 7; do run c5_t$c NTL_SKETCH_THRESH=$c -- --workload C5 --steps 3 --warmup 1; done
for c in 0 10 8; do run c2_t$c NTL_SKETCH_THRESH=$c -- --workload C2 --steps 40 --warmup 3; done
