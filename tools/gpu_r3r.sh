#!/bin/bash
# round 3: sketch_thresh_kernel with and without the staged keys (DIRECT), and on the w = 100 workloads
TAG=${1:-r03r}
O=gpurun_out/$TAG; mkdir -p $O
python __graft_entry__.py > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "full_size or threshold or variants" 2>&1 | tail -4 | tee $O/pytest_sel.log
run() { name=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 900 python bench.py --no-cpu-baseline --no-e2e --no-others "$@" > $O/bench_$name.json 2> $O/bench_$name.err
  python - $O/bench_$name.json <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{"metric'):
        j=json.loads(l); sp=j["config"].get("serial_pass",{}); print(sys.argv[1], j["value"], j["ms_per_step"], "serial", sp.get("ms_per_step"), "window ms/launch", j["roofline"]["avg_launch_ms"], "mask", sp["stage_ms_per_step"]["sketch_mask"], "redo", sp["stage_ms_per_step"]["sketch_redo"], j["config"]["read_minimizers_per_step"])
PY
}
run c3_d0 NTL_SKETCH_THRESH_DIRECT=0 -- --steps 6 --warmup 1
run c3_d1 NTL_SKETCH_THRESH_DIRECT=1 -- --steps 6 --warmup 1
run c5_d0 NTL_SKETCH_THRESH_DIRECT=0 -- --workload C5 --steps 3 --warmup 1
run c5_d2 NTL_SKETCH_THRESH_DIRECT=2 -- --workload C5 --steps 3 --warmup 1
run c5_d2_t8 NTL_SKETCH_THRESH_DIRECT=2 NTL_SKETCH_THRESH=8 -- --workload C5 --steps 3 --warmup 1
run c2_d0 NTL_SKETCH_THRESH_DIRECT=0 -- --workload C2 --steps 40 --warmup 3
run c2_d2 NTL_SKETCH_THRESH_DIRECT=2 -- --workload C2 --steps 40 --warmup 3
