#!/bin/bash
# prefetch of the strip ahead into L2 (sketch_thresh_kernel): parity, then C3 / C5 with and without (NTL_SKETCH_ABLATE=32)
TAG=${1:-r03aa}
O=gpurun_out/$TAG; mkdir -p $O
python __graft_entry__.py > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sketch or threshold or variants or window or full_size" 2>&1 | tail -4 | tee $O/pytest_sel.log
run() { name=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 900 python bench.py --no-cpu-baseline --no-e2e --no-others "$@" > $O/bench_$name.json 2> $O/bench_$name.err
  python - $O/bench_$name.json <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{"metric'):
        j=json.loads(l); sp=j["config"].get("serial_pass",{}); print(sys.argv[1], j["value"], j["ms_per_step"], "serial", sp.get("ms_per_step"), "window ms/launch", j["roofline"]["avg_launch_ms"], "mask", sp["stage_ms_per_step"]["sketch_mask"], "emit", sp["stage_ms_per_step"]["sketch_emit"], "spans", j["config"]["stage_ms_per_step"]["sketch_mask"], j["config"]["stage_ms_per_step"]["sketch_emit"], j["config"]["read_minimizers_per_step"])
PY
tail -1 $O/bench_$name.err
}
run c3_pf X=1 -- --steps 6 --warmup 1
run c3_nopf NTL_SKETCH_ABLATE=32 -- --steps 6 --warmup 1
run c5_pf X=1 -- --workload C5 --steps 3 --warmup 1
run c5_nopf NTL_SKETCH_ABLATE=32 -- --workload C5 --steps 3 --warmup 1
run c2_pf X=1 -- --workload C2 --steps 40 --warmup 3
