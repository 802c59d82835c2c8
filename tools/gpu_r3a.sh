#!/bin/bash
# round 3, first visit: the asynchronous two-stream runtime.  gpu tests, then the C3 / C5 bench lines with the pipeline on
# (default), off, and with the other stream-priority settings.
TAG=${1:-r03a}
O=gpurun_out/$TAG; mkdir -p $O
python __graft_entry__.py > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $O/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2 | tee $O/smoke.log
run() { # name, env..., -- args
  name=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 900 python bench.py --no-cpu-baseline --no-e2e "$@" > $O/bench_$name.json 2> $O/bench_$name.err
  python - $O/bench_$name.json $name <<'PY'
import json,sys
try:
    j=json.load(open(sys.argv[1])); c=j["config"]
    print(sys.argv[2], j["value"], "Gbases/s", j["ms_per_step"], "ms", c["stage_ms_per_step"])
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
  tail -3 $O/bench_$name.err
}
run c3_pipe X=1 -- --steps 8 --warmup 1
run c3_serial NTL_PIPELINE=0 -- --steps 8 --warmup 1
run c3_prio0 NTL_PIPELINE_PRIO=0 -- --steps 8 --warmup 1
run c3_prio2 NTL_PIPELINE_PRIO=2 -- --steps 8 --warmup 1
run c5_pipe X=1 -- --workload C5 --steps 3 --warmup 1
run c5_serial NTL_PIPELINE=0 -- --workload C5 --steps 3 --warmup 1
run c2_pipe X=1 -- --workload C2 --steps 50 --warmup 3
run c2_serial NTL_PIPELINE=0 -- --workload C2 --steps 50 --warmup 3
ls -la $O
