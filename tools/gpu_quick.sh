#!/bin/bash
# A short visit: build, optionally a test selection (TESTS="-k expr" or TESTS=all), then bench.py per workload without the CPU baseline,
# the end-to-end leg and the other workloads.  usage: [TESTS=..] [ENVS="A=1 B=2"] tools/gpu_quick.sh <tag> [workloads, default "C3"]
set -x
TAG=${1:-r05q}; shift
WL=${*:-C3}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
[ -f "$R/bench.py" ] || { echo "no bench.py under $R"; exit 1; }
cd "$R"
O=gpurun_out/$TAG; mkdir -p $O
python __graft_entry__.py > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
if [ "$TESTS" = all ]; then timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee $O/pytest_gpu.log
elif [ -n "$TESTS" ]; then timeout 1200 bash -c "python -m pytest tests -m gpu -x -q $TESTS" 2>&1 | tail -5 | tee $O/pytest_gpu.log; fi
for W in $WL; do
  env $ENVS timeout 600 python bench.py --workload $W --steps 6 --warmup 2 --no-cpu-baseline --no-e2e --no-others > $O/bench_$W.json 2> $O/bench_$W.err
  python - <<PY
import json
d=json.load(open("$O/bench_$W.json")); c=d["config"]; r=d["roofline"]
print("$W", d["value"], "Gbases/s", d["ms_per_step"], "ms/step; window in pipe", r["avg_launch_ms"], "ms; alone", r["kernels_alone"]["avg_launch_ms"])
print("  in pipe", c["stage_ms_per_step"]); print("  alone  ", c["serial_pass"]["ms_per_step"], c["serial_pass"]["stage_ms_per_step"])
PY
done
