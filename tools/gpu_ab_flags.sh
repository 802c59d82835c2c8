#!/bin/bash
# A/B of two builds of the kernels in one visit: the default flags, then NTL_EXTRA_HIPCC_FLAGS="$1" (ntlink_amd/build.py rebuilds when the
# flags change); bench.py per workload for each.  usage: tools/gpu_ab_flags.sh <tag> "<extra hipcc flags>" [workloads, default C3]
set -x
TAG=${1:-r05ab}; FLAGS=$2; shift; shift
WL=${*:-C3}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
[ -f "$R/bench.py" ] || { echo "no bench.py under $R"; exit 1; }
cd "$R"
O=gpurun_out/$TAG; mkdir -p $O
for V in default flags; do
  if [ $V = flags ]; then export NTL_EXTRA_HIPCC_FLAGS="$FLAGS"; fi
  python __graft_entry__.py > $O/build_$V.log 2>&1 || { tail -20 $O/build_$V.log; exit 1; }
  timeout 600 python -m pytest tests -m gpu -x -q -k strip_lists 2>&1 | tail -2
  for W in $WL; do
    timeout 600 python bench.py --workload $W --steps 8 --warmup 2 --no-cpu-baseline --no-e2e --no-others > $O/bench_${W}_$V.json 2> $O/bench_${W}_$V.err
    python - <<PY
import json
d=json.load(open("$O/bench_${W}_$V.json")); c=d["config"]; r=d["roofline"]
print("$V $W", d["value"], "Gbases/s", d["ms_per_step"], "ms/step; window in pipe", r["avg_launch_ms"], "ms; alone", r["kernels_alone"]["avg_launch_ms"])
print("  in pipe", c["stage_ms_per_step"]); print("  alone  ", c["serial_pass"]["ms_per_step"], c["serial_pass"]["stage_ms_per_step"])
PY
  done
done
