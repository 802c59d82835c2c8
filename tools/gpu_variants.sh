#!/bin/bash
# bench.py per workload for the default library and for every prebuilt variant (tools/build_variant.py) named in VARIANTS, in one visit.
# usage: [TESTS="-k expr"] VARIANTS="a b" tools/gpu_variants.sh <tag> [workloads, default C3]     (STEPS, default 6)
TAG=${1:-r06v}; shift
WL=${*:-C3}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
O=gpurun_out/$TAG; mkdir -p $O
python __graft_entry__.py > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
for V in default $VARIANTS; do
  unset NTLINK_AMD_LIB
  if [ $V != default ]; then
    export NTLINK_AMD_LIB=$R/ntlink_amd/build/var_$V/libntlink_hip.so
    [ -f "$NTLINK_AMD_LIB" ] || { echo "no $NTLINK_AMD_LIB"; continue; }
  fi
  if [ -n "$TESTS" ]; then timeout 900 bash -c "python -m pytest tests -m gpu -x -q $TESTS" 2>&1 | tail -3; fi
  for W in $WL; do
    timeout 600 python bench.py --workload $W --steps ${STEPS:-6} --warmup 2 --no-cpu-baseline --no-e2e --no-others > $O/bench_${W}_$V.json 2> $O/bench_${W}_$V.err
    python - <<PY
import json
try:
    d=json.load(open("$O/bench_${W}_$V.json")); c=d["config"]; r=d["roofline"]
    print("$V $W", d["value"], "Gbases/s", d["ms_per_step"], "ms/step; window in pipe", r["avg_launch_ms"], "ms; alone", r["kernels_alone"]["avg_launch_ms"])
    print("  in pipe", c["stage_ms_per_step"]); print("  alone  ", c["serial_pass"]["ms_per_step"], c["serial_pass"]["stage_ms_per_step"])
except Exception as e:
    print("$V $W failed:", e); print(open("$O/bench_${W}_$V.err").read()[-1500:])
PY
  done
done
