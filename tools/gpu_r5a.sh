#!/bin/bash
# Round 5, first visit: the strip-list path on hardware -- its tests, then C3 / C5 with lists against NTL_SKETCH_LISTS=0 (the bitmask).
# usage: tools/gpu_r5a.sh <tag>
set -x
TAG=${1:-r05a}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
[ -f "$R/bench.py" ] || { echo "no bench.py under $R"; exit 1; }
cd "$R"
O=gpurun_out/$TAG; mkdir -p $O
python __graft_entry__.py > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "strip_lists or variants or edge or fuzz_sketch or golden or full_size_assembly" 2>&1 | tail -8 | tee $O/pytest_lists.log
for W in C3 C5; do
  for L in 1 0; do
    NTL_SKETCH_LISTS=$L timeout 600 python bench.py --workload $W --steps 5 --warmup 2 --no-cpu-baseline --no-e2e --no-others > $O/bench_${W}_lists$L.json 2> $O/bench_${W}_lists$L.err
    python - <<PY
import json
d=json.load(open("$O/bench_${W}_lists$L.json"))
print("$W lists=$L", d["value"], d["ms_per_step"], json.dumps(d.get("serial_pass",{}))[:900])
PY
  done
done
ls $O
