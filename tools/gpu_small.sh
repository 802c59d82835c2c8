#!/bin/bash
# Round 6: the small-window pass on a GPU box -- its parity tests, the bench of both forms (tools/small_window_bench.py), optionally the soak.
# usage: [SOAK_S=240] tools/gpu_small.sh <tag> [k,w ...]
TAG=${1:-r07s}; shift
cd ${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=gpurun_out/$TAG; mkdir -p $O
python __graft_entry__.py > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout 1200 python -m pytest tests -m gpu -x -q -k "small_window or edge_cases or fuzz_sketch or tiny or overlap or variants_on_the_gpu or second_emit or cap_guess" 2>&1 | tail -3 | tee $O/pytest_small.log
timeout 600 python tools/small_window_bench.py "$@" > $O/small_window_bench.jsonl 2> $O/small_window_bench.err
python - <<PY
import json
for l in open("$O/small_window_bench.jsonl"):
    d = json.loads(l); print(d["k"], d["w"], "small" if d["sketch_small_kernel"] else "round-1", d["Gbases_per_s"], d["window_pass_Gbases_per_s"], d["hbm_frac"], d["stage_ms"])
PY
if [ -n "$SOAK_S" ]; then timeout 1200 python tests/gpu_small_soak.py $SOAK_S 9300 2>&1 | tail -3 | tee $O/small_soak.log; fi
