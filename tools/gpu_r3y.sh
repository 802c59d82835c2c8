#!/bin/bash
# does the start of a strip (dependent global loads) explain the window kernel's slowdown beside the emit kernel?
# ablation build (results WRONG): NTL_SKETCH_ABLATE 8 = no base-word load and no table lookups, 16 = no table lookups
TAG=${1:-r03y}
O=gpurun_out/$TAG; mkdir -p $O
NTL_EXTRA_HIPCC_FLAGS=-DNTL_SKETCH_ABLATION python __graft_entry__.py > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
for ab in 0 16 8; do
NTL_SKETCH_ABLATE=$ab timeout 900 python bench.py --no-cpu-baseline --no-e2e --no-others --steps 4 --warmup 1 > $O/bench_ab$ab.json 2> $O/bench_ab$ab.err
python - $O/bench_ab$ab.json <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{"metric'):
        j=json.loads(l); sp=j["config"].get("serial_pass",{}); print(sys.argv[1], j["value"], j["ms_per_step"], "serial", sp.get("ms_per_step"), "window alone", sp["stage_ms_per_step"]["sketch_mask"], "emit alone", sp["stage_ms_per_step"]["sketch_emit"], "pipelined spans", j["config"]["stage_ms_per_step"]["sketch_mask"], j["config"]["stage_ms_per_step"]["sketch_emit"], j["config"]["read_minimizers_per_step"])
PY
tail -1 $O/bench_ab$ab.err
done
