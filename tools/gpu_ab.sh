#!/bin/bash
# A/B of tuning knobs on one box: bench lines only
mkdir -p gpurun_out/ab
python __graft_entry__.py > gpurun_out/ab/build.log 2>&1 || { tail -20 gpurun_out/ab/build.log; exit 1; }
P='import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["config"]["stage_ms_per_step"])'
for v in 128 256 128 256; do
  echo "NTL_SKETCH_NT=$v"; NTL_SKETCH_NT=$v timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>> gpurun_out/ab/err.log | python -c "$P"
done
echo "C3-like w=250 (scale 0.2)"; for v in 256 128; do NTL_SKETCH_NT=$v timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --workload C3 --scale 0.2 2>> gpurun_out/ab/err.log | python -c "$P"; done
