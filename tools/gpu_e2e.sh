#!/bin/bash
mkdir -p gpurun_out/e2e
python __graft_entry__.py > gpurun_out/e2e/build.log 2>&1 || { tail -20 gpurun_out/e2e/build.log; exit 1; }
timeout 1500 python -m pytest tests/test_gpu_cli.py -x -q 2>&1 | tail -2
rm -f gpurun_out/e2e/e2e.jsonl
for args in "--scale 1.0 --stages" "--scale 1.0" "--scale 8.0" "--scale 8.0 --batch 64000000" "--scale 0.2 --gz" "--scale 2.0 --gz --files 32" "--scale 1.0 --pipe" "--scale 8.0 --pipe"; do
  timeout 1200 python tools/e2e_bench.py $args 2>> gpurun_out/e2e/err.log | tee -a gpurun_out/e2e/e2e.jsonl
done
timeout 600 python bench.py --steps 5 --warmup 2 2>> gpurun_out/e2e/err.log | tee gpurun_out/e2e/bench.json | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["config"]["stage_ms_per_step"])'
