#!/bin/bash
# Quick GPU visit: sketch/map parity subset + bench line.
set -x
mkdir -p gpurun_out/quick
python __graft_entry__.py > gpurun_out/quick/build.log 2>&1 || { tail -20 gpurun_out/quick/build.log; exit 1; }
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -8 | tee gpurun_out/quick/pytest.log
timeout 600 python bench.py --steps 5 --warmup 2 ${BENCH_ARGS} 2> gpurun_out/quick/bench.err | tee gpurun_out/quick/bench.json | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['stage_ms_per_step'], d['roofline']['avg_launch_ms'], d.get('cpu_baseline',{}).get('value'))"
