#!/bin/bash
mkdir -p gpurun_out/c3
python __graft_entry__.py > gpurun_out/c3/build.log 2>&1 || { tail -20 gpurun_out/c3/build.log; exit 1; }
P='import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["config"]["stage_ms_per_step"], d["config"]["workload"], d["config"]["hit_fraction"], d["config"]["gen_s"], d["config"]["upload_s"], d.get("cpu_baseline"))'
free -g | head -2; nproc
for sc in ${SCALES:-0.25}; do
  timeout 2400 python bench.py --steps 3 --warmup 1 --workload ${WL:-C3} --scale $sc 2> gpurun_out/c3/err_$sc.log | tee gpurun_out/c3/bench_${WL:-C3}_$sc.json | python -c "$P"
  tail -2 gpurun_out/c3/err_$sc.log
done
