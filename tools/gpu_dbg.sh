#!/bin/bash
mkdir -p gpurun_out/dbg
python __graft_entry__.py > gpurun_out/dbg/build.log 2>&1 || { tail -20 gpurun_out/dbg/build.log; exit 1; }
NTL_INDEX_BUCKET_BYTES=1 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "fixture or scenario or fuzz_map or full_size" 2>&1 | tail -2
P='import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["config"]["stage_ms_per_step"])'
for bb in 0 3 4 5 6; do
  for wl in C5 C3; do
    echo "bucket bits $bb $wl"
    if [ $bb = 0 ]; then export NTL_INDEX_BUCKET_BYTES=100000000000; unset NTL_INDEX_BUCKET_BITS; else export NTL_INDEX_BUCKET_BYTES=1; export NTL_INDEX_BUCKET_BITS=$bb; fi
    timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --workload $wl --scale 1.0 2>> gpurun_out/dbg/err.log | python -c "$P"
  done
done
