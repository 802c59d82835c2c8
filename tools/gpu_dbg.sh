#!/bin/bash
mkdir -p gpurun_out/dbg
python __graft_entry__.py > gpurun_out/dbg/build.log 2>&1 || { tail -20 gpurun_out/dbg/build.log; exit 1; }
timeout 900 python tools/bench_f3.py 2>&1 | tail -3 | tee gpurun_out/dbg/f3.json
