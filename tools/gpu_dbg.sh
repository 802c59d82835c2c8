#!/bin/bash
mkdir -p gpurun_out/dbg
python __graft_entry__.py > gpurun_out/dbg/build.log 2>&1 || { tail -20 gpurun_out/dbg/build.log; exit 1; }
df -h /tmp | tail -1
timeout 1500 python -u tools/e2e_bench.py --workload C3 --scale 1.0 --read-bases 8000000000 2>&1 | grep -v Printing | tail -2 | cut -c1-900 | tee gpurun_out/dbg/e2e_c3.json
