#!/bin/bash
mkdir -p gpurun_out/dbg
python __graft_entry__.py > gpurun_out/dbg/build.log 2>&1 || { tail -20 gpurun_out/dbg/build.log; exit 1; }
for args in "--scale 8.0" "--scale 8.0"; do
timeout 600 python -u tools/e2e_bench.py $args 2>&1 | grep -v Printing | tail -1 | cut -c1-700
done
