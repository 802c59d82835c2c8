#!/bin/bash
mkdir -p gpurun_out/dbg
python __graft_entry__.py > gpurun_out/dbg/build.log 2>&1 || { tail -20 gpurun_out/dbg/build.log; exit 1; }
timeout 900 python -m pytest tests/test_gpu_cli.py -x -q 2>&1 | tail -2
for args in "--scale 1.0 --pipe" "--scale 8.0 --pipe" "--scale 8.0"; do
timeout 600 python -u tools/e2e_bench.py $args 2>&1 | grep -v Printing | tail -2 | cut -c1-300
done
