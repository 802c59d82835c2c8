#!/bin/bash
mkdir -p gpurun_out/dbg
python __graft_entry__.py > gpurun_out/dbg/build.log 2>&1 || { tail -20 gpurun_out/dbg/build.log; exit 1; }
for args in "--scale 1.0" "--scale 1.0 --stages" ; do
  echo "== $args"
  timeout 600 python -u -X faulthandler tools/e2e_bench.py $args 2>&1 | tail -40
done
