#!/bin/bash
# build + chosen gpu tests + one bench line without the profiler passes
TAG=${1:-r02q}; shift
mkdir -p gpurun_out/$TAG
python __graft_entry__.py > gpurun_out/$TAG/build.log 2>&1 || { tail -20 gpurun_out/$TAG/build.log; exit 1; }
timeout 1500 python -m pytest tests -m gpu -x -q $PYTEST_ARGS 2>&1 | tail -15 | tee gpurun_out/$TAG/pytest_gpu.log
timeout 900 python bench.py $* > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err
cat gpurun_out/$TAG/bench.json; tail -5 gpurun_out/$TAG/bench.err
