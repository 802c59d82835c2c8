#!/bin/bash
# One GPU-box visit (round 2 form): build, all gpu tests, smoke, the default bench line (C3, full 90 Gbases per step),
# a single-rank torch.distributed.run line, rocprofv3 kernel trace + three PMC passes of the SAME command, each in its own run.
# usage: tools/gpu_round2.sh <tag> [extra bench args for the profiled runs]
set -x
TAG=${1:-r02b}; shift
mkdir -p gpurun_out/$TAG
python __graft_entry__.py > gpurun_out/$TAG/build.log 2>&1 || { tail -20 gpurun_out/$TAG/build.log; exit 1; }
if [ -z "$SKIP_TESTS" ]; then
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/$TAG/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2 | tee gpurun_out/$TAG/smoke.log
fi
timeout 1200 python bench.py $BENCH_ARGS > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err
cat gpurun_out/$TAG/bench.json; tail -5 gpurun_out/$TAG/bench.err
if [ -z "$SKIP_TORCHRUN" ]; then
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --no-e2e > gpurun_out/$TAG/bench_torchrun1.json 2> gpurun_out/$TAG/bench_torchrun1.err; tail -c 400 gpurun_out/$TAG/bench_torchrun1.json; tail -3 gpurun_out/$TAG/bench_torchrun1.err
fi
if [ -z "$SKIP_PROF" ]; then
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-e2e $*"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/trace -o kt -- $B > $R/gpurun_out/$TAG/bench_trace.json 2> $R/gpurun_out/$TAG/trace.err
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/$TAG/pmc_fetch -o p -- $B > /dev/null 2> $R/gpurun_out/$TAG/pmc_fetch.err
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/$TAG/pmc_write -o p -- $B > /dev/null 2> $R/gpurun_out/$TAG/pmc_write.err
timeout 900 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/$TAG/pmc_sq -o p -- $B > $R/gpurun_out/$TAG/bench_pmc_sq.json 2> $R/gpurun_out/$TAG/pmc_sq.err
cd $R
# the raw per-dispatch traces are large: keep the stats and the counter tables
find gpurun_out/$TAG -name '*kernel_trace.csv' -size +8M -delete
head -12 gpurun_out/$TAG/trace/kt_kernel_stats.csv
fi
ls -la gpurun_out/$TAG/
