#!/bin/bash
# GPU-box visit: all gpu tests, smoke, bench line, kernel trace, PMC passes (each in its own run).
set -x
TAG=${1:-r01b}
mkdir -p gpurun_out/$TAG
python __graft_entry__.py > gpurun_out/$TAG/build.log 2>&1 || { tail -20 gpurun_out/$TAG/build.log; exit 1; }
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/$TAG/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2 | tee gpurun_out/$TAG/smoke.log
timeout 900 python bench.py --steps 5 --warmup 2 > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err
cat gpurun_out/$TAG/bench.json
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/$TAG/bench_torchrun1.json 2> gpurun_out/$TAG/bench_torchrun1.err; tail -c 600 gpurun_out/$TAG/bench_torchrun1.json; tail -3 gpurun_out/$TAG/bench_torchrun1.err
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/trace -o kt -- $B > $R/gpurun_out/$TAG/bench_trace.json 2> $R/gpurun_out/$TAG/trace.err
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/$TAG/pmc_fetch -o p -- $B > /dev/null 2> $R/gpurun_out/$TAG/pmc_fetch.err
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/$TAG/pmc_write -o p -- $B > /dev/null 2> $R/gpurun_out/$TAG/pmc_write.err
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/$TAG/pmc_sq -o p -- $B > /dev/null 2> $R/gpurun_out/$TAG/pmc_sq.err
cd $R
cat gpurun_out/$TAG/trace/kt_kernel_stats.csv | head -12
ls -la gpurun_out/$TAG/*
