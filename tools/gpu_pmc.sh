#!/bin/bash
# bench line + one SQ PMC pass (LDS / VALU / wait counters) for the sketch kernel
set -x
TAG=${1:-pmc}
mkdir -p gpurun_out/$TAG
python __graft_entry__.py > gpurun_out/$TAG/build.log 2>&1 || { tail -20 gpurun_out/$TAG/build.log; exit 1; }
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2> gpurun_out/$TAG/bench.err | tee gpurun_out/$TAG/bench.json | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['stage_ms_per_step'])"
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY --output-format csv -d $R/gpurun_out/$TAG/pmc_sq2 -o p -- $B > /dev/null 2> $R/gpurun_out/$TAG/pmc_sq2.err
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/$TAG/pmc_sq3 -o p -- $B > /dev/null 2> $R/gpurun_out/$TAG/pmc_sq3.err
tail -3 $R/gpurun_out/$TAG/pmc_sq3.err
