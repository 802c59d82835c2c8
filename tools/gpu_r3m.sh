#!/bin/bash
# round 3: A/B of sketch_thresh_kernel (NTL_SKETCH_THRESH, threshold-sparsified windows) against the shipped window pass:
# parity on the GPU, C3 / C5 step times, the kernel alone (trace) and its SQ counters.
TAG=${1:-r03m}
O=gpurun_out/$TAG; mkdir -p $O
python __graft_entry__.py > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "variants or near or sketch" -o addopts= 2>&1 | tail -5 | tee $O/pytest_variants.log
run() { name=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 900 python bench.py --no-cpu-baseline --no-e2e --no-others "$@" > $O/bench_$name.json 2> $O/bench_$name.err
  python - $O/bench_$name.json <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{"metric'):
        j=json.loads(l); print(sys.argv[1], j["value"], j["ms_per_step"], "serial", j["config"].get("serial_pass",{}).get("ms_per_step"), "window ms/launch", j["roofline"]["avg_launch_ms"], "stages", j["config"].get("serial_pass",{}).get("stage_ms_per_step"))
PY
}
for c in ${CPWS:-0 1 10 8}; do
run c3_t$c NTL_SKETCH_THRESH=$c -- --steps 6 --warmup 1
done
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export NTL_PIPELINE=0
for c in ${PMCS:-1}; do
export NTL_SKETCH_THRESH=$c
B="python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-e2e --no-others --serial-steps 0 --workload C3"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace_C3_t$c -o kt -- $B > $R/$O/bench_trace_C3_t$c.json 2> $R/$O/trace_C3_t$c.err
timeout 900 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/$O/pmc_sq_C3_t$c -o p -- $B > /dev/null 2> $R/$O/pmc_sq_C3_t$c.err
done
cd $R
find $O -name '*kernel_trace.csv' -size +8M -delete
for c in ${PMCS:-1}; do head -6 $O/trace_C3_t$c/kt_kernel_stats.csv; done
ls $O
