#!/bin/bash
# One GPU-box visit (rounds 4-6): build, all gpu tests, smoke, the default bench line (C3 pipelined + serial pass + C2/C5 +
# cpu baseline + end to end), a single-rank torch.distributed.run line, then -- with NTL_PIPELINE=0, so that per-kernel
# figures are those of kernels running alone -- rocprofv3 kernel trace + three PMC passes per workload, each in its own run.
# usage: tools/gpu_round5.sh <tag> [workloads to profile, default "C3 C5 C2"]   (SKIP_TESTS / SKIP_TORCHRUN / SKIP_PROF / SKIP_BENCH=1)
set -x
TAG=${1:-r06z}; shift
WL=${*:-C3 C5 C2}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
[ -f "$R/bench.py" ] || { echo "no bench.py under $R"; exit 1; }
cd "$R"
O=gpurun_out/$TAG; mkdir -p $O
python __graft_entry__.py > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
if [ -z "$SKIP_TESTS" ]; then
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $O/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2 | tee $O/smoke.log
fi
if [ -z "$SKIP_BENCH" ]; then
timeout 1500 python bench.py $BENCH_ARGS > $O/bench.json 2> $O/bench.err
head -c 1500 $O/bench.json; echo; tail -5 $O/bench.err
fi
if [ -z "$SKIP_TORCHRUN" ]; then
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-others > $O/bench_torchrun1.json 2> $O/bench_torchrun1.err; tail -c 300 $O/bench_torchrun1.json; tail -3 $O/bench_torchrun1.err
fi
if [ -z "$SKIP_PROF" ]; then
cd /tmp && export TMPDIR=/tmp
# the default (two-stream) command under the kernel trace: the window kernel's average duration INSIDE the timed region (roofline.avg_launch_ms)
for W in $WL; do
B="python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-others --serial-steps 0 --workload $W"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace_${W}_pipe -o kt -- $B > $R/$O/bench_trace_${W}_pipe.json 2> $R/$O/trace_${W}_pipe.err
done
export NTL_PIPELINE=0
for W in $WL; do
B="python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-e2e --no-others --serial-steps 0 --workload $W"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace_$W -o kt -- $B > $R/$O/bench_trace_$W.json 2> $R/$O/trace_$W.err
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$O/pmc_fetch_$W -o p -- $B > /dev/null 2> $R/$O/pmc_fetch_$W.err
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$O/pmc_write_$W -o p -- $B > /dev/null 2> $R/$O/pmc_write_$W.err
timeout 900 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/$O/pmc_sq_$W -o p -- $B > $R/$O/bench_pmc_sq_$W.json 2> $R/$O/pmc_sq_$W.err
done
# for the record: the workgroup-per-strip threshold pass that sketch_wave_kernel replaced (NTL_SKETCH_WAVE=0)
export NTL_SKETCH_WAVE=0
B="python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-e2e --no-others --serial-steps 0 --workload C3"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace_C3_thresh -o kt -- $B > $R/$O/bench_trace_C3_thresh.json 2> $R/$O/trace_C3_thresh.err
unset NTL_SKETCH_WAVE
unset NTL_PIPELINE
cd $R
# the raw per-dispatch traces are large: keep the stats and the counter tables
find $O -name '*kernel_trace.csv' -size +8M -delete
head -12 $O/trace_C3/kt_kernel_stats.csv
fi
ls $O/
