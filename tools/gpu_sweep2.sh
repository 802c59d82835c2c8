R=$GRAFT_REPO_ROOT; O=gpurun_out/${TAG:-r04at}; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sketch or probe or full or fixture" 2>&1 | tail -2
timeout 400 python tools/share_sweep.py --workload C3 --steps 4 "" "" > $O/C3.jsonl 2> $O/C3.err; cut -c1-330 $O/C3.jsonl
timeout 400 python tools/share_sweep.py --workload C5 --steps 2 "" "" > $O/C5.jsonl 2> $O/C5.err; cut -c1-330 $O/C5.jsonl
cd /tmp && export TMPDIR=/tmp
export NTL_PIPELINE=0
for W in C3 C5; do
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace_$W -o kt -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-e2e --no-others --serial-steps 0 --workload $W > $R/$O/bench_$W.json 2> $R/$O/$W.err
find $R/$O -name '*kernel_trace.csv' -delete
head -6 $R/$O/trace_$W/kt_kernel_stats.csv | cut -c1-150
done
