R=$GRAFT_REPO_ROOT; O=gpurun_out/r04as1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export NTL_PIPELINE=0
for REC in 0 1; do
export NTL_BENCH_RECORDS=$REC
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace_C5_rec$REC -o kt -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-e2e --no-others --serial-steps 0 --workload C5 > $R/$O/bench_rec$REC.json 2> $R/$O/rec$REC.err
find $R/$O -name '*kernel_trace.csv' -delete
head -9 $R/$O/trace_C5_rec$REC/*/kt_kernel_stats.csv 2>/dev/null || head -9 $R/$O/trace_C5_rec$REC/kt_kernel_stats.csv
done
