set -x
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD
O=gpurun_out/r07e; mkdir -p $O
python __graft_entry__.py > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
for M in avx512 avx2 avx512 avx2; do
  NTL_IO_SIMD=$M timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-others --serial-steps 0 2> $O/e2e_$M.err | python -c "
import sys, json
d = json.loads(sys.stdin.read()); e = d['end_to_end']
print('$M', e['value'], e.get('seconds'), 'steady', e.get('steady_state', {}).get('value'), e.get('steady_state', {}).get('seconds'), 'gz', e.get('compressed_inputs'))
" | tee -a $O/e2e_simd_ab.txt
done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace_small -o kt -- python3 $R/tools/small_window_bench.py 15,5 20,10 > $R/$O/small_window_bench_profiled.jsonl 2> $R/$O/trace_small.err
cd $R
find $O -name '*kernel_trace.csv' -size +8M -delete
find $O -name 'kt_kernel_stats.csv' | head -1 | xargs head -12
