set -x
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r07a; mkdir -p $O
python __graft_entry__.py > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout 1200 python -m pytest tests -m gpu -x -q -k "small_window or edge_cases or fuzz_sketch or tiny or overlap or btllib or indexlr_pos or variants_on_the_gpu" 2>&1 | tail -5 | tee $O/pytest_small.log
timeout 600 python tools/small_window_bench.py > $O/small_window_bench.jsonl 2> $O/small_window_bench.err; cat $O/small_window_bench.jsonl; tail -3 $O/small_window_bench.err
VARIANTS="nib" STEPS=8 bash tools/gpu_variants.sh r07a C3 C5
