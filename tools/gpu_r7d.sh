set -x
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r07d; mkdir -p $O
python __graft_entry__.py > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
(grep -m1 "model name" /proc/cpuinfo; grep -m1 flags /proc/cpuinfo | tr ' ' '\n' | grep -E "avx512|bmi2|gfni|vbmi|avx2" | tr '\n' ' '; echo; nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null) | tee $O/host_cpu.txt
timeout 900 python tests/gpu_small_soak.py ${SOAK_S:-240} 9100 2>&1 | tail -3 | tee $O/small_soak.log
