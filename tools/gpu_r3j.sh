#!/bin/bash
TAG=${1:-r03j}
O=gpurun_out/$TAG; mkdir -p $O
python __graft_entry__.py > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
(echo "nproc $(nproc)"; echo "cpu.max $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"; echo "cfs_quota $(cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us 2>/dev/null) period $(cat /sys/fs/cgroup/cpu/cpu.cfs_period_us 2>/dev/null)"; grep -E "Cpus_allowed_list" /proc/self/status; cat /proc/loadavg; free -g | head -2; cat /sys/fs/cgroup/cpu.stat 2>/dev/null | head -6) | tee $O/host.txt
NTL_E2E_SWEEP2=1 timeout 1200 python tools/e2e_diag.py --bases 16e9 > $O/e2e_sweep2.jsonl 2> $O/e2e_sweep2.err
python - $O/e2e_sweep2.jsonl <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if not l.startswith('{"Gbases'): continue
    j=json.loads(l); print(j["Gbases_per_s"], j["seconds"], j["env"], "contigs", j["t_contigs"], "ingest-wait", j["t_ingest"], "device", j["t_device"], "handover", j["t_handover"], "write", j["t_write"], "reader", j["reader"])
PY
cat /sys/fs/cgroup/cpu.stat 2>/dev/null | head -6
