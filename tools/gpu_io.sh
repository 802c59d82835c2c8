#!/bin/bash
# host-side ingest diagnostics on the GPU box (reader thread sweep), then the e2e leg of the bench with stage times
TAG=${1:-r02io}
mkdir -p gpurun_out/$TAG
python __graft_entry__.py > gpurun_out/$TAG/build.log 2>&1 || { tail -20 gpurun_out/$TAG/build.log; exit 1; }
nproc; free -g | head -2; df -h /dev/shm | tail -1
timeout 900 python tools/io_diag.py > gpurun_out/$TAG/io_diag.json 2> gpurun_out/$TAG/io_diag.err; python3 - <<PY
import json
j = json.load(open("gpurun_out/$TAG/io_diag.json"))
print({k: v for k, v in j.items() if k != "reader"})
for r in j["reader"]:
    if r["dest"] == "reuse":
        print(r, "GB/s count %.1f write %.1f" % (j["file_bytes"] / r["count_pass_s"] / 1e9, j["file_bytes"] / r["write_pass_s"] / 1e9))
PY
tail -3 gpurun_out/$TAG/io_diag.err
