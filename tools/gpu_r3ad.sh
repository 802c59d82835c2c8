#!/bin/bash
TAG=${1:-r03ad}
O=gpurun_out/$TAG; mkdir -p $O
python __graft_entry__.py > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
env NTL_E2E_${SWEEP:-SWEEP3}=1 timeout 1500 python tools/e2e_diag.py --bases 32e9 > $O/e2e_sweep3.jsonl 2> $O/e2e_sweep3.err
python - $O/e2e_sweep3.jsonl <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if not l.startswith('{"Gbases'): continue
    j=json.loads(l); print(j["pass"], j["Gbases_per_s"], j["seconds"], {k:v for k,v in j["env"].items() if not k.startswith("NTL_E2E")}, j["batch_bases"]//1000000, "contigs", j["t_contigs"], "ingest-wait", j["t_ingest"], "device", j["t_device"], "handover", j["t_handover"], "write", j["t_write"], "tally", j["t_tally"], "reader", j["reader"])
PY
