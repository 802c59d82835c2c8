#!/bin/bash
# kernel trace of the PIPELINED C3 steps: who runs beside whom, and for how long
TAG=${1:-r03x}
O=gpurun_out/$TAG; mkdir -p $O
python __graft_entry__.py > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $R/$O/trace_pipe -o kt -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-others --serial-steps 0 --workload C3 > $R/$O/bench_trace_pipe.json 2> $R/$O/trace_pipe.err
cd $R
python3 - $O/trace_pipe/kt_kernel_trace.csv > $O/timeline_summary.txt <<'PY'
import csv, sys, collections
rows=[]
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ",""), r.get("Queue_Id") or r.get("Stream_Id")))
rows.sort()
# last 2 steps: take window kernels (thresh) and restrict to the last 46 read launches
win=[x for x in rows if "sketch_thresh_kernel" in x[2]]
win=win[-46:]
t0=win[0][0]; t1=win[-1][1]
print("window launches", len(win), "span ms", (t1-t0)/1e6)
durs=[(e-s)/1e6 for s,e,_,_ in win]
gaps=[(win[i+1][0]-win[i][1])/1e6 for i in range(len(win)-1)]
print("window dur avg", sum(durs)/len(durs), "min", min(durs), "max", max(durs))
print("gap between window kernels avg", sum(gaps)/len(gaps), "max", max(gaps), "sum", sum(gaps))
inwin=[x for x in rows if x[0]>=t0 and x[1]<=t1]
agg=collections.defaultdict(lambda:[0,0.0])
for s,e,k,q in inwin:
    agg[(k,q)][0]+=1; agg[(k,q)][1]+=(e-s)/1e6
for (k,q),(n,ms) in sorted(agg.items(), key=lambda kv:-kv[1][1])[:25]:
    print(f"{ms:9.2f} ms {n:5d}  q={q}  {k[:70]}")
# for each window kernel: what overlapped with it
for i in (10,11,12):
    s,e,_,_=win[i]
    ov=[(max(a,s),min(b,e),k) for a,b,k,_ in rows if b>s and a<e and "sketch_thresh" not in k]
    print("window", i, "dur", (e-s)/1e6, "overlaps:", [(k[:24], round((b-a)/1e6,2)) for a,b,k in ov if b-a>20000])
PY
cat $O/timeline_summary.txt
find $O -name '*kernel_trace.csv' -size +8M -delete
