#!/bin/bash
cd /tmp; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/paced_kt
SPH_PACE_PRED=${1:-0} SPH_TIME_WARMUP=5 timeout -k 5 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -o kt -- python $GRAFT_REPO_ROOT/scripts/gpu_time.py dam_break_1m 20 > $OUT.log 2>&1; echo "rc=$?"
python - <<PY
import csv,glob,collections,statistics
f=glob.glob("$OUT/**/*kernel_trace.csv",recursive=True)[0]
rows=sorted(((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"][:48]) for r in csv.DictReader(open(f))))
# last 20 steps: find k_header_ahead publishing boundaries... take the last 60% of rows
rows=rows[int(len(rows)*0.25):]
span=rows[-1][1]-rows[0][0]; busy=sum(e-s for s,e,_ in rows)
print("span %.2f ms busy %.2f ms (%.3f)"%(span/1e6,busy/1e6,busy/span))
gaps=collections.defaultdict(list)
for i in range(len(rows)-1):
    g=rows[i+1][0]-rows[i][1]
    gaps[(rows[i][2][:34],rows[i+1][2][:34])].append(g/1e3)
tot=sum(sum(v) for v in gaps.values())
print("total gap ms %.2f"%(tot/1e3))
for k,v in sorted(gaps.items(), key=lambda kv:-sum(kv[1]))[:10]:
    print("%-36s -> %-36s n=%4d mean %.2f us sum %.1f us"%(k[0],k[1],len(v),statistics.mean(v),sum(v)))
d=collections.defaultdict(list)
for s,e,n in rows: d[n].append((e-s)/1e3)
for k,v in sorted(d.items(), key=lambda kv:-sum(kv[1]))[:6]:
    print(f"{k:50s} n={len(v):5d} short(<8us)={sum(1 for x in v if x<8):4d} total_ms={sum(v)/1e3:8.2f}")
PY
