#!/bin/bash
cd /tmp; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/hiptrace
timeout -k 5 300 rocprofv3 --kernel-trace --hip-runtime-trace --output-format csv -d $OUT -o ht -- python $GRAFT_REPO_ROOT/scripts/gpu_time.py dam_break_1m 40 > $OUT.log 2>&1; echo "rc=$?"
ls $OUT
python - <<PY
import csv,glob
kt=glob.glob("$OUT/**/*kernel_trace.csv",recursive=True)[0]
ht=glob.glob("$OUT/**/*hip_api_trace.csv",recursive=True)[0]
K=sorted(((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"][:40]) for r in csv.DictReader(open(kt))))
H=sorted(((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Function"]) for r in csv.DictReader(open(ht))))
import bisect
hs=[h[0] for h in H]
rows=[]
for i in range(len(K)-1):
    if K[i][2].startswith("k_header_ahead"):
        end=K[i][1]; nxt=K[i+1]
        # first launch API call that begins after the kernel ended
        j=bisect.bisect_left(hs,end)
        while j<len(H) and "Launch" not in H[j][2]: j+=1
        if j<len(H): rows.append((H[j][0]-end, H[j][1]-H[j][0], nxt[0]-H[j][0], nxt[0]-end, nxt[2]))
rows=rows[len(rows)//2:]
import statistics
print("n",len(rows))
for k,name in enumerate(["header_ahead end -> first launch call begins","that launch call's duration","launch call begin -> kernel starts","total gap"]):
    print(name, "median %.1f us mean %.1f us"%(statistics.median(r[k] for r in rows)/1e3, statistics.mean(r[k] for r in rows)/1e3))
print(rows[-3:])
# what API calls happen in the turnaround window of the last boundary
PY
