#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3e
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 5 120 python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1; echo "build rc=$?"
timeout -k 5 90 ./scripts/ubench/bin/jacobi_lab 1024 0.15 50 > $OUT/jacobi_lab_j15.md 2>&1; echo "lab rc=$?"; cat $OUT/jacobi_lab_j15.md
timeout -k 5 90 ./scripts/ubench/bin/jacobi_lab 1024 0.0 50 > $OUT/jacobi_lab_j0.md 2>&1; echo "lab rc=$?"; cat $OUT/jacobi_lab_j0.md
timeout -k 5 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "level or dedicated" > $OUT/pytest_level.log 2>&1; echo "pytest level rc=$?"; tail -5 $OUT/pytest_level.log
LV="dict(level_estimation_method='EmptyAngle', maximum_surface_distance=0.2, particle_radius_fine=0.0005, particle_radius_base=0.002)"
timeout -k 5 120 python scripts/gpu_time.py dam_break_1m 20 "$LV" 2>&1 | tail -1
SPH_LEVEL_GENERIC=1 timeout -k 5 120 python scripts/gpu_time.py dam_break_1m 20 "$LV" 2>&1 | tail -1
cd /tmp; export TMPDIR=/tmp
SPH_TIME_WARMUP=20 timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/level_kt -o kt -- python $GRAFT_REPO_ROOT/scripts/gpu_time.py dam_break_1m 20 "$LV" > $OUT/level_kt.log 2>&1; echo "kt rc=$?"
python - <<PY
import csv,glob,collections,statistics
f=glob.glob("$OUT/level_kt/**/*kernel_trace.csv",recursive=True)[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)): d[r["Kernel_Name"][:70]].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k,v in sorted(d.items(), key=lambda kv:-sum(kv[1]))[:12]: print(f"{k:72s} n={len(v):5d} med={statistics.median(v):7.1f} total_ms={sum(v)/1e3:8.2f}")
PY
