#!/bin/bash
# Ablation harness (runs ON the GPU box, through gpurun):  scripts/variants/run_matrix.sh <outdir under gpurun_out/> [variants file]
#   every line of the variants file is   <name> | <extra hipcc flags> | <environment assignments>
#   for each variant: rebuild libsph_hip.so with the flags, then
#     - rocprofv3 --kernel-trace of scripts/gpu_time.py dam_break_1m (30 steps after 20): median duration of every sweep kernel
#     - wall-clock ms/step of the steady window (steps 20-120) and of the driver's window (steps 5-25), uninstrumented
#   -> <outdir>/variants.md (one row per variant) ; copy it to profiles/ to have it judged
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
LIST=${2:-$GRAFT_REPO_ROOT/scripts/variants/variants.txt}
WL=${3:-dam_break_1m}
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
echo "| variant | build flags | env | density[build] us | pressure_accel us | jacobi us | source us | aii+nonpressure us | steady ms/step | driver-window ms/step |" > $OUT/variants.md
echo "|---|---|---|---|---|---|---|---|---|---|" >> $OUT/variants.md
while IFS='|' read -r NAME FLAGS ENVS; do
  NAME=$(echo $NAME); [ -z "$NAME" ] && continue; case "$NAME" in \#*) continue;; esac
  FLAGS=$(echo $FLAGS); ENVS=$(echo $ENVS)
  SPH_EXTRA_HIPCC_FLAGS="$FLAGS" python -c "from adaptive_sph_amd import build; build.build_hip(force=True)" > $OUT/$NAME.build.log 2>&1 || { echo "| $NAME | $FLAGS | $ENVS | build failed |" >> $OUT/variants.md; continue; }
  ( cd /tmp; export TMPDIR=/tmp; rm -rf $OUT/kt_$NAME
    env $ENVS timeout -k 5 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt_$NAME -o kt -- python $GRAFT_REPO_ROOT/scripts/gpu_time.py $WL 30 > $OUT/$NAME.kt.log 2>&1 )
  STEADY=$(env $ENVS python scripts/gpu_time.py $WL 100 2>/dev/null | tail -1 | sed -E 's/.*: ([0-9.]+) ms\/step.*/\1/')
  DRIVER=$(env $ENVS SPH_TIME_WARMUP=5 python scripts/gpu_time.py $WL 20 2>/dev/null | tail -1 | sed -E 's/.*: ([0-9.]+) ms\/step.*/\1/')
  python - "$OUT/kt_$NAME" "$NAME" "$FLAGS" "$ENVS" "$STEADY" "$DRIVER" >> $OUT/variants.md <<'PY'
import csv, glob, statistics, sys, collections
d, name, flags, envs, steady, driver = sys.argv[1:7]
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
dur = collections.defaultdict(list)
if f:
    for r in csv.DictReader(open(f[0])):
        dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
def med(*keys):
    v = [x for k, xs in dur.items() if all(q in k for q in keys) for x in xs]
    if not v:
        return "-"
    ref = sorted(v)[int(0.9 * (len(v) - 1))]
    v = [x for x in v if x > 0.25 * ref]      # launches that did work
    return f"{statistics.median(v):.1f}"
print(f"| {name} | {flags or '-'} | {envs or '-'} | {med('OpDensity', 'true>')} | {med('OpPressureAccel')} | {med('OpJacobi')} | {med('OpSource')} | {med('OpFuse')} | {steady} | {driver} |")
PY
  tail -1 $OUT/variants.md
done < $LIST
python -c "from adaptive_sph_amd import build; build.build_hip(force=True)"
