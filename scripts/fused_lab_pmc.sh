#!/bin/bash
# VALU instructions actually issued by the fused-Jacobi laboratory's kernels (profiles/r6_variants.md section 1): one rocprofv3 --pmc pass
# usage (through gpurun): bash scripts/fused_lab_pmc.sh gpurun_out/r6_fused_pmc
OUT=$GRAFT_REPO_ROOT/$1; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout -k 5 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d $OUT/p -o p -- $GRAFT_REPO_ROOT/scripts/ubench/fused_lab 1024 0.0 3 > $OUT/run.log 2>&1; echo "rc=$?"
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/p/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
n = 1 << 20
print("| kernel | waves per launch | VALU instructions per launch | VALU per particle | LDS instructions per particle | VMEM reads per particle |")
print("|---|---|---|---|---|---|")
for k, c in acc.items():
    m = lambda name: sum(c[name]) / max(len(c[name]), 1)
    print(f"| {k} | {m('SQ_WAVES'):.0f} | {m('SQ_INSTS_VALU'):.3g} | {m('SQ_INSTS_VALU') * 64 / n:.1f} | {m('SQ_INSTS_LDS') * 64 / n:.1f} | {m('SQ_INSTS_VMEM_RD') * 64 / n:.1f} |")
PY
