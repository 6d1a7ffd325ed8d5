#!/usr/bin/env python3
"""Per-kernel table of PMC counters from scripts/pmc_run.sh output:  python scripts/pmc_table.py gpurun_out/pmcX"""
import csv, collections, glob, statistics, sys
tab = collections.defaultdict(dict)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if kn.startswith("k_sweep_off<"):
            kn = kn[len("k_sweep_off<"):-1].strip()
        n = kn.replace("k_sweep<", "").replace("<MathUniform, false>", "").replace("<MathUniform, true>", "<dist>").replace("<MathUniform>", "").replace(", false>", "").replace(", true>", "[build]")
        acc[(n, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (n, c), v in acc.items():
        big = [x for x in v if x >= 0.5 * max(v)] or v     # launches that did real work
        tab[n][c] = statistics.median(big)
names = [n for n in tab if n.startswith("Op")]
cs = sorted({c for n in names for c in tab[n]})
print(f"{'counter':28s}" + "".join(f"{n[:16]:>17s}" for n in names))
for c in cs:
    print(f"{c:28s}" + "".join(f"{tab[n].get(c, float('nan')):17.4g}" for n in names))
