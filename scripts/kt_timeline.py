"""Per-queue timeline of the last step(s) of a rocprofv3 kernel trace (runs of the same kernel compressed): python scripts/kt_timeline.py <kernel_trace.csv> [steps]"""
import csv, re, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
def short(n):
    m = re.search(r"(Op\w+|k_\w+|fillBuffer)", n)
    return m.group(1) if m else n[:30]
idx = [k for k, r in enumerate(rows) if "OpDensity" in r["Kernel_Name"]]
s0, s1 = idx[-1 - nsteps], idx[-1]
t0 = int(rows[s0]["Start_Timestamp"])
out = [((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, short(r["Kernel_Name"]), r["Queue_Id"]) for r in rows[s0 - 14:s1 + 1]]
for q in sorted(set(o[3] for o in out)):
    print("queue", q)
    oo = [o for o in out if o[3] == q]
    i = 0
    while i < len(oo):
        j = i
        while j + 1 < len(oo) and oo[j + 1][2] == oo[i][2]:
            j += 1
        print(f"  {oo[i][0]:9.1f} .. {oo[j][1]:9.1f}  x{j - i + 1:3d}  {oo[i][2]}")
        i = j + 1
