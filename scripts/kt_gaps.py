#!/usr/bin/env python3
"""GPU timeline of a rocprofv3 kernel trace: busy time, launch gaps and host waits.
   usage: kt_gaps.py <dir with *kernel_trace.csv> [skip_first_n_kernels_fraction]"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))), key=lambda t: t[0])
rows = rows[len(rows) // 3:]          # past the warm-up
span = rows[-1][1] - rows[0][0]
busy = sum(e - s for s, e, _ in rows)
gaps = [rows[i + 1][0] - rows[i][1] for i in range(len(rows) - 1)]
small = [g for g in gaps if 0 < g <= 6000]
mid = [g for g in gaps if 6000 < g <= 15000]
big = [g for g in gaps if g > 15000]
print(f"kernels {len(rows)}  span {span/1e6:.2f} ms  busy {busy/1e6:.2f} ms ({busy/span:.3f})")
print(f"gaps <= 6 us: {len(small)} sum {sum(small)/1e6:.2f} ms (mean {sum(small)/max(len(small),1)/1e3:.2f} us) | 6-15 us: {len(mid)} sum {sum(mid)/1e6:.2f} ms | > 15 us (host waits): {len(big)} sum {sum(big)/1e6:.2f} ms (mean {sum(big)/max(len(big),1)/1e3:.1f} us)")
after = collections.Counter()
for i, g in enumerate(gaps):
    if g > 15000:
        after[rows[i][2].split("(")[0][:60]] += 1
print("host waits follow:", after.most_common(5))
