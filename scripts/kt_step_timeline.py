#!/usr/bin/env python3
"""One step of a rocprofv3 kernel trace as a timeline: every launch with its offset, duration and the idle time before it,
   plus per-kernel sums over the step.  The step is the one of median length among those delimited by the marker kernel.
   usage: kt_step_timeline.py <dir with *kernel_trace.csv> [marker substring, default k_cell_start: one per step] [--full]"""
import csv, glob, sys, collections, statistics
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
marker = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else "k_cell_start"
full = "--full" in sys.argv
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))), key=lambda t: t[0])
def short(n):
    n = n.split("(")[0]
    for a, b in (("void ", ""), ("MathUniform", "U"), ("MathFast", "F"), ("MathExact", "E")):
        n = n.replace(a, b)
    return n[:70]
cuts = [i for i, r in enumerate(rows) if marker in r[2]]
steps = [(cuts[k] + 1, cuts[k + 1] + 1) for k in range(len(cuts) - 1)]
steps = steps[len(steps) // 2:]
lens = [rows[b - 1][1] - rows[a - 1][1] for a, b in steps]
med = statistics.median_low(lens)
a, b = steps[lens.index(med)]
t0 = rows[a - 1][1]
print(f"steps {len(steps)}; step lengths us: min {min(lens)/1e3:.1f} median {med/1e3:.1f} max {max(lens)/1e3:.1f}; launches in the median step: {b - a}")
agg = collections.OrderedDict()
prev_end = t0
idle = 0
for s, e, n in rows[a:b]:
    gap = s - prev_end
    idle += max(gap, 0)
    k = short(n)
    d = agg.setdefault(k, [0, 0, 0])
    d[0] += 1; d[1] += e - s; d[2] += max(gap, 0)
    if full:
        print(f"{(s - t0)/1e3:9.1f} us  +{gap/1e3:6.1f} idle  {(e - s)/1e3:7.1f} us  {k}")
    prev_end = max(prev_end, e)
print(f"idle in the step: {idle/1e3:.1f} us of {med/1e3:.1f}")
print(f"{'kernel':70s} {'n':>4s} {'busy us':>9s} {'idle before us':>14s}")
for k, (n, d, g) in sorted(agg.items(), key=lambda kv: -kv[1][1] - kv[1][2]):
    print(f"{k:70s} {n:4d} {d/1e3:9.1f} {g/1e3:14.1f}")
