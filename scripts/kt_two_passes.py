"""rocprofv3 --kernel-trace of ONE `bench.py --steps K --warmup W` run holds the timed window (uninstrumented) AND its instrumented repeat
(Profiler mode 4: the sweeps carry hipExtLaunchKernelGGL event pairs).  Per sweep kernel: rocprofv3's average duration over the working
launches of each pass, next to the line's own `avg_us` -- do the dispatch timestamps the line reads agree with rocprofv3's for the SAME
launches, and does instrumenting the queue change what a sweep takes?
usage: python scripts/kt_two_passes.py <dir with *kernel_trace.csv> <bench.json> [W=5] [K=20]"""
import collections
import csv
import json
import sys
from pathlib import Path

src, bench = Path(sys.argv[1]), json.load(open(sys.argv[2]))
W = int(sys.argv[3]) if len(sys.argv) > 3 else 5
K = int(sys.argv[4]) if len(sys.argv) > 4 else 20
rows = sorted(csv.DictReader(open(next(src.rglob("*kernel_trace.csv")))), key=lambda r: int(r["Start_Timestamp"]))
NAMES = {"jacobi_update": "OpJacobiU", "pressure_accel": "OpPressureAccelU", "density": "OpDensity", "source_term": "OpSource", "aii_nonpressure": "OpFuse"}
# step boundaries: one density BUILD launch per step; passes: steps [W, W+K) of the first context, [W+K+W, W+K+W+K) of the second
dens = [int(r["Start_Timestamp"]) for r in rows if "OpDensity" in r["Kernel_Name"] and "k_sweep<" in r["Kernel_Name"]]
print(f"{len(dens)} density launches in the trace (two passes of {W}+{K} steps = {2 * (W + K)} expected, then the other legs)")


def window(first_step, nsteps):
    t0 = dens[first_step]
    t1 = dens[first_step + nsteps] if first_step + nsteps < len(dens) else 1 << 62
    out = collections.defaultdict(list)
    for r in rows:
        s = int(r["Start_Timestamp"])
        if t0 <= s < t1:
            for k, pat in NAMES.items():
                if pat in r["Kernel_Name"]:
                    out[k].append((int(r["End_Timestamp"]) - s) / 1e3)
    return out


def working_avg(v):
    ref = sorted(v)[int(0.9 * (len(v) - 1))]
    w = [x for x in v if x > 0.25 * ref]
    return sum(w) / len(w), len(w)


timed, instr = window(W, K), window(W + K + W, K)
line = {k["name"]: k["avg_us"] for k in bench["kernels"]}
print(f"{'kernel':18s} {'rocprofv3, timed pass':>24s} {'rocprofv3, instrumented pass':>30s} {'line avg_us':>12s} {'line / rocprofv3 (instr.)':>26s}")
for k in NAMES:
    if timed.get(k) and instr.get(k):
        a, na = working_avg(timed[k])
        b, nb = working_avg(instr[k])
        print(f"{k:18s} {a:14.2f} us x{na:5d} {b:20.2f} us x{nb:5d} {line.get(k, float('nan')):12.2f} {line.get(k, float('nan')) / b:26.3f}")
spin = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if "k_spin_calib" in r["Kernel_Name"]]
if spin:
    print(f"k_spin_calib (10 us): rocprofv3 {sum(spin) / len(spin):.2f} us over {len(spin)} launches; line's dispatch-timed figure "
          f"{bench['roofline']['window'].get('calibration_spin10_dispatch_us')}")
