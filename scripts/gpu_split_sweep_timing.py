"""Split sweep A (slabs with neighbours: interior under the iteration's exchange + all-reduce) against the one-launch
form (SPH_OVERLAP=0) on ONE GPU: k slabs of configs[1] as a loopback group and as ranks on threads.  One GPU runs all ranks' kernels, so
this measures what the split costs (two launches, the totals kernel, three event dependencies per iteration) and what the loopback /
thread transports' host waits hide -- not what RCCL latency it hides on k GPUs.
With SPH_DEBUG_COMM_DELAY_US=<us> (loopback only) every exchange / all-reduce occupies its stream for that long first: the split form
hides it under the interior sweep, the one-launch form pays it on the critical path.
usage: gpu_split_sweep_timing.py [workload] [steps]"""
import os, sys, time
sys.path.insert(0, __file__.rsplit("/scripts/", 1)[0])
import numpy as np
import torch  # noqa: F401
from adaptive_sph_amd import distributed as D, ffi, scene as sc
from adaptive_sph_amd.workloads import WORKLOADS
wl = sys.argv[1] if len(sys.argv) > 1 else "dam_break_1m"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
scene_f, params_f, _ = WORKLOADS[wl]
scn, P = scene_f(), params_f()
pos, mass, vel = sc.init_particles(scn)
planes = sc.boundary_planes(scn.boundary)
lib = ffi.load_product()
p = P.to_ffi()
delay = os.environ.get("SPH_DEBUG_COMM_DELAY_US", "0")
print(f"{wl}: {len(mass)} particles, {steps} timed steps after 20, every loopback collective delayed by {delay} us")
for transport in (("loopback",) if (delay != "0" or len(mass) > 2_000_000) else ("loopback", "threads")):
    for k in ((2,) if (delay != "0" or len(mass) > 2_000_000) else (2, 4, 8)):
        row = {}
        for mode in ("1", "0"):
            os.environ["SPH_OVERLAP"] = mode
            thr = None
            if transport == "threads":
                thr = D.ThreadedGroup(lib, pos, mass, vel, planes, k)
                grp, step = thr.contexts, (lambda: thr.step(p))
            else:
                grp = D.make_loopback_group(lib, pos, mass, vel, planes, k)
                step = (lambda: ffi.group_step(grp, p))
            try:
                for _ in range(20):
                    step()
                for c in grp:
                    c.dist_get_stats(reset=True)
                t0 = time.perf_counter()
                its = 0
                for _ in range(steps):
                    st = step()
                    its += st[0].div_solver.iters + st[0].density_solver.iters
                dt = (time.perf_counter() - t0) / steps
                st0 = grp[0].dist_get_stats()
                waits = st0["host_waits"] / steps
                row[mode] = (dt * 1e3, its / steps, waits)
                coll = (st0["exchanges"] / steps, st0["allreduces"] / steps)
            finally:
                if thr:
                    thr.close()
                else:
                    for c in grp:
                        c.close()
        (a, ia, wa), (b, ib, wb) = row["1"], row["0"]
        print(f"{transport:9s} k={k}: split {a:.3f} ms/step ({ia:.1f} iterations, {wa:.1f} host waits)   one launch {b:.3f} ms/step ({ib:.1f}, {wb:.1f})   ratio {a / b:.3f}   [{coll[0]:.1f} exchanges + {coll[1]:.1f} all-reduces per step and rank]", flush=True)
