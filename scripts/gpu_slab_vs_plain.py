#!/usr/bin/env python3
"""A forced one-rank slab (slab driver + RCCL communicator of one rank: every collective of the N > 1 path, no neighbour) against the
plain context, same scene, same windows -- what the slab path costs a rank that runs the headline's kernels.
usage: gpu_slab_vs_plain.py [workload ...]      (default: dam_break_1m dam_break_8m)
Prints one markdown table row per workload and window; the driver window = steps 5..24 from rest, settled = steps 20..119."""
import ctypes as C
import os
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
import numpy as np  # noqa: E402
import torch  # noqa: F401,E402
from adaptive_sph_amd import distributed as D, ffi, scene as sc  # noqa: E402
from adaptive_sph_amd.workloads import WORKLOADS  # noqa: E402

lib = ffi.load_product()


def make(wl, slab, env):
    for k, v in env.items():
        os.environ[k] = v
    scene_f, params_f, _ = WORKLOADS[wl]
    scn, P = scene_f(), params_f()
    pos, mass, vel = sc.init_particles(scn)
    planes = sc.boundary_planes(scn.boundary, P.init_boundary_handler)
    if slab:
        os.environ["SPH_FORCE_SLAB_MODE"] = "1"
        c = ffi.Context(lib, len(mass) + 65536, planes)
        raw = (C.c_uint8 * 128)()
        lib.comm_unique_id(raw)
        c.dist_configure(0, 1, -D.INF, D.INF)
        c.comm_init(bytes(raw), 0, 1)
        os.environ.pop("SPH_FORCE_SLAB_MODE")
    else:
        c = ffi.Context(lib, len(mass), planes)
    c.upload(mass, pos, vel)
    for k in env:
        os.environ.pop(k)
    return c, P.to_ffi(), len(mass)


def window(c, p, warm, steps):
    for _ in range(warm):
        c.step(p)
    c.dist_get_stats(reset=True)
    t0 = time.perf_counter()
    its = []
    for _ in range(steps):
        st = c.step(p)
        its.append(int(st.div_solver.iters) + int(st.density_solver.iters) + 2)
    el = time.perf_counter() - t0
    w = c.dist_get_stats()
    return el / steps * 1e3, float(np.mean(its)), w["host_waits"] / steps, w["exchanges"] / steps


forms = [("plain context", False, {}), ("one-rank slab (records + paced: the default)", True, {}),
         ("one-rank slab, predicted queue (SPH_SLAB_PACED=0)", True, {"SPH_SLAB_PACED": "0"}),
         ("one-rank slab, generic sweep A + predicted queue (round 3's form)", True, {"SPH_SLAB_PACED": "0", "SPH_SLAB_RECORDS": "0"})]
print("| workload | form | driver window ms/step | iterations | waits/step | exchanges/step | settled ms/step | vs plain (driver / settled) |")
print("|---|---|---|---|---|---|---|---|")
for wl in (sys.argv[1:] or ["dam_break_1m", "dam_break_8m"]):
    base = None
    for name, slab, env in forms:
        c, p, n = make(wl, slab, env)
        d_ms, d_it, d_w, d_x = window(c, p, 5, 20)      # steps 5..24
        s_ms, _, _, _ = window(c, p, 0 if wl.endswith("8m") else 0, 40 if wl.endswith("8m") else 95)   # continues: steps 25..(119)
        c.close()
        if base is None:
            base = (d_ms, s_ms)
        print(f"| {wl} ({n}) | {name} | {d_ms:.3f} | {d_it:.1f} | {d_w:.2f} | {d_x:.1f} | {s_ms:.3f} | {d_ms / base[0]:.3f} / {s_ms / base[1]:.3f} |", flush=True)
