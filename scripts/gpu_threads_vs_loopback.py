"""Every rank on its own host thread (thread transport: each rank sees only its own counts and takes its own branches, as a rank of the RCCL
transport does) against the loopback group, in lockstep over many steps of a dam break with re-balancing -- including scenes whose solve
diverges for a few steps (side 384: particles are thrown across slabs, the refresh falls back, the hand-over repeats).  A rank that
takes a different branch from its peers shows up as a time-out / "different collective" error; a wrong exchange as a field that differs.
usage: gpu_threads_vs_loopback.py [steps] [ranks] [n_side] [level|plain] ["dict(param overrides)"]"""
import os, sys
sys.path.insert(0, __file__.rsplit("/scripts/", 1)[0])
import numpy as np
import torch  # noqa: F401
from adaptive_sph_amd import distributed as D, ffi, scene as sc
from adaptive_sph_amd.workloads import dam_break_params_scaled
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
k = int(sys.argv[2]) if len(sys.argv) > 2 else 4
side = int(sys.argv[3]) if len(sys.argv) > 3 else 256
level = len(sys.argv) > 4 and sys.argv[4] == "level"
scn = sc.dam_break_small(side, side, 1.0 / side)
pos, mass, vel = sc.init_particles(scn)
planes = sc.boundary_planes(scn.boundary)
kw = dict(level_estimation_method="EmptyAngle", maximum_surface_distance=0.2, particle_radius_fine=0.5 / side, particle_radius_base=2.0 / side) if level else {}
if len(sys.argv) > 5:
    kw.update(eval(sys.argv[5]))
p = dam_break_params_scaled(1.0 / side)(**kw).to_ffi()
lib = ffi.load_product()
A = D.make_loopback_group(lib, pos, mass, vel, planes, k)
T = D.ThreadedGroup(lib, pos, mass, vel, planes, k)
try:
    for c in A + T.contexts:
        c.dist_set_rebalance(25)
    for s in range(steps):
        a = ffi.group_step(A, p)
        b = T.step(p)
        assert all(x.dt == y.dt and x.div_solver.iters == y.div_solver.iters and x.density_solver.iters == y.density_solver.iters for x, y in zip(a, b)), s
        if s % 50 == 49 or s == steps - 1:
            bad = []
            for ca, cb in zip(A, T.contexts):
                assert ca.n == cb.n, (s, ca.n, cb.n)
                bad += [f for f in ("particle_id", "position", "velocity", "density", "pressure") if not np.array_equal(ca.download(f), cb.download(f))]
            print(f"step {s + 1}: owned {[c.n for c in A]}, max iterations so far {max(x.density_solver.iters for x in a)}, fields that differ: {bad or 'none'}", flush=True)
            assert not bad
    print("bit-identical over", steps, "steps")
finally:
    T.close()
