"""The loopback transport ordered by events (default) against its host-synchronous form (SPH_LOOPBACK_SYNC=1) over many steps: a missing
dependency between two members' streams would show up as a field that differs.  k slabs of a dam break with re-balancing, split sweep A
forced on in the event-ordered group; both groups step in lockstep and are compared every 50 steps.
usage: gpu_loopback_events_vs_host.py [steps] [ranks] [n_side]"""
import os, sys
sys.path.insert(0, __file__.rsplit("/scripts/", 1)[0])
import numpy as np
import torch  # noqa: F401
from adaptive_sph_amd import distributed as D, ffi, scene as sc
from adaptive_sph_amd.workloads import dam_break_params_scaled
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
k = int(sys.argv[2]) if len(sys.argv) > 2 else 4
side = int(sys.argv[3]) if len(sys.argv) > 3 else 256
scn = sc.dam_break_small(side, side, 1.0 / side)
pos, mass, vel = sc.init_particles(scn)
planes = sc.boundary_planes(scn.boundary)
p = dam_break_params_scaled(1.0 / side)().to_ffi()
lib = ffi.load_product()
A = D.make_loopback_group(lib, pos, mass, vel, planes, k)
B = D.make_loopback_group(lib, pos, mass, vel, planes, k)
for c in A + B:
    c.dist_set_rebalance(20)
for s in range(steps):
    os.environ["SPH_OVERLAP"] = "1"
    os.environ.pop("SPH_LOOPBACK_SYNC", None)
    a = ffi.group_step(A, p)
    os.environ["SPH_OVERLAP"] = "0"
    os.environ["SPH_LOOPBACK_SYNC"] = "1"
    b = ffi.group_step(B, p)
    assert all(x.dt == y.dt and x.div_solver.iters == y.div_solver.iters and x.density_solver.iters == y.density_solver.iters for x, y in zip(a, b)), s
    if s % 50 == 49 or s == steps - 1:
        bad = []
        for ca, cb in zip(A, B):
            assert ca.n == cb.n, (s, ca.n, cb.n)
            bad += [f for f in ("particle_id", "position", "velocity", "density", "pressure") if not np.array_equal(ca.download(f), cb.download(f))]
        print(f"step {s + 1}: owned {[c.n for c in A]}, cuts moved {A[1].dist_get_cuts()[2]} times, fields that differ: {bad or 'none'}", flush=True)
        assert not bad
print("bit-identical over", steps, "steps")
