"""Long loopback run of the slab decomposition with re-balancing: the 256 x 256 dam break through its collapse, 4 ranks on one
GPU, cuts re-set every 20 steps.  Checks: nothing lost, finite, inside the box, counts level, statistics close to a single context."""
import sys
sys.path.insert(0, __file__.rsplit("/scripts/", 1)[0])
import numpy as np
import torch  # noqa: F401
from adaptive_sph_amd import distributed as D, ffi, scene as sc
from adaptive_sph_amd.workloads import WORKLOADS
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
scene_f, params_f, _ = WORKLOADS["dam_break_64k"]
scn, P = scene_f(), params_f()
pos, mass, vel = sc.init_particles(scn)
planes = sc.boundary_planes(scn.boundary)
lib = ffi.load_product()
p = P.to_ffi()
single = ffi.Context(lib, len(mass), planes)
single.upload(mass, pos, vel)
grp = D.make_loopback_group(lib, pos, mass, vel, planes, 4)
for c in grp:
    c.dist_set_rebalance(20)
worst = 0
for s in range(steps):
    single.step(p)
    try:
        ffi.group_step(grp, p)
    except ffi.SphError as e:
        print("group failed at step", s, e)
        break
    if s % 500 == 499:
        n = [c.n for c in grp]
        worst = max(worst, max(n) - min(n))
        print("step", s + 1, "counts", n, "cuts", [round(c.dist_get_cuts()[1], 3) for c in grp[:-1]], flush=True)
n = len(mass)
ids = np.concatenate([c.download("particle_id") for c in grp])
x = D.gather_by_id(grp, "position", n)
v = D.gather_by_id(grp, "velocity", n)
xs, vs = single.download("position"), single.download("velocity")
print("ids complete:", np.array_equal(np.sort(ids), np.arange(n)), "finite:", bool(np.isfinite(x).all() and np.isfinite(v).all()),
      "inside box:", bool((np.abs(x[:, 0]) <= 2.001).all() and (np.abs(x[:, 1]) <= 1.001).all()))
print("kinetic energy group / single:", float((mass * (v ** 2).sum(1)).sum() / (mass * (vs ** 2).sum(1)).sum()),
      " centre of mass x group / single:", float((mass * x[:, 0]).sum() / mass.sum()), float((mass * xs[:, 0]).sum() / mass.sum()),
      " time", grp[0].time, single.time, " re-balances", grp[0].dist_get_cuts()[2], " worst imbalance", worst)
