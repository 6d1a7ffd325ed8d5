import os, sys
import numpy as np
sys.path.insert(0, '/root/repo')
from adaptive_sph_amd import ffi, scene as sc
from adaptive_sph_amd.workloads import dam_break_params
lib = ffi.load_product()
scn = sc.dam_break_small(128, 96, 1 / 64)
pos, mass, vel = sc.init_particles(scn)
P = dam_break_params()
p = P.to_ffi()
def run(env, steps=25):
    for k, v in env.items(): os.environ[k] = v
    g = ffi.Context(lib, len(mass), sc.boundary_planes(scn.boundary, P.init_boundary_handler))
    for k in env: del os.environ[k]
    g.upload(mass, pos, vel)
    its = []
    outs = []
    for s in range(steps):
        st = g.step(p)
        its.append((int(st.div_solver.iters), int(st.density_solver.iters)))
        outs.append({f: g.download(f) for f in ("position", "velocity", "pressure", "density")})
    return its, outs
ia, a = run({})
ib, b = run({"SPH_OFFSET_LISTS": "0"})
ic, c = run({"SPH_ACCEL_GENERIC": "1"})
idd, d = run({"SPH_ACCEL_GENERIC": "1", "SPH_OFFSET_LISTS": "0"})
for name, (ix, x) in {"offsets off": (ib, b), "accel generic": (ic, c), "accel generic + offsets off": (idd, d)}.items():
    first = None
    for s in range(len(a)):
        for f in a[s]:
            if not np.array_equal(a[s][f], x[s][f]):
                first = (s, f, int((a[s][f] != x[s][f]).sum()), float(np.abs(a[s][f] - x[s][f]).max()))
                break
        if first: break
    print(name, "iters equal", ia == ix, "first difference", first)
s = 5
dv = np.nonzero((a[s]["velocity"] != b[s]["velocity"]).any(axis=1))[0]
print("differing particles at step", s, dv[:40])
print("positions", a[s]["position"][dv[:8]])
print("velocity a", a[s]["velocity"][dv[:8]], "b", b[s]["velocity"][dv[:8]])
for s2 in range(0, 6):
    print(s2, "pressure equal", np.array_equal(a[s2]["pressure"], b[s2]["pressure"]), "velocity equal", np.array_equal(a[s2]["velocity"], b[s2]["velocity"]), ia[s2], ib[s2])
