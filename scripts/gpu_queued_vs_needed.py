import sys, time
sys.path.insert(0, "/root/repo")
from adaptive_sph_amd import ffi, scene as sc
from adaptive_sph_amd.workloads import WORKLOADS
wl = "dam_break_1m"
scene_f, params_f, _ = WORKLOADS[wl]
scn, P = scene_f(), params_f()
pos, mass, vel = sc.init_particles(scn)
g = ffi.Context(ffi.load_product(), len(mass), sc.boundary_planes(scn.boundary, P.init_boundary_handler))
g.upload(mass, pos, vel)
p = P.to_ffi()
g.profile_enable(1)
out = []
for s in range(30):
    g.profile_reset()
    st = g.step(p)
    pr = g.profile_get()
    out.append((int(st.div_solver.iters), int(st.density_solver.iters), pr.get("jacobi_update", (0,))[0], pr.get("pressure_accel", (0,))[0]))
print("(div iters, dens iters, B launches, A launches) per step:")
print(out)
need = sum(a + b for a, b, _, _ in out); got = sum(c for _, _, c, _ in out)
print("B sweeps needed", need, "queued", got)
