import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from adaptive_sph_amd import ffi, scene as sc
from adaptive_sph_amd.workloads import dam_break_params
plib = ffi.load_product()
scn = sc.dam_break_small(256, 256, 1 / 256)
pos, mass, vel = sc.init_particles(scn)
c = ffi.Context(plib, len(mass), sc.boundary_planes(scn.boundary))
c.upload(mass, pos, vel)
p = dam_break_params().to_ffi()
c.step(p)
c.profile_reset(); c.profile_enable(3)
for _ in range(3): c.step(p)
print(c.profile_get())
