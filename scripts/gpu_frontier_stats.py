import sys, ctypes
sys.path.insert(0,'/root/repo')
from adaptive_sph_amd import ffi, scene as sc
from adaptive_sph_amd.workloads import WORKLOADS
scene_f, params_f, _ = WORKLOADS["dam_break_1m"]
scn=scene_f(); P=params_f(level_estimation_method="EmptyAngle", maximum_surface_distance=0.2, particle_radius_fine=0.0005, particle_radius_base=0.002)
pos,mass,vel=sc.init_particles(scn)
lib=ffi.load_product()
g=ffi.Context(lib,len(mass),sc.boundary_planes(scn.boundary)); g.upload(mass,pos,vel); p=P.to_ffi()
raw=ctypes.CDLL(str(lib.path)) if hasattr(lib,'path') else lib.lib
out=(ctypes.c_ulonglong*8)()
for s in range(24):
    st=g.step(p)
    raw.sph_debug_frontier_stats(out)
    if s in (0,1,2,5,10,20,23): print(s, 'waves_with_work',out[1],'rounds',out[2],'popped',out[6],'cands',out[3],'pushes',out[4],'2nd trips',out[5], 'surface', int(g.download("flag_is_fluid_surface").sum()))
