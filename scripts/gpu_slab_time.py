"""Cost of the slab machinery itself, on ONE GPU: k loopback ranks stepping the same scene as a single context (the ranks
share the device, so ideal = the single context's time; the difference is partition / ghost / pack / host-wait overhead,
without any link latency).  usage: gpu_slab_time.py [workload] [k] [steps]"""
import sys, time
sys.path.insert(0, ".")
import torch  # noqa: F401
from adaptive_sph_amd import distributed as D, ffi, scene as sc
from adaptive_sph_amd.workloads import WORKLOADS

wl = sys.argv[1] if len(sys.argv) > 1 else "dam_break_1m"
k = int(sys.argv[2]) if len(sys.argv) > 2 else 2
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 60
scene_f, params_f, _ = WORKLOADS[wl]
scn, P = scene_f(), params_f()
pos, mass, vel = sc.init_particles(scn)
planes = sc.boundary_planes(scn.boundary)
lib = ffi.load_product()
p = P.to_ffi()
single = ffi.Context(lib, len(mass), planes)
single.upload(mass, pos, vel)
for _ in range(20):
    single.step(p)
t0 = time.perf_counter()
for _ in range(steps):
    single.step(p)
t1 = (time.perf_counter() - t0) / steps * 1e3
single.close()
print(f'single {t1:.3f} ms/step', flush=True)
grp = D.make_loopback_group(lib, pos, mass, vel, planes, k)
for w in range(20):
    ffi.group_step(grp, p)
    if w < 3: print('group warmup step', w, flush=True)
t0 = time.perf_counter()
for _ in range(steps):
    ffi.group_step(grp, p)
tk = (time.perf_counter() - t0) / steps * 1e3
print(f"{wl}: single {t1:.3f} ms/step; {k} loopback ranks on the same GPU {tk:.3f} ms/step (+{tk - t1:.3f} ms of slab machinery)")
grp[0].profile_reset(); [c.profile_enable(1) for c in grp]
for _ in range(10):
    ffi.group_step(grp, p)
prof = grp[0].profile_get()
tot = sum(v[1] for v in prof.values())
for name, (launches, ms) in sorted(prof.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"  rank 0  {name:24s} {launches / 10:7.1f} launches/step {ms / 10 * 1e3:9.1f} us/step")
