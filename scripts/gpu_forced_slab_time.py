"""ONE rank in forced slab mode (SPH_FORCE_SLAB_MODE=1, RCCL communicator of one rank): what the slab driver costs a rank per
step apart from the neighbour exchange.  usage: gpu_forced_slab_time.py [steps]"""
import os, sys, time, ctypes as C
os.environ["SPH_FORCE_SLAB_MODE"] = "1"
sys.path.insert(0, __file__.rsplit("/scripts/", 1)[0])
import torch  # noqa: F401
from adaptive_sph_amd import distributed as D, ffi, scene as sc
from adaptive_sph_amd.workloads import WORKLOADS
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
wl = sys.argv[2] if len(sys.argv) > 2 else "dam_break_1m"
scene_f, params_f, _ = WORKLOADS[wl]
scn, P = scene_f(), params_f()
pos, mass, vel = sc.init_particles(scn)
lib = ffi.load_product()
raw = (C.c_uint8 * 128)()
lib.comm_unique_id(raw)
c = ffi.Context(lib, len(mass) + 65536, sc.boundary_planes(scn.boundary))
c.dist_configure(0, 1, -D.INF, D.INF)
c.comm_init(bytes(raw), 0, 1)
c.upload(mass, pos, vel)
p = P.to_ffi()
for _ in range(20):
    c.step(p)
t0 = time.perf_counter()
for _ in range(steps):
    c.step(p)
w = c.dist_get_stats()
print(f"forced slab mode, 1 rank, {wl}: {(time.perf_counter() - t0) / steps * 1e3:.3f} ms/step; per step over the whole run: host waits {w['host_waits'] / (steps + 20):.2f}, "
      f"all-reduces {w['allreduces'] / (steps + 20):.2f}, exchanges {w['exchanges'] / (steps + 20):.2f}")
