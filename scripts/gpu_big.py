"""One large uniform dam break on one GPU: allocation, a few steps, device memory in use.  usage: gpu_big.py [side=8192] [steps=3]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import torch
from adaptive_sph_amd import ffi, scene as sc
from adaptive_sph_amd.workloads import dam_break_params
side = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
sp = 1.0 / side
scn = sc.dam_break_small(side, side, sp)
pos, mass, vel = sc.init_particles(scn)
n = len(mass)
P = dam_break_params(max_dt=0.002 * sp * 1024, max_iters=4)
free0, total = torch.cuda.mem_get_info()
g = ffi.Context(ffi.load_product(), n, sc.boundary_planes(scn.boundary, P.init_boundary_handler))
g.upload(mass, pos, vel)
p = P.to_ffi()
for s in range(steps):
    t0 = time.perf_counter(); st = g.step(p); ms = (time.perf_counter() - t0) * 1e3
    print(f"step {s}: {ms:.1f} ms, iterations {st.div_solver.iters} + {st.density_solver.iters}, dt {st.dt:.3e}", flush=True)
free1, _ = torch.cuda.mem_get_info()
x = g.download("position")
print(f"n = {n} ({n / 1e6:.1f} M), device memory in use by the context {(free0 - free1) / 2**30:.2f} GiB = {(free0 - free1) / n:.0f} B per particle; finite {bool(np.isfinite(x).all())}")
