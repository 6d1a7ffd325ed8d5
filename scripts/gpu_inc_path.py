"""Which cell sort each step's build queued ahead took (incremental merge or radix), step by step.
usage: python scripts/gpu_inc_path.py <workload> <steps>"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from adaptive_sph_amd import ffi, scene as sc
from adaptive_sph_amd.workloads import WORKLOADS
name, steps = sys.argv[1], int(sys.argv[2])
scene_f, params_f, _ = WORKLOADS[name]
scn, P = scene_f(), params_f()
pos, mass, vel = sc.init_particles(scn)
g = ffi.Context(ffi.load_product(), len(mass), sc.boundary_planes(scn.boundary, P.init_boundary_handler))
g.upload(mass, pos, vel)
g.profile_enable(1)
p = P.to_ffi()
prev = {}
for s in range(steps):
    st = g.step(p)
    prof = g.profile_get()
    d = {k: prof[k][0] - prev.get(k, (0, 0))[0] for k in prof}
    prev = prof
    print(f"step {s}: inc_reorder {d.get('inc_reorder', 0)} sort_scatter {d.get('sort_scatter', 0)} reorder {d.get('reorder', 0)} cell_start {d.get('cell_start', 0)} "
          f"tile_hmax {d.get('tile_hmax', 0)} dt {float(st.dt):.3g} iters {int(st.density_solver.iters)}", flush=True)
