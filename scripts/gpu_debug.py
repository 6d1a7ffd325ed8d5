"""Ad-hoc GPU-vs-oracle comparison (development aid; the real checks live in tests/)."""
import sys, time
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
import numpy as np
from adaptive_sph_amd import ffi, scene as sc, build
from tests.oracle_harness import load_oracle, uniform_params, csr_sets

build.build_hip()
olib = load_oracle()
plib = ffi.load_product()
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
scn = sc.dam_break_small(nx, nx, 1.0 / nx)
pos, mass, vel = sc.init_particles(scn)
planes = sc.boundary_planes(scn.boundary)
P = uniform_params(check_neighborhood=(nx <= 64))
p = P.to_ffi()
o = ffi.Context(olib, len(mass), planes); o.upload(mass, pos, vel)
g = ffi.Context(plib, len(mass), planes); g.upload(mass, pos, vel)
def rel(a, b):
    d = np.abs(a.astype(np.float64) - b.astype(np.float64)).max()
    s = np.abs(b.astype(np.float64)).max()
    return d / (s if s > 0 else 1.0)
for s in range(steps):
    so = o.step(p); sg = g.step(p)
    print(f"step {s}: dt {so.dt} {sg.dt} | div iters {so.div_solver.iters} {sg.div_solver.iters} | dens iters {so.density_solver.iters} {sg.density_solver.iters}")
    print("   avg err", so.div_solver.avg_error, sg.div_solver.avg_error, so.density_solver.avg_error, sg.density_solver.avg_error)
    for f in ["h2", "cell_index", "neighbor_count", "lambda_sum", "lambda_grad_sum", "density", "constant_field", "aii", "ppe_source_term", "pressure", "pressure_accel", "velocity", "position"]:
        a, b = g.download(f), o.download(f)
        if a.dtype.kind in "ui":
            print(f"   {f:18s} equal={np.array_equal(a, b)} ndiff={(a != b).sum()}")
        else:
            print(f"   {f:18s} rel={rel(a, b):.3e} bitexact={np.array_equal(a, b)}")
go, gi = g.download_neighbors(); oo, oi = o.download_neighbors()
same = np.array_equal(go, oo) and all(np.array_equal(x, y) for x, y in zip(csr_sets(go, gi), csr_sets(oo, oi)))
print("neighbor sets equal:", same, "total", len(gi), len(oi))
gr = g.grid(); orr = o.grid()
print("grid", gr.cell_size, gr.cells_min_x, gr.cells_min_y, gr.size_x, gr.size_y, "|", orr.cell_size, orr.cells_min_x, orr.cells_min_y, orr.size_x, orr.size_y)
