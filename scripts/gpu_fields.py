"""Per-field relative error GPU vs oracle on the 2:1 colliding-columns scene (development aid)."""
import sys
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
import numpy as np
from adaptive_sph_amd import ffi, scene as sc
from adaptive_sph_amd.workloads import dam_break_params
from tests.oracle_harness import load_oracle
olib = load_oracle(); plib = ffi.load_product()
scn = sc.SceneConfig(sc.SceneBoundary("box", 2.0, 2.0),
                     [sc.SceneFluidBlock([-0.95, -0.5], [0.55, 1.4], 0.03, 0.93, [0.5, 0]),
                      sc.SceneFluidBlock([-0.40, -0.5], [0.55, 1.4], 0.06, 0.93, [-0.5, 0])])
pos, mass, vel = sc.init_particles(scn)
planes = sc.boundary_planes(scn.boundary)
p = dam_break_params(hybrid_dfsph_max_avg_density_error=0.0, hybrid_dfsph_max_avg_divergence_error=0.0, max_iters=3).to_ffi()
g = ffi.Context(plib, len(mass), planes); o = ffi.Context(olib, len(mass), planes)
g.upload(mass, pos, vel); o.upload(mass, pos, vel)
def rel(a, b):
    a = a.astype(np.float64); b = b.astype(np.float64); s = np.abs(b).max()
    return np.abs(a - b).max() / (s if s > 0 else 1)
for s in range(8):
    sg, so = g.step(p), o.step(p)
    print(s, " ".join(f"{f}={rel(g.download(f), o.download(f)):.2e}" for f in ["density", "aii", "ppe_source_term", "pressure", "pressure_accel", "velocity", "position"]))
