"""Randomised check of the build queued ahead and of its incremental cell sort (free-running steps): random blocks of one or two
particle sizes thrown at each other, stepped by three contexts side by side -- the product's defaults, SPH_INC_SORT=0 (radix sort
queued ahead) and SPH_AHEAD_BUILD=0 (build at the step's start) -- which must agree in every field, bit for bit, at every step.
usage: gpu_fuzz_incsort.py [first_seed] [n_seeds] [steps] [spacing divisor: 4 = sixteen times the particles]"""
import os
import sys
import numpy as np
sys.path.insert(0, ".")
from adaptive_sph_amd import ffi, scene as sc
from adaptive_sph_amd.workloads import dam_break_params

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 10
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
finer = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0
lib = ffi.load_product()
bad = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    d = float(rng.choice([1 / 48, 1 / 64, 1 / 96])) / finer
    two = rng.random() < 0.5
    blocks = []
    x = -1.8
    for b in range(int(rng.integers(1, 4))):
        sp = d * (4 if (two and b == 1) else 1)
        w, h = float(rng.uniform(0.3, 0.9)), float(rng.uniform(0.3, 1.2))
        blocks.append(sc.SceneFluidBlock([x, float(rng.uniform(-0.95, -0.2))], [w, h], sp, 0.93, [float(rng.uniform(-5, 5)), float(rng.uniform(-3, 3))]))
        x += w + float(rng.uniform(0.02, 0.3))
    scn = sc.SceneConfig(sc.SceneBoundary("box", 4.0, 2.0), blocks)
    pos, mass, vel = sc.init_particles(scn)
    solver = str(rng.choice(["HybridDFSPH", "IISPH", "OnlyDivergence"]))
    P = dam_break_params(pressure_solver_method=solver, max_dt=float(rng.choice([0.0005, 0.001, 0.002])))
    planes = sc.boundary_planes(scn.boundary, P.init_boundary_handler)
    p = P.to_ffi()
    ctxs = []
    for env in ({}, {"SPH_INC_SORT": "0"}, {"SPH_AHEAD_BUILD": "0"}):
        os.environ.update(env)
        g = ffi.Context(lib, len(mass), planes)   # (the switches are read at sph_create)
        for k in env:
            del os.environ[k]
        g.upload(mass, pos, vel)
        g.profile_enable(1)
        ctxs.append(g)
    ok, fail = True, None
    try:
        for s in range(steps):
            sts = [g.step(p) for g in ctxs]
            ref = None
            for gi, g in enumerate(ctxs):
                f = {k: g.download(k) for k in ("position", "velocity", "pressure", "density", "cell_index", "neighbor_count")}
                f["its"] = np.array([int(sts[gi].div_solver.iters), int(sts[gi].density_solver.iters)])
                if ref is None:
                    ref = f
                else:
                    for k in f:
                        if not np.array_equal(f[k], ref[k]):
                            ok, fail = False, (s, gi, k)
            if not ok:
                break
    except ffi.SphError as e:   # (a scene that blows up: the same guard must fire everywhere -- not this script's subject)
        fail = ("guard", str(e)[:80])
    prof = ctxs[0].profile_get()
    inc, rad = prof.get("inc_reorder", (0, 0))[0], prof.get("sort_scatter", (0, 0))[0]
    print(f"seed {seed}: n {len(mass)} {'two sizes' if two else 'uniform'} {solver} steps {steps}: {'ok' if ok else 'MISMATCH'} {fail if fail else ''} "
          f"(merges {inc}, radix scatters {rad})", flush=True)
    bad += 0 if ok else 1
    for g in ctxs:
        g.close()
print("mismatching scenes:", bad)
sys.exit(1 if bad else 0)
