"""Evolving scenes with the device re-synchronised to the oracle's state before every step: identical inputs each step, so
cell indices, neighbour counts and neighbour SETS must stay bit-exact while the fluid collapses, splashes and hits the walls.
usage: gpu_follow.py [first_seed] [n_seeds] [steps] [--level]"""
import sys
import numpy as np
sys.path.insert(0, ".")
import torch  # noqa: F401
from adaptive_sph_amd import ffi, scene as sc
from adaptive_sph_amd.workloads import dam_break_params, default_params
from tests.oracle_harness import load_oracle, csr_sets, quadtree_scene

level = "--level" in sys.argv
after = "--after" in sys.argv
sys.argv = [a for a in sys.argv if not a.startswith("--")]
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 6
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 60
olib, glib = load_oracle(), ffi.load_product()
planes = sc.boundary_planes(sc.SceneBoundary("box", 4.0, 2.0), "AnalyticOverestimate")
bad = 0
for seed in range(first, first + count):
    pos, mass, vel, info = quadtree_scene(seed)
    if len(mass) > 30000:
        continue
    vel = (vel * 20).astype(np.float32)        # violent: ~1 m/s random velocities on top of gravity
    p = (default_params(merging=False, sharing=False, splitting=False, max_dt=0.002, max_iters=30, level_estimation_after_advection=after) if level else dam_break_params(max_dt=0.002, max_iters=30)).to_ffi()
    g, o = ffi.Context(glib, len(mass), planes), ffi.Context(olib, len(mass), planes)
    o.upload(mass, pos, vel)
    msgs = []
    for step in range(steps):
        g.upload(o.download("mass"), o.download("position"), o.download("velocity"))
        try:
            so = o.step(p)
        except ffi.SphError as e:
            try:
                g.step(p)
                msgs.append(f"step {step}: oracle {e.status}, device ok")
            except ffi.SphError as e2:
                if e2.status != e.status:
                    msgs.append(f"step {step}: oracle {e.status}, device {e2.status}")
            break
        sg = g.step(p)
        if sg.dt != so.dt:
            msgs.append(f"step {step}: dt {sg.dt} {so.dt}")
        for f in ("h2", "neighbor_count", "cell_index", "lambda_sum", "lambda_grad_sum"):
            if not np.array_equal(g.download(f), o.download(f)):
                msgs.append(f"step {step}: {f} differs in {(g.download(f) != o.download(f)).sum()} places")
        go, gi = g.download_neighbors(); oo, oi = o.download_neighbors()
        if not np.array_equal(go, oo) or any(not np.array_equal(a, b) for a, b in zip(csr_sets(go, gi), csr_sets(oo, oi))):
            msgs.append(f"step {step}: neighbour sets differ")
        if level:
            fa, fb = g.download("flag_is_fluid_surface"), o.download("flag_is_fluid_surface")
            if not np.array_equal(fa, fb):
                msgs.append(f"step {step}: {(fa != fb).sum()} surface flags differ")
            a, b = g.download("level_estimation"), o.download("level_estimation")
            if not np.array_equal(np.isnan(a), np.isnan(b)):
                msgs.append(f"step {step}: level NaN pattern differs")
            elif np.nanmax(np.abs(a - b)) > 1e-4 * max(np.nanmax(np.abs(b)), 1e-30):
                msgs.append(f"step {step}: level differs by {np.nanmax(np.abs(a - b)):.2e}")
        if len(msgs) > 4:
            break
    x = o.download("position")
    print(f"seed {seed}: n={len(mass)} {info} steps={step + 1} bbox x[{x[:,0].min():.2f},{x[:,0].max():.2f}] y[{x[:,1].min():.2f},{x[:,1].max():.2f}] "
          f"nmax={int(o.download('neighbor_count').max())} " + ("OK" if not msgs else "MISMATCH " + "; ".join(msgs[:5])), flush=True)
    bad += bool(msgs)
print("BAD" if bad else "ALL OK")
