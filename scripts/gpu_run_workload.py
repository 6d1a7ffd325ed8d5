"""Step one BASELINE workload on the product library and print ms/step (fault hunting / quick timings).
usage: python scripts/gpu_run_workload.py <workload> <warmup> <steps> [--oracle] [--speed] [key=value overrides ...]
--oracle: the CPU oracle instead of the product library (test infrastructure; what does the reference's algorithm do on this scene?)
--speed: also print the largest particle speed and the bounding box after every step (one download per step)"""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from adaptive_sph_amd import ffi, scene as sc  # noqa: E402
from adaptive_sph_amd.workloads import WORKLOADS  # noqa: E402


def main():
    name, warmup, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    over = {}
    flags = [a for a in sys.argv[4:] if a.startswith("-")]
    for kv in [a for a in sys.argv[4:] if not a.startswith("-")]:
        k, v = kv.split("=", 1)
        try:
            over[k] = int(v)
        except ValueError:
            try:
                over[k] = float(v)
            except ValueError:
                over[k] = {"True": True, "False": False}.get(v, v)
    scene_f, params_f, desc = WORKLOADS[name]
    scn, P = scene_f(), params_f(**over)
    pos, mass, vel = sc.init_particles(scn)
    if "--oracle" in flags:
        from tests.oracle_harness import load_oracle
        plib = load_oracle()
    else:
        plib = ffi.load_product()

    def extra():
        if "--speed" not in flags:
            return ""
        import numpy as np
        v = ctx.download("velocity")
        x = ctx.download("position")
        return f" vmax {float(np.sqrt((v.astype(np.float64) ** 2).sum(axis=1)).max()):.5g} box [{x[:, 0].min():.4g}, {x[:, 0].max():.4g}] x [{x[:, 1].min():.4g}, {x[:, 1].max():.4g}]"

    ctx = ffi.Context(plib, len(mass), sc.boundary_planes(scn.boundary, P.init_boundary_handler))
    ctx.upload(mass, pos, vel)
    p = P.to_ffi()
    for i in range(warmup):
        st = ctx.step(p)
        print(f"warm {i} div {int(st.div_solver.iters) + 1} dens {int(st.density_solver.iters) + 1} dt {float(st.dt):.6g}" + extra(), flush=True)
    t0 = time.perf_counter()
    for i in range(steps):
        st = ctx.step(p)
        if "-v" in flags or steps <= 40:
            print(f"step {warmup + i} div {int(st.div_solver.iters) + 1} dens {int(st.density_solver.iters) + 1} dt {float(st.dt):.6g}" + extra(), flush=True)
    dt = time.perf_counter() - t0
    print(f"{name}: {len(mass)} particles, {dt * 1e3 / max(steps, 1):.4f} ms/step over {steps} steps after {warmup}", flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
