"""configs[2]'s two blocks moved against each other (scene.dam_break_1m_adaptive_contact): what does the recipe do with a 4:1 interface
at rest, free-running, as a function of the gap between the blocks?  Device (and, with --oracle, the CPU oracle) step by step:
iterations, dt, largest speed, density range.  usage: python scripts/gpu_config2_contact.py [steps=40] [gaps in coarse spacings ...] [--oracle]"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from adaptive_sph_amd import ffi, scene as sc  # noqa: E402
from adaptive_sph_amd.workloads import dam_break_params  # noqa: E402


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("-")]
    steps = int(args[0]) if args else 40
    gaps = [float(a) for a in args[1:]] or [1.0]
    if "--oracle" in sys.argv:
        from tests.oracle_harness import load_oracle
        lib = load_oracle()
    else:
        lib = ffi.load_product()
    for gap in gaps:
        scn = sc.dam_break_1m_adaptive_contact(gap * 0.00390625)
        P = dam_break_params()
        pos, mass, vel = sc.init_particles(scn)
        ctx = ffi.Context(lib, len(mass), sc.boundary_planes(scn.boundary))
        ctx.upload(mass, pos, vel)
        p = P.to_ffi()
        print(f"## gap {gap} coarse spacings, {len(mass)} particles", flush=True)
        for s in range(steps):
            try:
                st = ctx.step(p)
            except Exception as e:  # noqa: BLE001
                print(f"step {s}: {e}", flush=True)
                break
            v = ctx.download("velocity").astype(np.float64)
            rho = ctx.download("density")
            print(f"step {s} div {int(st.div_solver.iters) + 1} dens {int(st.density_solver.iters) + 1} dt {float(st.dt):.4g} "
                  f"vmax {np.sqrt((v ** 2).sum(axis=1)).max():.4g} rho [{rho.min():.3f}, {rho.max():.3f}]", flush=True)
        ctx.close()


if __name__ == "__main__":
    main()
