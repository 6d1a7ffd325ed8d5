"""Evidence for retiring SURVEY.md section 8d's configs[3] scene (VERDICT r4 missing 2 / next 1; advisor r4).

The surveyed scene -- box 4 x 2, ONE block pos [-1.9995, -0.9995] size [1.4143, 1.4143] spacing 1/2048 -> 2896 x 2896 = 8 386 816
particles, configs[1]'s parameters (`scene.dam_break_8m_spec`) -- is stepped from rest for steps 0..5 on the CPU oracle (the reference's
algorithm restated, oracle/) AND on the device, for max_dt = 0.001, 0.0005 and 0.00025; after every step: the largest particle speed,
dt, the two solves' iteration counts, the lowest particle (the floor is y = -1), the largest density.

usage: python scripts/gpu_config3_divergence.py [--side device|oracle|both] [--max-dt a,b,c] [--steps 6] [--scene dam_break_8m_spec|dam_break_8m|dam_break_1m]
Writes gpurun_out/r5_config3_divergence_<side>.txt (copied to profiles/ by hand).  The oracle needs ~1.5 min per 200-iteration solve
of 8.4 M particles on 128 cores; --side lets the two halves run on different machines (the oracle here, the device on the GPU box)."""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from adaptive_sph_amd import ffi, scene as sc  # noqa: E402
from adaptive_sph_amd.workloads import dam_break_params  # noqa: E402


def arg(name, default):
    return sys.argv[sys.argv.index(name) + 1] if name in sys.argv else default


def main():
    side = arg("--side", "both")
    max_dts = [float(x) for x in arg("--max-dt", "0.001,0.0005,0.00025").split(",")]
    steps = int(arg("--steps", "6"))
    scene_name = arg("--scene", "dam_break_8m_spec")
    scn = getattr(sc, scene_name)()
    pos, mass, vel = sc.init_particles(scn)
    planes = sc.boundary_planes(scn.boundary)
    out = Path(__file__).resolve().parents[1] / "gpurun_out"
    out.mkdir(exist_ok=True)
    lines = []

    def say(s):
        print(s, flush=True)
        lines.append(s)
        (out / f"r5_config3_divergence_{scene_name}_{side}.txt").write_text("\n".join(lines) + "\n")

    libs = []
    if side in ("device", "both"):
        libs.append(("device", ffi.load_product()))
    if side in ("oracle", "both"):
        from tests.oracle_harness import load_oracle   # test infrastructure: the checker
        libs.append(("oracle", load_oracle()))
    say(f"scene {scene_name}: {len(mass)} particles, blocks {[sc.block_dims(b) for b in scn.blocks]}, spacing {scn.blocks[0].spacing}, box {scn.boundary.width} x {scn.boundary.height}")
    for max_dt in max_dts:
        p = dam_break_params(max_dt=max_dt).to_ffi()
        for name, lib in libs:
            ctx = ffi.Context(lib, len(mass), planes)
            ctx.upload(mass, pos, vel)
            say(f"== max_dt {max_dt:g}, {name}")
            for s in range(steps):
                t0 = time.perf_counter()
                try:
                    st = ctx.step(p)
                except ffi.SphError as e:
                    say(f"step {s}: {e}")
                    break
                v = ctx.download("velocity").astype(np.float64)
                x = ctx.download("position")
                rho = ctx.download("density")
                say(f"step {s}: dt {float(st.dt):.6g} div {int(st.div_solver.iters) + 1} dens {int(st.density_solver.iters) + 1} "
                    f"vmax {float(np.sqrt((v ** 2).sum(axis=1)).max()):.6g} ymin {float(x[:, 1].min()):.6g} below the floor {int((x[:, 1] < -1.0).sum())} "
                    f"rho_max {float(rho.max()):.6g}   ({time.perf_counter() - t0:.1f} s)")
            ctx.close()


if __name__ == "__main__":
    main()
