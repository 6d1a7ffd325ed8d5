#!/usr/bin/env python3
"""Regenerate tests/golden/*.npz.

The reference (Rust) cannot be built or imported in this environment and holds no field-level golden
data, so the committed field fixtures come from the CPU oracle -- which is itself
pinned to the reference's known-answer tests by tests/test_oracle_golden.py.  Each fixture stores the
scene inputs and the oracle's state after `steps` steps with the Jacobi iteration counts FORCED
(tolerances 0, max_iters = K => exactly K+1 iterations per solve) so that a free-running stop decision
cannot flip the comparison.

    python tests/golden/make_fixtures.py
"""
import sys
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(REPO))

from adaptive_sph_amd import ffi, scene as sc  # noqa: E402
from adaptive_sph_amd.workloads import dam_break_params  # noqa: E402
from tests.oracle_harness import load_oracle  # noqa: E402

FIELDS = ["h2", "cell_index", "neighbor_count", "lambda_sum", "lambda_grad_sum", "density", "constant_field", "aii",
          "ppe_source_term", "pressure", "pressure_accel", "velocity", "position", "density_error"]


def forced(**kw):
    return dam_break_params(hybrid_dfsph_max_avg_density_error=0.0, hybrid_dfsph_max_avg_divergence_error=0.0,
                            iisph_max_avg_density_error=0.0, **kw)


CASES = {
    # name: (scene, params, steps)
    "dam32_hybrid_k4": (sc.dam_break_small(32, 32, 1 / 32), forced(max_iters=4), 4),
    "dam32_iisph_k5": (sc.dam_break_small(32, 32, 1 / 32), forced(max_iters=5, pressure_solver_method="IISPH"), 3),
    "ratio2to1_hybrid_k3": (sc.SceneConfig(sc.SceneBoundary("box", 2.0, 2.0),
                                           [sc.SceneFluidBlock([-0.95, -0.5], [0.55, 1.4], 0.03, 0.93, [0, 0]),
                                            sc.SceneFluidBlock([0.4, -0.5], [0.55, 1.4], 0.06, 0.93, [0, 0])]),
                            forced(max_iters=3), 3),
}


def main():
    lib = load_oracle()
    for name, (scn, params, steps) in CASES.items():
        pos, mass, vel = sc.init_particles(scn)
        ctx = ffi.Context(lib, len(mass), sc.boundary_planes(scn.boundary))
        ctx.upload(mass, pos, vel)
        p = params.to_ffi()
        for _ in range(steps):
            st = ctx.step(p)
        out = {"in_position": pos, "in_mass": mass, "in_velocity": vel, "steps": np.int32(steps), "dt": np.float32(st.dt),
               "time": np.float32(st.time)}
        for f in FIELDS:
            out[f] = ctx.download(f)
        off, idx = ctx.download_neighbors()
        out["nb_offsets"], out["nb_indices"] = off, idx
        np.savez_compressed(REPO / "tests" / "golden" / f"{name}.npz", **out)
        print(name, len(mass), "particles", steps, "steps")


if __name__ == "__main__":
    main()
