"""where does the one-sided "normal" count of the bench window's first flip come from?

configs[1] (1 048 576 particles from rest), steps 0..3 free-running, then step 4 with the divergence solve's count FORCED to 4 --
the iteration the stop rule (simulation.rs:1453-1479) judged differently on device and oracle in round 4 (normal 97 724 device vs
77 272 oracle, residual sums equal).  BOTH sides of every pair start step 4 from the same state (the oracle's after step 3), uploaded
in one of four orders (a first version of this script sorted by the cells of step 0: 12 % of the particles have changed their cell by
step 4 -- the column settles by a third of a spacing -- so that was not the visiting order any more):

    order = host       the scene as add_fluid_block makes it (x outer, y inner: ascending index = up a COLUMN of the lattice)
    order = device     the device's visiting order for that very state: perm = argsort(cell_index, stable) of the step itself -- the
                       oracle's ascending-j neighbour sum is then taken in the order the device's sweeps visit (rows of cells bottom
                       to top, index ascending); the device's own cell sort is the identity on it
    order = reversed   host order backwards
    order = row-major  ascending index = along a row of the lattice
    policy = fast      the product's default arithmetic (v_rsq / v_rcp, truncated-power spline, fma)
    policy = exact     SPH_HIP_EXACT=1: IEEE division / sqrt in the reference's operation order

If summation ORIENTATION is what moves 20 000 rounding-level pressures to one side, the sorted/exact pair agrees (nearly) count for
count and the host/exact pair shows the round-4 gap; if an arithmetic difference in sweep B's near-zero branch is the cause, the gap
survives the reordering.

usage: python scripts/gpu_normal_count.py [--window]     (--window: the host and device orders then run on, free, through step 24)
Writes gpurun_out/r5_normal_count.txt."""
import os
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from adaptive_sph_amd import ffi, scene as sc  # noqa: E402
from adaptive_sph_amd.workloads import dam_break_params  # noqa: E402
from tests.oracle_harness import load_oracle  # noqa: E402  (test infrastructure: the checker)

OUT = Path(__file__).resolve().parents[1] / "gpurun_out"
OUT.mkdir(exist_ok=True)
lines = []


def say(s):
    print(s, flush=True)
    lines.append(s)
    (OUT / "r5_normal_count.txt").write_text("\n".join(lines) + "\n")


def stats(s):
    return (f"iters {int(s.iters)} normal {int(s.normal_count)} negative {int(s.negative_count)} singular {int(s.singular_count)} "
            f"avg {float(s.avg_error):.7g} sum {float(s.avg_error) * int(s.normal_count):.7g} max {float(s.max_error):.7g}")


def main():
    window = "--window" in sys.argv
    scn = sc.dam_break_1m()
    pos, mass, vel = sc.init_particles(scn)
    planes = sc.boundary_planes(scn.boundary)
    plib, olib = ffi.load_product(), load_oracle()
    free = dam_break_params().to_ffi()
    n = len(mass)

    def forced(k):
        return dam_break_params(hybrid_dfsph_max_avg_density_error=0.0, hybrid_dfsph_max_avg_divergence_error=0.0, max_iters=k).to_ffi()

    def make(lib, m, x, v):
        c = ffi.Context(lib, len(m), planes)
        c.upload(m, x, v)
        return c

    # ---- the state step 4 starts from: steps 0..3 free-running on the ORACLE (host order); device beside it for the record
    o = make(olib, mass, pos, vel)
    g = make(plib, mass, pos, vel)
    for s in range(4):
        sg, so = g.step(free), o.step(free)
        say(f"step {s}: div {int(sg.div_solver.iters)}/{int(so.div_solver.iters)} dens {int(sg.density_solver.iters)}/{int(so.density_solver.iters)} "
            f"dt {float(sg.dt):.9g}/{float(so.dt):.9g}   (device/oracle)")
    x3, v3 = o.download("position"), o.download("velocity")
    say(f"state after step 3 (the oracle's; both sides of every pair below start step 4 from it): max |x_dev - x_orc| {float(np.abs(g.download('position') - x3).max()):.3g}, "
        f"max |v_dev - v_orc| {float(np.abs(g.download('velocity') - v3).max()):.3g}")
    g.close()
    o.close()

    for policy in ("fast", "exact"):
        if policy == "exact":
            os.environ["SPH_HIP_EXACT"] = "1"    # read by sph_create
        else:
            os.environ.pop("SPH_HIP_EXACT", None)
        # the device's visiting order for that state: its stable cell sort of the host-order upload
        probe = make(plib, mass, x3, v3)
        sp_ = probe.step(forced(4))
        cell4 = probe.download("cell_index")
        probe.close()
        perm4 = np.argsort(cell4, kind="stable")
        orders = {"host": np.arange(n), "device": perm4,
                  "reversed": np.arange(n)[::-1].copy(),                                       # ascending index = DOWN a column, right to left
                  "row-major": np.lexsort((pos[:, 0], pos[:, 1]))}                             # ascending index = along a ROW of the lattice
        say(f"== policy {policy}: step 4 from the oracle's state, divergence AND density solve forced to 4 iterations")
        for name, idx in orders.items():
            g = make(plib, mass[idx], x3[idx], v3[idx])
            o = make(olib, mass[idx], x3[idx], v3[idx])
            sg, so = g.step(forced(4)), o.step(forced(4))
            say(f"  order {name}:")
            say(f"    device: {stats(sg.div_solver)}")
            say(f"    oracle: {stats(so.div_solver)}")
            pg, po = g.download("pressure"), o.download("pressure")
            rg, ro = g.download("density"), o.download("density")
            ag, ao = g.download("pressure_accel"), o.download("pressure_accel")
            say(f"    after the step: pressure bit-equal on {float((pg == po).mean()):.6f} of the particles, a^p on {float((ag == ao).all(axis=1).mean()):.6f}, "
                f"density on {float((rg == ro).mean()):.6f}; max |rho diff| {float(np.abs(rg.astype(np.float64) - ro).max()):.3g}; "
                f"density solve normal {int(sg.density_solver.normal_count)}/{int(so.density_solver.normal_count)}")
            if window and name in ("host", "device"):
                first, rows = None, []
                for s in range(5, 25):
                    sg, so = g.step(free), o.step(free)
                    d = (int(sg.div_solver.iters), int(so.div_solver.iters), int(sg.density_solver.iters), int(so.density_solver.iters))
                    rows.append(d)
                    if first is None and (d[0] != d[1] or d[2] != d[3]):
                        first = s
                rg, ro = g.download("density").astype(np.float64), o.download("density").astype(np.float64)
                it = np.array(rows)
                say(f"    free-running steps 5..24 behind it: first step with different counts: {first}; mean |count difference| "
                    f"{float((np.abs(it[:, 0] - it[:, 1]) + np.abs(it[:, 2] - it[:, 3])).mean() / 2):.3f}; p99 |rho diff| {float(np.quantile(np.abs(rg - ro), 0.99)):.3g}, "
                    f"median {float(np.median(np.abs(rg - ro))):.3g}")
                say("    div  device/oracle: " + " ".join(f"{a}/{b}" for a, b, _, _ in rows))
                say("    dens device/oracle: " + " ".join(f"{c}/{d}" for _, _, c, d in rows))
            g.close()
            o.close()


if __name__ == "__main__":
    main()
