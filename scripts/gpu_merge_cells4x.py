"""What the build queued ahead (the incremental merge: count, scan, place + reorder) costs on a cell table FOUR times as large -- the sorting grid
half-support cells would need.  configs[1]'s particles with a quarter of the mass: h halves, the cell
halves, the positions and the counts stay (the physics is beside the point: the column is under-dense and falls freely, so the movers per step
are FEWER than a real half-cell grid would see -- a lower bound for the merge).  usage: python scripts/gpu_merge_cells4x.py [steps=40]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from adaptive_sph_amd import ffi, scene as sc  # noqa: E402
from adaptive_sph_amd.workloads import dam_break_params  # noqa: E402


def run(fill, steps):
    scn = sc.dam_break_1m()
    scn.blocks[0].volume_fill_ratio = fill
    pos, mass, vel = sc.init_particles(scn)
    lib = ffi.load_product()
    ctx = ffi.Context(lib, len(mass), sc.boundary_planes(scn.boundary))
    ctx.upload(mass, pos, vel)
    p = dam_break_params().to_ffi()
    for _ in range(10):
        ctx.step(p)
    ctx.profile_reset()
    ctx.profile_enable(1)
    for _ in range(steps):
        ctx.step(p)
    prof = ctx.profile_get()
    ctx.profile_enable(0)
    g = ctx.grid()
    names = ("inc_count", "inc_scan", "inc_reorder", "header_ahead", "density", "rs_hist", "rs_scatter", "reorder", "cell_start")
    row = {k: (prof[k][1] * 1e3 / max(prof[k][0], 1), prof[k][0] / steps) for k in names if k in prof}
    print(f"fill ratio {fill}: grid {g.size_x} x {g.size_y} = {g.size_x * g.size_y} cells for {len(mass)} particles; us per launch (launches per step): "
          + ", ".join(f"{k} {v[0]:.1f} ({v[1]:.2f})" for k, v in row.items()), flush=True)
    ctx.close()


if __name__ == "__main__":
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    run(0.93, steps)
    run(0.93 / 4.0, steps)
