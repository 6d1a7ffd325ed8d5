"""The cell sort a step queues ahead for the next one (sph_step.hip: queue_ahead_build) as a MERGE of the particles that stay in
their cells with the few that do not (sph_sort.hip: incremental_cell_sort) against the stable radix sort it replaces
(SPH_INC_SORT=0): the same keys, permutation and cell ranges, so the same order of the arrays -- every field, every iteration
statistic and every neighbour LIST (order included: it is the order of the sorted array) bit for bit, over free-running steps."""
import numpy as np
import pytest

from adaptive_sph_amd import ffi, scene as sc
from adaptive_sph_amd.workloads import dam_break_params

pytestmark = pytest.mark.gpu


def scene_of(kind):
    if kind == "column":   # the headline's shape: a column at rest that starts to collapse (few movers per step)
        return sc.dam_break_small(128, 96, 1 / 64), {}
    if kind == "thrown":   # a block thrown through the box: many movers per step, a bounding box (and a grid origin) that travels
        d = 1 / 64
        return sc.SceneConfig(sc.SceneBoundary("box", 4.0, 2.0), [sc.SceneFluidBlock([-1.6, -0.6], [96 * d, 64 * d], d, 0.93, [3.0, 1.0])]), dict(max_dt=0.0005)
    # two blocks that run into each other: cells that receive several movers in one step, cells that empty
    d = 1 / 64
    return sc.SceneConfig(sc.SceneBoundary("box", 4.0, 2.0), [sc.SceneFluidBlock([-1.2, -0.9], [48 * d, 80 * d], d, 0.93, [4.0, 0.0]),
                                                               sc.SceneFluidBlock([-0.2, -0.9], [48 * d, 80 * d], d, 0.93, [-4.0, 0.5])]), dict(max_dt=0.0005)


@pytest.mark.parametrize("kind,solver,steps", [("column", "HybridDFSPH", 40), ("thrown", "HybridDFSPH", 30), ("collide", "IISPH", 30), ("thrown", "OnlyDivergence", 30)])
def test_incremental_cell_sort_is_the_radix_sort(lab_lib, monkeypatch, kind, solver, steps):
    scn, over = scene_of(kind)
    pos, mass, vel = sc.init_particles(scn)
    P = dam_break_params(pressure_solver_method=solver, **over)
    planes = sc.boundary_planes(scn.boundary, P.init_boundary_handler)
    p = P.to_ffi()
    out = {}
    for form in ("merge", "radix"):
        if form == "radix":
            monkeypatch.setenv("SPH_INC_SORT", "0")
        g = ffi.Context(lab_lib, len(mass), planes)   # (the switches are read at sph_create)
        if form == "radix":
            monkeypatch.delenv("SPH_INC_SORT")
        g.upload(mass, pos, vel)
        g.profile_enable(1)
        its, fields = [], []
        for s in range(steps):
            st = g.step(p)
            its.append((int(st.div_solver.iters), int(st.density_solver.iters), int(st.density_solver.normal_count),
                        np.float32(st.density_solver.avg_error).view(np.uint32).item(), np.float32(st.dt).view(np.uint32).item()))
            f = {k: g.download(k) for k in ("position", "velocity", "pressure", "density", "cell_index")}
            if s % 5 == 4 or s == steps - 1:
                f["nl_offsets"], f["nl_indices"] = g.download_neighbors()
            fields.append(f)
        prof = g.profile_get()
        out[form] = (its, fields, prof)
        g.close()
    # the forms really differ in what they launched: all but the first build were queued ahead
    assert out["merge"][2].get("inc_reorder", (0, 0))[0] >= steps - 2, out["merge"][2]
    assert "inc_reorder" not in out["radix"][2] and out["radix"][2]["sort_scatter"][0] >= 2 * (steps - 1)
    assert out["merge"][2].get("sort_scatter", (0, 0))[0] <= 4   # (the first build of the run, and nothing else)
    assert out["merge"][0] == out["radix"][0]
    for s, (fa, fb) in enumerate(zip(out["merge"][1], out["radix"][1])):
        for k in fa:
            assert np.array_equal(fa[k], fb[k]), (s, k)
    if kind != "column":   # these scenes move: the bounding box (hence the grid origin) travelled by more than a cell along the way
        x0, x1 = out["merge"][1][0]["position"], out["merge"][1][-1]["position"]
        assert np.abs(x1 - x0).max() > (1 / 64) * 2.2


def two_sizes_scene():
    fine = 0.02
    return sc.SceneConfig(sc.SceneBoundary("box", 3.0, 3.0),
                          [sc.SceneFluidBlock([-0.95, -0.5], [0.55, 1.4], fine, 0.93, [0.5, 0]),
                           sc.SceneFluidBlock([-0.40 + 0.3 * fine * 4, -0.5], [0.7, 1.4], fine * 4, 0.93, [-0.5, 0])])


LEVEL = dict(level_estimation_method="EmptyAngle", maximum_surface_distance=0.2, particle_radius_fine=0.004, particle_radius_base=0.02)


@pytest.mark.parametrize("kind,solver,level", [("column", "HybridDFSPH", None), ("two_sizes", "HybridDFSPH", None), ("two_sizes", "IISPH", None),
                                               ("column", "HybridDFSPH", dict(LEVEL)), ("two_sizes", "IISPH", dict(LEVEL)),
                                               ("column", "IISPH", dict(LEVEL, level_estimation_after_advection=True))])
def test_build_queued_ahead_is_the_build_at_the_step_start(lab_lib, monkeypatch, kind, solver, level):
    """The next step's neighbour build queued behind the integrating tail on a PREDICTED grid (another origin, a margin around the
    bounding box; in a multi-resolution scene other tiles, hence other stencil widths for the same lists) against the build at the
    start of the step (SPH_AHEAD_BUILD=0): every field, every iteration statistic and every neighbour list bit for bit.  With the
    level estimation on (before or after advection) the build moves the smoothed level values: it is queued behind the smoothing."""
    scn = sc.dam_break_small(128, 96, 1 / 64) if kind == "column" else two_sizes_scene()
    pos, mass, vel = sc.init_particles(scn)
    P = dam_break_params(pressure_solver_method=solver, **(level or {}))
    planes = sc.boundary_planes(scn.boundary, P.init_boundary_handler)
    p = P.to_ffi()
    steps = 25
    out = {}
    for form in ("ahead", "at-start"):
        if form == "at-start":
            monkeypatch.setenv("SPH_AHEAD_BUILD", "0")
        g = ffi.Context(lab_lib, len(mass), planes)
        if form == "at-start":
            monkeypatch.delenv("SPH_AHEAD_BUILD")
        g.upload(mass, pos, vel)
        g.profile_enable(1)
        its, fields = [], []
        for s in range(steps):
            st = g.step(p)
            its.append((int(st.div_solver.iters), int(st.density_solver.iters), int(st.density_solver.normal_count),
                        np.float32(st.density_solver.avg_error).view(np.uint32).item(), np.float32(st.dt).view(np.uint32).item()))
            f = {k: g.download(k) for k in ("position", "velocity", "pressure", "density", "cell_index", "neighbor_count")}
            if level:
                f.update({k: g.download(k) for k in ("level_estimation", "level_old", "flag_is_fluid_surface")})
            if s % 6 == 5 or s == steps - 1:
                f["nl_offsets"], f["nl_indices"] = g.download_neighbors()
            fields.append(f)
        out[form] = (its, fields, g.profile_get())
        g.close()
    assert out["ahead"][2].get("inc_reorder", (0, 0))[0] >= steps - 2, out["ahead"][2]     # queued ahead, as a merge ...
    assert out["ahead"][2].get("sort_scatter", (0, 0))[0] <= 4                               # ... and adopted: no sort at a step's start
    assert "inc_reorder" not in out["at-start"][2]
    if kind == "two_sizes":
        assert out["ahead"][2]["tile_hmax"][0] >= steps - 1
    assert out["ahead"][0] == out["at-start"][0]
    for s, (fa, fb) in enumerate(zip(out["ahead"][1], out["at-start"][1])):
        for k in fa:
            assert np.array_equal(fa[k], fb[k]), (s, k)


@pytest.mark.parametrize("k,two_sizes", [(2, False), (3, False), (2, True)])
def test_slab_ranks_sort_by_merging_their_arrivals(lab_lib, monkeypatch, k, two_sizes):
    """A slab rank behind its fused refresh: last step's sorted slots (some of them left: last step's ghosts, migrants), then this
    step's arrivals (migrants, the new ghost layer) unsorted behind them.  The merge with the arrivals as movers against the radix
    sort (SPH_INC_SORT=0) on a loopback group whose fluid is pushed across the cuts: every rank's arrays bit for bit."""
    from adaptive_sph_amd import distributed as D
    if two_sizes:
        scn = two_sizes_scene()
        pos, mass, vel = sc.init_particles(scn)
    else:
        scn = sc.dam_break_small(96, 48, 1 / 48)
        pos, mass, vel = sc.init_particles(scn)
        vel = vel.copy()
        vel[:, 0] = 0.8
    P = dam_break_params(max_iters=6)
    planes = sc.boundary_planes(scn.boundary, P.init_boundary_handler)
    p = P.to_ffi()
    out = {}
    for form in ("merge", "radix"):
        if form == "radix":
            monkeypatch.setenv("SPH_INC_SORT", "0")
        grp = D.make_loopback_group(lab_lib, pos, mass, vel, planes, k)
        if form == "radix":
            monkeypatch.delenv("SPH_INC_SORT")
        for c in grp:
            c.profile_enable(1)
        n0 = [c.n for c in grp]
        its = []
        for s in range(25):
            sts = ffi.group_step(grp, p)
            its.append(tuple((int(st.div_solver.iters), int(st.density_solver.iters), int(st.density_solver.normal_count)) for st in sts))
        fields = [{f: c.download(f) for f in ("particle_id", "position", "velocity", "pressure", "density", "neighbor_count", "cell_index")} for c in grp]
        out[form] = (its, fields, [c.profile_get() for c in grp], n0, [c.n for c in grp])
        for c in grp:
            c.close()
    # (two particle sizes in a small box: the ghost layer of a coarse cut is a large part of a rank's array -- above a third of it the
    #  rank keeps the radix sort)
    assert all(pr.get("inc_place", (0, 0))[0] >= (3 if two_sizes else 20) for pr in out["merge"][2]), out["merge"][2][0]
    assert all("inc_place" not in pr for pr in out["radix"][2])
    if not two_sizes:
        assert out["merge"][3] != out["merge"][4]   # particles did migrate
    assert out["merge"][0] == out["radix"][0] and out["merge"][4] == out["radix"][4]
    for r, (fa, fb) in enumerate(zip(out["merge"][1], out["radix"][1])):
        for f in fa:
            assert np.array_equal(fa[f], fb[f]), (r, f)
