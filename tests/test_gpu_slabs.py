"""Slab decomposition (the multi-GPU algorithm) verified on ONE GPU: k contexts of this process act as ranks
0..k-1 with the loopback transport (sph_group_step); the same driver code runs with the RCCL transport when
every rank is its own process.  Results must match a single context on the same scene."""
import numpy as np
import pytest

from adaptive_sph_amd import distributed as D, ffi, scene as sc
from adaptive_sph_amd.workloads import dam_break_params
from tests.oracle_harness import quadtree_scene

pytestmark = pytest.mark.gpu


def rel_err(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    s = np.abs(b).max()
    return np.abs(a - b).max() / (s if s > 0 else 1.0)


def forced(**kw):
    return dam_break_params(hybrid_dfsph_max_avg_density_error=0.0, hybrid_dfsph_max_avg_divergence_error=0.0,
                            iisph_max_avg_density_error=0.0, **kw)


@pytest.mark.parametrize("k", [2, 3, 4])
def test_loopback_group_matches_single_context(product_lib, k):
    scn = sc.dam_break_small(96, 48, 1 / 48)     # wide column: every slab is several support radii wide
    pos, mass, vel = sc.init_particles(scn)
    vel = vel.copy()
    vel[:, 0] = 0.8                               # push the fluid across the cuts so particles migrate
    planes = sc.boundary_planes(scn.boundary)
    p = forced(max_iters=4).to_ffi()
    single = ffi.Context(product_lib, len(mass), planes)
    single.upload(mass, pos, vel)
    grp = D.make_loopback_group(product_lib, pos, mass, vel, planes, k)
    n0 = [c.n for c in grp]
    assert sum(n0) == len(mass) and min(n0) > 0
    for s in range(25):
        st1 = single.step(p)
        sts = ffi.group_step(grp, p)
        assert all(st.dt == st1.dt for st in sts)
        assert all(st.div_solver.iters == st1.div_solver.iters for st in sts)
        # every rank reports the all-reduced statistics; a particle whose p' sits at the clamp may flip class
        # between two summation orders, hence the small tolerance
        assert abs(int(sts[0].div_solver.normal_count) - int(st1.div_solver.normal_count)) <= 8
        assert len({st.div_solver.normal_count for st in sts}) == 1
    assert sum(c.n for c in grp) == len(mass)                       # nothing lost or duplicated
    ids = np.concatenate([c.download("particle_id") for c in grp])
    assert np.array_equal(np.sort(ids), np.arange(len(mass)))
    assert [c.n for c in grp] != n0                                  # particles did migrate
    for f, tol in (("position", 1e-5), ("velocity", 1e-4), ("density", 1e-5), ("mass", 0.0)):
        got = D.gather_by_id(grp, f, len(mass))
        assert rel_err(got, single.download(f)) <= tol, f
    assert np.array_equal(D.gather_by_id(grp, "neighbor_count", len(mass)), single.download("neighbor_count"))
    # ownership follows the cuts
    cuts = D.slab_cuts(pos[:, 0], k)
    for r, c in enumerate(grp):
        x = c.download("position")[:, 0]
        lo = cuts[r] if r > 0 else -np.inf
        hi = cuts[r + 1] if r + 1 < k else np.inf
        # a particle may sit just outside after the last integrate (it migrates at the start of the next step)
        assert np.all(x > lo - 0.05) and np.all(x < hi + 0.05)


@pytest.mark.parametrize("seed,k", [(0, 2), (13, 2), (32, 2), (33, 2)])   # (two ghost rings of 2 h_max: these scenes are too narrow for 3 slabs)
def test_loopback_group_on_graded_distributions(product_lib, seed, k):
    """Multi-resolution stencils, ghost layers and migration together: graded quadtree distributions (size ratios up to
    32:1, 106k particles at seed 33) drifting across the cuts; scripts/gpu_fuzz_slabs.py runs more seeds."""
    pos, mass, vel, info = quadtree_scene(seed)
    vel = vel.copy()
    vel[:, 0] += 0.5
    planes = sc.boundary_planes(sc.SceneBoundary("box", 4.0, 2.0), "AnalyticOverestimate")
    p = forced(max_iters=3, max_dt=0.001).to_ffi()
    single = ffi.Context(product_lib, len(mass), planes)
    single.upload(mass, pos, vel)
    grp = D.make_loopback_group(product_lib, pos, mass, vel, planes, k)
    for _ in range(6):
        st1 = single.step(p)
        sts = ffi.group_step(grp, p)
        assert all(abs(st.dt - st1.dt) <= 1e-5 * st1.dt for st in sts)      # CFL-limited: v differs in the last bits
    assert sum(c.n for c in grp) == len(mass)
    assert np.array_equal(D.gather_by_id(grp, "neighbor_count", len(mass)), single.download("neighbor_count"))
    for f, tol in (("position", 1e-5), ("velocity", 1e-3), ("density", 1e-4)):
        assert rel_err(D.gather_by_id(grp, f, len(mass)), single.download(f)) <= tol, (f, info)


@pytest.mark.parametrize("k", [2, 4])
def test_rebalancing_moves_the_cuts_and_keeps_the_physics(product_lib, k):
    """sph_dist_set_rebalance: a column that drifts to the right leaves the static cuts behind (the left slab drains); with the
    cuts re-set to the x quantiles every 5 steps the counts stay level, particles cross as many slabs as they have to, and the
    fields still equal the single context's."""
    scn = sc.dam_break_small(96, 48, 1 / 48)
    pos, mass, vel = sc.init_particles(scn)
    vel = vel.copy()
    vel[:, 0] = 1.5
    planes = sc.boundary_planes(scn.boundary)
    p = forced(max_iters=4).to_ffi()
    single = ffi.Context(product_lib, len(mass), planes)
    single.upload(mass, pos, vel)
    static = D.make_loopback_group(product_lib, pos, mass, vel, planes, k)
    moving = D.make_loopback_group(product_lib, pos, mass, vel, planes, k)
    for c in moving:
        c.dist_set_rebalance(5)
    cuts0 = [c.dist_get_cuts()[:2] for c in moving]
    for s in range(60):
        st1 = single.step(p)
        ffi.group_step(static, p)
        sts = ffi.group_step(moving, p)
        assert all(abs(st.dt - st1.dt) <= 1e-5 * st1.dt for st in sts)
    n_static, n_moving = [c.n for c in static], [c.n for c in moving]
    assert sum(n_moving) == len(mass)
    # level again: to within a lattice column or two (48 particles each) plus the drift since the last re-set
    assert max(n_moving) - min(n_moving) < 0.12 * len(mass) / k
    assert max(n_static) - min(n_static) > 2 * (max(n_moving) - min(n_moving))   # the static cuts did drift out of balance
    cuts1 = [c.dist_get_cuts() for c in moving]
    assert all(c[2] >= 10 for c in cuts1)
    for r in range(k - 1):
        assert cuts1[r][1] == cuts1[r + 1][0] and cuts1[r][1] > cuts0[r][1]     # shared, and moved with the fluid
    ids = np.concatenate([c.download("particle_id") for c in moving])
    assert np.array_equal(np.sort(ids), np.arange(len(mass)))
    # 60 steps with the column hitting the right wall: summation-order differences grow to a few 1e-4 in v -- the group with
    # the static cuts shows the same
    # (position: 1.0e-5 .. 1.7e-5 depending on the pair arithmetic of the build -- the same particles summed in another order)
    for f, tol in (("position", 3e-5), ("velocity", 1e-3), ("density", 1e-4)):
        assert rel_err(D.gather_by_id(moving, f, len(mass)), single.download(f)) <= tol, f
        assert rel_err(D.gather_by_id(static, f, len(mass)), single.download(f)) <= tol, f
    assert (D.gather_by_id(moving, "neighbor_count", len(mass)) != single.download("neighbor_count")).mean() < 1e-3


@pytest.mark.parametrize("k", [2, 3])
def test_level_estimation_on_slabs(product_lib, k):
    """EmptyAngle detection, propagation, smoothing and size classes across the cuts: the ghost layer widens to the extended
    range, the ghosts' (level, when) follow their owners after every sweep, the stop decision is all-reduced.  Same flags,
    same distances, same classes as the single context (whose propagation runs in frontier form)."""
    scn = sc.dam_break_small(96, 48, 1 / 48)
    pos, mass, vel = sc.init_particles(scn)
    vel = vel.copy()
    vel[:, 0] = 0.8
    planes = sc.boundary_planes(scn.boundary)
    P = forced(max_iters=4, level_estimation_method="EmptyAngle", maximum_surface_distance=0.2, particle_radius_fine=0.004,
               particle_radius_base=0.02)
    P.fill_stash_with = "SurfaceDistanceMiddle"
    p = P.to_ffi()
    single = ffi.Context(product_lib, len(mass), planes)
    single.upload(mass, pos, vel)
    grp = D.make_loopback_group(product_lib, pos, mass, vel, planes, k)
    for s in range(12):
        single.step(p)
        ffi.group_step(grp, p)
    n = len(mass)
    for f in ("flag_is_fluid_surface", "flag_insufficient_neighs"):
        assert np.array_equal(D.gather_by_id(grp, f, n), single.download(f)), f
    assert 0 < single.download("flag_is_fluid_surface").sum() < n
    for f in ("level_estimation", "level_old", "stash"):
        a, b = D.gather_by_id(grp, f, n), single.download(f)
        assert np.array_equal(np.isnan(a), np.isnan(b)), f
        assert np.nanmax(np.abs(a - b)) <= 1e-4 * max(np.nanmax(np.abs(b)), 1e-30), f
    single.classify(p)
    for c in grp:
        c.classify(p)
    assert len(np.unique(single.download("particle_size_class"))) > 1
    assert (D.gather_by_id(grp, "particle_size_class", n) != single.download("particle_size_class")).mean() < 1e-3
    for f, tol in (("position", 1e-5), ("velocity", 1e-4), ("density", 1e-5)):
        assert rel_err(D.gather_by_id(grp, f, n), single.download(f)) <= tol, f


@pytest.mark.parametrize("k", [2, 3])
def test_slab_level_propagation_in_frontier_form_equals_the_plain_form(lab_lib, monkeypatch, k):
    """The propagation on slabs runs in the frontier form -- marked candidates plus every unassigned halo member probing its list,
    a probing lane that is assigned marks its neighbours in a second pass (OpLevelPropagate, mode 2) -- and moves (level, when) of
    the ghosts in ONE exchange per sweep.  SPH_SLAB_LEVEL_PLAIN=1 is the reference's form, every unassigned particle in every
    sweep: the same field, bit for bit, and the same number of sweeps (the exchanges of the two runs are equal)."""
    scn = sc.dam_break_small(96, 48, 1 / 48)
    pos, mass, vel = sc.init_particles(scn)
    vel = vel.copy()
    vel[:, 0] = 0.8
    planes = sc.boundary_planes(scn.boundary)
    P = forced(max_iters=4, level_estimation_method="EmptyAngle", maximum_surface_distance=0.2, particle_radius_fine=0.004,
               particle_radius_base=0.02)
    p = P.to_ffi()
    out = {}
    for form in ("frontier", "plain"):
        if form == "plain":
            monkeypatch.setenv("SPH_SLAB_LEVEL_PLAIN", "1")
        grp = D.make_loopback_group(lab_lib, pos, mass, vel, planes, k)
        for s in range(10):
            ffi.group_step(grp, p)
        out[form] = ({f: D.gather_by_id(grp, f, len(mass)) for f in ("level_estimation", "level_old", "flag_is_fluid_surface", "position")},
                     sum(c.dist_get_stats()["exchanges"] for c in grp))
        if form == "plain":
            monkeypatch.delenv("SPH_SLAB_LEVEL_PLAIN")
    (fa, xa), (fb, xb) = out["frontier"], out["plain"]
    assert xa == xb
    for f in fa:
        assert np.array_equal(fa[f], fb[f], equal_nan=True), f
    assert np.isfinite(fa["level_estimation"]).sum() > 0.9 * len(mass)


@pytest.mark.parametrize("method,extended", [("EmptyAngle", True), ("EmptyAngle", False), ("CenterDiff", True)])
def test_level_estimation_after_advection_on_slabs(product_lib, method, extended):
    """level_estimation_after_advection (simulation.rs:2678-2722) across the cuts: the ghosts' advected records come from their
    owners, the extended lists of the advected positions are gathered with every range widened by twice the largest displacement
    over all ranks -- which the ghost layer was widened for at the start of the step --, detection / propagation / smoothing run
    there.  Same flags, distances and trajectories as the single context."""
    scn = sc.dam_break_small(96, 48, 1 / 48)
    pos, mass, vel = sc.init_particles(scn)
    vel = vel.copy()
    vel[:, 0] = 0.8
    planes = sc.boundary_planes(scn.boundary)
    P = forced(max_iters=4, level_estimation_method=method, level_estimation_after_advection=True,
               use_extended_range_for_level_estimation=extended, maximum_surface_distance=0.2, particle_radius_fine=0.004, particle_radius_base=0.02)
    p = P.to_ffi()
    single = ffi.Context(product_lib, len(mass), planes)
    single.upload(mass, pos, vel)
    grp = D.make_loopback_group(product_lib, pos, mass, vel, planes, 3)
    for s in range(12):
        st1 = single.step(p)
        sts = ffi.group_step(grp, p)
        assert all(st.dt == st1.dt for st in sts)
    n = len(mass)
    assert sum(c.n for c in grp) == n
    fa, fb = D.gather_by_id(grp, "flag_is_fluid_surface", n), single.download("flag_is_fluid_surface")
    assert 0 < fb.sum() < n and (fa != fb).sum() <= 2          # (a pair ON the cone / range boundary may flip with the summation order)
    if np.array_equal(fa, fb):
        for f in ("level_estimation", "level_old"):
            a, b = D.gather_by_id(grp, f, n), single.download(f)
            assert np.array_equal(np.isnan(a), np.isnan(b)), f
            assert np.nanmax(np.abs(a - b)) <= 1e-4 * max(np.nanmax(np.abs(b)), 1e-30), f
    assert np.array_equal(D.gather_by_id(grp, "neighbor_count", n), single.download("neighbor_count"))
    for f, tol in (("position", 1e-5), ("velocity", 1e-4), ("density", 1e-5)):
        assert rel_err(D.gather_by_id(grp, f, n), single.download(f)) <= tol, f
    # what the reference's cache holds now (and its partner searches would iterate): with the extended range the lists of the
    # ADVECTED positions at that range, else the step's own -- exported per owned particle with global ids
    so, si = single.download_neighbors()
    differing = 0
    for c in grp:
        ids = c.download("particle_id")
        off, idx = c.download_neighbors()
        assert len(off) == c.n + 1
        for r, pid in enumerate(ids):
            differing += not np.array_equal(np.sort(idx[off[r]:off[r + 1]]), np.sort(si[so[pid]:so[pid + 1]]))
    assert differing <= 2          # (positions differ in the last bits between the two runs: a pair ON the range may flip)
    if extended:
        assert (np.diff(so.astype(np.int64)) > single.download("neighbor_count")).any()     # wider than the k = 2 lists


def test_constrain_neighborhood_count_on_slabs(product_lib, k=2):
    """simulation.rs:2145-2177 across a cut (two slabs: with three the scene's slabs are narrower than two ghost layers): the
    over-populated particles of three rings take their reduced smoothing length from their own lists, the reduced h travels to the ghosts before the density
    replay, the CFL step of the new header is min-reduced.  Same flags, smoothing lengths (bit for bit), dt and fields as the
    single context; the second step fires the reference's own assertion on every rank together."""
    from tests.oracle_harness import rings_and_block_scene
    scn, pos, mass, vel = rings_and_block_scene()
    planes = sc.boundary_planes(scn.boundary, "AnalyticOverestimate")
    p = forced(max_iters=3, constrain_neighborhood_count=True).to_ffi()
    single = ffi.Context(product_lib, len(mass), planes)
    single.upload(mass, pos, vel)
    grp = D.make_loopback_group(product_lib, pos, mass, vel, planes, k)
    st1 = single.step(p)
    sts = ffi.group_step(grp, p)
    assert all(st.dt == st1.dt for st in sts)
    n = len(mass)
    flag = single.download("flag_neighborhood_reduced")
    assert flag.sum() == 3 and np.array_equal(D.gather_by_id(grp, "flag_neighborhood_reduced", n), flag)
    for f in ("h2", "h2_next", "neighbor_count"):
        assert np.array_equal(D.gather_by_id(grp, f, n), single.download(f)), f
    for f, tol in (("position", 1e-6), ("velocity", 1e-5), ("density", 1e-6), ("aii", 1e-5)):
        assert rel_err(D.gather_by_id(grp, f, n), single.download(f)) <= tol, f
    with pytest.raises(ffi.SphError) as e:
        ffi.group_step(grp, p)
    assert e.value.status == 25          # SPH_ERR_CONSTRAIN_NOT_SMALLER, as on the single context


@pytest.mark.parametrize("mode", ["FromDistribution", "FromDistribution2"])
def test_support_length_from_distribution_on_slabs(product_lib, mode):
    """h2_next and the previous step's lambda sums travel with the particles (partition, hand-over to the neighbour, cell sort);
    check_neighborhood / check_aii run on the owned particles of every slab."""
    scn = sc.dam_break_small(96, 48, 1 / 48)
    pos, mass, vel = sc.init_particles(scn)
    vel = vel.copy()
    vel[:, 0] = 0.8
    planes = sc.boundary_planes(scn.boundary)
    p = forced(max_iters=4, support_length_estimation=mode, check_neighborhood=True, check_aii=True).to_ffi()
    single = ffi.Context(product_lib, len(mass), planes)
    single.upload(mass, pos, vel)
    grp = D.make_loopback_group(product_lib, pos, mass, vel, planes, 3)
    n0 = [c.n for c in grp]
    for s in range(15):
        st1 = single.step(p)
        sts = ffi.group_step(grp, p)
        assert all(abs(st.dt - st1.dt) <= 1e-5 * st1.dt for st in sts)
    assert [c.n for c in grp] != n0
    n = len(mass)
    h = single.download("h2")
    assert h.max() > h.min()                               # the estimate did move the smoothing lengths
    for f, tol in (("h2", 1e-5), ("h2_next", 1e-5), ("position", 1e-5), ("velocity", 1e-4), ("density", 1e-5)):
        assert rel_err(D.gather_by_id(grp, f, n), single.download(f)) <= tol, f
    assert np.array_equal(D.gather_by_id(grp, "neighbor_count", n), single.download("neighbor_count"))


def test_profiling_a_slab_group(product_lib):
    """The instrumented pass of bench.py on a decomposition.  The first step takes the general path (no h_max of a previous
    step to size the ghost layer with): its slab partition opens a profiler scope around the radix sort and the reorder, which
    open their own -- nested scopes once corrupted the profiler (hang); only the outermost is timed.  The other five steps take
    the fused refresh (two scopes per step: classify + scan, pack)."""
    scn = sc.dam_break_small(96, 48, 1 / 48)
    pos, mass, vel = sc.init_particles(scn)
    planes = sc.boundary_planes(scn.boundary)
    p = forced(max_iters=4).to_ffi()
    grp = D.make_loopback_group(product_lib, pos, mass, vel, planes, 2)
    for c in grp:
        c.profile_reset()
        c.profile_enable(1)
    for _ in range(6):
        ffi.group_step(grp, p)
    for c in grp:
        prof = c.profile_get()
        assert prof["slab_partition"][0] == 1 and prof["slab_refresh"][0] >= 6 and prof["jacobi_update"][0] >= 12 and "ghost_pack" in prof
        assert all(ms >= 0 for _, ms in prof.values())
        c.profile_enable(0)
    ffi.group_step(grp, p)


def test_fused_refresh_is_bit_identical_to_the_general_path(lab_lib, monkeypatch):
    """Ordinary steps maintain the slabs in one round trip (classify once, arrivals appended, the cell sort drops what left);
    the general path (SPH_SLAB_GENERAL: partition sort + reorder, then the halo selection -- also what the first step, re-balancing
    steps and FromDistribution* support lengths take) must leave the same particles in the same order: every field bit for bit."""
    scn = sc.dam_break_small(96, 48, 1 / 48)
    pos, mass, vel = sc.init_particles(scn)
    vel = vel.copy()
    vel[:, 0] = 0.8                               # particles cross the cuts in both paths
    planes = sc.boundary_planes(scn.boundary)
    p = forced(max_iters=4).to_ffi()
    groups = {}
    for name in ("fused", "general"):
        if name == "general":
            monkeypatch.setenv("SPH_SLAB_GENERAL", "1")
        grp = D.make_loopback_group(lab_lib, pos, mass, vel, planes, 3)
        for c in grp:
            c.profile_enable(1)
        moved = 0
        for s in range(30):
            before = [c.n for c in grp]
            ffi.group_step(grp, p)
            moved += sum(abs(a - c.n) for a, c in zip(before, grp))
        assert moved > 0
        prof = grp[1].profile_get()
        if name == "fused":
            assert prof["slab_partition"][0] == 1 and prof["slab_refresh"][0] >= 29     # only the first step took the general path
        else:
            assert prof["slab_partition"][0] == 30 and "slab_refresh" not in prof
        groups[name] = grp
    monkeypatch.delenv("SPH_SLAB_GENERAL")
    for a, b in zip(groups["fused"], groups["general"]):
        assert a.n == b.n
        for f in ("particle_id", "position", "velocity", "density", "neighbor_count", "mass"):
            assert np.array_equal(a.download(f), b.download(f)), f


@pytest.mark.parametrize("solver,extra", [("IISPH", {}), ("IISPH2", dict(max_dt=0.0005)), ("OnlyDivergence", {}),
                                          ("HybridDFSPH", dict(hybrid_dfsph_non_pressure_accel_before_divergence_free=False)),
                                          ("HybridDFSPH", dict(hybrid_dfsph_density_source_term="OnlyDensity", operator_discretization="ConsistentSymmetricGradient"))])
def test_every_solver_mode_on_slabs(product_lib, solver, extra):
    """The four sequencings of the step (simulation.rs:2262-2670) and the branches inside HybridDFSPH on a 3-rank loopback group
    against the single context: each mode has its own ghost refreshes (IISPH2 refreshes the rescaled pressures before its last
    pressure-acceleration sweep; forces behind the divergence solve add a velocity refresh) and its own final wait."""
    scn = sc.dam_break_small(96, 48, 1 / 48)
    pos, mass, vel = sc.init_particles(scn)
    vel = vel.copy()
    vel[:, 0] = 0.8
    planes = sc.boundary_planes(scn.boundary)
    p = forced(max_iters=4, pressure_solver_method=solver, **extra).to_ffi()
    single = ffi.Context(product_lib, len(mass), planes)
    single.upload(mass, pos, vel)
    grp = D.make_loopback_group(product_lib, pos, mass, vel, planes, 3)
    for s in range(15):
        st1 = single.step(p)
        sts = ffi.group_step(grp, p)
        assert all(st.dt == st1.dt for st in sts)
        assert all(st.div_solver.iters == st1.div_solver.iters and st.density_solver.iters == st1.density_solver.iters for st in sts)
    n = len(mass)
    ids = np.concatenate([c.download("particle_id") for c in grp])
    assert np.array_equal(np.sort(ids), np.arange(n))
    assert np.array_equal(D.gather_by_id(grp, "neighbor_count", n), single.download("neighbor_count"))
    for f, tol in (("position", 1e-5), ("velocity", 1e-4), ("density", 1e-5), ("pressure", 2e-3)):
        assert rel_err(D.gather_by_id(grp, f, n), single.download(f)) <= tol, f


@pytest.mark.parametrize("k", [2, 3])
def test_neighbour_lists_of_the_slabs_are_those_of_the_single_context(product_lib, k):
    """sph_download_neighbors on a slab context: one row per owned particle (the order of the particle_id download), neighbours
    named by their global ids -- ghosts included.  Over all ranks: exactly the single context's lists, as sets, bit for bit."""
    scn = sc.dam_break_small(96, 48, 1 / 48)
    pos, mass, vel = sc.init_particles(scn)
    vel = vel.copy()
    vel[:, 0] = 0.8
    planes = sc.boundary_planes(scn.boundary)
    p = forced(max_iters=3).to_ffi()
    single = ffi.Context(product_lib, len(mass), planes)
    single.upload(mass, pos, vel)
    grp = D.make_loopback_group(product_lib, pos, mass, vel, planes, k)
    for s in range(12):
        single.step(p)
        ffi.group_step(grp, p)
    so, si = single.download_neighbors()
    n = len(mass)
    seen = np.zeros(n, bool)
    crossing = 0
    for c in grp:
        ids = c.download("particle_id")
        off, idx = c.download_neighbors()
        assert len(off) == c.n + 1 and off[-1] == len(idx) and idx.max() < n
        assert np.array_equal(np.diff(off.astype(np.int64)), c.download("neighbor_count"))
        mine = np.zeros(n, bool)
        mine[ids] = True
        crossing += int((~mine[idx]).sum())                       # neighbours that are another rank's particles
        for r, pid in enumerate(ids):
            assert np.array_equal(np.sort(idx[off[r]:off[r + 1]]), np.sort(si[so[pid]:so[pid + 1]])), pid
        seen[ids] = True
    assert seen.all() and crossing > 0


def test_sparse_edits_on_slab_contexts(product_lib):
    """sph_apply_edits on the ranks of a decomposition: indices are rows of the owned order, survivors keep their global ids,
    an appended particle gets its id from the host.  The same logical script on the single context (by host index) and on the
    ranks that own the particles (by row) -- SET of mass / velocity, a swap-with-the-last deletion, an appended particle -- then
    both step on: same particles (matched by id), same fields."""
    scn = sc.dam_break_small(96, 48, 1 / 48)
    pos, mass, vel = sc.init_particles(scn)
    planes = sc.boundary_planes(scn.boundary)
    p = forced(max_iters=3).to_ffi()
    n = len(mass)
    single = ffi.Context(product_lib, n + 16, planes)
    single.upload(mass, pos, vel)
    grp = D.make_loopback_group(product_lib, pos, mass, vel, planes, 3)
    for _ in range(3):
        single.step(p)
        ffi.group_step(grp, p)
    ids_of = [c.download("particle_id") for c in grp]
    # the script, in global ids: particle A gets a new velocity, particle B (interior of rank 1) is deleted, a new particle
    # appears next to particle C on rank 2
    gpos = single.download("position")
    A, B, C = int(ids_of[0][len(ids_of[0]) // 2]), int(ids_of[1][len(ids_of[1]) // 3]), int(ids_of[2][len(ids_of[2]) // 2])
    new_pos = (gpos[C] + np.array([0.25 / 48, 0.1 / 48], np.float32)).astype(np.float32)
    new_mass, new_vel = float(mass[C]) * 0.5, (0.1, -0.2)

    def E(kind, a=0, b=0, **f):
        return {"set": ("set", a, f), "swap": ("swap", a, b), "truncate": ("truncate", a), "extend": ("extend", a)}[kind]

    # single context: host index == id so far; deleting B moves the last particle (id n - 1) into index B
    single.apply_edits([E("set", A, velocity=(0.3, 0.1)), E("swap", B, n - 1), E("truncate", n - 1), E("extend", 1),
                        E("set", n - 1, mass=new_mass, position=tuple(new_pos), velocity=new_vel, h2_next=float(single.download("h2_next")[C]))])
    label_single = np.arange(n)
    label_single[B] = n - 1           # index B now holds the particle that had id n - 1
    label_single[n - 1] = n           # the appended particle: label n
    # ranks: rows of their owned order
    for r, c in enumerate(grp):
        ids = ids_of[r]
        ops = []
        if A in ids:
            ops.append(E("set", int(np.nonzero(ids == A)[0][0]), velocity=(0.3, 0.1)))
        if B in ids:
            row, last = int(np.nonzero(ids == B)[0][0]), len(ids) - 1
            ops += [E("swap", row, last), E("truncate", last)]
        if C in ids:
            m_after = len(ids) - (1 if B in ids else 0)
            ops += [E("extend", 1), E("set", m_after, mass=new_mass, position=tuple(new_pos), velocity=new_vel, h2_next=float(c.download("h2_next")[np.nonzero(ids == C)[0][0]]))]
        c.apply_edits(ops)
        new_ids = c.download("particle_id")
        assert (new_ids == 0xffffffff).sum() == (1 if C in ids else 0)
        new_ids[new_ids == 0xffffffff] = n            # the host names the new particle
        c.upload_field("particle_id", new_ids)
    assert sum(c.n for c in grp) == n == single.n
    for _ in range(4):
        st1 = single.step(p)
        sts = ffi.group_step(grp, p)
        assert all(st.dt == st1.dt for st in sts)
    order_s = np.argsort(label_single)
    ids = np.concatenate([c.download("particle_id") for c in grp])
    order_g = np.argsort(ids)
    assert np.array_equal(np.sort(ids), np.sort(label_single))
    for f, tol in (("position", 1e-5), ("velocity", 1e-4), ("density", 1e-5), ("mass", 0.0)):
        a = np.concatenate([c.download(f) for c in grp])[order_g]
        b = single.download(f)[order_s]
        assert rel_err(a, b) <= tol, f


@pytest.mark.parametrize("transport", ["loopback", "threads"])
def test_a_particle_several_slabs_away_reaches_its_owner(product_lib, transport):
    """One hand-over moves a particle to the x-neighbour.  A particle that is further from its slab than that -- here: set down
    three slabs to the right by an edit; in a run: thrown across a slab by a solve that did not converge -- makes the one-round
    refresh fall back to the general path, which hands over until nobody moves: after ONE step every particle sits on the rank
    whose slab holds it, and the step is the single context's."""
    scn = sc.dam_break_small(128, 32, 1 / 64)
    pos, mass, vel = sc.init_particles(scn)
    planes = sc.boundary_planes(scn.boundary)
    p = forced(max_iters=3).to_ffi()
    n, k = len(mass), 4
    single = ffi.Context(product_lib, n, planes)
    single.upload(mass, pos, vel)
    thr = None
    if transport == "threads":
        thr = D.ThreadedGroup(product_lib, pos, mass, vel, planes, k)
        grp, step = thr.contexts, (lambda: thr.step(p))
    else:
        grp = D.make_loopback_group(product_lib, pos, mass, vel, planes, k)
        step = (lambda: ffi.group_step(grp, p))
    try:
        for _ in range(3):
            single.step(p)
            step()
        ids0 = grp[0].download("particle_id")
        x_hi = float(np.max(single.download("position")[:, 0]))
        movers = [int(ids0[j]) for j in (5, len(ids0) // 2, len(ids0) - 7)]
        # into the empty space right of the column, on the floor, a few spacings apart: rank 3's (open) slab
        targets = [(x_hi + 0.2 + 0.05 * t, float(pos[:, 1].min()) + 0.02) for t in range(3)]
        single.apply_edits([("set", g, dict(position=targets[t], velocity=(0.0, 0.0))) for t, g in enumerate(movers)])
        grp[0].apply_edits([("set", int(np.nonzero(ids0 == g)[0][0]), dict(position=targets[t], velocity=(0.0, 0.0))) for t, g in enumerate(movers)])
        for c in grp[1:]:
            c.apply_edits([])                       # every rank makes the same sequence of calls
        st1 = single.step(p)
        sts = step()
        assert all(st.dt == st1.dt for st in sts)
        owners = {g: [r for r, c in enumerate(grp) if g in c.download("particle_id")] for g in movers}
        assert all(v == [k - 1] for v in owners.values()), owners
        assert sum(c.n for c in grp) == n
        for _ in range(3):
            st1 = single.step(p)
            sts = step()
            assert all(st.dt == st1.dt for st in sts)
        for f, tol in (("position", 1e-5), ("velocity", 1e-4), ("density", 1e-5)):
            assert rel_err(D.gather_by_id(grp, f, n), single.download(f)) <= tol, f
    finally:
        if thr:
            thr.close()


@pytest.mark.parametrize("k,level", [(2, False), (3, False), (4, False), (3, True)])
def test_every_rank_on_its_own_thread_matches_the_loopback_group(product_lib, k, level):
    """The per-rank driver code -- what every process of a multi-GPU run executes: a group of ONE member, the rank's own counts
    and branches (an exchange only where it has something to send or receive), collectives it must enter together with the
    others -- run for real: k ranks, one host thread each, each calling sph_step on its own context, the collectives as
    rendezvous in host memory (thread transport, sph_ffi.h).  A collective that not every rank enters or a send without a
    matching receive would be an error here (a hang over RCCL).  Same arithmetic in the same order as the loopback group
    (sph_group_step drives all ranks from one loop): every field bit for bit, with particles migrating."""
    scn = sc.dam_break_small(96, 48, 1 / 48)
    pos, mass, vel = sc.init_particles(scn)
    vel = vel.copy()
    vel[:, 0] = 0.8
    planes = sc.boundary_planes(scn.boundary)
    kw = dict(level_estimation_method="EmptyAngle", maximum_surface_distance=0.2, particle_radius_fine=0.004, particle_radius_base=0.02) if level else {}
    p = dam_break_params(**kw).to_ffi()          # free-running iteration counts: mispredicted solves, chained and unchained steps
    loop = D.make_loopback_group(product_lib, pos, mass, vel, planes, k)
    thr = D.ThreadedGroup(product_lib, pos, mass, vel, planes, k)
    try:
        moved = 0
        for s in range(25):
            before = [c.n for c in thr.contexts]
            a = ffi.group_step(loop, p)
            b = thr.step(p)
            moved += sum(abs(x - c.n) for x, c in zip(before, thr.contexts))
            for sa, sb in zip(a, b):
                assert sa.dt == sb.dt and sa.div_solver.iters == sb.div_solver.iters and sa.density_solver.iters == sb.density_solver.iters, s
        assert moved > 0
        for ca, cb in zip(loop, thr.contexts):
            assert ca.n == cb.n
            for f in ("particle_id", "position", "velocity", "density", "pressure", "neighbor_count") + (("level_estimation", "flag_is_fluid_surface") if level else ()):
                x, y = ca.download(f), cb.download(f)
                assert np.array_equal(x, y, equal_nan=(x.dtype.kind == "f")), f
        st = thr.contexts[1].dist_get_stats()
        assert st["exchanges"] > 0 and st["bytes_sent"] > 0 and st["bytes_received"] > 0
    finally:
        thr.close()


def test_ranks_that_start_empty(product_lib):
    """Static cuts that leave two of four ranks without a single particle: an empty slab contributes zeros to the all-reduced totals,
    takes its decisions from them, exchanges nothing -- and starts to own particles when the collapsing column reaches it.  Loopback
    group and ranks on threads bit for bit; while the run is still regular (the first steps) also the single context's."""
    import ctypes as C
    from concurrent.futures import ThreadPoolExecutor
    from adaptive_sph_amd.workloads import dam_break_params_scaled
    side = 96
    scn = sc.dam_break_small(side, side, 1.0 / side)
    pos, mass, vel = sc.init_particles(scn)
    planes = sc.boundary_planes(scn.boundary)
    p = dam_break_params_scaled(1.0 / side)().to_ffi()
    x_hi = float(pos[:, 0].max())
    cuts = [-D.INF, float(np.median(pos[:, 0])), x_hi + 0.15, x_hi + 1.0, D.INF]    # ranks 2 and 3 start empty
    k = len(cuts) - 1

    def make(group):
        ctxs = []
        for r in range(k):
            sel = np.nonzero((pos[:, 0] >= cuts[r]) & (pos[:, 0] < cuts[r + 1]))[0]
            c = ffi.Context(product_lib, len(mass) + 4096, planes)
            c.dist_configure(r, k, cuts[r], cuts[r + 1])
            if group is not None:
                c.comm_init_threads(group, r, k)
            c.upload(mass[sel], pos[sel], vel[sel])
            c.upload_field("particle_id", sel.astype(np.uint32))
            ctxs.append(c)
        return ctxs

    group = C.c_void_p()
    assert product_lib.thread_group_create(k, C.byref(group)) == 0
    single = ffi.Context(product_lib, len(mass), planes)
    single.upload(mass, pos, vel)
    loop, thr = make(None), make(group)
    pool = ThreadPoolExecutor(k)
    try:
        assert [c.n for c in loop][2:] == [0, 0]
        for s in range(300):
            a = ffi.group_step(loop, p)
            b = [f.result() for f in [pool.submit(c.step, p) for c in thr]]
            assert all(x.dt == y.dt for x, y in zip(a, b)), s
            if s < 5:
                st1 = single.step(p)
                assert abs(a[0].dt - st1.dt) <= 1e-6 * st1.dt          # (CFL-limited: the sums run in another order on slabs)
                for f, tol in (("position", 1e-5), ("density", 1e-4)):
                    assert rel_err(D.gather_by_id(loop, f, len(mass)), single.download(f)) <= tol, (s, f)
            if loop[2].n > 50:
                break
        assert loop[2].n > 50, [c.n for c in loop]                    # the third rank owns particles by now
        assert sum(c.n for c in loop) == len(mass)
        for ca, cb in zip(loop, thr):
            assert ca.n == cb.n
            if ca.n:
                for f in ("particle_id", "position", "velocity", "density"):
                    assert np.array_equal(ca.download(f), cb.download(f)), f
    finally:
        pool.shutdown(wait=True)
        for c in thr + loop:
            c.close()
        product_lib.thread_group_destroy(group)


def test_ranks_on_threads_exchange_point_to_point(product_lib):
    """A collapsing column hands particles over at its right-hand cuts long before anything crosses the left-hand ones: for many steps
    some ranks exchange migrants with one neighbour while others have nobody to send to or receive from and go straight on to the
    ghost exchange.  That is legal point to point (ncclSend / ncclRecv pair up per neighbour) -- the thread transport meets per
    neighbour pair, as RCCL does, and the run is the loopback group's bit for bit."""
    from adaptive_sph_amd.workloads import dam_break_params_scaled
    side, k = 128, 4
    scn = sc.dam_break_small(side, side, 1.0 / side)
    pos, mass, vel = sc.init_particles(scn)
    planes = sc.boundary_planes(scn.boundary)
    p = dam_break_params_scaled(1.0 / side)().to_ffi()
    loop = D.make_loopback_group(product_lib, pos, mass, vel, planes, k)
    thr = D.ThreadedGroup(product_lib, pos, mass, vel, planes, k)
    try:
        n0 = [c.n for c in loop]
        uneven = 0
        for s in range(60):
            before = [c.n for c in loop]
            a = ffi.group_step(loop, p)
            b = thr.step(p)
            assert all(x.dt == y.dt for x, y in zip(a, b)), s
            changed = [c.n != m for c, m in zip(loop, before)]
            uneven += any(changed) and not all(changed)
        assert uneven > 0 and [c.n for c in loop] != n0      # steps in which only some ranks handed particles over
        for ca, cb in zip(loop, thr.contexts):
            assert ca.n == cb.n
            for f in ("particle_id", "position", "velocity", "density"):
                assert np.array_equal(ca.download(f), cb.download(f)), f
    finally:
        thr.close()


@pytest.mark.parametrize("mode", ["rebalance", "after_advection", "from_distribution", "general_path"])
def test_ranks_on_threads_other_step_variants(lab_lib, monkeypatch, mode):
    """The same per-rank execution for the steps that take other collectives: re-balancing (x range, histogram, migration
    rounds), level estimation after advection (all-reduced displacement, widened ghost layer), FromDistribution support lengths
    (header launch + two-round slab maintenance every step), and the general slab maintenance forced on every step."""
    scn = sc.dam_break_small(96, 48, 1 / 48)
    pos, mass, vel = sc.init_particles(scn)
    vel = vel.copy()
    vel[:, 0] = 0.8
    planes = sc.boundary_planes(scn.boundary)
    kw = {}
    if mode == "after_advection":
        kw = dict(level_estimation_method="EmptyAngle", level_estimation_after_advection=True, maximum_surface_distance=0.2, particle_radius_fine=0.004,
                  particle_radius_base=0.02)
    if mode == "from_distribution":
        kw = dict(support_length_estimation="FromDistribution")
    if mode == "general_path":
        monkeypatch.setenv("SPH_SLAB_GENERAL", "1")
    p = forced(max_iters=4, **kw).to_ffi()
    loop = D.make_loopback_group(lab_lib, pos, mass, vel, planes, 3)
    thr = D.ThreadedGroup(lab_lib, pos, mass, vel, planes, 3)
    try:
        if mode == "rebalance":
            for c in loop + thr.contexts:
                c.dist_set_rebalance(2)
        for s in range(12):
            a = ffi.group_step(loop, p)
            b = thr.step(p)
            assert all(sa.dt == sb.dt for sa, sb in zip(a, b)), s
        if mode == "rebalance":
            assert thr.contexts[1].dist_get_cuts()[2] > 0 and thr.contexts[1].dist_get_cuts() == loop[1].dist_get_cuts()
        for ca, cb in zip(loop, thr.contexts):
            assert ca.n == cb.n
            for f in ("particle_id", "position", "velocity", "density", "neighbor_count"):
                assert np.array_equal(ca.download(f), cb.download(f)), f
    finally:
        thr.close()


@pytest.mark.parametrize("solver,extra", [("IISPH", {}), ("IISPH2", dict(max_dt=0.0005)), ("OnlyDivergence", {}),
                                          ("HybridDFSPH", dict(hybrid_dfsph_non_pressure_accel_before_divergence_free=False)),
                                          ("HybridDFSPH", dict(check_neighborhood=True, check_aii=True))])
def test_ranks_on_threads_every_solver_mode(lab_lib, monkeypatch, solver, extra):
    """Every sequencing of the step, per rank on its own thread, chained solves forced on (SPH_CHAIN=1: the gated second solve and
    its short-fall path run on every rank together or not at all) -- bit for bit the loopback group."""
    monkeypatch.setenv("SPH_CHAIN", "1")
    scn = sc.dam_break_small(96, 48, 1 / 48)
    pos, mass, vel = sc.init_particles(scn)
    vel = vel.copy()
    vel[:, 0] = 0.8
    planes = sc.boundary_planes(scn.boundary)
    p = dam_break_params(pressure_solver_method=solver, **extra).to_ffi()
    loop = D.make_loopback_group(lab_lib, pos, mass, vel, planes, 3)
    thr = D.ThreadedGroup(lab_lib, pos, mass, vel, planes, 3)
    try:
        for s in range(15):
            a = ffi.group_step(loop, p)
            b = thr.step(p)
            assert all(x.dt == y.dt and x.div_solver.iters == y.div_solver.iters and x.density_solver.iters == y.density_solver.iters for x, y in zip(a, b)), s
        for ca, cb in zip(loop, thr.contexts):
            assert ca.n == cb.n
            for f in ("particle_id", "position", "velocity", "density", "pressure"):
                assert np.array_equal(ca.download(f), cb.download(f)), f
    finally:
        thr.close()


def test_ranks_on_threads_fail_together(product_lib):
    """A guard that fires on ONE rank (a NaN velocity uploaded there) ends the step on every rank -- through the all-reduced
    totals and the guard agreement, not through a time-out -- and poisons every context."""
    scn = sc.dam_break_small(96, 48, 1 / 48)
    pos, mass, vel = sc.init_particles(scn)
    planes = sc.boundary_planes(scn.boundary)
    p = dam_break_params().to_ffi()
    thr = D.ThreadedGroup(product_lib, pos, mass, vel, planes, 3)
    try:
        for _ in range(2):
            thr.step(p)
        c = thr.contexts[2]
        v = c.download("velocity")
        v[len(v) // 2] = np.nan
        m, x, ids = c.download("mass"), c.download("position"), c.download("particle_id")
        c.upload(m, x, v)
        c.upload_field("particle_id", ids)
        import time
        t0 = time.perf_counter()
        futs = [thr.pool.submit(cc.step, p) for cc in thr.contexts]
        errs = []
        for f in futs:
            with pytest.raises(ffi.SphError) as e:
                f.result()
            errs.append(e.value.status)
        assert time.perf_counter() - t0 < 30.0              # nobody waited for a rank that had left
        assert errs[2] in (14, 15, 17, 18, 19) and all(e != 0 for e in errs), errs
    finally:
        thr.close()


@pytest.mark.parametrize("overlap", ["1", "0"])
def test_loopback_without_host_waits_is_bit_identical_to_the_host_synchronous_form(product_lib, monkeypatch, overlap):
    """The loopback transport orders its copies by events between the members' streams and all-reduces the solver totals through
    mapped host memory -- no host wait inside an exchange or a Jacobi iteration.  SPH_LOOPBACK_SYNC=1 is its first form (the host
    waits for every member before and after every copy, sums the totals itself): same fields bit for bit, far fewer host waits.
    Particles cross the cuts (both exchanges of the refresh), with the split sweep A and without."""
    monkeypatch.setenv("SPH_OVERLAP", overlap)
    scn = sc.dam_break_small(96, 48, 1 / 48)
    pos, mass, vel = sc.init_particles(scn)
    vel = vel.copy()
    vel[:, 0] = 0.8
    planes = sc.boundary_planes(scn.boundary)
    p = dam_break_params().to_ffi()
    runs = {}
    for name in ("events", "host"):
        if name == "host":
            monkeypatch.setenv("SPH_LOOPBACK_SYNC", "1")
        grp = D.make_loopback_group(product_lib, pos, mass, vel, planes, 4)
        for _ in range(3):
            ffi.group_step(grp, p)
        for c in grp:
            c.dist_get_stats(reset=True)
        stats = [ffi.group_step(grp, p) for _ in range(20)]
        runs[name] = ([(x.dt, x.div_solver.iters, x.density_solver.iters) for st in stats for x in st],
                      [{f: c.download(f) for f in ("particle_id", "position", "velocity", "density", "pressure")} for c in grp],
                      grp[1].dist_get_stats()["host_waits"] / 20)
    monkeypatch.delenv("SPH_LOOPBACK_SYNC")
    assert runs["events"][0] == runs["host"][0]
    for a, b in zip(runs["events"][1], runs["host"][1]):
        for f in a:
            assert np.array_equal(a[f], b[f]), f
    assert runs["events"][2] <= 6 and runs["host"][2] > 3 * runs["events"][2], (runs["events"][2], runs["host"][2])


@pytest.mark.parametrize("transport", ["loopback", "threads"])
@pytest.mark.parametrize("solver,extra", [("HybridDFSPH", {}), ("IISPH", {}), ("OnlyDivergence", {}), ("IISPH2", dict(max_dt=0.0005))])
def test_split_sweep_a_is_bit_identical_to_the_unsplit_form(product_lib, monkeypatch, transport, solver, extra):
    """Large slabs with neighbours run sweep A of every Jacobi iteration in two launches -- the particles without a ghost in reach on
    the context's side stream, beside the ghost exchange and the all-reduce of the totals; the halo members and the first ghost ring
    once the ghosts arrived and the interior is done.  SPH_OVERLAP=0 keeps the one-launch form with everything on one stream: same
    particles, same arithmetic, same decisions -- every field bit for bit, and the split form really ran (its list launch shows up
    in the profile)."""
    scn = sc.dam_break_small(96, 48, 1 / 48)
    pos, mass, vel = sc.init_particles(scn)
    vel = vel.copy()
    vel[:, 0] = 0.8
    planes = sc.boundary_planes(scn.boundary)
    p = dam_break_params(pressure_solver_method=solver, **extra).to_ffi()
    runs = {}
    for name in ("split", "unsplit"):
        monkeypatch.setenv("SPH_OVERLAP", "1" if name == "split" else "0")   # (default: by slab size, these slabs are small)
        thr = None
        if transport == "threads":
            thr = D.ThreadedGroup(product_lib, pos, mass, vel, planes, 3)
            grp, step = thr.contexts, (lambda: thr.step(p))
        else:
            grp = D.make_loopback_group(product_lib, pos, mass, vel, planes, 3)
            step = (lambda: ffi.group_step(grp, p))
        try:
            grp[1].profile_enable(1)
            stats = [step() for _ in range(12)]
            prof = grp[1].profile_get()
            # (either form adds up the rank's totals in block 0 of the launch that packs the ghost values, so that the all-reduce travels
            #  with it: one collective call per iteration)
            assert ("pressure_accel_edge" in prof) == (name == "split") and ("solver_progress" in prof) == (name == "split") and "ghost_pack" in prof
            runs[name] = ([(x.dt, x.div_solver.iters, x.density_solver.iters) for st in stats for x in st],
                          [{f: c.download(f) for f in ("particle_id", "position", "velocity", "density", "pressure")} for c in grp])
        finally:
            if thr:
                thr.close()
    monkeypatch.delenv("SPH_OVERLAP")
    assert runs["split"][0] == runs["unsplit"][0]
    for a, b in zip(runs["split"][1], runs["unsplit"][1]):
        for f in a:
            assert np.array_equal(a[f], b[f]), f


@pytest.mark.parametrize("which", ["first step", "one-round refresh"])
def test_ranks_on_threads_capacity_failure_is_collective(product_lib, which):
    """One rank of three (each on its own thread) has no room for what its neighbours hand it -- at the first step (two-round slab
    maintenance: owned + ghosts do not fit) or at the second (one-round refresh: previous slots + arrivals + new ghosts do not fit,
    although the first step's owned + ghosts did).  Every rank's sph_step returns an error within that step; nobody waits for a
    collective the starved rank never enters."""
    import time
    scn = sc.dam_break_small(96, 48, 1 / 48)
    pos, mass, vel = sc.init_particles(scn)
    planes = sc.boundary_planes(scn.boundary)
    p = dam_break_params().to_ffi()
    probe = D.ThreadedGroup(product_lib, pos, mass, vel, planes, 3)
    try:
        probe.step(p)
        st = probe.contexts[1].dist_get_stats()
    finally:
        probe.close()
    owned, ghosts = st["n_owned"], sum(st["n_ghost"])
    assert ghosts > 64
    cap = owned + 8 if which == "first step" else owned + ghosts + ghosts // 2
    thr = D.ThreadedGroup(product_lib, pos, mass, vel, planes, 3, capacities=[None, cap, None])
    try:
        if which != "first step":
            thr.step(p)
        t0 = time.perf_counter()
        futs = [thr.pool.submit(c.step, p) for c in thr.contexts]
        errs = []
        for f in futs:
            with pytest.raises(ffi.SphError) as e:
                f.result()
            errs.append(e.value.status)
        assert time.perf_counter() - t0 < 30.0
        assert errs[1] == 3 and all(e != 0 for e in errs), errs               # SPH_ERR_CAPACITY on the starved rank, its status on the others
    finally:
        thr.close()


def test_a_slab_without_room_for_its_ghosts_says_so(product_lib):
    """Between the refresh and the cell sort a slab holds its previous slots, the arrivals and the new ghosts: a context that
    cannot fit them ends the step in SPH_ERR_CAPACITY (and is poisoned) -- it never writes past its arrays, and it does not leave
    the step alone: it receives what was agreed, drops it, raises the guard word, and every rank fails together at the step's end."""
    scn = sc.dam_break_small(96, 48, 1 / 48)
    pos, mass, vel = sc.init_particles(scn)
    planes = sc.boundary_planes(scn.boundary)
    p = forced(max_iters=3).to_ffi()
    cuts = D.slab_cuts(pos[:, 0], 2)
    parts = D.partition(pos[:, 0], cuts)
    grp = []
    for r in range(2):
        c = ffi.Context(product_lib, len(parts[r]) + 8, planes)          # room for the owned particles, not for a ghost layer
        c.dist_configure(r, 2, cuts[r], cuts[r + 1])
        c.upload(mass[parts[r]], pos[parts[r]], vel[parts[r]])
        c.upload_field("particle_id", parts[r].astype(np.uint32))
        grp.append(c)
    with pytest.raises(ffi.SphError) as e:
        for _ in range(3):
            ffi.group_step(grp, p)
    assert e.value.status == 3 and "room" in str(e.value)              # SPH_ERR_CAPACITY
    with pytest.raises(ffi.SphError):
        ffi.group_step(grp, p)                                          # poisoned until sph_upload


def test_group_of_one_is_the_plain_step(product_lib):
    scn = sc.dam_break_small(32, 32, 1 / 32)
    pos, mass, vel = sc.init_particles(scn)
    planes = sc.boundary_planes(scn.boundary)
    p = forced(max_iters=3).to_ffi()
    a = ffi.Context(product_lib, len(mass), planes)
    b = ffi.Context(product_lib, len(mass), planes)
    a.upload(mass, pos, vel)
    b.upload(mass, pos, vel)
    for _ in range(5):
        a.step(p)
        ffi.group_step([b], p)
    for f in ("position", "velocity", "density", "pressure"):
        assert np.array_equal(a.download(f), b.download(f)), f


def test_rccl_collectives_with_one_rank(product_lib, monkeypatch):
    """SPH_FORCE_SLAB_MODE: the slab driver with the RCCL transport on a communicator of ONE rank -- every ncclAllReduce of the
    step (CFL minimum, error agreement, Jacobi totals) and the empty send/recv groups really run through RCCL; only the
    neighbour exchange needs a second GPU.  Same trajectory as the plain context."""
    import ctypes as C
    scn = sc.dam_break_small(64, 48, 1 / 48)
    pos, mass, vel = sc.init_particles(scn)
    planes = sc.boundary_planes(scn.boundary)
    p = forced(max_iters=4).to_ffi()
    plain = ffi.Context(product_lib, len(mass), planes)
    plain.upload(mass, pos, vel)
    monkeypatch.setenv("SPH_FORCE_SLAB_MODE", "1")
    raw = (C.c_uint8 * 128)()
    assert product_lib.comm_unique_id(raw) == 0
    c = ffi.Context(product_lib, len(mass) + 1024, planes)
    c.dist_configure(0, 1, -D.INF, D.INF)
    c.comm_init(bytes(raw), 0, 1)
    monkeypatch.delenv("SPH_FORCE_SLAB_MODE")
    c.upload(mass, pos, vel)
    c.upload_field("particle_id", np.arange(len(mass), dtype=np.uint32))
    for _ in range(10):
        s0, s1 = plain.step(p), c.step(p)
        assert s0.dt == s1.dt and s0.div_solver.iters == s1.div_solver.iters
    assert c.n == len(mass)
    for f, tol in (("position", 1e-6), ("velocity", 1e-5), ("density", 1e-6)):
        assert rel_err(D.gather_by_id([c], f, len(mass)), plain.download(f)) <= tol, f


def test_rccl_single_rank_roundtrip(product_lib):
    """world_size 1 through the RCCL entry points (more ranks need more GPUs than this box has)."""
    import ctypes as C
    raw = (C.c_uint8 * 128)()
    assert product_lib.comm_unique_id(raw) == 0
    scn = sc.dam_break_small(16, 16, 1 / 16)
    pos, mass, vel = sc.init_particles(scn)
    c = ffi.Context(product_lib, len(mass), sc.boundary_planes(scn.boundary))
    c.dist_configure(0, 1, -D.INF, D.INF)
    c.comm_init(bytes(raw), 0, 1)
    c.upload(mass, pos, vel)
    c.step(dam_break_params().to_ffi())
    assert np.all(np.isfinite(c.download("position")))
