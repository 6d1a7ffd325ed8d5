"""BIT-FOR-BIT parity of the HIP path with the CPU oracle (round 5).

The reference's neighbour ORDER (R*-tree traversal) and reduce order (rayon) are unpinned, so every other parity test of this suite
compares fields within a tolerance.  Two things separate the device from the oracle on identical inputs: the ORDER a particle's
neighbour sums are taken in (device: ascending slot of the cell-sorted array -- rows of cells bottom to top; oracle: ascending host
index) and the ARITHMETIC of the default math policy (v_rsq / v_rcp, fma, truncated-power spline).  This file removes both:

  * SPH_HIP_EXACT=1 -- the library's EXACT policy: IEEE division / sqrt in the reference's operation order, no fma;
  * the particles are uploaded to BOTH sides in the device's visiting order (stable sort by the cell of the sorting grid: the device's
    own sort is then the identity, and the oracle's ascending-index sum IS the device's ascending-slot sum).

What is left must be equal to the last bit -- tolerance 0 on every field the step produces -- or there is an arithmetic difference
between the HIP kernels and the restated reference.  scripts/gpu_normal_count.py found this on configs[1] (profiles/r5_normal_count.md:
step 4 of the bench window, 1 048 576 particles: pressure, a^p, density, counts of the residual classes all identical, where the
round-4 comparison in host order saw 97 724 vs 77 272 "normal" particles); here it is asserted, step by step over the window bench.py
times and over the solver modes / discretisations / viscosities / boundary terms on small scenes.

Multi-step runs re-upload every step (the oracle's state of the step before, re-sorted): a particle that changes its cell changes the
device's slot order but not the oracle's index order.
"""
import numpy as np
import pytest

from adaptive_sph_amd import ffi, scene as sc
from adaptive_sph_amd.workloads import dam_break_params, default_params

pytestmark = pytest.mark.gpu

STEP_FIELDS = ["position", "velocity", "density", "pressure", "pressure_accel", "aii", "ppe_source_term", "constant_field",
               "lambda_sum", "lambda_grad_sum", "h2", "neighbor_count", "cell_index"]


def h_from_mass(mass, rest_density=1.0):
    """h_next_from_mass (simulation.rs:1865-1871) in f32: 1.9 sqrt((m / rho0) (1 / pi))"""
    vol = mass.astype(np.float32) / np.float32(rest_density)
    return np.float32(1.9) * np.sqrt(vol * np.float32(0.318309873342514038086), dtype=np.float32)


def device_order(pos, h):
    """The device's slot order for particles at `pos`: stable sort by the cell of the SORTING grid -- cell = the support 2 h of the
    smallest particle, cell index floor(x / cs) per axis (IEEE f32 division, neighborhood_search.rs:253-255), cells ordered x fastest
    (:383-395); ties keep the upload order."""
    cs = np.float32(2.0) * np.float32(h.min())
    cx = np.floor(pos[:, 0].astype(np.float32) / cs)
    cy = np.floor(pos[:, 1].astype(np.float32) / cs)
    return np.lexsort((cx, cy))


def forced(base=dam_break_params, **kw):
    return base(hybrid_dfsph_max_avg_density_error=0.0, hybrid_dfsph_max_avg_divergence_error=0.0, iisph_max_avg_density_error=0.0, **kw)


def assert_bit_identical(g, o, fields, where=""):
    for f in fields:
        a, b = g.download(f), o.download(f)
        if not np.array_equal(a, b):
            bad = np.nonzero((a != b).reshape(len(a), -1).any(axis=1))[0]
            d = np.abs(a.astype(np.float64) - b.astype(np.float64)).max()
            raise AssertionError(f"{where}{f}: {len(bad)} of {len(a)} particles differ (first {bad[:5]}), max |difference| {d:.3g}")


def solver_counts(s):
    return (int(s.iters), int(s.normal_count), int(s.negative_count), int(s.singular_count))


def stepwise(product_lib, oracle_lib, mass, pos, vel, planes, params, steps, fields=STEP_FIELDS, counts=True):
    """`steps` steps; every step starts on BOTH sides from the oracle's state of the step before, uploaded in the device's visiting
    order, and must end bit-identical."""
    p = params.to_ffi()
    m, x, v = mass.copy(), pos.copy(), vel.copy()
    for s in range(steps):
        perm = device_order(x, h_from_mass(m, params.rest_density))
        m, x, v = m[perm], x[perm], v[perm]
        g = ffi.Context(product_lib, len(m), planes)
        o = ffi.Context(oracle_lib, len(m), planes)
        g.upload(m, x, v)
        o.upload(m, x, v)
        sg, so = g.step(p), o.step(p)
        where = f"step {s}: "
        assert sg.dt == so.dt, where
        # the device's own cell sort found nothing to do: the upload order IS its visiting order
        ci = g.download("cell_index").astype(np.int64)
        if np.ptp(h_from_mass(m)) == 0:
            assert (np.diff(ci) >= 0).all(), where + "the upload order is not the device's slot order"
        assert_bit_identical(g, o, fields, where)
        if counts:   # the residual classes of the last Jacobi iteration: integer counts of bit-identical pressures
            assert solver_counts(sg.div_solver) == solver_counts(so.div_solver), where
            assert solver_counts(sg.density_solver) == solver_counts(so.density_solver), where
            # the residual sums are reduced in different orders (per-block partials on the device): a relative bar, not bits
            for a, b in ((sg.div_solver, so.div_solver), (sg.density_solver, so.density_solver)):
                if a.normal_count:   # (no "normal" particle: the average is NaN on both sides)
                    assert abs(a.avg_error - b.avg_error) <= 1e-5 * abs(b.avg_error) + 1e-12, where
                assert a.max_error == b.max_error, where
        x, v = o.download("position"), o.download("velocity")
        g.close()
        o.close()


@pytest.fixture
def exact(monkeypatch):
    monkeypatch.setenv("SPH_HIP_EXACT", "1")   # read by sph_create


def test_bench_window_step_by_step_is_the_oracle_bit_for_bit(product_lib, oracle_lib, exact):
    """BASELINE configs[1] at FULL size, the 25 steps bench.py's driver flags time (from rest, through the violent start): every step
    bit-identical to the oracle on every field, iteration counts forced to 6 + 6 (the free-running counts follow below)."""
    scn = sc.dam_break_1m()
    pos, mass, vel = sc.init_particles(scn)
    assert len(mass) == 1048576
    stepwise(product_lib, oracle_lib, mass, pos, vel, sc.boundary_planes(scn.boundary), forced(max_iters=6), 25)


def test_bench_window_free_running_in_the_device_order(product_lib, oracle_lib, exact):
    """... and FREE-RUNNING (configs[1]'s own tolerances): with bit-identical pressures the residual classes are identical, so the two
    sides stop at the same iteration unless the differently reduced residual SUM straddles the threshold in its last digits.  Every
    step restarts from the oracle's state, so a flip cannot propagate: counts equal at >= 22 of the 25 steps (measured: 25), and at every
    step with equal counts every field is bit-identical."""
    scn = sc.dam_break_1m()
    pos, mass, vel = sc.init_particles(scn)
    planes = sc.boundary_planes(scn.boundary)
    P = dam_break_params()
    p = P.to_ffi()
    m, x, v = mass.copy(), pos.copy(), vel.copy()
    same = 0
    rows = []
    for s in range(25):
        perm = device_order(x, h_from_mass(m))
        m, x, v = m[perm], x[perm], v[perm]
        g = ffi.Context(product_lib, len(m), planes)
        o = ffi.Context(oracle_lib, len(m), planes)
        g.upload(m, x, v)
        o.upload(m, x, v)
        sg, so = g.step(p), o.step(p)
        rows.append((int(sg.div_solver.iters), int(so.div_solver.iters), int(sg.density_solver.iters), int(so.density_solver.iters)))
        if rows[-1][0] == rows[-1][1] and rows[-1][2] == rows[-1][3]:
            same += 1
            assert sg.dt == so.dt
            assert_bit_identical(g, o, STEP_FIELDS, f"step {s}: ")
            assert solver_counts(sg.div_solver) == solver_counts(so.div_solver) and solver_counts(sg.density_solver) == solver_counts(so.density_solver)
        x, v = o.download("position"), o.download("velocity")
        g.close()
        o.close()
    assert same >= 22, rows
    assert sum(r[1] + r[3] for r in rows[5:]) / 20 + 2 > 10, rows   # the violent window: ~11 + ~9 iterations per step


def small_scene(jitter=0.0, nx=40, ny=40, seed=3):
    scn = sc.dam_break_small(nx, ny, 1.0 / nx)
    pos, mass, vel = sc.init_particles(scn)
    if jitter:
        rng = np.random.default_rng(seed)
        pos = (pos + rng.uniform(-jitter, jitter, pos.shape).astype(np.float32) * np.float32(1.0 / nx)).astype(np.float32)
        vel = rng.normal(0, 0.05, vel.shape).astype(np.float32)
    return scn, pos, mass, vel


@pytest.mark.parametrize("solver", ["HybridDFSPH", "IISPH", "OnlyDivergence", "IISPH2"])
@pytest.mark.parametrize("jitter", [0.0, 0.2])
def test_solver_modes_bit_for_bit(product_lib, oracle_lib, exact, solver, jitter):
    scn, pos, mass, vel = small_scene(jitter)
    stepwise(product_lib, oracle_lib, mass, pos, vel, sc.boundary_planes(scn.boundary), forced(max_iters=5, pressure_solver_method=solver), 8)


@pytest.mark.parametrize("op", ["ConsistentSimpleGradient", "ConsistentSymmetricGradient", "Winchenbach2020"])
@pytest.mark.parametrize("visc", ["ApproxLaplace", "WCSPH"])
def test_discretisations_and_viscosities_bit_for_bit(product_lib, oracle_lib, exact, op, visc):
    scn, pos, mass, vel = small_scene(0.2, 32, 32)
    P = forced(max_iters=4, operator_discretization=op, viscosity_type=visc, viscosity=0.01)
    stepwise(product_lib, oracle_lib, mass, pos, vel, sc.boundary_planes(scn.boundary), P, 5)


@pytest.mark.parametrize("pen", ["None", "Linear", "Quadratic1", "Quadratic2"])
def test_boundary_penalty_terms_bit_for_bit(product_lib, oracle_lib, exact, pen):
    scn, pos, mass, vel = small_scene(0.1, 24, 24)
    stepwise(product_lib, oracle_lib, mass, pos, vel, sc.boundary_planes(scn.boundary), forced(max_iters=3, boundary_penalty_term=pen), 4)


def test_polygon_boundary_and_source_term_flags_bit_for_bit(product_lib, oracle_lib, exact):
    scn, pos, mass, vel = small_scene(0.1, 24, 24)
    planes = sc.boundary_planes(scn.boundary, "AnalyticUnderestimate")
    for kw in (dict(), dict(hybrid_dfsph_density_source_term="OnlyDensity"), dict(hybrid_dfsph_non_pressure_accel_before_divergence_free=False),
               dict(hybrid_dfsph_factor=0.0)):
        stepwise(product_lib, oracle_lib, mass, pos, vel, planes, forced(max_iters=3, init_boundary_handler="AnalyticUnderestimate", **kw), 3)


def test_two_particle_sizes_bit_for_bit(product_lib, oracle_lib, exact):
    """4:1 radius ratio: the device sorts by the FINE particles' grid and walks wide stencils for the interface particles (explicit index
    lists, ascending slot); the oracle sums in ascending index = the same order."""
    scn = sc.SceneConfig(sc.SceneBoundary("box", 4.0, 2.0),
                         [sc.SceneFluidBlock([-1.99, -0.99], [0.5, 0.4], 1.0 / 64, 0.93, [0.0, 0.0]),
                          sc.SceneFluidBlock([-1.48, -0.985], [0.5, 0.5], 1.0 / 16, 0.93, [0.0, 0.0])])
    pos, mass, vel = sc.init_particles(scn)
    fields = [f for f in STEP_FIELDS if f != "cell_index"]   # (download: the reference's coarse grid either way; compared below)
    stepwise(product_lib, oracle_lib, mass, pos, vel, sc.boundary_planes(scn.boundary), forced(max_iters=4), 6, fields + ["cell_index"])


def test_level_estimation_bit_for_bit(product_lib, oracle_lib, exact):
    """One step with the EmptyAngle level estimation on its extended-range lists (surface detection, propagation, smoothing)."""
    scn, pos, mass, vel = small_scene(0.15, 36, 36)
    P = forced(base=default_params, max_iters=3, merging=False, sharing=False, splitting=False, level_estimation_method="EmptyAngle",
               maximum_surface_distance=0.3, max_dt=0.002)
    stepwise(product_lib, oracle_lib, mass, pos, vel, sc.boundary_planes(scn.boundary), P, 1,
             STEP_FIELDS + ["level_estimation", "flag_is_fluid_surface", "stash"])


def test_math_policy_through_the_abi(product_lib, oracle_lib, monkeypatch):
    """sph_set_math_policy (include/sph_ffi.h): the EXACT policy asked for through the ABI a Rust host binds
    -- no environment variable -- is the oracle bit for bit; switching policies between steps drops what the last step derived and
    nothing else: a context switched in mid-run continues bit for bit like a FRESH context of the new policy that received the same state
    in the switched context's slot order (the summation order: the device sorts stably by cell, so the slot order of a step is the
    stable sort of the previous step's by the cells of that step -- replayed here from the downloaded cell indices)."""
    monkeypatch.delenv("SPH_HIP_EXACT", raising=False)
    scn, pos, mass, vel = small_scene(0.2)
    planes = sc.boundary_planes(scn.boundary)
    perm = device_order(pos, h_from_mass(mass))
    m, x, v = mass[perm], pos[perm], vel[perm]
    p = forced(max_iters=5).to_ffi()
    g, o = ffi.Context(product_lib, len(m), planes), ffi.Context(oracle_lib, len(m), planes)
    assert g.math_policy() == "fast"
    g.set_math_policy("exact")
    assert g.math_policy() == "exact"
    g.upload(m, x, v)
    o.upload(m, x, v)
    order = np.arange(len(m))   # host indices in g's slot order

    def g_step():
        nonlocal order
        st = g.step(p)
        c = g.download("cell_index").astype(np.int64)   # the cells the step sorted by
        order = order[np.argsort(c[order], kind="stable")]
        return st

    sg, so = g_step(), o.step(p)
    assert sg.dt == so.dt
    assert_bit_identical(g, o, STEP_FIELDS, "EXACT through the ABI: ")
    # a FAST step is NOT the oracle to the bit (the switch does something) ...
    f = ffi.Context(product_lib, len(m), planes)
    f.upload(m, x, v)
    f.step(p)
    assert not np.array_equal(f.download("pressure"), o.download("pressure"))
    f.close()
    # ... and switching in mid-run: EXACT -> FAST -> EXACT, each leg against a fresh context of that policy started from g's state
    moved = 0
    for policy in ("fast", "exact"):
        at = order.copy()
        state = [g.download(k)[at] for k in ("mass", "position", "velocity")]
        g.set_math_policy(policy)
        fresh = ffi.Context(product_lib, len(m), planes)
        fresh.set_math_policy(policy)
        fresh.upload(*state)
        for s in range(2):
            a, b = g_step(), fresh.step(p)
            assert a.dt == b.dt and solver_counts(a.density_solver) == solver_counts(b.density_solver), (policy, s)
        moved += int((order != at).sum())
        for k in STEP_FIELDS:
            ga, fa = g.download(k)[at], fresh.download(k)
            assert np.array_equal(ga, fa), f"switched to {policy}: {k}: {int((ga != fa).reshape(len(ga), -1).any(axis=1).sum())} particles differ"
        fresh.close()
    assert moved > 0   # (particles did change cells on the way: the replayed slot order was needed)
    with pytest.raises(Exception):
        g.set_math_policy(7)
