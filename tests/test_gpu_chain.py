"""HybridDFSPH on one context, three ways of queueing the same launches (sph_step.hip):
  paced     (default) the host queues every iteration against the device's published stop decisions: no prediction, no gate, one
            host wait per step;
  chained   (SPH_PACED=0, SPH_CHAIN=1) predicted counts, the density solve queued behind the divergence solve with its launches
            gated on the device -- the form slab decompositions use; a divergence solve that falls short re-queues both;
  two-wait  (SPH_PACED=0, SPH_CHAIN=0) predicted counts, a host wait behind each solve.
Whatever the queue does, the step must be the step: every field and every iteration count bit for bit the same."""
import numpy as np
import pytest

from adaptive_sph_amd import ffi, scene as sc
from adaptive_sph_amd.workloads import dam_break_params

pytestmark = pytest.mark.gpu

FORMS = {"paced": dict(SPH_PACED="1"), "chained": dict(SPH_PACED="0", SPH_CHAIN="1"), "two-wait": dict(SPH_PACED="0", SPH_CHAIN="0")}


def run(lab_lib, monkeypatch, form, steps, **overrides):
    for k, v in FORMS[form].items():
        monkeypatch.setenv(k, v)
    scn = sc.dam_break_small(128, 96, 1 / 64)
    pos, mass, vel = sc.init_particles(scn)
    P = dam_break_params(**overrides)
    g = ffi.Context(lab_lib, len(mass), sc.boundary_planes(scn.boundary, P.init_boundary_handler))
    g.upload(mass, pos, vel)
    p = P.to_ffi()
    its = []
    for _ in range(steps):
        st = g.step(p)
        its.append((int(st.div_solver.iters), int(st.density_solver.iters), int(st.div_solver.normal_count), int(st.density_solver.normal_count),
                    np.float32(st.div_solver.avg_error).view(np.uint32).item(), np.float32(st.density_solver.avg_error).view(np.uint32).item()))
    waits = g.dist_get_stats()["host_waits"]
    for k in FORMS[form]:
        monkeypatch.delenv(k)
    return g, its, waits


@pytest.mark.parametrize("overrides", [dict(), dict(hybrid_dfsph_density_source_term="OnlyDensity")])
def test_paced_chained_and_two_wait_solves_are_bit_identical(lab_lib, monkeypatch, overrides):
    steps = 40
    a, ia, wa = run(lab_lib, monkeypatch, "chained", steps, **overrides)
    b, ib, wb = run(lab_lib, monkeypatch, "two-wait", steps, **overrides)
    c, ic, wc = run(lab_lib, monkeypatch, "paced", steps, **overrides)
    assert ia == ib == ic
    div = [t[0] for t in ia]
    assert max(div) > min(div) and any(div[k + 1] > div[k] for k in range(len(div) - 1))   # a chained divergence solve fell short at least once
    assert wa < wb                                                                        # fewer host waits when chained
    assert wc <= steps + 1 and wc <= wa                                                   # paced: the step's one wait (+ the first step's header)
    for f in ("position", "velocity", "density", "pressure", "aii", "ppe_source_term", "neighbor_count"):
        assert np.array_equal(a.download(f), b.download(f)), f
        assert np.array_equal(a.download(f), c.download(f)), f


@pytest.mark.parametrize("solver", ["IISPH", "IISPH2", "OnlyDivergence"])
def test_paced_single_solve_modes_match_the_predicted_queue(lab_lib, monkeypatch, solver):
    """pressure_iterations() of the one-solve modes, paced and with the predicted queue"""
    out = {}
    for form in ("paced", "two-wait"):
        g, its, waits = run(lab_lib, monkeypatch, form, 25, pressure_solver_method=solver)
        out[form] = (g, its, waits)
    assert out["paced"][1] == out["two-wait"][1]
    assert out["paced"][2] <= out["two-wait"][2]
    for f in ("position", "velocity", "density", "pressure"):
        assert np.array_equal(out["paced"][0].download(f), out["two-wait"][0].download(f)), f


@pytest.mark.parametrize("solver", ["HybridDFSPH", "IISPH", "OnlyDivergence"])
def test_sweep_a_on_combined_records_is_bit_identical_to_the_generic_sweep(lab_lib, monkeypatch, solver):
    """Uniform-h scenes on one context: the solves store {x, y, p / rho^2, p} records and sweep A gathers one record per neighbour
    (OpPressureAccelU); SPH_ACCEL_GENERIC=1 keeps p / rho^2 as a field and gathers the particle record beside it.  Equal masses:
    the same arithmetic on the same values, so every field and every iteration count agree bit for bit."""
    def go(generic):
        if generic:
            monkeypatch.setenv("SPH_ACCEL_GENERIC", "1")
        out = run(lab_lib, monkeypatch, "paced", 25, pressure_solver_method=solver)
        if generic:
            monkeypatch.delenv("SPH_ACCEL_GENERIC")
        return out
    (a, ia, _), (b, ib, _) = go(False), go(True)
    assert ia == ib
    for f in ("position", "velocity", "density", "pressure", "aii", "ppe_source_term"):
        assert np.array_equal(a.download(f), b.download(f)), f


@pytest.mark.parametrize("np_before", [True, False])
@pytest.mark.parametrize("solver", ["HybridDFSPH", "IISPH", "OnlyDivergence"])
def test_source_term_on_position_velocity_records_is_bit_identical_to_the_generic_sweep(lab_lib, monkeypatch, solver, np_before):
    """Uniform-h scenes on one context: the sweep of the non-pressure forces and the divergence solve's tail leave {x, y, v} records and
    the source-term sweep gathers one record per neighbour (OpSourceU); SPH_SOURCE_GENERIC=1 gathers the particle record and the
    velocity (OpSource).  Equal masses: the same arithmetic on the same values -- every field, every iteration count and the
    residual statistics agree bit for bit.  np_before = False (HybridDFSPH): the forces behind the divergence solve -- its source
    term has no current record (the generic sweep runs), the density solve's has the tail's, then the forces'."""
    def go(generic):
        if generic:
            monkeypatch.setenv("SPH_SOURCE_GENERIC", "1")
        out = run(lab_lib, monkeypatch, "paced", 25, pressure_solver_method=solver, hybrid_dfsph_non_pressure_accel_before_divergence_free=np_before)
        if generic:
            monkeypatch.delenv("SPH_SOURCE_GENERIC")
        return out
    (a, ia, _), (b, ib, _) = go(False), go(True)
    assert ia == ib
    for f in ("position", "velocity", "density", "pressure", "aii", "ppe_source_term", "density_error"):
        assert np.array_equal(a.download(f), b.download(f)), f


@pytest.mark.parametrize("paced", ["1", "0"])
def test_default_policy_waits_once_per_step_when_the_iteration_count_repeats(lab_lib, monkeypatch, paced):
    monkeypatch.delenv("SPH_CHAIN", raising=False)
    monkeypatch.setenv("SPH_PACED", paced)
    scn = sc.dam_break_small(128, 96, 1 / 64)
    pos, mass, vel = sc.init_particles(scn)
    P = dam_break_params(hybrid_dfsph_max_avg_density_error=0.0, hybrid_dfsph_max_avg_divergence_error=0.0, max_iters=3)   # the divergence solve's count pinned
    g = ffi.Context(lab_lib, len(mass), sc.boundary_planes(scn.boundary, P.init_boundary_handler))
    g.upload(mass, pos, vel)
    p = P.to_ffi()
    for _ in range(4):
        g.step(p)
    g.dist_get_stats(reset=True)
    for _ in range(10):
        st = g.step(p)
        assert st.div_solver.iters == 3
    assert g.dist_get_stats()["host_waits"] == 10      # one wait per step: header from the previous step's tail, solves paced (or chained)


def test_chained_solves_on_slabs_are_bit_identical_to_the_two_wait_form(lab_lib, monkeypatch):
    """The same on a slab decomposition (loopback transport, 3 ranks): the gated solve all-reduces into totals of its own, the
    exchanges behind a closed gate re-send what the ghosts already hold."""
    from adaptive_sph_amd import distributed as D
    scn = sc.dam_break_small(128, 96, 1 / 64)      # from rest: the iteration counts of the first steps jump about
    pos, mass, vel = sc.init_particles(scn)
    planes = sc.boundary_planes(scn.boundary)
    p = dam_break_params().to_ffi()
    out = {}
    for chain in ("1", "0"):
        monkeypatch.setenv("SPH_CHAIN", chain)
        grp = D.make_loopback_group(lab_lib, pos, mass, vel, planes, 3)
        its = []
        for _ in range(40):
            sts = ffi.group_step(grp, p)
            its.append((int(sts[0].div_solver.iters), int(sts[0].density_solver.iters), int(sts[1].div_solver.normal_count), int(sts[2].density_solver.normal_count)))
        out[chain] = (grp, its, sum(c.dist_get_stats()["host_waits"] for c in grp))
    monkeypatch.delenv("SPH_CHAIN")
    (ga, ia, wa), (gb, ib, wb) = out["1"], out["0"]
    assert ia == ib
    div = [t[0] for t in ia]
    # a chained divergence solve fell short at least once (the first step predicts 2 iterations)
    assert div[0] > 2 or any(div[k + 1] > max(div[k], 2) for k in range(len(div) - 1)), div
    # (no statement about host waits here: the loopback transport waits in every exchange)
    for a, b in zip(ga, gb):
        assert a.n == b.n
        for f in ("particle_id", "position", "velocity", "density", "pressure", "neighbor_count"):
            assert np.array_equal(a.download(f), b.download(f)), f


def test_record_sweeps_with_masses_one_ulp_apart_stay_within_rounding_of_the_generic_sweeps(lab_lib, monkeypatch):
    """OpPressureAccelU / OpJacobiU / OpSourceU take ONE mass for every neighbour (slot 0's, respectively the particle's own): exact when the masses
    are equal.  h = 1.9 sqrt(m / (rho0 pi)) loses a bit, so masses that are neighbouring floats still give bit-identical smoothing
    lengths -- the record path stays on -- and the substitution is then a relative error of one ulp per pair.  Bound it: the same
    steps through the generic sweeps (per-neighbour m_j) agree to a few 1e-6 of the field's scale, counts equal."""
    scn = sc.dam_break_small(128, 96, 1 / 64)
    pos, mass, vel = sc.init_particles(scn)
    rng = np.random.default_rng(7)
    up = np.nextafter(mass, np.float32(np.inf)).astype(np.float32)
    mass2 = np.where(rng.random(len(mass)) < 0.5, mass, up).astype(np.float32)
    h = lambda m: (np.float32(1.9) * np.sqrt(m / np.float32(1.0) / np.float32(np.pi), dtype=np.float32))   # noqa: E731
    if not np.array_equal(h(mass), h(mass2)):   # the perturbation must not break the uniform-h condition (else nothing is tested)
        mass2 = np.where(h(mass2) == h(mass), mass2, mass)
    assert (mass2 != mass).any()
    P = dam_break_params()
    p = P.to_ffi()
    out = {}
    for generic in (False, True):
        if generic:
            monkeypatch.setenv("SPH_ACCEL_GENERIC", "1")
            monkeypatch.setenv("SPH_JACOBI_GENERIC", "1")
            monkeypatch.setenv("SPH_SOURCE_GENERIC", "1")   # (advisor r5: OpSourceU's single mass -- particle 0's -- is bounded by the same bars)
        g = ffi.Context(lab_lib, len(mass2), sc.boundary_planes(scn.boundary, P.init_boundary_handler))
        g.upload(mass2, pos, vel)
        its = []
        for _ in range(10):
            st = g.step(p)
            its.append((int(st.div_solver.iters), int(st.density_solver.iters)))
        out[generic] = (g, its)
        if generic:
            monkeypatch.delenv("SPH_ACCEL_GENERIC")
            monkeypatch.delenv("SPH_JACOBI_GENERIC")
            monkeypatch.delenv("SPH_SOURCE_GENERIC")
    (a, ia), (b, ib) = out[False], out[True]
    assert ia == ib
    for f, tol in (("position", 2e-6), ("velocity", 2e-5), ("density", 2e-6), ("pressure", 2e-4)):
        x, y = a.download(f).astype(np.float64), b.download(f).astype(np.float64)
        assert np.abs(x - y).max() <= tol * max(np.abs(y).max(), 1e-30), (f, np.abs(x - y).max() / np.abs(y).max())


@pytest.mark.parametrize("compression", [1.0, 0.85, 0.66])
@pytest.mark.parametrize("solver", ["HybridDFSPH", "IISPH"])
def test_offset_lists_are_bit_identical_to_the_mask_words(lab_lib, monkeypatch, solver, compression):
    """The two sweeps of a Jacobi iteration on 16-bit relative offsets (k_sweep_off: no mask decoding, no row bases; padding slots =
    the particle itself) against the replay of the mask words (SPH_OFFSET_LISTS=0): same visiting order, same arithmetic, so every
    field and every iteration statistic agree bit for bit over free-running steps.  compression < 1: a lattice squeezed below its
    rest spacing -- 0.85: lists of 17-24 neighbours (the third quad of offsets), 0.66: more than 24 (no offset list: those lanes
    replay their mask words inside the same launch)."""
    nx, ny = 96, 80
    d0 = 1.0 / 64
    # (a squeezed lattice pushes itself apart violently: short steps, few of them -- the point is the list forms, not the flow)
    squeezed = dict(max_dt=0.00002, max_iters=4, hybrid_dfsph_max_avg_density_error=0.0, hybrid_dfsph_max_avg_divergence_error=0.0,
                    iisph_max_avg_density_error=0.0)
    P = dam_break_params(pressure_solver_method=solver, **(dict(max_dt=0.0005) if compression == 1.0 else squeezed))
    scn = sc.dam_break_small(nx, ny, d0)
    pos, mass, vel = sc.init_particles(scn)
    if compression != 1.0:   # the same masses (the same h) on a tighter lattice
        pos = (pos[0] + (pos - pos[0]) * np.float32(compression)).astype(np.float32)
    planes = sc.boundary_planes(scn.boundary, P.init_boundary_handler)
    p = P.to_ffi()
    out = {}
    for form in ("offsets", "masks"):
        if form == "masks":
            monkeypatch.setenv("SPH_OFFSET_LISTS", "0")
        g = ffi.Context(lab_lib, len(mass), planes)   # (the switches are read at sph_create)
        if form == "masks":
            monkeypatch.delenv("SPH_OFFSET_LISTS")
        g.upload(mass, pos, vel)
        its, fields = [], []
        for _ in range(4 if compression != 1.0 else 25):
            st = g.step(p)
            its.append((int(st.div_solver.iters), int(st.density_solver.iters), int(st.density_solver.normal_count),
                        np.float32(st.density_solver.avg_error).view(np.uint32).item(), np.float32(st.dt).view(np.uint32).item()))
            # (a_ii and the constant field: the fused force sweep is on the lists too, with the particle's own W(0) term put where the
            #  mask replay has it)
            fields.append({f: g.download(f) for f in ("position", "velocity", "pressure", "density", "neighbor_count", "aii", "constant_field")})
        out[form] = (its, fields)
        g.close()
    assert out["offsets"][0] == out["masks"][0]
    for s, (fa, fb) in enumerate(zip(out["offsets"][1], out["masks"][1])):
        for f in fa:
            assert np.array_equal(fa[f], fb[f]), (s, f)
    nc = out["offsets"][1][0]["neighbor_count"].max() - 1   # (the reference counts the particle itself)
    assert {1.0: nc <= 16, 0.85: 16 < nc <= 24, 0.66: nc > 24}[compression], nc


@pytest.mark.parametrize("solver", ["HybridDFSPH", "IISPH"])
def test_offset_lists_in_a_multi_resolution_scene(lab_lib, monkeypatch, solver):
    """Two particle sizes (4:1): the bulk has mask lists and replays them as offsets (FAST math: h_ij from the gathered records), the
    interface particles keep their explicit index lists inside the same launches.  Bit for bit the mask replay."""
    fine = 0.02
    scn = sc.SceneConfig(sc.SceneBoundary("box", 3.0, 3.0),
                         [sc.SceneFluidBlock([-0.95, -0.5], [0.55, 1.4], fine, 0.93, [0.5, 0]),
                          sc.SceneFluidBlock([-0.40 + 0.3 * fine * 4, -0.5], [0.7, 1.4], fine * 4, 0.93, [-0.5, 0])])
    pos, mass, vel = sc.init_particles(scn)
    P = dam_break_params(pressure_solver_method=solver)
    planes = sc.boundary_planes(scn.boundary, P.init_boundary_handler)
    p = P.to_ffi()
    out = {}
    for form in ("offsets", "masks"):
        if form == "masks":
            monkeypatch.setenv("SPH_OFFSET_LISTS", "0")
        g = ffi.Context(lab_lib, len(mass), planes)
        if form == "masks":
            monkeypatch.delenv("SPH_OFFSET_LISTS")
        g.upload(mass, pos, vel)
        its, fields = [], []
        for _ in range(12):
            st = g.step(p)
            its.append((int(st.div_solver.iters), int(st.density_solver.iters), np.float32(st.dt).view(np.uint32).item()))
            fields.append({f: g.download(f) for f in ("position", "velocity", "pressure", "density", "aii", "constant_field")})
        forms = g.profile_list_forms()
        out[form] = (its, fields, forms)
        g.close()
    assert out["offsets"][0] == out["masks"][0]
    for s, (fa, fb) in enumerate(zip(out["offsets"][1], out["masks"][1])):
        for f in fa:
            assert np.array_equal(fa[f], fb[f]), (s, f)
    forms = out["offsets"][2]
    assert forms["n_mask"] > 0 and forms["n_index"] > 0, forms     # both list forms are in the scene


def test_the_laboratory_build_runs_the_products_defaults(product_lib, lab_lib):
    """libsph_lab.so is libsph_hip.so's sources with the ablation switches compiled in (-DSPH_LAB); with no switch set it must BE the
    product: the same exported symbols, and 25 free-running steps of a dam break bit for bit -- so that what the tests above establish
    about the laboratory build's default forms holds for the library that ships."""
    assert [n for n in ffi.ABI_SYMBOLS if not hasattr(lab_lib.lib, "sph_" + n)] == []
    scn = sc.dam_break_small(128, 96, 1 / 64)
    pos, mass, vel = sc.init_particles(scn)
    planes = sc.boundary_planes(scn.boundary)
    p = dam_break_params().to_ffi()
    a, b = ffi.Context(product_lib, len(mass), planes), ffi.Context(lab_lib, len(mass), planes)
    a.upload(mass, pos, vel)
    b.upload(mass, pos, vel)
    for s in range(25):
        sa, sb = a.step(p), b.step(p)
        assert (sa.dt, int(sa.div_solver.iters), int(sa.density_solver.iters)) == (sb.dt, int(sb.div_solver.iters), int(sb.density_solver.iters)), s
    for f in ("position", "velocity", "density", "pressure", "aii", "neighbor_count"):
        assert np.array_equal(a.download(f), b.download(f)), f
    # ... and the product IGNORES the laboratory's switches: sph_set_sweep_variant refuses the LDS-staged forms there
    assert product_lib.set_sweep_variant(3) == 30 and lab_lib.set_sweep_variant(0) == 0
