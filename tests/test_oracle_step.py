"""The oracle's step against the reference's built-in self-checks and conservation properties (no GPU).

check_neighborhood (neighborhood_search.rs:187-238, simulation.rs:1810-1863) and check_aii
(simulation.rs:1347-1375, tol 0.01) are the reference's own runtime oracles; enabling them makes the
oracle verify its lists against the O(N^2) definition and its analytic a_ii against apply-operator-to-e_i.
"""
import numpy as np
import pytest

from adaptive_sph_amd import ffi, scene as sc
from adaptive_sph_amd.workloads import dam_break_params, default_params
from tests.oracle_harness import csr_sets, REPO


def _ctx(lib, scn, handler="AnalyticOverestimate"):
    pos, mass, vel = sc.init_particles(scn)
    c = ffi.Context(lib, len(mass), sc.boundary_planes(scn.boundary, handler))
    c.upload(mass, pos, vel)
    return c, pos, mass, vel


@pytest.mark.parametrize("op", ["ConsistentSimpleGradient", "ConsistentSymmetricGradient", "Winchenbach2020"])
def test_self_checks_pass(oracle_lib, op):
    c, pos, mass, vel = _ctx(oracle_lib, sc.dam_break_small(24, 24, 1 / 24))
    p = dam_break_params(check_neighborhood=True, check_aii=True, operator_discretization=op).to_ffi()
    for _ in range(3):
        st = c.step(p)
    rho = c.download("density")
    assert np.all(np.isfinite(rho)) and rho.max() < 1.2
    assert st.div_solver.iters >= 2     # minimum 3 iterations: returned index >= 2
    assert np.all(c.download("aii") >= 0)


def test_neighbor_lists_symmetric_self_included_sorted(oracle_lib):
    scn = sc.SceneConfig.from_yaml(str(REPO / "tests" / "golden" / "default-scene.yaml"))   # 2:1 radii
    c, pos, mass, vel = _ctx(oracle_lib, scn)
    c.step(dam_break_params().to_ffi())
    off, idx = c.download_neighbors()
    sets = csr_sets(off, idx)
    cnt = c.download("neighbor_count")
    assert np.array_equal(np.diff(off), cnt)
    for i, s in enumerate(sets):
        assert i in s and len(np.unique(s)) == len(s)
    pairs = {(i, int(j)) for i, s in enumerate(sets) for j in s}
    assert all((j, i) in pairs for (i, j) in pairs)


def test_rest_lattice_has_13_neighbors(oracle_lib):
    c, *_ = _ctx(oracle_lib, sc.dam_break_small(32, 32, 1 / 32))
    c.step(dam_break_params().to_ffi())
    cnt = c.download("neighbor_count").reshape(32, 32)
    assert np.all(cnt[3:-3, 3:-3] == 13)          # k = 13 on the rest lattice
    g = c.grid()
    h = c.download("h2")
    assert g.cell_size == np.float32(h.max()) * np.float32(2.0)


def test_mass_and_momentum_sanity(oracle_lib):
    c, pos, mass, vel = _ctx(oracle_lib, sc.dam_break_small(24, 24, 1 / 24))
    p = dam_break_params().to_ffi()
    t = 0.0
    for _ in range(20):
        st = c.step(p)
        t += st.dt
    assert abs(c.time - t) < 1e-6
    assert np.array_equal(c.download("mass"), mass)
    v = c.download("velocity")
    assert v[:, 1].mean() < 0            # gravity pulls the column down
    x = c.download("position")
    assert x[:, 0].min() > -2.01 and x[:, 1].min() > -1.01   # stays inside the box


def test_solver_modes_and_errors(oracle_lib):
    scn = sc.dam_break_small(16, 16, 1 / 16)
    for mode in ("IISPH", "OnlyDivergence", "HybridDFSPH"):
        c, *_ = _ctx(oracle_lib, scn)
        for _ in range(2):
            st = c.step(dam_break_params(pressure_solver_method=mode).to_ffi())
        assert np.all(np.isfinite(c.download("position")))
    c, *_ = _ctx(oracle_lib, scn)
    with pytest.raises(ffi.SphError) as e:
        c.step(dam_break_params(viscosity_type="XSPH").to_ffi())
    assert e.value.status == 20
    c, *_ = _ctx(oracle_lib, scn, "NoBoundary")
    with pytest.raises(ffi.SphError) as e:
        c.step(dam_break_params().to_ffi())
    assert e.value.status == 4


def test_level_estimation_default_config(oracle_lib):
    """configs[0]: default-config.yaml + default-scene.yaml (EmptyAngle level set) steps on the CPU path."""
    scn = sc.SceneConfig.from_yaml(str(REPO / "tests" / "golden" / "default-scene.yaml"))
    c, pos, mass, vel = _ctx(oracle_lib, scn)
    assert len(mass) == 1035
    p = default_params().to_ffi()
    for _ in range(3):
        c.step(p)
    lvl = c.download("level_estimation")
    assert np.all(np.isfinite(lvl)) and np.all(lvl <= 1e-6)     # smoothed level set: everyone has a distance <= 0
    surf = c.download("flag_is_fluid_surface")
    assert 0 < surf.sum() < len(mass)
    cls = c.download("particle_size_class")
    assert set(np.unique(cls)) <= {0, 1, 2, 3, 4}
