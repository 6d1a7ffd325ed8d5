"""The oracle's step against the reference's built-in self-checks and conservation properties (no GPU).

check_neighborhood (neighborhood_search.rs:187-238, simulation.rs:1810-1863) and check_aii
(simulation.rs:1347-1375, tol 0.01) are the reference's own runtime oracles; enabling them makes the
oracle verify its lists against the O(N^2) definition and its analytic a_ii against apply-operator-to-e_i.
"""
import numpy as np
import pytest

from adaptive_sph_amd import ffi, scene as sc
from adaptive_sph_amd.workloads import dam_break_params, default_params
from tests.oracle_harness import csr_sets, ring_scene, REPO


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
    assert (c.download("particle_size_class") == 2).all()    # the step never classifies (simulation.rs:2749-2778): default Optimal
    c.classify(p)
    cls = c.download("particle_size_class")
    assert set(np.unique(cls)) <= {0, 1, 2, 3, 4} and len(np.unique(cls)) > 1


def _edit_script(rng, n):
    """What merge (value writes, swap-to-end deletes, truncate) and split (extend, child writes) do, as a script."""
    ops, length = [], n
    for i in rng.choice(n, 12, replace=False):          # sharing / merging: mass, position, velocity, h2_next of survivors
        ops.append(("set", int(i), {"mass": float(rng.uniform(1e-4, 3e-4)), "position": rng.uniform(-0.5, 0.5, 2),
                                    "velocity": rng.uniform(-1, 1, 2), "h2_next": float(rng.uniform(0.01, 0.03))}))
    last = n - 1
    for i in sorted(rng.choice(n // 2, 9, replace=False)):   # merge_particles: delete by swapping to the end
        ops.append(("swap", int(i), last))
        last -= 1
    ops.append(("truncate", last + 1))
    length = last + 1
    ops.append(("extend", 7))                           # split_particles: children appended, then written
    for q in range(7):
        ops.append(("set", length + q, {"mass": 1.5e-4, "position": rng.uniform(-0.5, 0.5, 2), "velocity": [0.1, -0.2],
                                        "h2_next": 0.02, "level_estimation": -0.05, "level_old": -0.04}))
    ops.append(("set", 3, {"h2": 0.0123, "level_old": -0.5}))
    return ops, length + 7


def _apply_model(ops, fields):
    """The same script on plain Python lists (Vec semantics)."""
    defaults = {"mass": 0.0, "position": [0.0, 0.0], "velocity": [0.0, 0.0], "h2": 0.0, "h2_next": 0.0,
                "level_estimation": float("nan"), "level_old": 0.0}
    rows = [{k: (list(v[i]) if np.ndim(v[i]) else float(v[i])) for k, v in fields.items()} for i in range(len(fields["mass"]))]
    for op in ops:
        if op[0] == "set":
            for k, v in op[2].items():
                rows[op[1]][k] = [float(np.float32(x)) for x in v] if np.ndim(v) else float(np.float32(v))
        elif op[0] == "swap":
            rows[op[1]], rows[op[2]] = rows[op[2]], rows[op[1]]
        elif op[0] == "truncate":
            del rows[op[1]:]
        else:
            rows += [dict((k, list(v) if isinstance(v, list) else v) for k, v in defaults.items()) for _ in range(op[1])]
    return {k: np.array([r[k] for r in rows], dtype=np.float32) for k in defaults}


EDIT_FIELDS = ["mass", "position", "velocity", "h2", "h2_next", "level_estimation", "level_old"]


def test_sparse_edits_follow_vec_semantics(oracle_lib):
    """sph_apply_edits (include/sph_ffi.h): SET / SWAP / TRUNCATE / EXTEND in host index space, like the ParticleVec calls of
    merge_particles and split_particles (particle_merging.rs:341-370, splitting.rs:52-79)."""
    scn = sc.dam_break_small(20, 20, 1 / 20)
    c, pos, mass, vel = _ctx(oracle_lib, scn)
    p = dam_break_params(level_estimation_method="EmptyAngle", maximum_surface_distance=0.2).to_ffi()
    c.step(p)
    before = {f: c.download(f) for f in EDIT_FIELDS}
    ops, n_new = _edit_script(np.random.default_rng(5), len(mass))
    c.apply_edits(ops)
    assert c.n == n_new
    want = _apply_model(ops, before)
    for f in EDIT_FIELDS:
        assert np.array_equal(c.download(f), want[f], equal_nan=True), f
    with pytest.raises(ffi.SphError):
        c.apply_edits([("swap", 0, n_new)])          # out of bounds, like the Vec index panic


def test_constrain_neighborhood_count(oracle_lib):
    """simulation.rs:2145-2177.  (a) nobody above 19 neighbours: the step equals the unconstrained one, h2_next := h2,
    no flags.  (b) the ring: the centre takes fringe[21 - 19] = 0.4 h.  (c) a compressed lattice: the reference's own
    assertion `*p_h_next < h` fires (the far neighbours' fringe values are ~2 h)."""
    scn = sc.dam_break_small(24, 24, 1 / 24)
    ca, *_ = _ctx(oracle_lib, scn)
    cb, *_ = _ctx(oracle_lib, scn)
    for _ in range(2):
        sa = ca.step(dam_break_params().to_ffi())
        sb = cb.step(dam_break_params(constrain_neighborhood_count=True).to_ffi())
    assert sa.dt == sb.dt
    for f in ("position", "velocity", "density", "h2"):
        assert np.array_equal(ca.download(f), cb.download(f)), f
    assert np.array_equal(cb.download("h2_next"), cb.download("h2"))
    assert not cb.download("flag_neighborhood_reduced").any()

    pos, mass, vel = ring_scene()
    planes = sc.boundary_planes(sc.SceneBoundary("box", 4.0, 4.0), "AnalyticOverestimate")
    c = ffi.Context(oracle_lib, len(mass), planes)
    c.upload(mass, pos, vel)
    p = dam_break_params(constrain_neighborhood_count=True, gravity=0.0)
    st = c.step(p.to_ffi())
    h0 = np.float32(0.05)
    h = c.download("h2")
    flag = c.download("flag_neighborhood_reduced")
    cnt = c.download("neighbor_count")
    assert cnt[0] == 21 and cnt[1:].max() <= 19
    assert flag[0] == 1 and not flag[1:].any()
    assert abs(h[0] - 0.4 * h0) < 1e-5 and np.allclose(h[1:], h0, rtol=1e-6)
    assert np.allclose(c.download("h2_next"), h0, rtol=1e-6)      # mem::swap: the unconstrained values
    assert st.dt <= np.float32(p.max_dt) and np.isfinite(c.download("density")).all()

    sq = sc.dam_break_small(24, 24, 1 / 24)
    pos, mass, vel = sc.init_particles(sq)
    pos = (pos * np.float32(0.8)).astype(np.float32)               # 1.56 x the rest density: ~21 neighbours
    c = ffi.Context(oracle_lib, len(mass), sc.boundary_planes(sq.boundary, "AnalyticOverestimate"))
    c.upload(mass, pos, vel)
    with pytest.raises(ffi.SphError) as e:
        c.step(dam_break_params(constrain_neighborhood_count=True).to_ffi())
    assert e.value.status == 25      # SPH_ERR_CONSTRAIN_NOT_SMALLER
