"""Every recipe of the reference's media/ folder (58 distinct `update_attributes` x scene combinations, collected as DATA into
tests/golden/media_recipes.json by tests/golden/make_media_fixture.py): default-config.yaml + the recipe's overrides + its scene,
stepped on the HIP path and on the CPU oracle.  The host-side adaptivity is switched off (tests/test_gpu_adaptivity.py steps it),
everything else is the recipe's own: solver mode, boundary handler, support-length estimation, sizing, level estimation."""
import json
from pathlib import Path

import numpy as np
import pytest

from adaptive_sph_amd import ffi, scene as sc
from adaptive_sph_amd.workloads import default_params

pytestmark = pytest.mark.gpu

RECIPES = json.loads((Path(__file__).parent / "golden" / "media_recipes.json").read_text())
REL_TOL_FIELDS = 1e-4
TOL = {"velocity": 1e-3}


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    s = np.abs(b).max()
    return np.abs(a - b).max() / (s if s > 0 else 1.0)


@pytest.mark.parametrize("k", range(len(RECIPES)), ids=[r["recipe"] for r in RECIPES])
def test_media_recipe(product_lib, oracle_lib, k):
    r = RECIPES[k]
    attrs = dict(r["update_attributes"])
    attrs.update(merging=False, sharing=False, splitting=False)
    # (a key default-config.yaml does not hold -- `fill_stash_with` in surface-distance.yaml -- is the reference's own
    #  panic "not able to find attribute", animation/mod.rs:89-95, and the mirror's KeyError; it is set directly here so that
    #  the recipe still runs)
    from adaptive_sph_amd.workloads import DEFAULT_CONFIG
    extra = {k: attrs.pop(k) for k in list(attrs) if k not in DEFAULT_CONFIG}
    if extra:
        with pytest.raises(KeyError, match="not able to find attribute"):
            default_params(**dict(attrs, **extra))
    P = default_params(**attrs).replace(**extra)
    scn = sc.SceneConfig.from_mapping(r["scene"])
    pos, mass, vel = sc.init_particles(scn)
    planes = sc.boundary_planes(scn.boundary, P.init_boundary_handler)
    g, o = ffi.Context(product_lib, len(mass), planes), ffi.Context(oracle_lib, len(mass), planes)
    g.upload(mass, pos, vel)
    o.upload(mass, pos, vel)
    P = P.replace(max_iters=4, iisph_max_avg_density_error=0.0, hybrid_dfsph_max_avg_density_error=0.0, hybrid_dfsph_max_avg_divergence_error=0.0)
    p = P.to_ffi()
    def lists_equal():
        assert np.array_equal(g.download("neighbor_count"), o.download("neighbor_count"))
        assert np.array_equal(g.download("cell_index"), o.download("cell_index"))
        go, gi = g.download_neighbors()
        oo, oi = o.download_neighbors()
        assert np.array_equal(go, oo)
        starts = go[:-1].astype(np.int64)
        for power in (1, 2):
            assert np.array_equal(np.add.reduceat(gi.astype(np.uint64) ** power, starts), np.add.reduceat(oi.astype(np.uint64) ** power, starts))

    def fields_close(fields, tol_scale=1.0):
        for f in fields:
            assert rel_err(g.download(f), o.download(f)) < tol_scale * TOL.get(f, REL_TOL_FIELDS), f

    status = []
    for c in (g, o):
        try:
            st = c.step(p)
            status.append((0, st.dt, int(st.div_solver.iters), int(st.density_solver.iters)))
        except ffi.SphError as e:
            status.append((e.status, None, None, None))
    if status[0][0] or status[1][0]:
        assert status[0][0] == status[1][0], status      # a recipe the reference itself refuses (its assert / todo): the same code on both sides
        return
    # ---- step 1: identical inputs -> cells, counts and neighbour sets bit for bit, every field within tolerance
    assert status[0][1] == status[1][1] and status[0][2:] == status[1][2:]
    if P.support_length_estimation == "FromMass":
        lists_equal()
    else:   # FromDistribution*: h comes out of a sum already in the first step
        assert (g.download("neighbor_count") != o.download("neighbor_count")).mean() < 0.02
        assert rel_err(g.download("h2"), o.download("h2")) < 1e-5
    fields_close(("position", "velocity", "density", "aii", "ppe_source_term", "lambda_sum"))
    level = P.level_estimation_method != "None" and P.support_length_estimation == "FromMass"
    if level:
        assert np.array_equal(g.download("flag_is_fluid_surface"), o.download("flag_is_fluid_surface"))
        a, b = g.download("level_estimation"), o.download("level_estimation")
        assert np.array_equal(np.isnan(a), np.isnan(b))
        if np.isfinite(b).any():
            assert float(np.nanmax(np.abs(a - b))) <= REL_TOL_FIELDS * max(float(np.nanmax(np.abs(b))), 1e-30)
    # ---- two more steps, free-running: positions and densities stay within the bar (the two sides' inputs now differ in the
    # last bits, so a particle ON a cell or support boundary may fall on either side: no bit-exact statement here)
    for s in range(2):
        sg, so = g.step(p), o.step(p)
        assert abs(sg.dt - so.dt) <= 1e-5 * so.dt
    fields_close(("position", "density"))
    fields_close(("velocity",), 10.0)       # carries the unconverged (4 iterations), clamped pressure field of three steps


ADAPTIVE = [k for k, r in enumerate(RECIPES) if any(r["update_attributes"].get(f) for f in ("merging", "sharing", "splitting"))]


@pytest.mark.parametrize("k", ADAPTIVE, ids=[RECIPES[k]["recipe"] for k in ADAPTIVE])
def test_media_recipe_with_its_adaptivity(product_lib, k):
    """The recipes that resample (merging / sharing / splitting as the recipe sets them), the way the reference runs them:
    single_step = the step on the device + single_step_adaptivity (decisions on the host, data on the device, split patterns from
    the reference's table).  Eight steps; what the reference itself asserts afterwards -- mass conservation (simulation.rs:2792,
    tolerance 5e-3 per call) -- plus finite positions inside the box."""
    from adaptive_sph_amd import adaptivity as A
    from adaptive_sph_amd.simulation import init_fluid_sim
    from adaptive_sph_amd.workloads import DEFAULT_CONFIG
    r = RECIPES[k]
    attrs = {kk: v for kk, v in r["update_attributes"].items() if kk in DEFAULT_CONFIG}
    P = default_params(**attrs).replace(**{kk: v for kk, v in r["update_attributes"].items() if kk not in DEFAULT_CONFIG})
    scn = sc.SceneConfig.from_mapping(r["scene"])
    sp = A.SplitPatterns.load_from_file(Path(__file__).parent / "golden" / "split-patterns.yaml")
    n0 = len(sc.init_particles(scn)[1])
    sim = init_fluid_sim(P, scn, lib=product_lib, split_patterns=sp, n_capacity=max(40 * n0, 60000) if n0 < 20000 else 4 * n0)
    m0 = float(sim.particles.mass.sum(dtype=np.float64))
    counts = []
    try:
        for s in range(8):
            sim.single_step(P)
            counts.append(sim.num_fluid_particles())
    except ffi.SphError as e:
        # the reference's own refusals for a recipe as written (e.g. CenterDiff before advection): nothing to compare
        assert e.status in (1, 27), e
        return
    x = sim.particles.position
    half_w, half_h = 0.5 * scn.boundary.width, 0.5 * scn.boundary.height
    # (a split places children around the parent and the semi-analytic wall is soft: a particle may sit a few percent outside)
    assert np.isfinite(x).all() and np.abs(x[:, 0]).max() < 1.05 * half_w and np.abs(x[:, 1]).max() < 1.05 * half_h
    assert abs(float(sim.particles.mass.sum(dtype=np.float64)) - m0) < 0.005 * 8
    assert min(counts) > 0
