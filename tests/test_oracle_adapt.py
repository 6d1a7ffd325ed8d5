"""The apply half of single_step_adaptivity in the CPU oracle (oracle/adapt.c) against small models written straight from the
reference's text (adaptivity/particle_sharing.rs:152-253, particle_merging.rs:270-385, splitting.rs:19-82), and the host-side
decision code (adaptive_sph_amd/adaptivity.py) on the oracle: no GPU."""
from pathlib import Path

import numpy as np
import pytest

from adaptive_sph_amd import adaptivity as A, ffi, scene as sc
from adaptive_sph_amd.workloads import default_params

REPO = Path(__file__).resolve().parent.parent
AV, DEL = ffi.MERGE_PARTNER_AVAILABLE, ffi.MERGE_PARTNER_DELETE


def level_ctx(oracle_lib, n_capacity=None, **overrides):
    """default scene stepped twice on the oracle: every particle has a level value, two particle sizes"""
    scn = sc.SceneConfig.from_yaml(str(REPO / "tests" / "golden" / "default-scene.yaml"))
    pos, mass, vel = sc.init_particles(scn)
    c = ffi.Context(oracle_lib, n_capacity or 4 * len(mass), sc.boundary_planes(scn.boundary))
    c.upload(mass, pos, vel)
    P = default_params(**overrides)
    p = P.to_ffi()
    for _ in range(2):
        st = c.step(p)
    return c, P, p, float(st.dt)


def test_split_patterns_file():
    sp = A.SplitPatterns.load_from_file(REPO / "tests" / "golden" / "split-patterns.yaml")
    assert sp.get_max_num_children() == 59 and sp.get(2).shape == (2, 2) and sp.get(59).shape == (59, 2)
    with pytest.raises(KeyError):
        sp.get(60)
    # a 1-to-2 split puts the children on opposite sides of the parent
    assert np.allclose(sp.get(2)[0], -sp.get(2)[1], atol=1e-6)


def merge_model(mass, pos, vel, h2n, partner, counter, min_partners, rho0):
    """particle_merging.rs:270-370 on python lists of per-particle records (everything swaps together)."""
    f32 = np.float32
    n = len(mass)
    mass, pos, vel, h2n = mass.copy(), pos.copy(), vel.copy(), h2n.copy()
    ident = np.arange(n)
    partner, counter = partner.copy(), counter.copy()
    m0, x0, v0 = mass.copy(), pos.copy(), vel.copy()
    for i in range(n):
        j = partner[i]
        if j in (AV, DEL) or counter[j] < min_partners:
            continue
        mass_n = f32(m0[j]) / f32(counter[j])
        m = f32(m0[i]) + mass_n
        vel[i] = (f32(m0[i]) * v0[i] + mass_n * v0[j]) / m
        pos[i] = (f32(m0[i]) * x0[i] + mass_n * x0[j]) / m
        mass[i] = m
        h2n[i] = f32(1.9) * np.sqrt((m / f32(rho0)) * f32(1 / np.pi), dtype=np.float32)
    last, i = n - 1, 0
    while i <= last:
        if partner[i] == DEL and counter[i] >= min_partners:
            mass[i] = mass[i] - mass[i]
            if mass[i] < 0.000001:
                for a in (mass, pos, vel, h2n, ident, partner, counter):
                    a[[i, last]] = a[[last, i]]
                last -= 1
                continue
        i += 1
    k = last + 1
    return mass[:k], pos[:k], vel[:k], h2n[:k], ident[:k]


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_merge_particles_matches_the_sequential_model(oracle_lib, seed):
    rng = np.random.default_rng(seed)
    c, P, p, dt = level_ctx(oracle_lib)
    n = c.n
    # random donors with 1-3 receivers each (any particles: the apply half does not look at classes or distances)
    partner = np.full(n, AV, np.uint32)
    counter = np.zeros(n, np.uint16)
    free = list(rng.permutation(n))
    for _ in range(60 if seed else 1):
        d = free.pop()
        k = int(rng.integers(1, 4))
        partner[d] = DEL
        for _ in range(k):
            partner[free.pop()] = d
        counter[d] = k
    if seed == 3:   # donors at the very end of the vector: holes and tail interleave
        for d in range(n - 5, n):
            if partner[d] == AV:
                r = free.pop()
                while r >= n - 5:
                    r = free.pop()
                partner[d], partner[r], counter[d] = DEL, d, 1
    before = {f: c.download(f) for f in ("mass", "position", "velocity", "h2_next")}
    ap = A.adapt_params(P, dt)
    ap.minimum_merge_partners = 2 if seed == 2 else 0
    c.merge_particles(p, ap, partner, counter)
    m, x, v, hn, ident = merge_model(before["mass"], before["position"], before["velocity"], before["h2_next"], partner, counter,
                                     int(ap.minimum_merge_partners), P.rest_density)
    assert c.n == len(m) < n
    assert np.array_equal(c.download("mass"), m)
    assert np.array_equal(c.download("position"), x) and np.array_equal(c.download("velocity"), v)
    assert np.allclose(c.download("h2_next"), hn, rtol=1e-6)
    assert abs(float(m.sum()) - float(before["mass"].sum())) < 1e-6 * before["mass"].sum()     # mass conserved


def test_share_particles(oracle_lib):
    c, P, p, dt = level_ctx(oracle_lib)
    n = c.n
    mass, pos, lvl = c.download("mass"), c.download("position"), c.download("level_estimation")
    target = A.target_mass(lvl, P)
    donors = np.nonzero(mass > target)[0][:20]
    assert len(donors) == 20
    partner = np.full(n, AV, np.uint32)
    counter = np.zeros(n, np.uint16)
    others = [i for i in range(n) if i not in set(donors.tolist())]
    for k, d in enumerate(donors):
        partner[d] = DEL
        partner[others[2 * k]] = d
        partner[others[2 * k + 1]] = d
        counter[d] = 2
    ap = A.adapt_params(P, dt)
    c.share_particles(p, ap, partner, counter)
    m2 = c.download("mass")
    f32 = np.float32
    for d in donors:
        dropped = min(f32(mass[d]) - target[d], target[d] * f32(P.max_mass_transfer_sharing) * f32(dt))
        assert m2[d] == f32(mass[d]) - dropped
        for r in np.nonzero(partner == d)[0]:
            assert m2[r] == f32(mass[r]) + dropped / f32(2)
    assert c.n == n and abs(float(m2.sum()) - float(mass.sum())) < 1e-6 * mass.sum()
    untouched = (partner == AV)
    assert np.array_equal(c.download("position")[untouched], pos[untouched])


def test_split_particles(oracle_lib):
    sp = A.SplitPatterns.load_from_file(REPO / "tests" / "golden" / "split-patterns.yaml")
    c, P, p, dt = level_ctx(oracle_lib, n_capacity=70000)
    n = c.n
    c.set_split_patterns(sp.patterns)
    c.classify(p)
    cls = c.download("particle_size_class")
    mass, pos, vel, lvl, lvo = (c.download(f) for f in ("mass", "position", "velocity", "level_estimation", "level_old"))
    parents = np.nonzero(cls == 4)[0]
    assert len(parents) > 10
    target = A.target_mass(lvl, P)
    nchild = np.minimum(np.round(mass[parents] / target[parents]).astype(np.int64), sp.get_max_num_children())
    ap = A.adapt_params(P, dt)
    c.split_particles(p, ap)
    assert c.n == n + int((nchild - 1).sum())
    m2, x2, v2, l2, lo2, h2, h2n = (c.download(f) for f in ("mass", "position", "velocity", "level_estimation", "level_old", "h2", "h2_next"))
    q = n
    f32 = np.float32
    for par, k in zip(parents, nchild):
        pat = sp.get(int(k))
        radius = np.sqrt((f32(mass[par]) / f32(1.0)) * f32(1 / np.pi), dtype=np.float32)
        cm = f32(mass[par]) / f32(k)
        assert m2[par] == cm and np.array_equal(x2[par], pos[par] + pat[0] * radius)
        assert lo2[par] == lvo[par] and h2[par] == h2n[par] > 0
        for child in range(1, int(k)):                       # appended in the order of the parents' indices
            assert m2[q] == cm and np.array_equal(x2[q], pos[par] + pat[child] * radius)
            assert np.array_equal(v2[q], vel[par]) and l2[q] == lvl[par]
            assert h2[q] == 0.0 and lo2[q] == 0.0 and h2n[q] == h2n[par]      # the reference's quirk (splitting.rs:73, 76)
            q += 1
    assert q == c.n
    assert abs(float(m2.sum()) - float(mass.sum())) < 1e-6 * mass.sum()
    # fail_on_missing_split_pattern with a table that is too small
    c2, P2, p2, dt2 = level_ctx(oracle_lib, n_capacity=70000)
    c2.set_split_patterns(sp.patterns[:1])
    c2.classify(p2)
    ap2 = A.adapt_params(P2.replace(fail_on_missing_split_pattern=True), dt2)
    with pytest.raises(ffi.SphError) as e:
        c2.split_particles(p2, ap2)
    assert e.value.status == 27


def test_adaptive_steps_on_the_oracle(oracle_lib):
    """configs[0] as the reference runs it: default-config.yaml (merging, sharing, splitting on) + default-scene.yaml, the
    decisions by adaptivity.py, the data path by the oracle: particles are split at the surface and merged in the bulk, the
    mass sum stays (single_step_adaptivity's own assertion), every step keeps working on the edited vector."""
    from adaptive_sph_amd.simulation import init_fluid_sim
    P = default_params()
    scn = sc.SceneConfig.from_yaml(str(REPO / "tests" / "golden" / "default-scene.yaml"))
    sp = A.SplitPatterns.load_from_file(REPO / "tests" / "golden" / "split-patterns.yaml")
    sim = init_fluid_sim(P, scn, lib=oracle_lib, split_patterns=sp, n_capacity=40000)
    m0 = float(sim.particles.mass.sum())
    n0, ns = sim.num_fluid_particles(), []
    events = {"shares": 0, "merges": 0, "splits": 0}
    for s in range(6):
        dt = sim.single_step_without_adaptivity(P)
        info = sim.single_step_adaptivity(P, dt)
        for k in events:
            events[k] += info[k]
        ns.append(sim.num_fluid_particles())
    assert n0 == 1035 and ns[-1] != n0
    assert events["splits"] > 0 and (events["merges"] > 0 or events["shares"] > 0), events
    assert abs(float(sim.particles.mass.sum()) - m0) < 0.005
    x = sim.particles.position
    assert np.isfinite(x).all() and np.abs(x).max() < 1.0
