"""Adaptivity data path on the device (sph_share_particles / sph_merge_particles / sph_split_particles, sph_ffi.h) against the
CPU oracle's statement-by-statement restatement of share_particles / merge_particles / split_particles (oracle/adapt.c), with
the partner decisions taken ONCE (adaptivity.py, from the oracle's state) and applied to both sides; then whole adaptive runs
(single_step = step + single_step_adaptivity) of BASELINE configs[0] and of configs[4]'s scene."""
from pathlib import Path

import numpy as np
import pytest

from adaptive_sph_amd import adaptivity as A, ffi, scene as sc
from adaptive_sph_amd.simulation import init_fluid_sim
from adaptive_sph_amd.workloads import WORKLOADS, default_params

pytestmark = pytest.mark.gpu
REPO = Path(__file__).resolve().parent.parent
PATTERNS = REPO / "tests" / "golden" / "split-patterns.yaml"


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    s = np.abs(b).max()
    return np.abs(a - b).max() / (s if s > 0 else 1.0)


def pair(product_lib, oracle_lib, cap=70000, steps=2, **overrides):
    scn = sc.SceneConfig.from_yaml(str(REPO / "tests" / "golden" / "default-scene.yaml"))
    pos, mass, vel = sc.init_particles(scn)
    planes = sc.boundary_planes(scn.boundary)
    g, o = ffi.Context(product_lib, cap, planes), ffi.Context(oracle_lib, cap, planes)
    g.upload(mass, pos, vel)
    o.upload(mass, pos, vel)
    P = default_params(**overrides)
    p = P.to_ffi()
    for _ in range(steps):
        sg, so = g.step(p), o.step(p)
    return g, o, P, p, float(so.dt)


def same_state(g, o, tol=1e-5):
    assert g.n == o.n
    for f in ("mass", "position", "velocity", "h2_next", "level_old"):
        assert rel_err(g.download(f), o.download(f)) < tol, f
    a, b = g.download("level_estimation"), o.download("level_estimation")
    assert np.array_equal(np.isnan(a), np.isnan(b))
    assert np.nanmax(np.abs(a - b)) <= 1e-4 * max(np.nanmax(np.abs(b)), 1e-30)


def decisions(o, kind, P, p, dt):
    o.classify(p)
    cls = o.download("particle_size_class")
    off, idx = o.download_neighbors()
    mp, mc = A._find_partners(kind, cls, o.download("mass"), o.download("level_estimation"), o.download("position"), o.download("h2"), off, idx, P, dt)
    A.validate_partners(kind, cls, mp, mc, off, idx)
    return cls, off, idx, mp, mc


@pytest.mark.parametrize("kind", ["share", "merge"])
def test_share_and_merge_match_the_oracle(product_lib, oracle_lib, kind):
    # radii that put the two particle sizes of the default scene on both sides of the class thresholds
    g, o, P, p, dt = pair(product_lib, oracle_lib, particle_radius_fine=0.012, particle_radius_base=0.05, maximum_surface_distance=0.3)
    cls, off, idx, mp, mc = decisions(o, kind, P, p, dt)
    assert mc.sum() >= (5 if kind == "share" else 100), (np.bincount(cls, minlength=5), mc.sum())
    # the compiled partner search of the library is the same loop
    mp2, mc2 = A.find_partners_native(product_lib, kind, cls, o.download("mass"), o.download("level_estimation"), o.download("position"),
                                      o.download("h2"), off, idx, P, dt)
    assert np.array_equal(mp, mp2) and np.array_equal(mc, mc2)
    ap = A.adapt_params(P, dt)
    n0, m0 = o.n, float(o.download("mass").sum())
    for c in (g, o):
        (c.share_particles if kind == "share" else c.merge_particles)(p, ap, mp, mc)
    assert (o.n < n0) == (kind == "merge")
    same_state(g, o)
    assert abs(float(g.download("mass").sum()) - m0) < 1e-5 * m0
    with pytest.raises(ffi.SphError):
        g.download_neighbors()                          # the lists belong to the state before the transfer
    for _ in range(2):                                  # both sides keep stepping on the edited vector
        sg, so = g.step(p), o.step(p)
        assert abs(sg.dt - so.dt) <= 1e-5 * so.dt
    assert (g.download("neighbor_count") != o.download("neighbor_count")).mean() < 0.01
    for f in ("position", "velocity", "density"):
        assert rel_err(g.download(f), o.download(f)) < 1e-3, f


def test_merge_deletion_order_with_random_partners(product_lib, oracle_lib):
    """The swap-with-the-last loop of merge_particles in its closed form (prefix sums) against the sequential loop: random
    donors anywhere in the vector, also in its tail, minimum_merge_partners leaving some donors alone."""
    for seed in range(4):
        rng = np.random.default_rng(seed)
        g, o, P, p, dt = pair(product_lib, oracle_lib)
        n = o.n
        partner = np.full(n, ffi.MERGE_PARTNER_AVAILABLE, np.uint32)
        counter = np.zeros(n, np.uint16)
        free = list(rng.permutation(n))
        tail_first = sorted(free, reverse=True)[:40] if seed % 2 else []
        for d in tail_first + [free.pop() for _ in range(150)]:
            if partner[d] != ffi.MERGE_PARTNER_AVAILABLE:
                continue
            k = int(rng.integers(1, 4))
            recv = []
            while len(recv) < k and free:
                r = free.pop()
                if partner[r] == ffi.MERGE_PARTNER_AVAILABLE and r != d:
                    recv.append(r)
            partner[d] = ffi.MERGE_PARTNER_DELETE
            for r in recv:
                partner[r] = d
            counter[d] = len(recv)
        ap = A.adapt_params(P, dt)
        ap.minimum_merge_partners = 2 if seed >= 2 else 0
        ident = np.arange(n, dtype=np.float32)           # level_old carries each particle's old index through the reordering
        for c in (g, o):
            c.upload_field("level_old", ident)
            c.merge_particles(p, ap, partner, counter)
        assert g.n == o.n < n
        assert np.array_equal(g.download("level_old"), o.download("level_old"))     # the same particle in every slot
        same_state(g, o)


def test_split_matches_the_oracle(product_lib, oracle_lib):
    sp = A.SplitPatterns.load_from_file(PATTERNS)
    g, o, P, p, dt = pair(product_lib, oracle_lib)
    for c in (g, o):
        c.set_split_patterns(sp.patterns)
    # identical inputs for the rounding in num_children = round(mass / target_mass): the oracle's level field and classes
    o.classify(p)
    g.upload_field("level_estimation", o.download("level_estimation"))
    g.upload_field("particle_size_class", o.download("particle_size_class"))
    n0 = o.n
    ap = A.adapt_params(P, dt)
    for c in (g, o):
        c.split_particles(p, ap)
    assert g.n == o.n > 2 * n0
    for f in ("mass", "h2", "h2_next", "particle_size_class"):
        assert np.array_equal(g.download(f), o.download(f)), f          # IEEE operations on identical inputs
    lo_g, lo_o = g.download("level_old"), o.download("level_old")        # carried (each side's own smoothed field), 0 for the appended children
    assert np.array_equal(lo_g == 0, lo_o == 0) and rel_err(lo_g, lo_o) < 1e-4
    assert rel_err(g.download("position"), o.download("position")) < 1e-6
    assert rel_err(g.download("velocity"), o.download("velocity")) < 1e-5
    for _ in range(2):
        sg, so = g.step(p), o.step(p)
        assert abs(sg.dt - so.dt) <= 1e-4 * so.dt
    assert rel_err(g.download("density"), o.download("density")) < 1e-3
    # no patterns set / table too small with fail_on_missing_split_pattern
    g2, o2, P2, p2, dt2 = pair(product_lib, oracle_lib, cap=5000)
    g2.classify(p2)
    with pytest.raises(ffi.SphError) as e:
        g2.split_particles(p2, A.adapt_params(P2, dt2))
    assert e.value.status == 27
    g2.set_split_patterns(sp.patterns[:2])
    with pytest.raises(ffi.SphError) as e:
        g2.split_particles(p2, A.adapt_params(P2.replace(fail_on_missing_split_pattern=True), dt2))
    assert e.value.status == 27


def test_adaptive_run_of_the_default_config(product_lib, oracle_lib):
    """BASELINE configs[0] the way the reference runs it (default-config.yaml: merging, sharing, splitting on): 12 calls of
    single_step on the device and on the oracle.  The two runs take their own decisions from their own (1e-7 different) states,
    so they are compared through what the reference itself asserts -- mass conservation -- and through the particle counts."""
    P = default_params()
    scn = sc.SceneConfig.from_yaml(str(REPO / "tests" / "golden" / "default-scene.yaml"))
    sp = A.SplitPatterns.load_from_file(PATTERNS)
    sims = [init_fluid_sim(P, scn, lib=lib, split_patterns=sp, n_capacity=120000) for lib in (product_lib, oracle_lib)]
    m0 = float(sims[0].particles.mass.sum())
    counts = [[], []]
    for s in range(12):
        for k, sim in enumerate(sims):
            sim.single_step(P)
            counts[k].append(sim.num_fluid_particles())
    assert counts[0][0] > 1035                              # step 1 splits
    assert max(abs(a - b) for a, b in zip(*counts)) <= 0.02 * max(counts[1]), counts
    for sim in sims:
        assert abs(float(sim.particles.mass.sum()) - m0) < 0.005 * 12
        x = sim.particles.position
        assert np.isfinite(x).all() and np.abs(x).max() < 1.0


def test_config4_ratio_stress_4m_adaptive_steps(product_lib):
    """BASELINE configs[4] WITH its adaptivity at full size: the 4 004 343-particle scene (50:1 radii, IISPH, Sdf2D box, EmptyAngle
    level estimation), sizing radii scaled with the scene (fine = the fine particles' radius, base = the coarse ones'), four calls
    of single_step: sharing every step, splitting on odd and merging on even step numbers -- decisions by the library's compiled
    sequential partner search, data on the device.  Checked through the reference's own invariants."""
    scene_f, params_f, _ = WORKLOADS["ratio_stress_4m"]
    scn = scene_f()
    r_fine = float(np.sqrt(np.float32(0.0004385) ** 2 * 0.93 / np.pi))
    P = params_f(level_estimation_method="EmptyAngle", merging=True, sharing=True, splitting=True, particle_radius_fine=r_fine,
                 particle_radius_base=50 * r_fine, maximum_surface_distance=0.3)
    sp = A.SplitPatterns.load_from_file(PATTERNS)
    sim = init_fluid_sim(P, scn, lib=product_lib, split_patterns=sp, n_capacity=6000000)
    n0 = sim.num_fluid_particles()
    assert n0 == 4004343
    m0 = float(sim.particles.mass.sum(dtype=np.float64))
    events = {"shares": 0, "merges": 0, "splits": 0}
    for s in range(4):
        dt = sim.single_step_without_adaptivity(P)
        info = sim.single_step_adaptivity(P, dt)
        for k in events:
            events[k] += info[k]
    assert events["merges"] > 1000 and events["splits"] > 0, events     # the bulk of the fine block merges, coarse surface particles split
    assert sim.num_fluid_particles() != n0
    assert abs(float(sim.particles.mass.sum(dtype=np.float64)) - m0) < 1e-4 * m0
    x = sim.particles.position
    assert np.isfinite(x).all() and np.abs(x).max() < 1.0
    sim.single_step_without_adaptivity(P)                   # and the edited vector steps


def test_config4_first_adaptive_steps_with_the_oracles_decisions(product_lib, oracle_lib):
    """BASELINE configs[4] at its full 4 004 343 particles, the adaptive half checked against the ORACLE (the invariants of the test
    above cannot tell a wrong transfer from a right one): both sides step once (iteration counts pinned), then run
    single_step_adaptivity of an odd step (simulation.rs:2732-2796: share, then split) and, after a second step, of an even one
    (share, then merge) with the partner DECISIONS taken once, from the oracle's state, and applied to both sides -- the device's
    gather-form transfers, closed-form deletion order and appended children against oracle/adapt.c's statement-by-statement
    share_particles / merge_particles / split_particles on 4M particles."""
    scene_f, params_f, _ = WORKLOADS["ratio_stress_4m"]
    scn = scene_f()
    r_fine = float(np.sqrt(np.float32(0.0004385) ** 2 * 0.93 / np.pi))
    P = params_f(level_estimation_method="EmptyAngle", merging=True, sharing=True, splitting=True, particle_radius_fine=r_fine,
                 particle_radius_base=50 * r_fine, maximum_surface_distance=0.3, max_iters=3, iisph_max_avg_density_error=0.0)
    sp = A.SplitPatterns.load_from_file(PATTERNS)
    pos, mass, vel = sc.init_particles(scn)
    planes = sc.boundary_planes(scn.boundary, P.init_boundary_handler)
    g, o = ffi.Context(product_lib, 6000000, planes), ffi.Context(oracle_lib, 6000000, planes)
    for c in (g, o):
        c.upload(mass, pos, vel)
        c.set_split_patterns(sp.patterns)
    p = P.to_ffi()
    m0 = float(mass.sum(dtype=np.float64))

    def decide(kind, dt):
        """classes and partners from the ORACLE's state; the device gets the oracle's level field and classes, so that both sides
        round num_children = round(mass / target_mass) and test `TooLarge` on identical inputs"""
        o.classify(p)
        cls = o.download("particle_size_class")
        g.upload_field("particle_size_class", cls)
        if kind == "split":
            g.upload_field("level_estimation", o.download("level_estimation"))
            return cls, None, None
        mp, mc = A.find_partners_native(product_lib, kind, cls, o.download("mass"), o.download("level_estimation"), o.download("position"),
                                        o.download("h2"), *lists, P, dt)
        return cls, mp, mc

    def same(tol=1e-5):
        assert g.n == o.n
        for f in ("mass", "position", "velocity", "h2_next"):
            assert rel_err(g.download(f), o.download(f)) < tol, f
        a, b = g.download("level_estimation"), o.download("level_estimation")
        assert np.array_equal(np.isnan(a), np.isnan(b))
        assert np.nanmax(np.abs(a - b)) <= 1e-4 * max(np.nanmax(np.abs(b)), 1e-30)

    # ---- step 1 (odd): share, then split
    sg, so = g.step(p), o.step(p)
    assert abs(sg.dt - so.dt) <= 1e-6 * so.dt and int(so.step_number) == 1
    dt = float(so.dt)
    ap = A.adapt_params(P, dt)
    lists = o.download_neighbors()
    cls, mp, mc = decide("share", dt)
    n_share = int(mc.sum())
    for c in (g, o):
        c.share_particles(p, ap, mp, mc)
    same()
    cls, _, _ = decide("split", dt)
    n0 = o.n
    for c in (g, o):
        c.split_particles(p, ap)
    assert g.n == o.n > n0                                   # the coarse surface particles split
    same()
    assert np.array_equal(g.download("particle_size_class"), o.download("particle_size_class"))
    # ---- step 2 (even): share, then merge -- on the vector the first adaptive step left behind
    sg, so = g.step(p), o.step(p)
    assert abs(sg.dt - so.dt) <= 1e-5 * so.dt and int(so.step_number) == 2
    assert (g.download("neighbor_count") != o.download("neighbor_count")).mean() < 1e-3
    assert rel_err(g.download("density"), o.download("density")) < 1e-3
    dt = float(so.dt)
    ap = A.adapt_params(P, dt)
    lists = o.download_neighbors()
    cls, mp, mc = decide("share", dt)
    n_share += int(mc.sum())
    for c in (g, o):
        c.share_particles(p, ap, mp, mc)
    same(1e-4)
    cls, mp, mc = decide("merge", dt)
    n_merge = int(mc.sum())
    n1 = o.n
    for c in (g, o):
        c.merge_particles(p, ap, mp, mc)
    assert g.n == o.n < n1 and n_merge > 1000               # the bulk of the fine block merges
    same(1e-4)
    assert n_share >= 0
    for c in (g, o):
        assert abs(float(c.download("mass").sum(dtype=np.float64)) - m0) < 1e-4 * m0
    sg, so = g.step(p), o.step(p)                            # and both edited vectors step
    assert abs(sg.dt - so.dt) <= 1e-4 * so.dt
    assert rel_err(g.download("density"), o.download("density")) < 2e-3


def test_adaptive_run_on_a_slab_group(product_lib):
    """single_step = step + single_step_adaptivity on a 2-rank slab group (distributed.group_single_step_adaptivity: the ranks'
    particles and neighbour lists gathered in global index order, the decisions and the device-side share / merge / split on ONE
    context, the result scattered back) against the single context's adaptive run of BASELINE configs[0].  The two runs take
    their decisions from states that differ in the last digits, so they are compared like the single-context run is compared with
    the oracle: particle counts step by step, mass conservation, positions in the box; the first adaptive step (inputs equal to
    1e-7) must take the same number of splits."""
    from adaptive_sph_amd import distributed as D
    P = default_params()
    scn = sc.SceneConfig.from_yaml(str(REPO / "tests" / "golden" / "default-scene.yaml"))
    sp = A.SplitPatterns.load_from_file(PATTERNS)
    single = init_fluid_sim(P, scn, lib=product_lib, split_patterns=sp, n_capacity=120000)
    pos, mass, vel = sc.init_particles(scn)
    planes = sc.boundary_planes(scn.boundary, P.init_boundary_handler)
    grp = D.make_loopback_group(product_lib, pos, mass, vel, planes, 2)
    m0 = float(mass.sum(dtype=np.float64))
    p = P.to_ffi()
    counts = [[], []]
    events = {"shares": 0, "merges": 0, "splits": 0}
    for s in range(12):
        single.single_step(P)
        counts[0].append(single.num_fluid_particles())
        sts = ffi.group_step(grp, p)
        info = D.group_single_step_adaptivity(product_lib, grp, planes, P, float(sts[0].dt), int(sts[0].step_number), split_patterns=sp, capacity=120000)
        for k in events:
            events[k] += info[k]
        counts[1].append(sum(c.n for c in grp))
        assert info["n_after"] == counts[1][-1]
    assert counts[0][0] == counts[1][0] > 1035                         # step 1 splits, the same particles on both sides
    assert max(abs(a - b) for a, b in zip(*counts)) <= 0.02 * max(counts[0]), counts
    assert events["splits"] > 0 and events["merges"] + events["shares"] > 0, events
    n = counts[1][-1]
    ids = np.concatenate([c.download("particle_id") for c in grp])
    assert np.array_equal(np.sort(ids), np.arange(n))                  # ids are the global indices again
    assert abs(float(D.gather_by_id(grp, "mass", n).sum(dtype=np.float64)) - m0) < 0.005 * 12
    x = D.gather_by_id(grp, "position", n)
    assert np.isfinite(x).all() and np.abs(x).max() < 1.05


def test_config4_ratio_stress_4m_adaptive_steps_on_two_slabs(product_lib):
    """BASELINE configs[4] -- the 4 004 343-particle 50:1 scene, IISPH, Sdf2D box, EmptyAngle level estimation, merging / sharing /
    splitting on -- as a slab group (two ranks: three would be narrower than two ghost layers of the COARSE particles): three calls
    of step + group_single_step_adaptivity against the single context's first three adaptive steps.  Same event counts on the first
    step (inputs equal to 1e-7), particle counts within 2 % afterwards, mass conserved, everybody inside the box."""
    from adaptive_sph_amd import distributed as D
    scene_f, params_f, _ = WORKLOADS["ratio_stress_4m"]
    scn = scene_f()
    r_fine = float(np.sqrt(np.float32(0.0004385) ** 2 * 0.93 / np.pi))
    P = params_f(level_estimation_method="EmptyAngle", merging=True, sharing=True, splitting=True, particle_radius_fine=r_fine,
                 particle_radius_base=50 * r_fine, maximum_surface_distance=0.3)
    sp = A.SplitPatterns.load_from_file(PATTERNS)
    single = init_fluid_sim(P, scn, lib=product_lib, split_patterns=sp, n_capacity=6000000)
    pos, mass, vel = sc.init_particles(scn)
    planes = sc.boundary_planes(scn.boundary, P.init_boundary_handler)
    grp = D.make_loopback_group(product_lib, pos, mass, vel, planes, 2)
    m0 = float(mass.sum(dtype=np.float64))
    p = P.to_ffi()
    counts = [[], []]
    first = []
    for s in range(3):
        dt = single.single_step_without_adaptivity(P)
        i1 = single.single_step_adaptivity(P, dt)
        counts[0].append(single.num_fluid_particles())
        sts = ffi.group_step(grp, p)
        assert abs(float(sts[0].dt) - dt) <= 1e-5 * dt
        i2 = D.group_single_step_adaptivity(product_lib, grp, planes, P, float(sts[0].dt), int(sts[0].step_number), split_patterns=sp, capacity=6000000)
        counts[1].append(sum(c.n for c in grp))
        if s == 0:
            first = [(i1[k], i2[k]) for k in ("shares", "merges", "splits")]
    assert all(abs(a - b) <= 0.001 * max(a, 1) for a, b in first), first
    assert max(abs(a - b) for a, b in zip(*counts)) <= 0.02 * max(counts[0]), counts
    assert counts[1][-1] != 4004343
    n = counts[1][-1]
    assert abs(float(D.gather_by_id(grp, "mass", n).sum(dtype=np.float64)) - m0) < 1e-4 * m0
    x = D.gather_by_id(grp, "position", n)
    assert np.isfinite(x).all() and np.abs(x).max() < 1.0
    ffi.group_step(grp, p)                                   # and the re-uploaded slabs step


def _fields_by_id(grp, n):
    from adaptive_sph_amd import distributed as D
    return {f: D.gather_by_id(grp, f, n) for f in ("mass", "position", "velocity", "h2_next", "level_estimation", "level_old", "particle_size_class")}


@pytest.mark.parametrize("transport", ["loopback", "threads"])
def test_slab_form_of_the_apply_equals_the_gather_path(product_lib, transport):
    """share / merge / split applied ON the slabs (sph_group_adapt; per rank: the slab form of sph_share_particles / sph_merge_particles
    / sph_split_particles through the context's own transport) against the round-3 path that gathers every particle to ONE context,
    applies there and re-uploads: BASELINE configs[0]'s adaptive run on two ranks.  Up to adaptive step t - 1 both groups take the
    gather path (identical states); at step t one of them applies on its slabs: same decisions (the same host code on the same
    gathered fields), and then the same particle under the same id with the same bits in every field adaptivity writes."""
    from adaptive_sph_amd import distributed as D
    from concurrent.futures import ThreadPoolExecutor
    P = default_params()
    scn = sc.SceneConfig.from_yaml(str(REPO / "tests" / "golden" / "default-scene.yaml"))
    sp = A.SplitPatterns.load_from_file(PATTERNS)
    pos, mass, vel = sc.init_particles(scn)
    planes = sc.boundary_planes(scn.boundary, P.init_boundary_handler)
    p = P.to_ffi()
    k = 2      # (the 1035-particle scene is too narrow for a rank with two ghost layers)
    seen = {"shares": 0, "merges": 0, "splits": 0}
    for t in (1, 2, 3, 4, 6):
        a = D.make_loopback_group(product_lib, pos, mass, vel, planes, k)
        thr = D.ThreadedGroup(product_lib, pos, mass, vel, planes, k) if transport == "threads" else None
        b = thr.contexts if thr else D.make_loopback_group(product_lib, pos, mass, vel, planes, k)
        for c in b:
            c.set_split_patterns(sp.patterns)
        step_b = (lambda: thr.step(p)) if thr else (lambda: ffi.group_step(b, p))
        try:
            for s in range(1, t + 1):
                sa, sb = ffi.group_step(a, p), step_b()
                dt, num = float(sa[0].dt), int(sa[0].step_number)
                assert float(sb[0].dt) == dt
                ia = D.group_single_step_adaptivity(product_lib, a, planes, P, dt, num, split_patterns=sp, capacity=120000)
                if s < t:
                    # (a ThreadedGroup's contexts are ordinary slab contexts: the gather path re-uploads them all the same)
                    ib = D.group_single_step_adaptivity(product_lib, b, planes, P, dt, num, split_patterns=sp, capacity=120000)
                    for c in b:
                        c.set_split_patterns(sp.patterns)
                elif thr is None:
                    ib = D.group_single_step_adaptivity_on_slabs(product_lib, b, P, dt, num)
                else:
                    # every rank calls the per-rank entry points from its own thread; the decisions are taken once, here
                    ap = A.adapt_params(P, dt)
                    ids = [c.download("particle_id") for c in b]
                    n = int(sum(len(i) for i in ids))
                    off, idx = D.assemble_lists(ids, [c.download_neighbors() for c in b], n)
                    lists = (off.astype(np.uint32), idx)
                    ib = {"shares": 0, "merges": 0, "splits": 0}

                    def all_ranks(fn):
                        with ThreadPoolExecutor(k) as pool:
                            for f in [pool.submit(fn, c) for c in b]:
                                f.result()

                    def gathered():
                        all_ranks(lambda c: c.classify(p))
                        return {f: D.gather_by_id(b, f, n) for f in ("particle_size_class", "mass", "level_estimation", "position", "h2")}
                    if P.sharing:
                        mp, mc = D._decide_on_gathered(product_lib, "share", gathered(), lists, P, dt)
                        ib["shares"] = int(mc.sum())
                        all_ranks(lambda c: c.share_particles(p, ap, mp, mc))
                    if num % 2 == 0:
                        mp, mc = D._decide_on_gathered(product_lib, "merge", gathered(), lists, P, dt)
                        ib["merges"] = int(mc.sum())
                        all_ranks(lambda c: c.merge_particles(p, ap, mp, mc))
                    else:
                        all_ranks(lambda c: c.classify(p))
                        all_ranks(lambda c: c.split_particles(p, ap))
                        ib["splits"] = int(sum(c.n for c in b)) - n
                assert (ia["shares"], ia["merges"], ia["splits"]) == (ib["shares"], ib["merges"], ib["splits"]), (t, s, ia, ib)
            for key in seen:
                seen[key] += ib[key]
            n = sum(c.n for c in a)
            assert n == sum(c.n for c in b)
            ids = np.concatenate([c.download("particle_id") for c in b])
            assert np.array_equal(np.sort(ids), np.arange(n))              # the ids are the reference's indices 0 .. n-1 again
            fa, fb = _fields_by_id(a, n), _fields_by_id(b, n)
            for f in fa:
                assert np.array_equal(fa[f], fb[f], equal_nan=f == "level_estimation"), (t, f)
            step_b()                                                       # and the slabs step with what the apply left them
        finally:
            for c in a:
                c.close()
            if thr:
                thr.close()
            else:
                for c in b:
                    c.close()
    assert seen["splits"] > 0 and seen["merges"] + seen["shares"] > 0, seen


def test_config4_adaptive_steps_with_the_oracles_decisions_on_eight_slabs(product_lib, oracle_lib):
    """BASELINE configs[4] as BASELINE.json states it: the 4 004 343-particle 50:1 scene on EIGHT x-slabs (loopback transport: the
    driver code the RCCL transport runs), adaptive half included -- the test above this file's single-context one, with the device
    side a slab group that applies share / split / merge on its slabs (sph_group_adapt).  Decisions from the ORACLE's state, applied
    to both sides; fields compared by global id."""
    from adaptive_sph_amd import distributed as D
    scene_f, params_f, _ = WORKLOADS["ratio_stress_4m"]
    scn = scene_f()
    r_fine = float(np.sqrt(np.float32(0.0004385) ** 2 * 0.93 / np.pi))
    P = params_f(level_estimation_method="EmptyAngle", merging=True, sharing=True, splitting=True, particle_radius_fine=r_fine,
                 particle_radius_base=50 * r_fine, maximum_surface_distance=0.3, max_iters=3, iisph_max_avg_density_error=0.0)
    sp = A.SplitPatterns.load_from_file(PATTERNS)
    pos, mass, vel = sc.init_particles(scn)
    planes = sc.boundary_planes(scn.boundary, P.init_boundary_handler)
    o = ffi.Context(oracle_lib, 6000000, planes)
    o.upload(mass, pos, vel)
    o.set_split_patterns(sp.patterns)
    grp = D.make_loopback_group(product_lib, pos, mass, vel, planes, 8)
    for c in grp:
        c.set_split_patterns(sp.patterns)
    p = P.to_ffi()
    m0 = float(mass.sum(dtype=np.float64))

    def to_ranks(field, values):
        for c in grp:
            c.upload_field(field, values[c.download("particle_id")])

    def decide(kind, dt, lists):
        o.classify(p)
        cls = o.download("particle_size_class")
        to_ranks("particle_size_class", cls)
        if kind == "split":
            to_ranks("level_estimation", o.download("level_estimation"))
            return None, None
        return A.find_partners_native(product_lib, kind, cls, o.download("mass"), o.download("level_estimation"), o.download("position"),
                                      o.download("h2"), *lists, P, dt)

    def same(tol):
        n = o.n
        assert sum(c.n for c in grp) == n
        ids = np.concatenate([c.download("particle_id") for c in grp])
        assert np.array_equal(np.sort(ids), np.arange(n))
        for f in ("mass", "position", "velocity", "h2_next"):
            assert rel_err(D.gather_by_id(grp, f, n), o.download(f)) < tol, f
        a, b = D.gather_by_id(grp, "level_estimation", n), o.download("level_estimation")
        assert np.array_equal(np.isnan(a), np.isnan(b))
        assert np.nanmax(np.abs(a - b)) <= 1e-4 * max(np.nanmax(np.abs(b)), 1e-30)

    # ---- step 1 (odd): share, then split
    sg, so = ffi.group_step(grp, p), o.step(p)
    assert abs(sg[0].dt - so.dt) <= 1e-6 * so.dt and int(so.step_number) == 1
    dt = float(so.dt)
    ap = A.adapt_params(P, dt)
    lists = o.download_neighbors()
    mp, mc = decide("share", dt, lists)
    o.share_particles(p, ap, mp, mc)
    ffi.group_adapt(grp, "share", p, ap, mp, mc)
    same(1e-5)
    decide("split", dt, lists)
    n0 = o.n
    o.split_particles(p, ap)
    ffi.group_adapt(grp, "split", p, ap)
    assert o.n > n0
    same(1e-5)
    # ---- step 2 (even): share, then merge -- on the vector the first adaptive step left behind
    sg, so = ffi.group_step(grp, p), o.step(p)
    assert abs(sg[0].dt - so.dt) <= 1e-5 * so.dt and int(so.step_number) == 2
    dt = float(so.dt)
    ap = A.adapt_params(P, dt)
    lists = o.download_neighbors()
    mp, mc = decide("share", dt, lists)
    o.share_particles(p, ap, mp, mc)
    ffi.group_adapt(grp, "share", p, ap, mp, mc)
    same(1e-4)
    mp, mc = decide("merge", dt, lists)
    n1, n_merge = o.n, int(mc.sum())
    o.merge_particles(p, ap, mp, mc)
    ffi.group_adapt(grp, "merge", p, ap, mp, mc)
    assert o.n < n1 and n_merge > 1000
    same(1e-4)
    assert abs(float(sum(c.download("mass").sum(dtype=np.float64) for c in grp)) - m0) < 1e-4 * m0
    sg, so = ffi.group_step(grp, p), o.step(p)                # and both edited vectors step
    assert abs(sg[0].dt - so.dt) <= 1e-4 * so.dt
    assert rel_err(D.gather_by_id(grp, "density", o.n), o.download("density")) < 2e-3


def test_exports_into_persistent_host_buffers_are_the_fresh_exports(product_lib):
    """Round 6 (profiles/r6_export_time.txt): the adaptive driver exports into persistent host buffers (ffi.HostBuffers) with ONE library
    call for the lists -- same offsets, indices and fields as the two-call export into fresh arrays, before and after the particle count
    changed, and a list that outgrew its buffer is fetched with a second call."""
    from adaptive_sph_amd.workloads import default_params
    scn = sc.dam_break_small(96, 64, 1.0 / 64)
    pos, mass, vel = sc.init_particles(scn)
    P = default_params(merging=False, sharing=False, splitting=False, level_estimation_method="None")
    g = ffi.Context(product_lib, 2 * len(mass), sc.boundary_planes(scn.boundary))
    g.upload(mass, pos, vel)
    p = P.to_ffi()
    host = ffi.HostBuffers()
    for n_now in (len(mass), len(mass) // 2):
        if n_now != len(mass):
            g.upload(mass[:n_now], pos[:n_now], vel[:n_now])     # the vector shrank: the same buffers, shorter views
        g.step(p)
        off0, idx0 = g.download_neighbors()
        off1, idx1 = g.download_neighbors(host)
        assert np.array_equal(off0, off1) and np.array_equal(idx0, idx1) and len(idx1) == int(off1[-1])
        off2, idx2 = g.download_neighbors(host)                  # again: the library's own staging is kept across calls
        assert off2.ctypes.data == off1.ctypes.data and np.array_equal(idx0, idx2)
        for f in ("mass", "position", "h2", "density", "neighbor_count"):
            a, b = g.download(f), g.download(f, host)
            assert a.dtype == b.dtype and a.shape == b.shape and np.array_equal(a, b), f
    # a buffer that is too small: the call reports the total, the wrapper grows the buffer and asks again
    tiny = ffi.HostBuffers()
    tiny._bufs["csr:indices"] = np.zeros(64, np.uint8)
    real_view = tiny.view
    tiny.view = lambda key, dt, count: real_view(key, dt, 16 if key == "csr:indices" and count >= 16 * g.n else count)   # (defeat the 16 n default once)
    off3, idx3 = g.download_neighbors(tiny)
    assert np.array_equal(off3, off0) and np.array_equal(idx3, idx0)
    g.close()
