"""Host-side half of ``single_step_adaptivity`` (reference: src/simulation/simulation.rs:2732-2796).

The reference takes the share / merge partner DECISIONS in a sequential loop over the particles
(adaptivity/particle_sharing.rs:14-117, particle_merging.rs:16-125) -- that stays on the host, here as on the Rust side --
and the particle DATA stays on the device: the host reads the handful of fields the decision needs plus the neighbour lists,
fills ``merge_partner`` / ``merge_counter`` exactly as the reference does, and hands the two arrays to the library
(``sph_share_particles`` / ``sph_merge_particles`` / ``sph_split_particles``, include/sph_ffi.h), which applies them to the
device-resident state with the reference's Vec semantics.

All comparisons are made on float32 values with the reference's operation order (numpy float32 scalars).
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np
import yaml

from . import ffi
from .simulation_parameters import SimulationParams

f32 = np.float32

# ParticleSizeClass (adaptivity/mod.rs:12-23)
TOO_SMALL, SMALL, OPTIMAL, LARGE, TOO_LARGE = 0, 1, 2, 3, 4
PARTICLE_SIZE_FACTOR_LARGE = f32(1.1)          # adaptivity/mod.rs:26
MERGE_PARTNER_AVAILABLE = ffi.MERGE_PARTNER_AVAILABLE
MERGE_PARTNER_DELETE = ffi.MERGE_PARTNER_DELETE
PI = f32(np.pi)


class SplitPatterns:
    """``SplitPatterns<2>`` (splitting.rs:84-120): entry k is the 1-to-(k+2) split, child offsets in parent radii."""

    def __init__(self, patterns: Sequence[np.ndarray]):
        for i, q in enumerate(patterns):
            if np.asarray(q).shape != (i + 2, 2):
                raise ValueError("assertion failed: sp.pos_s.len() == i + 2")
        self.patterns: List[np.ndarray] = [np.asarray(q, np.float32) for q in patterns]

    @classmethod
    def load_from_file(cls, path) -> "SplitPatterns":
        """load_split_patterns_from_file (simulation.rs:3000-3004): the serde_yaml form of Vec<SplitPattern> (mass_s, pos_s, h_s)."""
        with open(path, "r") as fh:
            doc = yaml.safe_load(fh)
        out = []
        for k, entry in enumerate(doc):
            pos = np.asarray(entry["pos_s"], np.float32)
            if len(entry["mass_s"]) != len(pos) or len(entry["h_s"]) != len(pos):
                raise ValueError(f"split pattern {k}: mass_s / pos_s / h_s lengths differ")
            out.append(pos)
        return cls(out)

    def get_max_num_children(self) -> int:
        return len(self.patterns) + 1

    def get(self, num_children: int) -> np.ndarray:
        assert num_children > 1
        if num_children - 2 >= len(self.patterns):
            raise KeyError(f"no split pattern for a 1-to-{num_children} split")
        return self.patterns[num_children - 2]


def radius_to_sphere_volume(r):
    return PI * r * r          # DimensionUtils2d (sph_kernels.rs:208-211)


def target_mass(level: np.ndarray, P: SimulationParams) -> np.ndarray:
    """LevelEstimationState::target_mass (simulation.rs:213-237), vectorised in float32.  NaN (FluidInterior) stays NaN."""
    msd, rho0 = f32(P.maximum_surface_distance), f32(P.rest_density)
    fine, base = f32(P.particle_radius_fine), f32(P.particle_radius_base)
    level = np.maximum(np.asarray(level, np.float32), -msd)
    interp = level / -msd
    one = f32(1.0)
    if P.sizing_function == "Mass":
        return (radius_to_sphere_volume(fine) * rho0) * (one - interp) + (radius_to_sphere_volume(base) * rho0) * interp
    if P.sizing_function == "Radius":
        r = fine * (one - interp) + base * interp
        return radius_to_sphere_volume(r) * rho0
    e = f32(0.5)
    r = fine * (one - np.power(interp, e, dtype=np.float32)) + base * np.power(interp, e, dtype=np.float32)
    return radius_to_sphere_volume(r) * rho0


def mass_base(P: SimulationParams) -> np.float32:
    return radius_to_sphere_volume(f32(P.particle_radius_base)) * f32(P.rest_density)     # simulation_parameters.rs:129-131


def _find_partners(kind: str, size_class, mass, level, position, h2, offsets, indices, P: SimulationParams, dt: float) -> Tuple[np.ndarray, np.ndarray]:
    """find_share_partner_sequential (particle_sharing.rs:14-117) / find_merge_partner_sequential (particle_merging.rs:16-125):
    the same greedy loop, in particle order, over each particle's neighbour list in list order."""
    n = len(mass)
    merge_partner = np.full(n, MERGE_PARTNER_AVAILABLE, np.uint32)
    merge_counter = np.zeros(n, np.uint16)
    mass = np.asarray(mass, np.float32)
    target = target_mass(level, P)
    mbase = mass_base(P)
    share = kind == "share"
    donors = np.nonzero(np.asarray(size_class) == (LARGE if share else TOO_SMALL))[0]
    max_dist_factor = f32(P.max_share_distance if share else P.max_merge_distance)
    dtf = f32(dt)
    for i in donors:
        i = int(i)
        if share:
            dropped_i = min(mass[i] - target[i], target[i] * f32(P.max_mass_transfer_sharing) * dtf)   # dropped_mass_sharing
        else:
            dropped_i = mass[i]                                                                         # dropped_mass_merging
        for j in indices[offsets[i]:offsets[i + 1]]:
            j = int(j)
            if i == j:
                continue
            cj = size_class[j]
            if share:
                can = (cj == SMALL) or (cj == TOO_SMALL and P.allow_share_with_too_small_particle) or \
                      (cj == OPTIMAL and P.allow_share_with_optimal_particle)
            else:
                can = (cj in (SMALL, TOO_SMALL)) or (cj == OPTIMAL and P.allow_merge_with_optimal_particle)
                if P.allow_merge_on_size_difference and mass[j] > f32(5.0) * mass[i]:
                    can = True
            if not can:
                continue
            # the partner must lie within max_{share,merge}_distance mean smoothing lengths (particle_sharing.rs:60-67, particle_merging.rs:71-78)
            dx, dy = position[i, 0] - position[j, 0], position[i, 1] - position[j, 1]
            max_dist = ((h2[i] + h2[j]) * f32(0.5)) * max_dist_factor
            if dx * dx + dy * dy > max_dist * max_dist:
                continue
            new_mass_j = mass[j] + dropped_i / f32(int(merge_counter[i]) + 1)
            if new_mass_j >= target[j] * PARTICLE_SIZE_FACTOR_LARGE:
                continue
            if new_mass_j > mbase:
                continue
            if merge_partner[j] != MERGE_PARTNER_AVAILABLE:
                continue        # the neighbouring particle is being used as a partner already
            if merge_counter[i] == 0:
                if merge_partner[i] != MERGE_PARTNER_AVAILABLE:
                    continue    # this particle is itself somebody's partner
                merge_partner[i] = MERGE_PARTNER_DELETE
            merge_partner[j] = i
            merge_counter[i] += 1
            assert merge_counter[i] < 1000
    return merge_partner, merge_counter


def find_share_partner_sequential(size_class, mass, level, position, h2, offsets, indices, P: SimulationParams, dt: float):
    return _find_partners("share", size_class, mass, level, position, h2, offsets, indices, P, dt)


def find_merge_partner_sequential(size_class, mass, level, position, h2, offsets, indices, P: SimulationParams, dt: float):
    return _find_partners("merge", size_class, mass, level, position, h2, offsets, indices, P, dt)


def validate_partners(kind: str, size_class, merge_partner, merge_counter, offsets, indices) -> int:
    """validate_share_partners (particle_sharing.rs:119-150) / validate_merge_partners (particle_merging.rs:226-268)."""
    n = len(merge_counter)
    donors = 0
    want = LARGE if kind == "share" else TOO_SMALL
    for i in np.nonzero(merge_counter > 0)[0]:
        i = int(i)
        assert size_class[i] == want
        donors += 1
        assert merge_partner[i] == MERGE_PARTNER_DELETE
        nb = indices[offsets[i]:offsets[i + 1]]
        assert int((merge_partner[nb] == i).sum()) == int(merge_counter[i])
    rest = np.nonzero(merge_counter == 0)[0]
    assert not (merge_partner[rest] == MERGE_PARTNER_DELETE).any()
    recv = rest[(merge_partner[rest] != MERGE_PARTNER_AVAILABLE)]
    assert (merge_partner[merge_partner[recv]] == MERGE_PARTNER_DELETE).all()
    assert n == len(merge_partner)
    return donors


def adapt_params(P: SimulationParams, dt: float) -> ffi.SphAdaptParams:
    ap = ffi.SphAdaptParams()
    ap.dt = dt
    ap.max_mass_transfer_sharing = P.max_mass_transfer_sharing
    ap.minimum_share_partners = P.minimum_share_partners
    ap.minimum_merge_partners = P.minimum_merge_partners
    ap.fail_on_missing_split_pattern = int(P.fail_on_missing_split_pattern)
    ap.max_share_distance, ap.max_merge_distance = P.max_share_distance, P.max_merge_distance
    ap.allow_share_with_optimal_particle = int(P.allow_share_with_optimal_particle)
    ap.allow_share_with_too_small_particle = int(P.allow_share_with_too_small_particle)
    ap.allow_merge_with_optimal_particle = int(P.allow_merge_with_optimal_particle)
    ap.allow_merge_on_size_difference = int(P.allow_merge_on_size_difference)
    return ap


def find_partners_native(lib: ffi.SphLibrary, kind: str, size_class, mass, level, position, h2, offsets, indices, P: SimulationParams, dt: float, host=None):
    """The same sequential loops as `_find_partners`, compiled (sph_host_find_partners): for million-particle scenes.  `host`
    (ffi.HostBuffers): merge_partner / merge_counter land in persistent memory (views, overwritten by the next search)."""
    import ctypes as C
    n = len(mass)
    arrs = [np.ascontiguousarray(size_class, np.uint8), np.ascontiguousarray(mass, np.float32), np.ascontiguousarray(level, np.float32),
            np.ascontiguousarray(position, np.float32), np.ascontiguousarray(h2, np.float32), np.ascontiguousarray(offsets, np.uint32),
            np.ascontiguousarray(indices, np.uint32)]
    mp, mc = (np.empty(n, np.uint32), np.empty(n, np.uint16)) if host is None else (host.view("merge_partner", np.uint32, n), host.view("merge_counter", np.uint16, n))
    p, ap, tot = P.to_ffi(), adapt_params(P, dt), C.c_uint64(0)
    rc = lib.host_find_partners(0 if kind == "share" else 1, n, *[a.ctypes.data for a in arrs], C.byref(p), C.byref(ap), mp.ctypes.data, mc.ctypes.data,
                                C.byref(tot))
    if rc != 0:
        raise ffi.SphError(rc, "the partner search's validation failed (validate_share_partners / validate_merge_partners)")
    return mp, mc


class AdaptivityDriver:
    """single_step_adaptivity (simulation.rs:2732-2796) on a context that has just run single_step_without_adaptivity: sharing
    every step, merging on even step numbers, splitting on odd ones (`step_number` is FluidSimulation.step_number AFTER the step,
    :2725); mass is conserved to 0.005 (asserted like the reference).  The step's neighbour lists are read once and kept on the
    host across the passes, as the reference's NeighborhoodCache is: share_particles does not touch it, and the merge decision
    that follows still iterates the lists of the step."""

    def __init__(self, ctx: ffi.Context, split_patterns: SplitPatterns = None, log=None):
        self.ctx = ctx
        self.log = log
        self.host = ffi.HostBuffers()   # the exports land in the same host memory every step (round 6: the 26 ms "download" of configs[4]'s adaptive step were mostly page faults of fresh arrays)
        if ctx.n:
            self.host.reserve(ctx.n)
        if split_patterns is not None:
            ctx.set_split_patterns(split_patterns.patterns)

    def single_step_adaptivity(self, P: SimulationParams, dt: float, step_number: int, lists=None) -> dict:
        """`lists` = (offsets, indices): the step's neighbour lists when they do not live in `ctx` (slab decomposition: the ranks'
        exports assembled in global index order, distributed.group_single_step_adaptivity)."""
        import time as _t
        ctx, log = self.ctx, self.log
        p, ap = P.to_ffi(), adapt_params(P, dt)
        info = {"n_before": ctx.n, "shares": 0, "merges": 0, "splits": 0}
        # what the adaptive half of a step costs, by phase (bench.py reports it): device -> host of the lists and the five fields a
        # decision reads, the sequential partner searches on the host, the apply calls on the device
        tm = info["seconds"] = {"download": 0.0, "host_decide": 0.0, "apply": 0.0, "mass_check": 0.0}
        # particles.mass.iter().cloned().sum() (:2745, 2791) is a SEQUENTIAL f32 sum.  At the reference's own scene sizes (1e3..1e5
        # particles) that is accurate to ~1e-5 and the 0.005 bar means "mass is conserved".  At millions of particles it is not a
        # measurement any more: adding 1.8e-7 to a running total of 1.4 rounds to 1 or 2 ulp of the total every time (4M particles of
        # configs[4]: the sequential sums before and after a merge pass differ by > 0.005 although the mass is conserved to 1e-7, and
        # the reference would panic there).  The mirror keeps the assertion's MEANING: the sums are taken in f64.
        seq_sum = lambda a: float(np.sum(a, dtype=np.float64))   # noqa: E731
        t0 = _t.perf_counter()
        host = self.host
        m1 = ctx.download("mass", host)
        off, idx = lists if lists is not None else ctx.download_neighbors(host)   # the lists single_step_without_adaptivity left behind (self.neighs)
        t1 = _t.perf_counter()
        tm["download"] += t1 - t0
        total_mass1 = seq_sum(m1)   # (before the next download of the masses overwrites the persistent buffer)
        tm["mass_check"] += _t.perf_counter() - t1

        def decide(kind):
            ta = _t.perf_counter()
            ctx.classify(p)
            t0 = _t.perf_counter()
            tm["apply"] += t0 - ta   # (classify_particles on the device: the apply side's device work)
            cls = ctx.download("particle_size_class", host)
            fields = (cls, ctx.download("mass", host), ctx.download("level_estimation", host), ctx.download("position", host), ctx.download("h2", host))
            t1 = _t.perf_counter()
            tm["download"] += t1 - t0
            try:
                if getattr(ctx.lib, "host_find_partners", None) is not None:
                    return find_partners_native(ctx.lib, kind, *fields, off, idx, P, dt, host)     # (validates like the reference does)
                mp, mc = _find_partners(kind, *fields, off, idx, P, dt)
                validate_partners(kind, cls, mp, mc, off, idx)
                return mp, mc
            finally:
                tm["host_decide"] += _t.perf_counter() - t1

        def apply(f, *a):
            t0 = _t.perf_counter()
            f(*a)
            tm["apply"] += _t.perf_counter() - t0

        if P.sharing:
            mp, mc = decide("share")
            info["shares"] = int(mc.sum())
            if log:
                log(f"SEQUENTIAL SHARE {info['shares']} shares")
            apply(ctx.share_particles, p, ap, mp, mc)
        if step_number % 2 == 0:
            if P.merging:
                mp, mc = decide("merge")
                info["merges"] = int(mc.sum())
                if log:
                    log(f"SEQUENTIAL MERGE {info['merges']} merges")
                apply(ctx.merge_particles, p, ap, mp, mc)
        elif P.splitting:
            n0 = ctx.n
            apply(lambda: (ctx.classify(p), ctx.split_particles(p, ap)))
            info["splits"] = ctx.n - n0
        t0 = _t.perf_counter()
        m2 = ctx.download("mass", host)
        t1 = _t.perf_counter()
        tm["download"] += t1 - t0
        total_mass2 = seq_sum(m2)
        tm["mass_check"] += _t.perf_counter() - t1
        if not abs(total_mass1 - total_mass2) <= 0.005:             # assert_ft_approx_eq(total_mass1, total_mass2, 0.005, "mass sum")
            raise AssertionError(f"mass sum: {total_mass1} vs {total_mass2}")
        info["n_after"] = ctx.n
        return info
