"""TEST-ONLY helpers around the CPU oracle (oracle/).  Nothing in the product package imports this."""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

from adaptive_sph_amd import ffi

REPO = Path(__file__).resolve().parent.parent
ORACLE_DIR = REPO / "oracle"
ORACLE_LIB = ORACLE_DIR / "liboracle.so"

_ORACLE = None


def build_oracle():
    r = subprocess.run(["make", "-C", str(ORACLE_DIR)], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("oracle build failed:\n" + r.stdout + r.stderr)
    return ORACLE_LIB


def load_oracle() -> ffi.SphLibrary:
    global _ORACLE
    if _ORACLE is None:
        build_oracle()
        lib = ffi.SphLibrary(ORACLE_LIB, "oracle_")
        L = lib.lib
        L.oracle_cubic_kernel_2d.restype = C.c_float
        L.oracle_cubic_kernel_2d.argtypes = [C.c_float, C.c_float]
        L.oracle_cubic_kernel_2d_deriv.restype = None
        L.oracle_cubic_kernel_2d_deriv.argtypes = [C.c_float, C.c_float, C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        for nm in ("oracle_sphere_volume_to_radius", "oracle_radius_to_sphere_volume"):
            getattr(L, nm).restype = C.c_float
            getattr(L, nm).argtypes = [C.c_float]
        L.oracle_h_from_mass.restype = C.c_float
        L.oracle_h_from_mass.argtypes = [C.c_float, C.c_float]
        for nm in ("oracle_lambda2", "oracle_dlambda2"):
            getattr(L, nm).restype = C.c_double
            getattr(L, nm).argtypes = [C.c_double]
        L.oracle_lambda_luts.restype = None
        L.oracle_lambda_luts.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_lut_get.restype = C.c_float
        L.oracle_lut_get.argtypes = [C.c_void_p, C.c_int, C.c_float]
        L.oracle_add_fluid_block.restype = C.c_uint64
        L.oracle_add_fluid_block.argtypes = [C.c_float] * 8 + [C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_boundary_box.restype = None
        L.oracle_boundary_box.argtypes = [C.c_float, C.c_float, C.POINTER(ffi.SphPlane)]
        L.oracle_build_neighbors.restype = C.c_int
        L.oracle_build_neighbors.argtypes = [C.c_void_p, C.c_float]
        L.oracle_check_neighborhood.restype = C.c_int
        L.oracle_check_neighborhood.argtypes = [C.c_void_p]
        L.oracle_num_threads.restype = C.c_int
        L.oracle_set_num_threads.argtypes = [C.c_int]
        _ORACLE = lib
    return _ORACLE


def oracle_kernel_deriv(lib, dx, dy, h):
    gx, gy = C.c_float(), C.c_float()
    lib.lib.oracle_cubic_kernel_2d_deriv(dx, dy, h, C.byref(gx), C.byref(gy))
    return gx.value, gy.value


def oracle_scene_block(lib, block):
    """add_fluid_block through the oracle's own C restatement."""
    args = [block.pos[0], block.pos[1], block.size[0], block.size[1], block.spacing, block.volume_fill_ratio,
            block.velocity[0], block.velocity[1]]
    n = int(lib.lib.oracle_add_fluid_block(*args, 0, None, None, None))
    pos = np.empty((n, 2), np.float32)
    mass = np.empty(n, np.float32)
    vel = np.empty((n, 2), np.float32)
    lib.lib.oracle_add_fluid_block(*args, n, pos.ctypes.data, mass.ctypes.data, vel.ctypes.data)
    return pos, mass, vel


def oracle_box(lib, width, height):
    arr = (ffi.SphPlane * 4)()
    lib.lib.oracle_boundary_box(width, height, arr)
    return [(p.dir_x, p.dir_y, p.delta) for p in arr]


def csr_sets(offsets, indices):
    """list of sorted index arrays per particle"""
    return [np.sort(indices[offsets[i]:offsets[i + 1]]) for i in range(len(offsets) - 1)]


def uniform_params(**kw):
    """default-config.yaml with the uniform dam-break overrides of SURVEY.md section 8d config 2
    (media/motivation-video.yaml:42-57)."""
    from adaptive_sph_amd.workloads import dam_break_params
    return dam_break_params(**kw)


def ring_scene(n_ring=20, h=0.05, radius_in_h=1.2, centre=(0.0, 0.0), jitter=0.0, seed=0):
    """A particle in the middle of a ring of `n_ring` others at ~1.2 h: n_ring + 1 entries on its list (> 19) and every fringe
    value 2 |x_ij| - 2 h_j in [0, h) -- the one geometry in which constrain_neighborhood_count (simulation.rs:2145-2177)
    passes its own two assertions.  The ring particles see at most ~17 neighbours and keep their h.  `jitter` spreads the
    ring radii (distinct fringe values)."""
    rng = np.random.default_rng(seed)
    ang = np.arange(n_ring, dtype=np.float64) * (2 * np.pi / n_ring)
    rad = radius_in_h * h * (1.0 + jitter * rng.uniform(-1.0, 1.0, n_ring))
    pos = np.zeros((n_ring + 1, 2), np.float32)
    pos[:, 0] = centre[0]
    pos[:, 1] = centre[1]
    pos[1:, 0] += (rad * np.cos(ang)).astype(np.float32)
    pos[1:, 1] += (rad * np.sin(ang)).astype(np.float32)
    m = np.float32(np.pi * (h / 1.9) ** 2)            # h = 1.9 sqrt(m / pi) at rest_density 1
    mass = np.full(n_ring + 1, m, np.float32)
    vel = np.zeros((n_ring + 1, 2), np.float32)
    return pos, mass, vel


def rings_and_block_scene():
    """Three jittered rings (21, 23, 25 list entries in the middle: ranks 2, 4, 6 of the descending fringe order) beside a
    rest-lattice block in which nobody exceeds 19 neighbours."""
    from adaptive_sph_amd import scene as sc
    parts = [ring_scene(20, 0.05, 1.2, (-1.0, 0.5), 0.08, 1), ring_scene(22, 0.04, 1.2, (0.0, 0.6), 0.08, 2),
             ring_scene(24, 0.06, 1.2, (1.0, 0.4), 0.08, 3)]
    scn = sc.dam_break_small(24, 24, 1 / 24)
    parts.append(sc.init_particles(scn))
    pos = np.concatenate([p[0] for p in parts]).astype(np.float32)
    mass = np.concatenate([p[1] for p in parts]).astype(np.float32)
    vel = np.concatenate([p[2] for p in parts]).astype(np.float32)
    return scn, pos, mass, vel


def quadtree_scene(seed):
    rng = np.random.default_rng(seed)
    levels = int(rng.integers(1, 6))                 # size ratio up to 32:1
    s_max = float(rng.choice([0.08, 0.1, 0.16]))
    kind = int(rng.integers(0, 4))
    w, hgt = float(rng.uniform(0.8, 2.4)), float(rng.uniform(0.5, 1.4))
    x0, y0 = -1.95 + float(rng.uniform(0, 0.3)), -0.95 + float(rng.uniform(0, 0.2))
    cx, cy = x0 + w * rng.uniform(0.2, 0.8), y0 + hgt * rng.uniform(0.2, 0.8)
    ang = rng.uniform(0, np.pi)

    def size_at(x, y):          # wanted particle spacing
        if kind == 0:           # fine disc in a coarse bath
            d = np.hypot(x - cx, y - cy)
            t = np.clip(d / (0.35 * min(w, hgt)), 0, 1)
        elif kind == 1:         # sharp slanted interface
            t = 1.0 if (x - cx) * np.cos(ang) + (y - cy) * np.sin(ang) > 0 else 0.0
        elif kind == 2:         # smooth gradient
            t = np.clip((x - x0) / w, 0, 1)
        else:                   # fine layer at the top ("free surface")
            t = np.clip((y0 + hgt - y) / (0.5 * hgt), 0, 1)
        return s_max * 2.0 ** (-levels * (1.0 - t))

    pts, sizes = [], []

    def rec(x, y, s, depth):
        if depth < levels and size_at(x + s / 2, y + s / 2) < s / 1.41:
            for dx in (0, 1):
                for dy in (0, 1):
                    rec(x + dx * s / 2, y + dy * s / 2, s / 2, depth + 1)
        else:
            j = rng.uniform(-0.15, 0.15, 2) * s
            pts.append((x + s / 2 + j[0], y + s / 2 + j[1]))
            sizes.append(s)

    nx, ny = max(int(w / s_max), 1), max(int(hgt / s_max), 1)
    for ix in range(nx):
        for iy in range(ny):
            rec(x0 + ix * s_max, y0 + iy * s_max, s_max, 0)
    pos = np.array(pts, np.float32)
    sizes = np.array(sizes, np.float32)
    mass = (np.float32(0.93) * sizes * sizes).astype(np.float32)
    vel = (rng.normal(0, 0.05, pos.shape)).astype(np.float32)
    return pos, mass, vel, dict(levels=levels, kind=kind, s_max=s_max)


def displacement_bars(x_gpu, x_oracle, x0, rel=1e-3):
    """Positions compared through the DISPLACEMENT from the uploaded positions x0 -- relative to max|x| (~2) a bar of 1e-4 is
    0.2 mm absolute, more than a particle at rest-lattice spacing 1/1024 moves in the first steps, so that bar alone cannot fail.
    Two figures, both against `rel` x the oracle's own displacement plus ONE ulp of the largest coordinate (positions are f32: two
    correct integrations may round a coordinate to neighbouring floats, and that quantum is 2e-3 of a 6e-5 displacement):
      * max over the particles of |dx_gpu - dx_oracle|  vs  rel * max|dx_oracle|  (the fastest particles)
      * median of the same                              vs  rel * median|dx_oracle|  (the bulk, which the ejected corners do not hide)
    Returns (ok, report); `report` carries the numbers for the assertion message.  A kernel that never moved a particle has
    error == displacement and fails both by three orders of magnitude."""
    xg, xo, x0 = (np.asarray(a, np.float64) for a in (x_gpu, x_oracle, x0))
    dg, do = xg - x0, xo - x0
    err = np.abs(dg - do).max(axis=1)
    mag = np.abs(do).max(axis=1)
    ulp = float(np.spacing(np.float32(np.abs(xo).max())))
    rep = {"max_err": float(err.max()), "max_disp": float(mag.max()), "median_err": float(np.median(err)), "median_disp": float(np.median(mag)), "ulp": ulp}
    meaningful = rep["max_disp"] > 50 * ulp and rep["median_disp"] > 10 * ulp      # the scene moved by more than rounding
    ok = meaningful and rep["max_err"] <= rel * rep["max_disp"] + ulp and rep["median_err"] <= rel * rep["median_disp"] + ulp
    return ok, rep


def same_sets(g, o):
    """Equal CSR offsets, per particle equal sums and sums of squares of the neighbour indices (a cheap first look), then the SETS
    entry by entry: the order inside a list is unspecified (cell-sorted on the device, ascending in the oracle)."""
    go, gi = g.download_neighbors()
    oo, oi = o.download_neighbors()
    assert np.array_equal(go, oo)
    starts = go[:-1].astype(np.int64)
    assert (np.diff(go.astype(np.int64)) > 0).all()          # every particle is on its own list
    for power in (1, 2):
        a = np.add.reduceat(gi.astype(np.uint64) ** power, starts)
        b = np.add.reduceat(oi.astype(np.uint64) ** power, starts)
        assert np.array_equal(a, b), power
    # ... and the sets themselves: (row, index) packed into one 64-bit key per entry, the device's entries sorted (the oracle's lists
    # are ascending already), compared whole -- 13 M entries at configs[1], 110 M at configs[3]
    rows = np.repeat(np.arange(len(go) - 1, dtype=np.uint64), np.diff(go.astype(np.int64))) << np.uint64(32)
    key_o = rows | oi.astype(np.uint64)
    assert (np.diff(key_o.astype(np.int64)) > 0).all()
    key_g = rows | gi.astype(np.uint64)
    key_g.sort()
    assert np.array_equal(key_g, key_o)
