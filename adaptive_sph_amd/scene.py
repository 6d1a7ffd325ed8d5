"""Host-side mirror of the scene format and initialiser.

Reference: ``SceneConfig`` / ``SceneBoundary`` / ``SceneFluidBlock`` (src/simulation/simulation.rs:3052-3072),
``add_fluid_block`` (:2915-2983), ``init_fluid_sim`` (:3074-3231), ``SdfPlane::new_boundary_box``
(src/simulation/sdf/sdf_plane.rs:13-20).

All arithmetic is float32 exactly as in the reference: ``floor(size / spacing)`` particles per axis,
positions ``idx * spacing + min`` (x outer loop, y inner loop), mass ``spacing^2 * volume_fill_ratio * 1``.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Sequence, Tuple

import numpy as np
import yaml

f32 = np.float32


@dataclass
class SceneBoundary:
    type: str
    width: float
    height: float


@dataclass
class SceneFluidBlock:
    pos: Sequence[float]
    size: Sequence[float]
    spacing: float
    volume_fill_ratio: float
    velocity: Sequence[float]


@dataclass
class SceneConfig:
    boundary: SceneBoundary
    blocks: List[SceneFluidBlock] = field(default_factory=list)

    @classmethod
    def from_mapping(cls, m) -> "SceneConfig":
        b = m["boundary"]
        return cls(SceneBoundary(str(b["type"]), float(b["width"]), float(b["height"])),
                   [SceneFluidBlock(list(x["pos"]), list(x["size"]), float(x["spacing"]),
                                    float(x["volume_fill_ratio"]), list(x["velocity"])) for x in m["blocks"]])

    @classmethod
    def from_yaml(cls, text_or_path) -> "SceneConfig":
        s = str(text_or_path)
        if "\n" not in s and (s.endswith(".yaml") or s.endswith(".yml")):
            with open(s, "r") as fh:
                s = fh.read()
        return cls.from_mapping(yaml.safe_load(s))


def block_dims(block: SceneFluidBlock) -> Tuple[int, int]:
    """num_particles_x / _y of add_fluid_block (simulation.rs:2968-2971), float32 floor."""
    min_x, min_y = f32(block.pos[0]), f32(block.pos[1])
    max_x, max_y = min_x + f32(block.size[0]), min_y + f32(block.size[1])  # init_fluid_sim :3090-3091
    s = f32(block.spacing)
    nx = int(np.floor((max_x - min_x) / s))
    ny = int(np.floor((max_y - min_y) / s))
    return max(nx, 0), max(ny, 0)


def add_fluid_block(block: SceneFluidBlock):
    """-> (position[n,2], mass[n], velocity[n,2]) float32, particle order x-outer / y-inner."""
    nx, ny = block_dims(block)
    s = f32(block.spacing)
    min_x, min_y = f32(block.pos[0]), f32(block.pos[1])
    xs = np.arange(nx, dtype=np.float32) * s + min_x
    ys = np.arange(ny, dtype=np.float32) * s + min_y
    pos = np.empty((nx, ny, 2), dtype=np.float32)
    pos[:, :, 0] = xs[:, None]
    pos[:, :, 1] = ys[None, :]
    particle_volume = s * s * f32(block.volume_fill_ratio)
    particle_mass = particle_volume * f32(1.0)  # INIT_REST_DENSITY (simulation.rs:344)
    n = nx * ny
    mass = np.full(n, particle_mass, dtype=np.float32)
    vel = np.empty((n, 2), dtype=np.float32)
    vel[:, 0] = f32(block.velocity[0])
    vel[:, 1] = f32(block.velocity[1])
    return pos.reshape(n, 2), mass, vel


@dataclass
class BoundaryPolygon:
    """One Sdf2D connected component (sdf/sdf2d.rs): closed polygon, air on the left of every edge."""
    points: List[Tuple[float, float]]


def boundary_planes(boundary: SceneBoundary, init_boundary_handler: str = "AnalyticOverestimate"):
    """The boundary handler built by init_fluid_sim (:3137-3213): planes (dir_x, dir_y, delta) of the SdfPlane box
    (AnalyticOverestimate), or the BoundaryPolygon of Sdf2D::new_boundary_box (AnalyticUnderestimate, sdf2d.rs:167-179)."""
    if init_boundary_handler == "NoBoundary":
        return []
    if init_boundary_handler == "AnalyticUnderestimate":
        if boundary.type != "box":
            raise NotImplementedError(f"boundary type {boundary.type!r}")
        w, h = f32(boundary.width), f32(boundary.height)
        min_x, min_y = f32(0.0) - w / f32(2.0), f32(0.0) - h / f32(2.0)
        max_x, max_y = f32(0.0) + w / f32(2.0), f32(0.0) + h / f32(2.0)
        return BoundaryPolygon([(float(min_x), float(min_y)), (float(max_x), float(min_y)), (float(max_x), float(max_y)),
                                (float(min_x), float(max_y))])
    if init_boundary_handler != "AnalyticOverestimate":
        # Particles = Akinci boundary particles (ParticleBasedBoundaryHandler): not on the covered path
        raise NotImplementedError(f"init_boundary_handler={init_boundary_handler} is outside the covered path "
                                  f"(the SdfPlane box of AnalyticOverestimate and the Sdf2D box of AnalyticUnderestimate are)")
    if boundary.type != "box":
        raise NotImplementedError(f"boundary type {boundary.type!r}")
    w, h = f32(boundary.width), f32(boundary.height)
    min_x, min_y = f32(0.0) - w / f32(2.0), f32(0.0) - h / f32(2.0)
    max_x, max_y = f32(0.0) + w / f32(2.0), f32(0.0) + h / f32(2.0)
    return [(1.0, 0.0, float(-min_x)), (-1.0, 0.0, float(max_x)), (0.0, 1.0, float(-min_y)), (0.0, -1.0, float(max_y))]


def init_particles(scene: SceneConfig):
    """Concatenated blocks in scene order (init_fluid_sim :3088-3099)."""
    ps, ms, vs = [], [], []
    for b in scene.blocks:
        p, m, v = add_fluid_block(b)
        ps.append(p)
        ms.append(m)
        vs.append(v)
    if not ps:
        return (np.zeros((0, 2), np.float32), np.zeros(0, np.float32), np.zeros((0, 2), np.float32))
    return np.concatenate(ps), np.concatenate(ms), np.concatenate(vs)


# ---- the BASELINE.json workloads as SceneConfig ---------------------------

def dam_break_1m() -> SceneConfig:
    """configs[1]: 1024 x 1024 = 1 048 576 uniform particles, box 4 x 2."""
    return SceneConfig(SceneBoundary("box", 4.0, 2.0),
                       [SceneFluidBlock([-1.999, -0.999], [1.0005, 1.0005], 0.0009765625, 0.93, [0.0, 0.0])])


def dam_break_1m_adaptive() -> SceneConfig:
    """configs[2]: 942 080 fine + 58 880 coarse particles, radius ratio 4:1."""
    return SceneConfig(SceneBoundary("box", 4.0, 2.0),
                       [SceneFluidBlock([-1.999, -0.999], [1.0005, 0.8989], 0.0009765625, 0.93, [0.0, 0.0]),
                        SceneFluidBlock([1.0, -0.996], [0.9, 1.0005], 0.00390625, 0.93, [0.0, 0.0])])


def dam_break_1m_adaptive_contact(gap: float = 0.00390625) -> SceneConfig:
    """configs[2]'s two blocks -- the same sizes, spacings and counts (942 080 fine + 58 880 coarse, 4:1 radii) -- with the coarse block
    moved against the fine one: its first column stands one coarse spacing (1/256) right of the fine block's last column (x = -1.999 +
    1023/1024), so the symmetric (h_i + h_j) / 2 rule of sph_kernels.rs:273-278 is at work from step 0 (BASELINE's placement leaves 2.0
    between the blocks: they meet after thousands of steps).  Same floor offsets as configs[2]."""
    return SceneConfig(SceneBoundary("box", 4.0, 2.0),
                       [SceneFluidBlock([-1.999, -0.999], [1.0005, 0.8989], 0.0009765625, 0.93, [0.0, 0.0]),
                        SceneFluidBlock([-1.999 + 1023.0 / 1024.0 + gap, -0.996], [0.9, 1.0005], 0.00390625, 0.93, [0.0, 0.0])])


def dam_break_8m() -> SceneConfig:
    """configs[3]: 8192 x 1024 = 8 388 608 particles: configs[1]'s column eight times as wide (same spacing, same height, same
    parameters), in a box 16 x 2.  (Until round 4 this was a 2896 x 2896 column at spacing 1/2048 with max_dt = 0.001: that scene
    DIVERGES at step 3 -- largest speed 6 -> 74 -> 566 -> 1580 -> 3800 m/s, dt 1e-7, particles below the floor -- in the CPU oracle
    identically: the relaxed Jacobi solves stop on AVERAGE errors long before 2896 layers have seen the floor.  Round 4 wrote "and for
    max_dt = 0.0005 and 0.00025 as well"; the record committed in round 5 (profiles/r5_config3_divergence.md) shows 0.0005 diverging
    ten times milder and 0.00025 NOT diverging -- the surveyed geometry is back as `dam_break_8m_spec` with that value.)"""
    return dam_break_weak(8)


def dam_break_8m_spec() -> SceneConfig:
    """SURVEY.md section 8d config 4 AS WRITTEN: box 4 x 2, one block pos [-1.9995, -0.9995], size [1.4143, 1.4143], spacing 1/2048
    -> 2896 x 2896 = 8 386 816 particles.  Kept under its own name since configs[3] became `dam_break_8m` (round 4).  With max_dt 0.001
    this column blows up at step 3 on the reference's algorithm (3 800 m/s, dt 1e-7), with 0.0005 at step 6 (107 m/s); with 0.00025
    it behaves like configs[1] (8-10 m/s, density 1.7): scripts/gpu_config3_divergence.py steps it on the CPU oracle and on the device
    for the three values, output in profiles/r5_config3_divergence.md.  workloads.py runs it with 0.00025 (bench.py: strong_8m_spec)."""
    return SceneConfig(SceneBoundary("box", 4.0, 2.0),
                       [SceneFluidBlock([-1.9995, -0.9995], [1.4143, 1.4143], 0.00048828125, 0.93, [0.0, 0.0])])


def ratio_stress_4m() -> SceneConfig:
    """configs[4]: the geometry of media/ratio-stress-test-scene.yaml at 50:1 radius ratio with 4 002 768 fine + 1 575 coarse
    particles."""
    return SceneConfig(SceneBoundary("box", 2.0, 2.0),
                       [SceneFluidBlock([0.4, -0.5], [0.55, 1.4], 0.021925, 0.93, [0.0, 0.0]),
                        SceneFluidBlock([-0.95, -0.5], [0.55, 1.4], 0.0004385, 0.93, [0.0, 0.0])])


def ratio_stress_4m_settled() -> SceneConfig:
    """configs[4]'s two blocks STANDING ON THE FLOOR (VERDICT r4 missing 4 / next 6): the reference scene hangs both blocks 0.5 above the
    floor (media/ratio-stress-test-scene.yaml:5-15), so its first ~0.3 s are free fall -- IISPH's Jacobi loop sees all-negative pressures
    and leaves after one iteration.  Here the same blocks (same x positions, spacings and sizes: 4 002 768 fine + 1 575 coarse particles)
    start one fine spacing above the floor: hydrostatic pressure builds from step 0 and the solver iterates.  (Moving the blocks AGAINST
    each other as well was tried and is no benchmark scene: with h_ij = (h_i + h_j) / 2 a coarse particle adds m_c W(r, h_c / 2) ~ 1.5 rho_0 to
    every fine particle within half its smoothing length, IISPH answers the density error with velocities ~ 1 / dt -- 64, 159, 319 m/s
    at step 0 for max_dt 2.5e-4, 1e-4, 5e-5 -- and the state is NaN by step 10, on the oracle alike; the interface at 50:1 is covered by
    the forced-count parity test test_config4_ratio_stress_4m_blocks_in_contact.)"""
    return SceneConfig(SceneBoundary("box", 2.0, 2.0),
                       [SceneFluidBlock([0.4, -0.97755], [0.55, 1.4], 0.021925, 0.93, [0.0, 0.0]),
                        SceneFluidBlock([-0.95, -0.99955], [0.55, 1.4], 0.0004385, 0.93, [0.0, 0.0])])


def dam_break_weak(n_gpus: int) -> SceneConfig:
    """Weak-scaling family between configs[1] (1 GPU, 1M) and configs[3] (8 GPUs, 8M): configs[1]'s column n times as wide --
    (1024 n) x 1024 particles at spacing 1/1024, one configs[1] per x-slab -- in a box of twice the column's width (4 x 2 up to
    n = 2).  (Taller columns at finer spacing were tried first -- 1448 x 1448 at 1/1448, 2048 x 2048 and 2896 x 2896 at 1/2048 with
    proportionally scaled max_dt: they all diverge within four steps, in the CPU oracle identically, so they are no benchmark
    scenes.)"""
    n = max(int(n_gpus), 1)
    if n == 1:
        return dam_break_1m()
    width = 4.0 if n <= 2 else 2.0 * n
    return SceneConfig(SceneBoundary("box", width, 2.0),
                       [SceneFluidBlock([-0.5 * width + 0.001, -0.999], [n + 0.0005, 1.0005], 0.0009765625, 0.93, [0.0, 0.0])])


def dam_break_small(nx: int = 64, ny: int = 64, spacing: float = 1.0 / 64.0) -> SceneConfig:
    """A small dam break with the proportions of configs[1] (column one spacing off the left/bottom wall),
    for parity tests."""
    off = spacing * 1.024   # configs[1]: 0.001 / (1/1024)
    eps = spacing * 0.5
    return SceneConfig(SceneBoundary("box", 4.0, 2.0),
                       [SceneFluidBlock([-2.0 + off, -1.0 + off], [nx * spacing + eps, ny * spacing + eps], spacing, 0.93,
                                        [0.0, 0.0])])
