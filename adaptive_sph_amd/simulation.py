"""Host-side mirror of the reference's simulation API for the hot path.

Reference (src/simulation/simulation.rs):
  ``FluidSimulation``                       :471-533   (particles, neighs, boundary_handler, time)
  ``init_fluid_sim`` / ``init_simulation_params``   :3074-3256
  ``single_step_without_adaptivity``        :1980-2730  -> dt     <== the call that crosses the C ABI
  ``single_step``                           :1973-1978
  ``write_statistics``                      :3279-3359

Same names, same argument meaning, same error behaviour (the reference panics; here `ffi.SphError`
carries the status code of the guard that fired).  All arithmetic of the step lives in the HIP library;
this file only moves arrays across the boundary and keeps the reference's counters.
"""
from __future__ import annotations

import time as _time
from typing import Dict, List, Optional

import numpy as np

from . import ffi
from .scene import SceneConfig, boundary_planes, init_particles
from .simulation_parameters import SimulationParams


class _ParticleView:
    """``fluid_simulation.particles.<field>`` -> numpy array downloaded in host particle order
    (ParticleVec, simulation.rs:284-334)."""

    def __init__(self, ctx: ffi.Context):
        object.__setattr__(self, "_ctx", ctx)

    def __getattr__(self, name: str) -> np.ndarray:
        if name in ffi.FIELDS:
            return self._ctx.download(name)
        raise AttributeError(name)

    def __setattr__(self, name: str, value):
        if name in ffi.FIELDS:
            self._ctx.upload_field(name, value)
        else:
            raise AttributeError(name)


def rust_display(x, f32: bool = False) -> str:
    """What Rust's `{}` prints for an f32 / f64: the shortest digits that round-trip in that type, never an exponent, no
    trailing ".0" (`1035f32` -> "1035", `0.1f32` -> "0.1"), "NaN" / "inf" for the non-finite."""
    v = np.float32(x) if f32 else np.float64(x)
    if np.isnan(v):
        return "NaN"
    if np.isinf(v):
        return "inf" if v > 0 else "-inf"
    return np.format_float_positional(v, unique=True, trim="-")


class _Counter:
    """Counter<FT> (values: f32; avg = sequential f32 sum / f32 count, simulation.rs:96-106)."""

    def __init__(self):
        self.values: List[float] = []

    def add_value(self, v: float):
        self.values.append(float(np.float32(v)))

    def avg(self):
        if not self.values:
            return float("nan")
        acc = np.float32(0)
        for v in self.values:
            acc = np.float32(acc + np.float32(v))
        return float(np.float32(acc / np.float32(len(self.values))))

    def min(self):
        return float(min(self.values)) if self.values else float(np.finfo(np.float32).max)     # fold from FT::max_value()

    def max(self):
        return float(max(self.values)) if self.values else float(np.finfo(np.float32).min)


class _PCounter:
    """Counter<Duration> (simulation.rs:108-135): whole nanoseconds; avg = sum / count in integer nanoseconds."""

    def __init__(self):
        self.values: List[int] = []

    def add_value(self, ms: float):
        self.values.append(int(round(float(ms) * 1e6)))

    def add_to_last(self, ms: float):
        """end_add_to_last (simulation.rs:159-189): the duration joins the LAST sample instead of becoming one."""
        if self.values:
            self.values[-1] += int(round(float(ms) * 1e6))
        else:
            self.add_value(ms)

    def sum_secs(self) -> float:
        return sum(self.values) / 1e9

    def avg_secs(self) -> float:
        return (sum(self.values) // len(self.values)) / 1e9 if self.values else float("nan")


class FluidSimulation:
    """Drop-in for ``FluidSimulation<DimensionUtils2d, 2>`` on the step path."""

    def __init__(self, position, velocity, mass, planes, counters_enabled: bool = False,
                 lib: Optional[ffi.SphLibrary] = None, device_id: int = 0, n_capacity: Optional[int] = None, split_patterns=None):
        self.lib = lib if lib is not None else ffi.load_product()
        mass = np.ascontiguousarray(mass, dtype=np.float32)
        n = mass.shape[0]
        self.ctx = ffi.Context(self.lib, n_capacity if n_capacity is not None else max(n, 1), planes, device_id)
        self.ctx.upload(mass, position, velocity)
        self.particles = _ParticleView(self.ctx)
        self.split_patterns = split_patterns       # adaptivity.SplitPatterns (FluidSimulation.split_patterns, simulation.rs:476)
        self._adaptivity = None
        self.counters_enabled = counters_enabled
        if counters_enabled and self.lib.profile_enable is not None:
            # PerformanceCounters::new(counters_enabled) (simulation.rs:137-189): per-phase times need the library's event
            # instrumentation (neighbourhood / level estimation / the two solves); it perturbs dispatch, like `-p` does
            self.ctx.profile_enable(1)
        self.pcounters: Dict[str, _PCounter] = {}
        self.vcounters: Dict[str, _Counter] = {}
        self.step_number = 0
        self.last_stats: Optional[ffi.SphStepStats] = None

    # ---- reference surface ---------------------------------------------------------------------
    @property
    def time(self) -> float:
        return self.ctx.time

    def num_fluid_particles(self) -> int:
        return self.ctx.n

    def single_step_without_adaptivity(self, simulation_params: SimulationParams) -> float:
        """simulation.rs:1980-2730; returns dt (:2729)."""
        p = simulation_params.to_ffi() if isinstance(simulation_params, SimulationParams) else simulation_params
        t0 = _time.perf_counter()
        st = self.ctx.step(p)
        wall_ms = (_time.perf_counter() - t0) * 1e3
        self.last_stats = st
        self.step_number = int(st.step_number)
        if self.counters_enabled:
            self._v("particle-count", st.n_particles)               # :1990-1991
            self._v("dt", st.dt)                                    # :2202
            if st.div_solver.iters > 0:
                self._v("div-iterations", st.div_solver.iters)      # :2542-2544
            if st.density_solver.iters > 0:
                self._v("density-iterations", st.density_solver.iters)  # :2617-2619
            self._p("simulation-step", wall_ms)
            self._p("neighborhood", st.ms_neighborhood)
            self._p("level-estimation", st.ms_level_estimation)
            self._p("div-solver", st.ms_div_solver)
            self._p("density-solver", st.ms_density_solver)
        return float(st.dt)

    def single_step(self, simulation_params: SimulationParams) -> None:
        """simulation.rs:1973-1978: the step, then single_step_adaptivity with its dt."""
        dt = self.single_step_without_adaptivity(simulation_params)
        self.single_step_adaptivity(simulation_params, dt)

    def single_step_adaptivity(self, simulation_params: SimulationParams, dt: float) -> dict:
        """simulation.rs:2732-2796.  The partner decisions are taken here on the host (the reference's sequential loops,
        adaptivity.py); the particle data stays on the device (sph_share_particles / sph_merge_particles / sph_split_particles)."""
        P = simulation_params
        if not (P.sharing or P.merging or P.splitting):
            return {"n_before": self.ctx.n, "n_after": self.ctx.n, "shares": 0, "merges": 0, "splits": 0}
        if self._adaptivity is None:
            from .adaptivity import AdaptivityDriver
            if P.splitting and self.split_patterns is None:
                raise RuntimeError("splitting needs split patterns: FluidSimulation(..., split_patterns=SplitPatterns.load_from_file('split-patterns.yaml'))")
            self._adaptivity = AdaptivityDriver(self.ctx, self.split_patterns)
        t0 = _time.perf_counter()
        info = self._adaptivity.single_step_adaptivity(P, dt, self.step_number)
        if self.counters_enabled:
            ms = (_time.perf_counter() - t0) * 1e3
            self._p("adaptivity", ms)                                        # pcounters "adaptivity" (:2734, 2794)
            # :2733, 2795: begin("simulation-step") ... end_add_to_last("simulation-step") -- the adaptivity time belongs to the
            # step's own sample (and so to `simulation-time`), it does not open a new one
            self.pcounters.setdefault("simulation-step", _PCounter()).add_to_last(ms)
        return info

    def neighbors(self):
        """NeighborhoodCache as CSR (offsets, indices), host particle order."""
        return self.ctx.download_neighbors()

    # ---- counters / statistics (simulation.rs:137-189, 3279-3359) -------------------------------
    def _v(self, key, v):
        self.vcounters.setdefault(key, _Counter()).add_value(v)

    def _p(self, key, ms):
        self.pcounters.setdefault(key, _PCounter()).add_value(ms)

    def write_statistics(self) -> str:
        """write_statistics (simulation.rs:3279-3359), number for number: `{:.2}` / `{:.02}` for the LaTeX row, Rust's `{}` for
        everything else (rust_display); counters sorted by label."""
        pc, vc = self.pcounters, self.vcounters
        sim_s = pc["simulation-step"].sum_secs()
        lines = []
        lines.append("${:.2f}\\si{{\\second}}$ & {} & {:.2f} & {:.2f} & - \\\\".format(
            sim_s, int(np.copysign(np.floor(abs(vc["particle-count"].avg()) + 0.5), vc["particle-count"].avg())),   # f32::round: half away from zero
            vc["div-iterations"].avg() if "div-iterations" in vc else float("nan"),
            vc["density-iterations"].avg() if "density-iterations" in vc else float("nan")))
        lines.append("")
        lines.append(f"simulation-time: {rust_display(sim_s * 1000.)}ms")
        lines.append("")
        for label in sorted(pc):
            lines.append(f"{label}: avg:{rust_display(pc[label].avg_secs() * 1000.)}ms")
        lines.append("")
        for label in sorted(vc):
            c = vc[label]
            lines.append(f"{label}: min:{rust_display(c.min(), True)} max:{rust_display(c.max(), True)} avg:{rust_display(c.avg(), True)}")
        return "\n".join(lines) + "\n"

    def close(self):
        self.ctx.close()


def init_simulation_params(simulation_params: SimulationParams, scene_config: SceneConfig) -> SimulationParams:
    """simulation.rs:3233-3256 (adaptive build: h is unused and set to 0)."""
    return simulation_params.replace(h=0.0)


def init_fluid_sim(simulation_params: SimulationParams, scene_config: SceneConfig, counters_enabled: bool = False,
                   lib: Optional[ffi.SphLibrary] = None, device_id: int = 0, split_patterns=None,
                   n_capacity: Optional[int] = None) -> FluidSimulation:
    """simulation.rs:3074-3231.  `n_capacity`: room for the particles splitting will add (the reference's Vecs grow on demand;
    the device arrays are sized once)."""
    pos, mass, vel = init_particles(scene_config)
    planes = boundary_planes(scene_config.boundary, simulation_params.init_boundary_handler)
    return FluidSimulation(pos, vel, mass, planes, counters_enabled, lib=lib, device_id=device_id, split_patterns=split_patterns,
                           n_capacity=n_capacity)


def run_until(fluid_simulation: FluidSimulation, simulation_params: SimulationParams, max_seconds: float,
              max_steps: Optional[int] = None) -> int:
    """The GUI-free loop of the image harness (platform/desktop/animation/mod.rs:138-272) /
    `run --max-seconds` (main_loop.rs:346-350): step until simulated time >= max_seconds."""
    steps = 0
    while fluid_simulation.time < max_seconds and (max_steps is None or steps < max_steps):
        fluid_simulation.single_step_without_adaptivity(simulation_params)
        steps += 1
    return steps
