"""Host-side mirror of ``SimulationParams`` (reference: src/simulation/simulation_parameters.rs:26-108).

Same field names, same enum spellings, same YAML behaviour:
  * the file is parsed to a mapping first, then ``update_attributes`` / ``-c`` overrides are applied and
    must hit an EXISTING key (main_loop.rs:113-126, animation/mod.rs:89-96 panic otherwise);
  * every non-``Option`` field is mandatory (serde derive without defaults);
  * ``Option`` fields (pull_fluid_to, fill_stash_with, operator_discretization_for_diagonal) default to None.
Only the subset the step reads crosses the C ABI (`to_ffi`, include/sph_ffi.h `sph_params`).
"""
from __future__ import annotations

from dataclasses import dataclass, fields
from typing import Any, Mapping, Optional, Sequence

import yaml

from . import ffi

_OPTIONAL = {"pull_fluid_to", "fill_stash_with", "operator_discretization_for_diagonal"}

_ENUMS = {
    "viscosity_type": ffi.VISCOSITY_TYPE,
    "level_estimation_method": ffi.LEVEL_ESTIMATION_METHOD,
    "neighborhood_search_algorithm": {"Grid": 0, "RStar": 1},
    "init_boundary_handler": {"Particles": 0, "AnalyticUnderestimate": 1, "AnalyticOverestimate": 2, "NoBoundary": 3},
    "support_length_estimation": ffi.SUPPORT_LENGTH_ESTIMATION,
    "pressure_solver_method": ffi.PRESSURE_SOLVER_METHOD,
    "hybrid_dfsph_density_source_term": ffi.HYBRID_DFSPH_DENSITY_SOURCE_TERM,
    "boundary_penalty_term": ffi.BOUNDARY_PENALTY_TERM,
    "sizing_function": ffi.SIZING_FUNCTION,
    "operator_discretization": ffi.OPERATOR_DISCRETIZATION,
}


@dataclass
class SimulationParams:
    # order follows simulation_parameters.rs:26-108
    rest_density: float
    cfl_factor: float
    max_dt: float
    h: float
    use_iisph: bool
    viscosity: float
    viscosity_type: str
    gravity: float
    check_aii: bool
    level_estimation_method: str
    maximum_range: float
    jacobi_omega: float
    eos_stiffness: float
    eos_power: int
    neighborhood_search_algorithm: str
    init_boundary_handler: str
    support_length_estimation: str
    sdf_gradient_eps: float
    fail_on_missing_split_pattern: bool
    constrain_neighborhood_count: bool
    particle_radius_fine: float
    particle_radius_base: float
    maximum_surface_distance: float
    minimum_share_partners: int
    minimum_merge_partners: int
    merging: bool
    sharing: bool
    splitting: bool
    max_mass_transfer_sharing: float
    max_mass_transfer_merging: float
    max_share_distance: float
    max_merge_distance: float
    allow_merge_with_optimal_particle: bool
    allow_share_with_optimal_particle: bool
    allow_share_with_too_small_particle: bool
    allow_merge_on_size_difference: bool
    boundary_is_fluid_surface: bool
    use_extended_range_for_level_estimation: bool
    pressure_solver_method: str
    iisph_max_avg_density_error: float
    hybrid_dfsph_factor: float
    hybrid_dfsph_max_avg_density_error: float
    hybrid_dfsph_max_avg_divergence_error: float
    hybrid_dfsph_density_source_term: str
    hybrid_dfsph_non_pressure_accel_before_divergence_free: bool
    check_neighborhood: bool
    boundary_penalty_term: str
    sizing_function: str
    level_estimation_after_advection: bool
    level_estimation_range: float
    operator_discretization: str
    max_iters: int
    pull_fluid_to: Optional[Sequence[float]] = None
    fill_stash_with: Optional[str] = None
    operator_discretization_for_diagonal: Optional[str] = None

    # ---- construction -------------------------------------------------------------------------
    @classmethod
    def from_mapping(cls, m: Mapping[str, Any]) -> "SimulationParams":
        kw = {}
        for f in fields(cls):
            if f.name in m:
                v = m[f.name]
            elif f.name in _OPTIONAL:
                v = None
            else:
                raise KeyError(f"failed to unpack SimulationParams: missing field `{f.name}`")
            if f.name in _ENUMS:
                v = "None" if v is None else str(v)  # a YAML null can only mean the variant `None`
                if v not in _ENUMS[f.name]:
                    raise ValueError(f"unknown variant `{v}` for {f.name}, expected one of {list(_ENUMS[f.name])}")
            elif f.type == "bool":
                if not isinstance(v, bool):
                    raise TypeError(f"invalid type for {f.name}: expected a boolean, got {v!r}")
            elif f.type == "int":
                v = int(v)
            elif f.type == "float":
                v = float(v)
            kw[f.name] = v
        return cls(**kw)

    @classmethod
    def from_yaml(cls, text_or_path, update_attributes: Optional[Mapping[str, Any]] = None) -> "SimulationParams":
        m = load_yaml_mapping(text_or_path)
        if update_attributes:
            apply_overrides(m, update_attributes)
        return cls.from_mapping(m)

    def replace(self, **kw) -> "SimulationParams":
        d = {f.name: getattr(self, f.name) for f in fields(self)}
        for k, v in kw.items():
            if k not in d:
                raise KeyError(f"not able to find attribute {k}")
            d[k] = v
        return SimulationParams.from_mapping(d)

    # ---- boundary crossing --------------------------------------------------------------------
    def to_ffi(self) -> ffi.SphParams:
        if self.neighborhood_search_algorithm == "Grid":
            # build_neighborhood_list (neighborhood_search.rs:334-342): the grid search exists only in the uniform-particle-sizes
            # build; the default (adaptive) build the library replaces panics on it
            raise ValueError("assertion failed: PARTICLE_SIZES == ParticleSizes::Uniform (neighborhood_search_algorithm: Grid)")
        p = ffi.SphParams()
        p.rest_density = self.rest_density
        p.cfl_factor = self.cfl_factor
        p.max_dt = self.max_dt
        p.viscosity = self.viscosity
        p.viscosity_type = ffi.VISCOSITY_TYPE[self.viscosity_type]
        p.gravity = self.gravity
        p.jacobi_omega = self.jacobi_omega
        p.level_estimation_method = ffi.LEVEL_ESTIMATION_METHOD[self.level_estimation_method]
        p.maximum_range = self.maximum_range
        p.support_length_estimation = ffi.SUPPORT_LENGTH_ESTIMATION[self.support_length_estimation]
        p.sdf_gradient_eps = self.sdf_gradient_eps
        if self.pull_fluid_to is not None:
            p.has_pull_fluid_to = 1
            for k in range(3):
                p.pull_fluid_to[k] = float(self.pull_fluid_to[k])
        p.maximum_surface_distance = self.maximum_surface_distance
        p.boundary_is_fluid_surface = int(self.boundary_is_fluid_surface)
        p.use_extended_range_for_level_estimation = int(self.use_extended_range_for_level_estimation)
        p.level_estimation_after_advection = int(self.level_estimation_after_advection)
        p.level_estimation_range = self.level_estimation_range
        p.pressure_solver_method = ffi.PRESSURE_SOLVER_METHOD[self.pressure_solver_method]
        p.iisph_max_avg_density_error = self.iisph_max_avg_density_error
        p.hybrid_dfsph_factor = self.hybrid_dfsph_factor
        p.hybrid_dfsph_max_avg_density_error = self.hybrid_dfsph_max_avg_density_error
        p.hybrid_dfsph_max_avg_divergence_error = self.hybrid_dfsph_max_avg_divergence_error
        p.hybrid_dfsph_density_source_term = ffi.HYBRID_DFSPH_DENSITY_SOURCE_TERM[self.hybrid_dfsph_density_source_term]
        p.hybrid_dfsph_non_pressure_accel_before_divergence_free = int(
            self.hybrid_dfsph_non_pressure_accel_before_divergence_free)
        p.boundary_penalty_term = ffi.BOUNDARY_PENALTY_TERM[self.boundary_penalty_term]
        p.operator_discretization = ffi.OPERATOR_DISCRETIZATION[self.operator_discretization]
        p.max_iters = self.max_iters
        p.check_neighborhood = int(self.check_neighborhood)
        p.check_aii = int(self.check_aii)
        p.constrain_neighborhood_count = int(self.constrain_neighborhood_count)
        p.fill_stash_with = ffi.FILL_STASH_WITH[self.fill_stash_with]
        p.sizing_function = ffi.SIZING_FUNCTION[self.sizing_function]
        p.particle_radius_fine = self.particle_radius_fine
        p.particle_radius_base = self.particle_radius_base
        return p


def load_yaml_mapping(text_or_path) -> dict:
    s = str(text_or_path)
    if "\n" not in s and (s.endswith(".yaml") or s.endswith(".yml")):
        with open(s, "r") as fh:
            s = fh.read()
    m = yaml.safe_load(s)
    if not isinstance(m, dict):
        raise TypeError("cannot get parsed simulation parameters as mapping")
    # YAML 1.1 turns the bare word `None` into the string "None" (kept) but `null`/`~` into None
    return m


def apply_overrides(mapping: dict, overrides: Mapping[str, Any]) -> None:
    """main_loop.rs:113-126 / animation/mod.rs:89-96: an override must hit an existing key."""
    for k, v in overrides.items():
        if k not in mapping:
            raise KeyError(f"not able to find attribute {k}")
        mapping[k] = v
