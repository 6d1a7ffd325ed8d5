"""BASELINE.json workloads: parameter sets and scenes.

`DEFAULT_CONFIG` is the content of the reference's default-config.yaml as a mapping (tests check it
against tests/golden/default-config.yaml); the dam-break configs apply the reference's own uniform
recipe on top (media/motivation-video.yaml:42-57 + section 8d overrides).
"""
from __future__ import annotations

from . import scene as sc
from .simulation_parameters import SimulationParams, apply_overrides

DEFAULT_CONFIG = dict(
    rest_density=1, cfl_factor=0.4, max_dt=0.006, h=0.0, use_iisph=True, eos_power=7, eos_stiffness=80,
    viscosity_type="ApproxLaplace", viscosity=0.003, jacobi_omega=0.5, gravity=-9.81, check_neighborhood=False,
    maximum_range=5.0, level_estimation_method="EmptyAngle", neighborhood_search_algorithm="RStar",
    init_boundary_handler="AnalyticOverestimate", support_length_estimation="FromMass",
    constrain_neighborhood_count=False, maximum_surface_distance=8.0, particle_radius_base=0.7,
    particle_radius_fine=0.005, merging=True, sharing=True, splitting=True, minimum_share_partners=0,
    minimum_merge_partners=0, max_mass_transfer_sharing=400000, max_mass_transfer_merging=100,
    allow_share_with_optimal_particle=False, allow_share_with_too_small_particle=False,
    allow_merge_with_optimal_particle=False, allow_merge_on_size_difference=False, boundary_is_fluid_surface=False,
    max_merge_distance=1.6, max_share_distance=1.6, hybrid_dfsph_factor=0.0, hybrid_dfsph_max_avg_density_error=0.01,
    hybrid_dfsph_max_avg_divergence_error=0.001, hybrid_dfsph_density_source_term="DensityAndDivergence",
    hybrid_dfsph_non_pressure_accel_before_divergence_free=True, iisph_max_avg_density_error=0.002,
    sdf_gradient_eps=0.00001, fail_on_missing_split_pattern=False, use_extended_range_for_level_estimation=True,
    boundary_penalty_term="Quadratic1", sizing_function="Radius", level_estimation_after_advection=False,
    level_estimation_range=5.5, max_iters=1000, operator_discretization="ConsistentSimpleGradient", check_aii=False,
    pressure_solver_method="HybridDFSPH",
)

DAM_BREAK_OVERRIDES = dict(
    merging=False, sharing=False, splitting=False, level_estimation_method="None",
    support_length_estimation="FromMass", pressure_solver_method="HybridDFSPH", hybrid_dfsph_factor=20000000.0,
    max_dt=0.002, viscosity=0.001, max_iters=200,
)


def default_params(**overrides) -> SimulationParams:
    m = dict(DEFAULT_CONFIG)
    apply_overrides(m, overrides)
    return SimulationParams.from_mapping(m)


def dam_break_params(**overrides) -> SimulationParams:
    m = dict(DEFAULT_CONFIG)
    apply_overrides(m, DAM_BREAK_OVERRIDES)
    apply_overrides(m, overrides)
    return SimulationParams.from_mapping(m)


def dam_break_params_scaled(spacing: float):
    """configs[1]'s recipe with max_dt scaled to keep dt/spacing of configs[1] (0.002 at spacing 1/1024).
    The literal max_dt = 0.002 diverges at spacing 1/2048 -- in the CPU oracle as well: the divergence solve
    hits max_iters at step 2, the density solve at step 3, and particles leave the box -- so the finer
    weak-scaling scenes use 0.002 * spacing * 1024."""
    def f(**overrides):
        kw = dict(max_dt=0.002 * spacing * 1024.0)
        kw.update(overrides)
        return dam_break_params(**kw)
    return f


WORKLOADS = {
    # name: (scene factory, params factory, description)
    "dam_break_1m": (sc.dam_break_1m, dam_break_params, "2D dam-break, 1024x1024 = 1 048 576 uniform-h particles, HybridDFSPH"),
    "dam_break_1m_adaptive": (sc.dam_break_1m_adaptive, dam_break_params, "2D dam-break, 1 000 960 particles, 4:1 radius ratio"),
    "dam_break_1m_adaptive_contact": (sc.dam_break_1m_adaptive_contact, dam_break_params,
                                      "configs[2]'s two blocks (1 000 960 particles, 4:1 radius ratio) one coarse spacing apart: the mixed-h interface from step 0"),
    "dam_break_1m_adaptive_colliding": (lambda: sc.dam_break_1m_adaptive_contact(1.5 * 0.00390625), dam_break_params,
                                        "configs[2]'s two blocks 1.5 coarse spacings apart: no mixed-h pair at rest, the collapsing fine column reaches the coarse block within ~13 steps"),
    "dam_break_8m": (sc.dam_break_8m, dam_break_params, "2D dam-break, 8192x1024 = 8 388 608 particles (configs[1]'s column eight times as wide)"),
    "dam_break_8m_spec": (sc.dam_break_8m_spec, lambda **kw: dam_break_params(**dict(dict(max_dt=0.00025), **kw)),
                          "SURVEY 8d config 4 as written: 2896x2896 = 8 386 816 particles at spacing 1/2048, box 4x2, max_dt 0.00025 (0.001 blows up at step 3, 0.0005 at step 6: profiles/r5_config3_divergence.md)"),
    "dam_break_2m": (lambda: sc.dam_break_weak(2), dam_break_params, "2D dam-break, 2048x1024 = 2 097 152 particles (configs[1]'s column twice as wide)"),
    "dam_break_4m": (lambda: sc.dam_break_weak(4), dam_break_params, "2D dam-break, 4096x1024 = 4 194 304 particles (configs[1]'s column four times as wide)"),
    # configs[4] without the host-side adaptivity: media/ratio-stress-test-video.yaml's IISPH recipe on the 4M-particle scene
    "ratio_stress_4m": (sc.ratio_stress_4m, lambda **kw: default_params(**dict(dict(
        merging=False, sharing=False, splitting=False, support_length_estimation="FromMass", pressure_solver_method="IISPH",
        cfl_factor=0.2, max_dt=0.001, iisph_max_avg_density_error=0.001, init_boundary_handler="AnalyticUnderestimate",
        level_estimation_method="None"), **kw)), "ratio-stress-test geometry, 4 004 343 particles at 50:1 radii, IISPH, Sdf2D box"),
    "ratio_stress_4m_settled": (sc.ratio_stress_4m_settled, lambda **kw: default_params(**dict(dict(
        merging=False, sharing=False, splitting=False, support_length_estimation="FromMass", pressure_solver_method="IISPH",
        cfl_factor=0.2, max_dt=0.001, iisph_max_avg_density_error=0.001, init_boundary_handler="AnalyticUnderestimate",
        level_estimation_method="None"), **kw)), "configs[4]'s blocks standing on the floor, in contact: 4 004 343 particles at 50:1 radii, IISPH under hydrostatic load"),
    "dam_break_64k": (lambda: sc.dam_break_small(256, 256, 1.0 / 256), dam_break_params, "2D dam-break, 256x256 particles (smoke)"),
}
