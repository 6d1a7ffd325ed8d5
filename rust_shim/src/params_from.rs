// Drop this into the reference crate (e.g. src/simulation/hip_params.rs, `mod hip_params;`): the by-value copy of
// SimulationParams (simulation_parameters.rs:26-108) that sph_step takes every step.  Enum numbering = include/sph_ffi.h.
use sph_hip::ffi::*;

use crate::simulation_parameters::*;

impl From<&SimulationParams> for SphParams {
    fn from(p: &SimulationParams) -> SphParams {
        SphParams {
            rest_density: p.rest_density as f32,
            cfl_factor: p.cfl_factor as f32,
            max_dt: p.max_dt as f32,
            viscosity: p.viscosity as f32,
            viscosity_type: match p.viscosity_type {
                ViscosityType::WCSPH => SPH_VISC_WCSPH,
                ViscosityType::ApproxLaplace => SPH_VISC_APPROX_LAPLACE,
                ViscosityType::XSPH => SPH_VISC_XSPH,
            },
            gravity: p.gravity as f32,
            jacobi_omega: p.jacobi_omega as f32,
            level_estimation_method: match p.level_estimation_method {
                LevelEstimationMethod::None => SPH_LEVEL_NONE,
                LevelEstimationMethod::CenterDiff => SPH_LEVEL_CENTER_DIFF,
                LevelEstimationMethod::EmptyAngle => SPH_LEVEL_EMPTY_ANGLE,
            },
            maximum_range: p.maximum_range as f32,
            support_length_estimation: match p.support_length_estimation {
                SupportLengthEstimation::FromDistribution => SPH_H_FROM_DISTRIBUTION,
                SupportLengthEstimation::FromDistributionClamped1 => SPH_H_FROM_DISTRIBUTION_CLAMPED1,
                SupportLengthEstimation::FromDistributionClamped2 => SPH_H_FROM_DISTRIBUTION_CLAMPED2,
                SupportLengthEstimation::FromDistribution2 => SPH_H_FROM_DISTRIBUTION2,
                SupportLengthEstimation::FromMass => SPH_H_FROM_MASS,
            },
            sdf_gradient_eps: p.sdf_gradient_eps as f32,
            has_pull_fluid_to: p.pull_fluid_to.is_some() as i32,
            pull_fluid_to: p.pull_fluid_to.map(|v| [v.x as f32, v.y as f32, v.z as f32]).unwrap_or([0.0; 3]),
            maximum_surface_distance: p.maximum_surface_distance as f32,
            boundary_is_fluid_surface: p.boundary_is_fluid_surface as i32,
            use_extended_range_for_level_estimation: p.use_extended_range_for_level_estimation as i32,
            level_estimation_after_advection: p.level_estimation_after_advection as i32,
            level_estimation_range: p.level_estimation_range as f32,
            pressure_solver_method: match p.pressure_solver_method {
                PressureSolverMethod::IISPH => SPH_SOLVER_IISPH,
                PressureSolverMethod::IISPH2 => SPH_SOLVER_IISPH2,
                PressureSolverMethod::HybridDFSPH => SPH_SOLVER_HYBRID_DFSPH,
                PressureSolverMethod::OnlyDivergence => SPH_SOLVER_ONLY_DIVERGENCE,
            },
            iisph_max_avg_density_error: p.iisph_max_avg_density_error as f32,
            hybrid_dfsph_factor: p.hybrid_dfsph_factor as f32,
            hybrid_dfsph_max_avg_density_error: p.hybrid_dfsph_max_avg_density_error as f32,
            hybrid_dfsph_max_avg_divergence_error: p.hybrid_dfsph_max_avg_divergence_error as f32,
            hybrid_dfsph_density_source_term: match p.hybrid_dfsph_density_source_term {
                HybridDfsphDensitySourceTerm::DensityAndDivergence => SPH_DENSITY_AND_DIVERGENCE,
                HybridDfsphDensitySourceTerm::OnlyDensity => SPH_ONLY_DENSITY,
            },
            hybrid_dfsph_non_pressure_accel_before_divergence_free: p.hybrid_dfsph_non_pressure_accel_before_divergence_free as i32,
            boundary_penalty_term: match p.boundary_penalty_term {
                BoundaryPenaltyTerm::None => SPH_PENALTY_NONE,
                BoundaryPenaltyTerm::Linear => SPH_PENALTY_LINEAR,
                BoundaryPenaltyTerm::Quadratic1 => SPH_PENALTY_QUADRATIC1,
                BoundaryPenaltyTerm::Quadratic2 => SPH_PENALTY_QUADRATIC2,
            },
            operator_discretization: match p.operator_discretization {
                OperatorDiscretization::ConsistentSimpleGradient => SPH_OP_SIMPLE_GRADIENT,
                OperatorDiscretization::ConsistentSymmetricGradient => SPH_OP_SYMMETRIC_GRADIENT,
                OperatorDiscretization::Winchenbach2020 => SPH_OP_WINCHENBACH2020,
            },
            max_iters: p.max_iters as u32,
            check_neighborhood: p.check_neighborhood as i32,
            check_aii: p.check_aii as i32,
            constrain_neighborhood_count: p.constrain_neighborhood_count as i32,
            fill_stash_with: match p.fill_stash_with {
                None => SPH_STASH_NONE,
                Some(FillStashWith::SurfaceDistanceFirstIteration) => SPH_STASH_SURFACE_DISTANCE_FIRST,
                Some(FillStashWith::SurfaceDistanceMiddle) => SPH_STASH_SURFACE_DISTANCE_MIDDLE,
            },
            sizing_function: match p.sizing_function {
                SizingFunction::Radius2 => SPH_SIZING_RADIUS2,
                SizingFunction::Radius => SPH_SIZING_RADIUS,
                SizingFunction::Mass => SPH_SIZING_MASS,
            },
            particle_radius_fine: p.particle_radius_fine as f32,
            particle_radius_base: p.particle_radius_base as f32,
        }
    }
}
