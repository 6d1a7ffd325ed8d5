//! Rust side of `include/sph_ffi.h`.
//!
//! `ffi` is the raw `extern "C"` surface (struct layouts are those of the header, field for field);
//! `HipStep` is the thin safe wrapper `FluidSimulation` holds instead of running its rayon sweeps:
//! `single_step_without_adaptivity` (simulation.rs:1980-2730) becomes `HipStep::step`, every `panic!`/`assert!`
//! on that path comes back as a non-zero status that `HipStep` turns into the same `panic!` (so the
//! `catch_unwind` of main_loop.rs:300-311 keeps working).  `params_from.rs` (next to this file) is the
//! `From<&SimulationParams>` a maintainer drops into the reference crate.
#![allow(non_camel_case_types)]

pub mod ffi {
    use std::os::raw::{c_char, c_int, c_void};

    // ---- enums of simulation_parameters.rs as the header numbers them -------------------------------------
    pub const SPH_MATH_FAST: i32 = 0;
    pub const SPH_MATH_EXACT: i32 = 1;
    pub const SPH_VISC_WCSPH: i32 = 0;
    pub const SPH_VISC_APPROX_LAPLACE: i32 = 1;
    pub const SPH_VISC_XSPH: i32 = 2;
    pub const SPH_LEVEL_NONE: i32 = 0;
    pub const SPH_LEVEL_CENTER_DIFF: i32 = 1;
    pub const SPH_LEVEL_EMPTY_ANGLE: i32 = 2;
    pub const SPH_H_FROM_DISTRIBUTION: i32 = 0;
    pub const SPH_H_FROM_DISTRIBUTION_CLAMPED1: i32 = 1;
    pub const SPH_H_FROM_DISTRIBUTION_CLAMPED2: i32 = 2;
    pub const SPH_H_FROM_DISTRIBUTION2: i32 = 3;
    pub const SPH_H_FROM_MASS: i32 = 4;
    pub const SPH_SOLVER_IISPH: i32 = 0;
    pub const SPH_SOLVER_IISPH2: i32 = 1;
    pub const SPH_SOLVER_HYBRID_DFSPH: i32 = 2;
    pub const SPH_SOLVER_ONLY_DIVERGENCE: i32 = 3;
    pub const SPH_DENSITY_AND_DIVERGENCE: i32 = 0;
    pub const SPH_ONLY_DENSITY: i32 = 1;
    pub const SPH_PENALTY_NONE: i32 = 0;
    pub const SPH_PENALTY_LINEAR: i32 = 1;
    pub const SPH_PENALTY_QUADRATIC1: i32 = 2;
    pub const SPH_PENALTY_QUADRATIC2: i32 = 3;
    pub const SPH_OP_SIMPLE_GRADIENT: i32 = 0;
    pub const SPH_OP_SYMMETRIC_GRADIENT: i32 = 1;
    pub const SPH_OP_WINCHENBACH2020: i32 = 2;
    pub const SPH_STASH_NONE: i32 = 0;
    pub const SPH_STASH_SURFACE_DISTANCE_FIRST: i32 = 1;
    pub const SPH_STASH_SURFACE_DISTANCE_MIDDLE: i32 = 2;
    pub const SPH_SIZING_RADIUS2: i32 = 0;
    pub const SPH_SIZING_RADIUS: i32 = 1;
    pub const SPH_SIZING_MASS: i32 = 2;

    /// `sph_params`: 37 four-byte fields, 148 bytes (tests/test_abi.py pins the same number on the Python side).
    #[repr(C)]
    #[derive(Clone, Copy, Debug, Default)]
    pub struct SphParams {
        pub rest_density: f32,
        pub cfl_factor: f32,
        pub max_dt: f32,
        pub viscosity: f32,
        pub viscosity_type: i32,
        pub gravity: f32,
        pub jacobi_omega: f32,
        pub level_estimation_method: i32,
        pub maximum_range: f32,
        pub support_length_estimation: i32,
        pub sdf_gradient_eps: f32,
        pub has_pull_fluid_to: i32,
        pub pull_fluid_to: [f32; 3],
        pub maximum_surface_distance: f32,
        pub boundary_is_fluid_surface: i32,
        pub use_extended_range_for_level_estimation: i32,
        pub level_estimation_after_advection: i32,
        pub level_estimation_range: f32,
        pub pressure_solver_method: i32,
        pub iisph_max_avg_density_error: f32,
        pub hybrid_dfsph_factor: f32,
        pub hybrid_dfsph_max_avg_density_error: f32,
        pub hybrid_dfsph_max_avg_divergence_error: f32,
        pub hybrid_dfsph_density_source_term: i32,
        pub hybrid_dfsph_non_pressure_accel_before_divergence_free: i32,
        pub boundary_penalty_term: i32,
        pub operator_discretization: i32,
        pub max_iters: u32,
        pub check_neighborhood: i32,
        pub check_aii: i32,
        pub constrain_neighborhood_count: i32,
        pub fill_stash_with: i32,
        pub sizing_function: i32,
        pub particle_radius_fine: f32,
        pub particle_radius_base: f32,
    }

    #[repr(C)]
    #[derive(Clone, Copy, Debug, Default)]
    pub struct SphPlane {
        pub dir_x: f32,
        pub dir_y: f32,
        pub delta: f32,
    }

    #[repr(C)]
    #[derive(Clone, Copy, Debug, Default)]
    pub struct SphSolverStats {
        pub iters: u32,
        pub converged: i32,
        pub normal_count: u32,
        pub singular_count: u32,
        pub negative_count: u32,
        pub avg_error: f32,
        pub max_error: f32,
    }

    #[repr(C)]
    #[derive(Clone, Copy, Debug, Default)]
    pub struct SphStepStats {
        pub dt: f32,
        pub time: f32,
        pub step_number: u64,
        pub n_particles: u64,
        pub div_solver: SphSolverStats,
        pub density_solver: SphSolverStats,
        pub ms_simulation_step: f64,
        pub ms_neighborhood: f64,
        pub ms_level_estimation: f64,
        pub ms_div_solver: f64,
        pub ms_density_solver: f64,
    }

    #[repr(C)]
    #[derive(Clone, Copy, Debug, Default)]
    pub struct SphGridInfo {
        pub cell_size: f32,
        pub cells_min_x: i32,
        pub cells_min_y: i32,
        pub size_x: i32,
        pub size_y: i32,
    }

    // ---- field ids (SPH_F_*) ---------------------------------------------------------------------------------
    pub const SPH_F_MASS: c_int = 0;
    pub const SPH_F_POSITION: c_int = 1;
    pub const SPH_F_VELOCITY: c_int = 2;
    pub const SPH_F_PRESSURE_ACCEL: c_int = 3;
    pub const SPH_F_DENSITY: c_int = 4;
    pub const SPH_F_PPE_SOURCE_TERM: c_int = 5;
    pub const SPH_F_PRESSURE: c_int = 6;
    pub const SPH_F_AII: c_int = 7;
    pub const SPH_F_DENSITY_ERROR: c_int = 8;
    pub const SPH_F_H2: c_int = 9;
    pub const SPH_F_H2_NEXT: c_int = 10;
    pub const SPH_F_CONSTANT_FIELD: c_int = 11;
    pub const SPH_F_NEIGHBOR_COUNT: c_int = 12;
    pub const SPH_F_LEVEL_ESTIMATION: c_int = 13;
    pub const SPH_F_LEVEL_OLD: c_int = 14;
    pub const SPH_F_STASH: c_int = 15;
    pub const SPH_F_FLAG_IS_FLUID_SURFACE: c_int = 16;
    pub const SPH_F_FLAG_INSUFFICIENT_NEIGHS: c_int = 17;
    pub const SPH_F_PARTICLE_SIZE_CLASS: c_int = 18;
    pub const SPH_F_FLAG_NEIGHBORHOOD_REDUCED: c_int = 23;

    pub const SPH_OK: c_int = 0;

    // ---- sparse edits between steps (sph_apply_edits) ------------------------------------------------------------
    pub const SPH_EDIT_SET: i32 = 0;
    pub const SPH_EDIT_SWAP: i32 = 1;
    pub const SPH_EDIT_TRUNCATE: i32 = 2;
    pub const SPH_EDIT_EXTEND: i32 = 3;
    pub const SPH_EDIT_F_MASS: u32 = 1;
    pub const SPH_EDIT_F_POSITION: u32 = 2;
    pub const SPH_EDIT_F_VELOCITY: u32 = 4;
    pub const SPH_EDIT_F_H2: u32 = 8;
    pub const SPH_EDIT_F_H2_NEXT: u32 = 16;
    pub const SPH_EDIT_F_LEVEL_ESTIMATION: u32 = 32;
    pub const SPH_EDIT_F_LEVEL_OLD: u32 = 64;

    #[repr(C)]
    #[derive(Clone, Copy, Debug, Default)]
    pub struct SphEditOp {
        pub kind: i32,
        pub a: u32,
        pub b: u32,
        pub fields: u32,
        pub mass: f32,
        pub position: [f32; 2],
        pub velocity: [f32; 2],
        pub h2: f32,
        pub h2_next: f32,
        pub level_estimation: f32,
        pub level_old: f32,
    }
    /// sph_adapt_params: the SimulationParams fields the apply half of single_step_adaptivity reads (+ the step's dt)
    #[repr(C)]
    #[derive(Clone, Copy, Default)]
    pub struct SphAdaptParams {
        pub dt: f32,
        pub max_mass_transfer_sharing: f32,
        pub minimum_share_partners: u32,
        pub minimum_merge_partners: u32,
        pub fail_on_missing_split_pattern: i32,
        pub max_share_distance: f32,
        pub max_merge_distance: f32,
        pub allow_share_with_optimal_particle: i32,
        pub allow_share_with_too_small_particle: i32,
        pub allow_merge_with_optimal_particle: i32,
        pub allow_merge_on_size_difference: i32,
    }


    extern "C" {
        pub fn sph_create(n_capacity: u64, device_id: c_int, planes: *const SphPlane, n_planes: c_int, out: *mut *mut c_void) -> c_int;
        pub fn sph_set_boundary_polygon(ctx: *mut c_void, points_xy: *const f32, n_points: c_int) -> c_int;
        pub fn sph_set_math_policy(ctx: *mut c_void, policy: c_int) -> c_int;
        pub fn sph_get_math_policy(ctx: *const c_void) -> c_int;
        pub fn sph_destroy(ctx: *mut c_void);
        pub fn sph_upload(ctx: *mut c_void, n: u64, mass: *const f32, position_xy: *const f32, velocity_xy: *const f32) -> c_int;
        pub fn sph_upload_field(ctx: *mut c_void, field: c_int, src: *const c_void, src_bytes: u64) -> c_int;
        pub fn sph_apply_edits(ctx: *mut c_void, ops: *const SphEditOp, n_ops: u64) -> c_int;
        pub fn sph_download(ctx: *mut c_void, field: c_int, dst: *mut c_void, dst_bytes: u64) -> c_int;
        pub fn sph_download_neighbors(ctx: *mut c_void, offsets: *mut u32, indices: *mut u32, indices_capacity: u64, n_indices: *mut u64) -> c_int;
        pub fn sph_num_particles(ctx: *const c_void) -> u64;
        pub fn sph_time(ctx: *const c_void) -> f32;
        pub fn sph_set_time(ctx: *mut c_void, time: f32, step_number: u64) -> c_int;
        pub fn sph_step(ctx: *mut c_void, params: *const SphParams, out: *mut SphStepStats) -> c_int;
        pub fn sph_classify(ctx: *mut c_void, params: *const SphParams) -> c_int;
        pub fn sph_share_particles(ctx: *mut c_void, params: *const SphParams, ap: *const SphAdaptParams, merge_partner: *const u32,
                                   merge_counter: *const u16) -> c_int;
        pub fn sph_merge_particles(ctx: *mut c_void, params: *const SphParams, ap: *const SphAdaptParams, merge_partner: *const u32,
                                   merge_counter: *const u16) -> c_int;
        pub fn sph_set_split_patterns(ctx: *mut c_void, n_patterns: u32, pos_s_xy: *const f32) -> c_int;
        pub fn sph_split_particles(ctx: *mut c_void, params: *const SphParams, ap: *const SphAdaptParams) -> c_int;
        pub fn sph_last_error(ctx: *const c_void) -> *const c_char;
        pub fn sph_grid(ctx: *const c_void, out: *mut SphGridInfo) -> c_int;
        /// the sequential partner searches as compiled host code (no context; the reference keeps its own Rust loops)
        pub fn sph_host_find_partners(kind: c_int, n: u64, particle_size_class: *const u8, mass: *const f32, level_estimation: *const f32,
                                      position_xy: *const f32, h2: *const f32, offsets: *const u32, indices: *const u32, params: *const SphParams,
                                      ap: *const SphAdaptParams, merge_partner: *mut u32, merge_counter: *mut u16, n_transfers: *mut u64) -> c_int;
        // ---- multi-GPU: one process (= one context) per GPU, 1-D slabs along x (sph_ffi.h, "multi-GPU") --------------
        pub fn sph_dist_configure(ctx: *mut c_void, rank: c_int, n_ranks: c_int, cut_lo: f32, cut_hi: f32) -> c_int;
        pub fn sph_dist_set_rebalance(ctx: *mut c_void, every_n_steps: c_int) -> c_int;
        pub fn sph_dist_get_cuts(ctx: *mut c_void, cut_lo: *mut f32, cut_hi: *mut f32, n_rebalances: *mut u32) -> c_int;
        pub fn sph_dist_get_stats(ctx: *mut c_void, out: *mut SphDistStats, reset: c_int) -> c_int;
        pub fn sph_comm_unique_id(id_out: *mut u8 /* [128] */) -> c_int;
        pub fn sph_comm_init(ctx: *mut c_void, id: *const u8 /* [128] */, rank: c_int, n_ranks: c_int) -> c_int;
        /// k contexts of ONE process as ranks 0..k-1 (plain copies as transport): the single-GPU check of the decomposition
        pub fn sph_group_step(ctxs: *mut *mut c_void, n: c_int, params: *const SphParams, outs: *mut SphStepStats) -> c_int;
        /// share (op 0) / merge (1) / split (2) for those k contexts in their slab form: `merge_partner` / `merge_counter` of the WHOLE
        /// vector by global particle id, null for op 2.  (One process per rank: the three apply calls themselves, collectively.)
        pub fn sph_group_adapt(ctxs: *mut *mut c_void, n: c_int, op: c_int, params: *const SphParams, ap: *const SphAdaptParams,
                               merge_partner: *const u32, merge_counter: *const u16) -> c_int;
        /// ranks as processes of one node without RCCL: rendezvous and staging through a POSIX shared-memory segment
        pub fn sph_comm_init_shm(ctx: *mut c_void, name: *const c_char, rank: c_int, n_ranks: c_int, bytes_per_side: u64, create: c_int) -> c_int;
        /// ... and on top of it the peer-mapped push transport: export this rank's inbox (64-byte hipIpc handle), all-gather the handles
        /// with the launcher, map the others'
        pub fn sph_comm_ipc_export(ctx: *mut c_void, bytes_per_side: u64, handle_out: *mut u8 /* [64] */) -> c_int;
        pub fn sph_comm_init_ipc(ctx: *mut c_void, handles: *const u8 /* n_ranks x 64 */, n_ranks: c_int) -> c_int;
    }

    /// `sph_dist_stats`
    #[repr(C)]
    #[derive(Clone, Copy, Debug, Default)]
    pub struct SphDistStats {
        pub steps: u64,
        pub exchanges: u64,
        pub bytes_sent: u64,
        pub bytes_received: u64,
        pub allreduces: u64,
        pub host_waits: u64,
        pub n_owned: u64,
        pub n_halo: [u32; 2],
        pub n_ghost: [u32; 2],
        pub transport: u32,
        pub comm_ranks: u32,
    }
}

use std::ffi::CStr;
use std::os::raw::c_void;

/// Owner of one `sph_ctx`.  Not `Sync`: the reference calls the step from exactly one thread (main_loop.rs:152-181).
pub struct HipStep {
    ctx: *mut c_void,
    /// host arrays changed since the last upload (single_step_adaptivity ran, or the GUI restarted the scene)
    pub dirty: bool,
}

/// The boundary handler init_fluid_sim builds (simulation.rs:3135-3213).
pub enum Boundary<'a> {
    /// AnalyticOverestimate: `SdfPlane::new_boundary_box` -> (dir.x, dir.y, delta) per plane
    Planes(&'a [ffi::SphPlane]),
    /// AnalyticUnderestimate: the points of `Sdf2D::new_boundary_box`, x0 y0 x1 y1 ...
    Polygon(&'a [f32]),
}

impl HipStep {
    pub fn new(capacity: usize, device_id: i32, boundary: Boundary) -> HipStep {
        let mut ctx: *mut c_void = std::ptr::null_mut();
        let (planes, n_planes): (*const ffi::SphPlane, i32) = match boundary {
            Boundary::Planes(p) => (p.as_ptr(), p.len() as i32),
            Boundary::Polygon(_) => (std::ptr::null(), 0),
        };
        let rc = unsafe { ffi::sph_create(capacity as u64, device_id, planes, n_planes, &mut ctx) };
        if rc != ffi::SPH_OK {
            panic!("sph_create failed with status {}", rc);
        }
        let s = HipStep { ctx, dirty: true };
        if let Boundary::Polygon(pts) = boundary {
            s.check(unsafe { ffi::sph_set_boundary_polygon(s.ctx, pts.as_ptr(), (pts.len() / 2) as i32) });
        }
        s
    }

    /// `panic!` with the message of the reference assertion that fired (status != 0).
    fn check(&self, status: i32) {
        if status != ffi::SPH_OK {
            let msg = unsafe { CStr::from_ptr(ffi::sph_last_error(self.ctx)) }.to_string_lossy().into_owned();
            panic!("{}", msg);
        }
    }

    /// FluidSimulation::new arguments / whatever adaptivity left in ParticleVec.  `position` and `velocity` are the
    /// `Vec<VF<2>>` storage viewed as 2n floats (nalgebra SVector<f32,2> is 8 contiguous bytes, x then y).
    pub fn upload(&mut self, mass: &[f32], position_xy: &[f32], velocity_xy: &[f32]) {
        assert!(position_xy.len() == 2 * mass.len() && velocity_xy.len() == 2 * mass.len());
        self.check(unsafe { ffi::sph_upload(self.ctx, mass.len() as u64, mass.as_ptr(), position_xy.as_ptr(), velocity_xy.as_ptr()) });
        self.dirty = false;
    }

    pub fn upload_field<T: Copy>(&mut self, field: i32, values: &[T]) {
        self.check(unsafe {
            ffi::sph_upload_field(self.ctx, field, values.as_ptr() as *const c_void, (values.len() * std::mem::size_of::<T>()) as u64)
        });
    }

    /// The edit log of one single_step_adaptivity (element writes + ParticleVec::swap / truncate / extend, in call order).
    /// Cheaper than `upload` when few particles changed: one small host->device copy and one gather on the device.
    pub fn apply_edits(&mut self, log: &[ffi::SphEditOp]) {
        self.check(unsafe { ffi::sph_apply_edits(self.ctx, log.as_ptr(), log.len() as u64) });
        self.dirty = false;
    }

    pub fn download<T: Copy>(&self, field: i32, out: &mut [T]) {
        self.check(unsafe { ffi::sph_download(self.ctx, field, out.as_mut_ptr() as *mut c_void, (out.len() * std::mem::size_of::<T>()) as u64) });
    }

    /// NeighborhoodCache of the positions the last step started from, as CSR.
    pub fn download_neighbors(&self) -> (Vec<u32>, Vec<u32>) {
        let n = unsafe { ffi::sph_num_particles(self.ctx) } as usize;
        let mut offsets = vec![0u32; n + 1];
        let mut total: u64 = 0;
        self.check(unsafe { ffi::sph_download_neighbors(self.ctx, offsets.as_mut_ptr(), std::ptr::null_mut(), 0, &mut total) });
        let mut indices = vec![0u32; total as usize];
        self.check(unsafe { ffi::sph_download_neighbors(self.ctx, offsets.as_mut_ptr(), indices.as_mut_ptr(), total, &mut total) });
        (offsets, indices)
    }

    /// single_step_without_adaptivity: returns dt and the statistics that feed vcounters / pcounters.
    pub fn step(&mut self, params: &ffi::SphParams) -> ffi::SphStepStats {
        assert!(!self.dirty, "host arrays changed: call upload() first");
        let mut st = ffi::SphStepStats::default();
        self.check(unsafe { ffi::sph_step(self.ctx, params, &mut st) });
        st
    }

    /// classify_particles (adaptivity/mod.rs:50-59) on the device-resident state; read the classes back with
    /// `download(SPH_F_PARTICLE_SIZE_CLASS, ..)` for the partner searches.
    pub fn classify(&mut self, params: &ffi::SphParams) {
        self.check(unsafe { ffi::sph_classify(self.ctx, params) });
    }

    /// share_particles (particle_sharing.rs:152-240) with the merge_partner / merge_counter arrays find_share_partner_sequential
    /// filled (AtomicU32 has the layout of u32): the particle data never leaves the device.
    pub fn share_particles(&mut self, params: &ffi::SphParams, ap: &ffi::SphAdaptParams, merge_partner: &[u32], merge_counter: &[u16]) {
        assert!(merge_partner.len() == self.num_particles() && merge_counter.len() == merge_partner.len());
        self.check(unsafe { ffi::sph_share_particles(self.ctx, params, ap, merge_partner.as_ptr(), merge_counter.as_ptr()) });
    }

    /// merge_particles (particle_merging.rs:270-370): transfer + the swap-with-the-last deletion; `num_particles()` is the new length.
    pub fn merge_particles(&mut self, params: &ffi::SphParams, ap: &ffi::SphAdaptParams, merge_partner: &[u32], merge_counter: &[u16]) {
        assert!(merge_partner.len() == self.num_particles() && merge_counter.len() == merge_partner.len());
        self.check(unsafe { ffi::sph_merge_particles(self.ctx, params, ap, merge_partner.as_ptr(), merge_counter.as_ptr()) });
    }

    /// SplitPatterns (splitting.rs:84-120): `pos_s_xy` = the pos_s of every pattern, concatenated (pattern k has k + 2 children).
    pub fn set_split_patterns(&mut self, n_patterns: usize, pos_s_xy: &[f32]) {
        self.check(unsafe { ffi::sph_set_split_patterns(self.ctx, n_patterns as u32, pos_s_xy.as_ptr()) });
    }

    /// split_particles (splitting.rs:19-82) for the TooLarge particles of the last `classify`.
    pub fn split_particles(&mut self, params: &ffi::SphParams, ap: &ffi::SphAdaptParams) {
        self.check(unsafe { ffi::sph_split_particles(self.ctx, params, ap) });
    }

    pub fn time(&self) -> f32 {
        unsafe { ffi::sph_time(self.ctx) }
    }

    /// `SPH_MATH_EXACT`: the reference's IEEE operations in the reference's order (bit for bit the Rust sweeps when the particles
    /// are in the device's visiting order); `SPH_MATH_FAST` (default): hardware rsq / rcp, fma -- within 1e-4 relative.
    pub fn set_math_policy(&mut self, policy: c_int) {
        self.check(unsafe { ffi::sph_set_math_policy(self.ctx, policy) });
    }

    pub fn set_time(&mut self, time: f32, step_number: u64) {
        self.check(unsafe { ffi::sph_set_time(self.ctx, time, step_number) });
    }

    pub fn num_particles(&self) -> usize {
        unsafe { ffi::sph_num_particles(self.ctx) as usize }
    }
}

impl Drop for HipStep {
    fn drop(&mut self) {
        unsafe { ffi::sph_destroy(self.ctx) }
    }
}

#[cfg(test)]
mod layout {
    use super::ffi::*;
    #[test]
    fn struct_sizes_match_the_header() {
        assert_eq!(std::mem::size_of::<SphParams>(), 37 * 4);
        assert_eq!(std::mem::size_of::<SphPlane>(), 12);
        assert_eq!(std::mem::size_of::<SphSolverStats>(), 28);
        assert_eq!(std::mem::size_of::<SphStepStats>(), 8 + 8 + 8 + 28 + 28 + 5 * 8);
        assert_eq!(std::mem::size_of::<SphGridInfo>(), 20);
    }
}
