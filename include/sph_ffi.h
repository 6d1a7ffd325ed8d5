/*
 * sph_ffi.h -- C-ABI boundary of the MI355X-native SPH particle loop.
 *
 * This is the drop-in seam for ONE method of the reference (kaegi/adaptive-sph):
 *
 *     FluidSimulation::<DimensionUtils2d,2>::single_step_without_adaptivity(&mut self, SimulationParams) -> FT
 *         src/simulation/simulation.rs:1980-2730
 *
 * plus the state it reads/mutates (`FluidSimulation.particles`, `.neighs`, `.boundary_handler`,
 * `.time`; simulation.rs:471-477).  The reference has no FFI of its own (it is one Rust crate), so
 * every entry point below cites the Rust item it replaces.  INTEGRATION.md shows the Rust
 * `extern "C"` block and the ~40-line patch a maintainer would add.
 *
 * Conventions
 *   - plain pointers and sizes only; the library never retains a host pointer after the call returns;
 *   - `Vec<VF<2>>` (nalgebra SVector<f32,2>, 8 contiguous bytes x,y) is passed as `const float*` of
 *     length 2*n (`as_ptr() as *const f32`);
 *   - every function returns an `int` status (0 = SPH_OK); the reference signals the same
 *     conditions by panic!/assert! (caught by catch_unwind at platform/desktop/main_loop.rs:300-311);
 *     unwinding across `extern "C"` is UB, so the Rust shim turns non-zero into `panic!`;
 *   - one host thread per context at a time (the reference's `fluid` thread, main_loop.rs:152-181).
 *
 * Two libraries implement this header with different symbol prefixes:
 *   libsph_hip.so      sph_*      the product: hand-written HIP kernels for gfx950 (adaptive_sph_amd/csrc)
 *   liboracle.so       oracle_*   TEST INFRASTRUCTURE ONLY: scalar CPU restatement (oracle/)
 */
#ifndef SPH_FFI_H
#define SPH_FFI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- enums: 1:1 with src/simulation/simulation_parameters.rs ------------------------------- */
enum { SPH_VISC_WCSPH = 0, SPH_VISC_APPROX_LAPLACE = 1, SPH_VISC_XSPH = 2 };          /* :148-153 */
enum { SPH_LEVEL_NONE = 0, SPH_LEVEL_CENTER_DIFF = 1, SPH_LEVEL_EMPTY_ANGLE = 2 };    /* :183-194 */
enum {                                                                                 /* :170-181 */
    SPH_H_FROM_DISTRIBUTION = 0, SPH_H_FROM_DISTRIBUTION_CLAMPED1 = 1, SPH_H_FROM_DISTRIBUTION_CLAMPED2 = 2,
    SPH_H_FROM_DISTRIBUTION2 = 3, SPH_H_FROM_MASS = 4
};
enum { SPH_SOLVER_IISPH = 0, SPH_SOLVER_IISPH2 = 1, SPH_SOLVER_HYBRID_DFSPH = 2, SPH_SOLVER_ONLY_DIVERGENCE = 3 }; /* :196-206 */
enum { SPH_DENSITY_AND_DIVERGENCE = 0, SPH_ONLY_DENSITY = 1 };                         /* :208-212 */
enum { SPH_PENALTY_NONE = 0, SPH_PENALTY_LINEAR = 1, SPH_PENALTY_QUADRATIC1 = 2, SPH_PENALTY_QUADRATIC2 = 3 }; /* :17-23 */
enum { SPH_OP_SIMPLE_GRADIENT = 0, SPH_OP_SYMMETRIC_GRADIENT = 1, SPH_OP_WINCHENBACH2020 = 2 };               /* :110-122 */
enum { SPH_STASH_NONE = 0, SPH_STASH_SURFACE_DISTANCE_FIRST = 1, SPH_STASH_SURFACE_DISTANCE_MIDDLE = 2 };     /* :4-8 */
enum { SPH_SIZING_RADIUS2 = 0, SPH_SIZING_RADIUS = 1, SPH_SIZING_MASS = 2 };            /* :10-15 */

/* ---- per-step parameters: POD mirror of every SimulationParams field the path reads --------
 * (simulation_parameters.rs:26-108).  Passed BY VALUE EVERY STEP because the GUI thread may edit
 * any of them between steps (main_loop.rs:280).  Fields the path never reads (use_iisph,
 * eos_*, merge/share knobs, ...) stay on the Rust side. */
typedef struct sph_params {
    float    rest_density;
    float    cfl_factor;
    float    max_dt;
    float    viscosity;
    int32_t  viscosity_type;
    float    gravity;
    float    jacobi_omega;
    int32_t  level_estimation_method;
    float    maximum_range;
    int32_t  support_length_estimation;
    float    sdf_gradient_eps;
    int32_t  has_pull_fluid_to;          /* Option<VF<3>>: 0 = None */
    float    pull_fluid_to[3];
    float    maximum_surface_distance;
    int32_t  boundary_is_fluid_surface;
    int32_t  use_extended_range_for_level_estimation;
    int32_t  level_estimation_after_advection;
    float    level_estimation_range;
    int32_t  pressure_solver_method;
    float    iisph_max_avg_density_error;
    float    hybrid_dfsph_factor;
    float    hybrid_dfsph_max_avg_density_error;
    float    hybrid_dfsph_max_avg_divergence_error;
    int32_t  hybrid_dfsph_density_source_term;
    int32_t  hybrid_dfsph_non_pressure_accel_before_divergence_free;
    int32_t  boundary_penalty_term;
    int32_t  operator_discretization;
    uint32_t max_iters;
    int32_t  check_neighborhood;
    int32_t  check_aii;
    int32_t  constrain_neighborhood_count;
    int32_t  fill_stash_with;
    /* read by LevelEstimationState::target_mass / classify_particle (simulation.rs:213-237,
     * adaptivity/mod.rs:32-59) -- the first host-side consumer of the step's output */
    int32_t  sizing_function;
    float    particle_radius_fine;
    float    particle_radius_base;
} sph_params;

/* ---- boundary description: SdfPlane{dir,delta} (sdf/sdf_plane.rs:4-7); a box is the 4 inward
 * planes of SdfPlane::new_boundary_box (sdf_plane.rs:13-20).  n_planes = 0 is NoBoundaryHandler,
 * which cannot step in the reference either (compute_aii hits unimplemented!(),
 * boundary_handler/mod.rs:35-46) -> sph_step returns SPH_ERR_NO_BOUNDARY. */
typedef struct sph_plane {
    float dir_x, dir_y, delta;
} sph_plane;

/* ---- pressure-solver statistics of the LAST Jacobi iteration (simulation.rs:397-469) -------- */
typedef struct sph_solver_stats {
    uint32_t iters;              /* value returned by iisph_pressure_iterations: index of last iteration */
    int32_t  converged;
    uint32_t normal_count;
    uint32_t singular_count;
    uint32_t negative_count;
    float    avg_error;          /* avg_error_times_normal_particle_count / normal_particle_count */
    float    max_error;
} sph_solver_stats;

/* ---- what one step reports back; feeds vcounters "dt", "div-iterations",
 * "density-iterations", "particle-count" (simulation.rs:1990-1991, 2202, 2542-2544, 2617-2619)
 * and pcounters (ids as in the reference: simulation.rs:1993, 2023-2058, 2517-2545, 2578-2620) */
typedef struct sph_step_stats {
    float    dt;                 /* return value of single_step_without_adaptivity (sim.rs:2729) */
    float    time;               /* FluidSimulation.time after the step (sim.rs:2724) */
    uint64_t step_number;
    uint64_t n_particles;
    sph_solver_stats div_solver;      /* HybridDFSPH / OnlyDivergence */
    sph_solver_stats density_solver;  /* IISPH / HybridDFSPH */
    double   ms_simulation_step;
    double   ms_neighborhood;
    double   ms_level_estimation;
    double   ms_div_solver;
    double   ms_density_solver;
} sph_step_stats;

/* ---- field ids for sph_upload_field / sph_download (ParticleVec, simulation.rs:284-334) ----- */
enum {
    SPH_F_MASS = 0,            /* f32[n]   */
    SPH_F_POSITION = 1,        /* f32[2n]  */
    SPH_F_VELOCITY = 2,        /* f32[2n]  */
    SPH_F_PRESSURE_ACCEL = 3,  /* f32[2n]  */
    SPH_F_DENSITY = 4,         /* f32[n]   */
    SPH_F_PPE_SOURCE_TERM = 5, /* f32[n]   */
    SPH_F_PRESSURE = 6,        /* f32[n]   */
    SPH_F_AII = 7,             /* f32[n]   */
    SPH_F_DENSITY_ERROR = 8,   /* f32[n]   */
    SPH_F_H2 = 9,              /* f32[n]   */
    SPH_F_H2_NEXT = 10,        /* f32[n]   */
    SPH_F_CONSTANT_FIELD = 11, /* f32[n]   */
    SPH_F_NEIGHBOR_COUNT = 12, /* u32[n]   (usize in the reference) */
    SPH_F_LEVEL_ESTIMATION = 13, /* f32[n]: FluidSurface(x) -> x, FluidInterior -> NaN (sim.rs:197-201) */
    SPH_F_LEVEL_OLD = 14,      /* f32[n]   */
    SPH_F_STASH = 15,          /* f32[n]   */
    SPH_F_FLAG_IS_FLUID_SURFACE = 16,     /* u8[n] */
    SPH_F_FLAG_INSUFFICIENT_NEIGHS = 17,  /* u8[n] */
    SPH_F_PARTICLE_SIZE_CLASS = 18,       /* u8[n]: adaptivity/mod.rs:12-23 order; default Optimal (simulation.rs:318).  Written only by
                                           * sph_classify (the step never classifies, like single_step_without_adaptivity) or uploaded */
    /* BoundaryWinchenbach2020.lambda folded per particle (boundary_winchenbach2020.rs:27):
     * sum of lambda and sum of grad-lambda over the planes -- all downstream uses are linear */
    SPH_F_LAMBDA_SUM = 19,     /* f32[n]   */
    SPH_F_LAMBDA_GRAD_SUM = 20,/* f32[2n]  */
    /* spatial-hash cell of each particle in the reference's CellGrid convention
     * (neighborhood_search.rs:253-255, 273-274, 383-395): linear index, x fastest */
    SPH_F_CELL_INDEX = 21,     /* u32[n]   */
    /* slab (multi-GPU) contexts: particles migrate between ranks, so fields come back in device order
     * and this field carries their identity (uploadable right after sph_upload; default 0..n-1) */
    SPH_F_PARTICLE_ID = 22,    /* u32[n]   */
    SPH_F_FLAG_NEIGHBORHOOD_REDUCED = 23, /* u8[n]: constrain_neighborhood_count, sim.rs:2145-2171 */
    SPH_F_COUNT_ = 24
};

/* ---- status codes: one per reference guard ------------------------------------------------- */
enum {
    SPH_OK = 0,
    SPH_ERR_INVALID_ARGUMENT = 1,
    SPH_ERR_DEVICE = 2,                 /* HIP / RCCL runtime failure */
    SPH_ERR_CAPACITY = 3,               /* n > n_capacity; on slabs: owned + arrivals + ghosts do not fit (every rank's step fails together) */
    SPH_ERR_NO_BOUNDARY = 4,            /* boundary_handler/mod.rs:35-46 unimplemented!() */
    SPH_ERR_DENSITY_NOT_FINITE = 10,    /* sim.rs:1046 */
    SPH_ERR_DENSITY_TOO_SMALL = 11,     /* sim.rs:1047  density > 0.0001 */
    SPH_ERR_AII_NOT_FINITE = 12,        /* sim.rs:1106 */
    SPH_ERR_AII_NEGATIVE = 13,          /* sim.rs:1391-1400 */
    SPH_ERR_AP_NOT_FINITE = 14,         /* sim.rs:1269-1271 */
    SPH_ERR_PRESSURE_NOT_FINITE = 15,   /* sim.rs:1279-1281 */
    SPH_ERR_TOO_MANY_NEIGHBORS = 16,    /* neighborhood_search.rs:149-151 (20000) */
    SPH_ERR_VELOCITY_NOT_FINITE = 17,   /* sim.rs:2443, 2558 */
    SPH_ERR_POSITION_NOT_FINITE = 18,   /* sim.rs:2667 */
    SPH_ERR_VISCOSITY_NOT_FINITE = 19,  /* sim.rs:987, 995 */
    SPH_ERR_XSPH_TODO = 20,             /* sim.rs:2673-2676 todo!() */
    SPH_ERR_CHECK_NEIGHBORHOOD = 21,    /* sim.rs:1810-1863 */
    SPH_ERR_CHECK_AII = 22,             /* sim.rs:1347-1375 */
    SPH_ERR_LEVEL_WEIGHT = 23,          /* sim.rs:843-845 */
    SPH_ERR_VOLUME_ESTIMATE = 24,       /* sim.rs:1903-1909, 1961  volume_estimate >= 0 */
    SPH_ERR_CONSTRAIN_NOT_SMALLER = 25, /* sim.rs:2163  *p_h_next < smoothing_length_single(h2, i) */
    SPH_ERR_CONSTRAIN_NEGATIVE = 26,    /* sim.rs:2165  *p_h_next >= 0 */
    SPH_ERR_NO_SPLIT_PATTERN = 27,      /* splitting.rs:35, 41, 108: no split pattern for a 1-to-n split / num_children > 1 */
    SPH_ERR_UNSUPPORTED = 30,           /* a SimulationParams combination this build does not cover */
    SPH_ERR_POISONED = 31               /* an earlier step failed inside the step: state undefined until sph_upload */
};

typedef struct sph_ctx sph_ctx;

/* ---- lifecycle ------------------------------------------------------------------------------
 * sph_create  <->  FluidSimulation::new (sim.rs:487-533) + BoundaryWinchenbach2020::new
 *                  (boundary_winchenbach2020.rs:33-46: builds the two 10001-entry lambda LUTs).
 * `device_id` is the HIP device ordinal this context owns (one process per GPU).
 * "Restart" in the GUI = destroy + create (main_loop.rs:269-278). */
int  sph_create(uint64_t n_capacity, int device_id, const sph_plane* planes, int n_planes, sph_ctx** out);

/* Replace the boundary by ONE Sdf2D connected component (sdf/sdf2d.rs:36-210): the closed polygon p0 .. p(n-1), air on the
 * left-hand side of every edge p_i -> p_(i+1).  init_fluid_sim's AnalyticUnderestimate handler is the box polygon
 * (min.x,min.y), (max.x,min.y), (max.x,max.y), (min.x,max.y) (simulation.rs:3195-3206, sdf2d.rs:167-179): one SDF whose
 * value is the distance to the NEAREST wall, where AnalyticOverestimate sums four planes.  Fails with
 * SPH_ERR_INVALID_ARGUMENT on the conditions Sdf2DConnectedComponents::from_points asserts (sdf2d.rs:43, 59). */
#define SPH_MAX_POLYGON_POINTS 16
int  sph_set_boundary_polygon(sph_ctx* ctx, const float* points_xy, int n_points);

/* Which arithmetic the sweeps evaluate the reference's formulas in (no counterpart in the reference: it has ONE arithmetic, f32 IEEE
 * operations in the order of the Rust source -- sph_kernels.rs:23-71, simulation.rs:1007-1322):
 *   SPH_MATH_FAST   (default) v_rsq / v_rcp, fma, the truncated-power form of the cubic spline, m_j -> m_i in uniform scenes, boundary
 *                   entries folded per particle: within 1e-4 relative of the reference after N steps (tests/test_gpu_parity.py);
 *   SPH_MATH_EXACT  IEEE division / sqrt, no fma, every operation in the reference's order, per-SDF boundary entries kept apart: the
 *                   reference's results BIT FOR BIT when the particles are uploaded in the device's visiting order
 *                   (tests/test_gpu_bitexact.py), at the price bench.py prints under other_configs ("EXACT policy").
 * Allowed between steps at any time: the particle state stays, the lists and per-step outputs of the previous step are dropped
 * (the next step rebuilds them).  The environment variable SPH_HIP_EXACT=1 only sets the initial value at sph_create. */
enum sph_math_policy { SPH_MATH_FAST = 0, SPH_MATH_EXACT = 1 };
int  sph_set_math_policy(sph_ctx* ctx, int policy);
int  sph_get_math_policy(const sph_ctx* ctx);
void sph_destroy(sph_ctx* ctx);

/* Replace the whole particle set (FluidSimulation::new arguments, and what the host must do
 * whenever single_step_adaptivity (sim.rs:2732-2796) changed N or permuted indices).
 * h2_next is initialised from mass as in sim.rs:505-520; all other fields zero/default;
 * time and step_number are NOT reset (use sph_set_time). */
int  sph_upload(sph_ctx* ctx, uint64_t n, const float* mass, const float* position_xy, const float* velocity_xy);

/* Overwrite one host-written field (mass, position, velocity, h2, h2_next, level_estimation,
 * level_old -- the fields adaptivity writes: splitting.rs:60-79, particle_merging.rs:319-368,
 * particle_sharing.rs:202-237) without changing N. */
int  sph_upload_field(sph_ctx* ctx, int field, const void* src, uint64_t src_bytes);

/* ---- sparse edits between two steps -------------------------------------------------------------------------------
 * What particle sharing / merging / splitting do to ParticleVec (adaptivity/particle_sharing.rs:202-237,
 * particle_merging.rs:319-370, splitting.rs:52-79), replayed on the device-resident state instead of re-uploading all of it:
 * the Rust side records the element writes and the ParticleVec::swap / truncate / extend calls (simulation.rs:248-271) it
 * makes -- the same calls reach neighs and boundary_handler -- and hands the script over once.  The ops apply in order, in
 * HOST index space, with exactly the Vec semantics:
 *   SET       particle a gets the values whose bit is set in `fields` (the fields adaptivity writes)
 *   SWAP      particles a and b trade places (every field, the boundary handler's lambda entries included)
 *   TRUNCATE  the vector keeps its first a particles
 *   EXTEND    a particles with ParticleVec's defaults are appended (mass 0, zero vectors, h2 = h2_next = 0,
 *             LevelEstimationState::FluidInterior, level_old 0, no boundary terms)
 * Afterwards sph_num_particles() is the new length and downloads speak the new indices.  Per-step outputs (density,
 * pressure, neighbour lists, ...) are those of the last step and no longer line up with the edited vector.
 * Slab context: the index space is the rank's OWNED particles in the order sph_download returns them (the ghosts are dropped
 * by the call; the next step selects new ones); a surviving particle keeps its global id (SPH_F_PARTICLE_ID), an appended one
 * has the id 0xffffffff until the host uploads the id field -- unique ids are the host's business, as are particles whose new
 * position lies in another rank's slab (the next step hands them over). */
enum { SPH_EDIT_SET = 0, SPH_EDIT_SWAP = 1, SPH_EDIT_TRUNCATE = 2, SPH_EDIT_EXTEND = 3 };
enum {
    SPH_EDIT_F_MASS = 1, SPH_EDIT_F_POSITION = 2, SPH_EDIT_F_VELOCITY = 4, SPH_EDIT_F_H2 = 8, SPH_EDIT_F_H2_NEXT = 16,
    SPH_EDIT_F_LEVEL_ESTIMATION = 32, SPH_EDIT_F_LEVEL_OLD = 64
};
typedef struct sph_edit_op {
    int32_t  kind;
    uint32_t a, b;
    uint32_t fields;
    float    mass;
    float    position[2];
    float    velocity[2];
    float    h2, h2_next;
    float    level_estimation;   /* NaN = FluidInterior */
    float    level_old;
} sph_edit_op;
int  sph_apply_edits(sph_ctx* ctx, const sph_edit_op* ops, uint64_t n_ops);

/* ---- adaptivity data path (single_step_adaptivity, simulation.rs:2732-2796) ------------------------------------------------
 * The DECISIONS stay on the host, where the reference takes them sequentially (find_share_partner_sequential,
 * particle_sharing.rs:14-117; find_merge_partner_sequential, particle_merging.rs:16-125): the host reads the fields and the
 * neighbour lists it needs (sph_download, sph_download_neighbors), fills merge_partner / merge_counter exactly as the reference
 * does and hands the two arrays over.  The DATA never leaves the device: the gather-form mass / momentum transfers, the
 * swap-to-end deletion order and the appended split children are computed there, with the Vec semantics of the reference
 * (host indices after the call are the reference's indices).  Afterwards per-step outputs and neighbour lists belong to the old
 * vector, as after sph_apply_edits.
 * SLAB CONTEXTS (one process per GPU, or the k contexts of sph_group_step through sph_group_adapt): the same three calls, collective --
 * every rank calls them in the same order, right behind a step.  The particles stay on their ranks.  merge_partner / merge_counter
 * are then the arrays of the WHOLE vector, indexed by global particle id (SPH_F_PARTICLE_ID = the reference's Vec index), the same on
 * every rank: the decisions are taken in one place, as the reference takes them.  A receiver reads its donor from the step's ghost
 * layer (a donor is a neighbour of its receiver; the ghosts' records are refreshed from their owners first), merge_particles' new
 * indices are derived from the two arrays by every rank alike, split_particles' child counts are all-reduced over the ids; afterwards
 * a rank holds owned particles only, their ids = the reference's indices after the call (SPH_ERR_INVALID_ARGUMENT if a donor is not
 * among the receiver's owned particles and ghosts, or if no step came before). */
typedef struct sph_adapt_params {
    float    dt;                          /* the step's dt (single_step, simulation.rs:1973-1978) */
    float    max_mass_transfer_sharing;   /* dropped_mass_sharing, particle_sharing.rs:242-253 */
    uint32_t minimum_share_partners;      /* particle_sharing.rs:176, 220 */
    uint32_t minimum_merge_partners;      /* particle_merging.rs:287, 345 */
    int32_t  fail_on_missing_split_pattern;   /* splitting.rs:33-39 */
    /* read by the partner searches only (sph_host_find_partners) */
    float    max_share_distance, max_merge_distance;                  /* particle_sharing.rs:63-67, particle_merging.rs:75-79 */
    int32_t  allow_share_with_optimal_particle, allow_share_with_too_small_particle;   /* particle_sharing.rs:52-58 */
    int32_t  allow_merge_with_optimal_particle, allow_merge_on_size_difference;       /* particle_merging.rs:59-67 */
} sph_adapt_params;
#define SPH_MERGE_PARTNER_AVAILABLE 0xFFFFFFFFu   /* adaptivity/mod.rs:29 */
#define SPH_MERGE_PARTNER_DELETE    0xFFFFFFFEu   /* adaptivity/mod.rs:30 */
/* share_particles (particle_sharing.rs:152-240): a receiver i (merge_partner[i] = donor index j) takes
 * dropped_mass_sharing(j) / merge_counter[j] of the donor's mass and momentum, position mass-weighted; a donor
 * (merge_partner = DELETE) then loses what it dropped; h2_next from the new masses.  n is unchanged. */
int  sph_share_particles(sph_ctx* ctx, const sph_params* params, const sph_adapt_params* ap, const uint32_t* merge_partner,
                         const uint16_t* merge_counter);
/* merge_particles (particle_merging.rs:270-370): the same transfer with the donor's WHOLE mass, then the donors whose mass
 * fell below 1e-6 are deleted by the reference's swap-with-the-last loop (the k-th hole from the front receives the k-th
 * surviving particle from the back) and the vector is truncated.  sph_num_particles() is the new length. */
int  sph_merge_particles(sph_ctx* ctx, const sph_params* params, const sph_adapt_params* ap, const uint32_t* merge_partner,
                         const uint16_t* merge_counter);
/* SplitPatterns (splitting.rs:84-120; split-patterns.yaml): pattern k (k = 0 .. n_patterns-1) holds the k + 2 child offsets
 * pos_s of a 1-to-(k+2) split in units of the parent's radius, concatenated as (x, y) pairs. */
int  sph_set_split_patterns(sph_ctx* ctx, uint32_t n_patterns, const float* pos_s_xy);
/* split_particles (splitting.rs:19-82): every TooLarge particle (by the classes sph_classify left) becomes
 * round(mass / target_mass) children (at most the largest pattern, or SPH_ERR_NO_SPLIT_PATTERN with
 * fail_on_missing_split_pattern): child 0 replaces the parent, the others are appended in the order of the parents' indices.
 * Reproduces the reference's quirk that children >= 1 write h2 and level_old into the PARENT's slot (splitting.rs:73, 76):
 * an appended child keeps h2 = 0 and level_old = 0 (harmless under FromMass, where h2 is recomputed every step). */
int  sph_split_particles(sph_ctx* ctx, const sph_params* params, const sph_adapt_params* ap);

/* find_share_partner_sequential (particle_sharing.rs:14-117, kind = 0) / find_merge_partner_sequential (particle_merging.rs:16-125,
 * kind = 1) as HOST code: the reference's sequential greedy loop over the particles, each over its neighbour list in list order,
 * on host arrays (what sph_download / sph_download_neighbors returned) -- no device, no context.  For hosts that do not bring
 * their own implementation (the Rust side has one; the Python mirror uses this for million-particle scenes).  Writes
 * merge_partner[n] / merge_counter[n] and runs the reference's validate_*_partners assertions (SPH_ERR_INVALID_ARGUMENT if one
 * fails). */
int  sph_host_find_partners(int kind, uint64_t n, const uint8_t* particle_size_class, const float* mass, const float* level_estimation,
                            const float* position_xy, const float* h2, const uint32_t* offsets, const uint32_t* indices,
                            const sph_params* params, const sph_adapt_params* ap, uint32_t* merge_partner, uint16_t* merge_counter,
                            uint64_t* n_transfers);

/* Read one field back in HOST particle order. */
int  sph_download(sph_ctx* ctx, int field, void* dst, uint64_t dst_bytes);

/* Current neighbour lists (NeighborhoodCache.neighs, neighborhood_search.rs:13-15) as CSR in
 * host particle order: offsets[n+1], indices[offsets[n]].  Call with indices == NULL to get the
 * total in *n_indices first.  Set: { j : |x_ij|^2 < ((h_i+h_j)/2 * 2)^2 }, self included
 * (neighborhood_search.rs:143-146, 187-238).  Order within a list is unspecified (the reference's
 * is R*-tree traversal order).  After a step with level_estimation_after_advection the cache holds what
 * the reference rebuilt at the end of the step (simulation.rs:2678-2689): the lists of the ADVECTED positions
 * with range (h_i+h_j)/2 * level_estimation_range / 1.9.
 * Slab context: one row per OWNED particle in the order of sph_download(SPH_F_PARTICLE_ID); the indices are global particle ids
 * (an owned particle's neighbours are all among the rank's owned particles and ghosts, and ghost records carry their ids). */
int  sph_download_neighbors(sph_ctx* ctx, uint32_t* offsets, uint32_t* indices, uint64_t indices_capacity,
                            uint64_t* n_indices);

uint64_t sph_num_particles(const sph_ctx* ctx);
float    sph_time(const sph_ctx* ctx);
int      sph_set_time(sph_ctx* ctx, float time, uint64_t step_number);

/* ---- THE hot path ---------------------------------------------------------------------------
 * sph_step  <->  single_step_without_adaptivity (sim.rs:1980-2730).  Synchronous: returns after
 * dt and the statistics are on the host.  `out` may be NULL.
 * A non-zero status that comes from a guard INSIDE the step (the reference panics there and its caller drops the
 * simulation, main_loop.rs:300-311) leaves the particle state undefined -- velocities may be advanced and positions not:
 * the context is then POISONED: sph_step / sph_group_step / sph_apply_edits and the adaptivity entry points return
 * SPH_ERR_POISONED until sph_upload replaces the particle set; downloads still work (diagnostics).  Refusals taken before
 * anything is launched (SPH_ERR_INVALID_ARGUMENT, SPH_ERR_NO_BOUNDARY, SPH_ERR_UNSUPPORTED for a parameter combination)
 * leave the context as it was. */
int  sph_step(sph_ctx* ctx, const sph_params* params, sph_step_stats* out);

/* classify_particles (adaptivity/mod.rs:50-59): particle_size_class from LevelEstimationState::target_mass
 * (simulation.rs:213-237) and the thresholds 0.5, 1/1.1, 1.1, 2.  The reference calls it from single_step_adaptivity only
 * (simulation.rs:2749, 2764, 2778), never from the step: IISPH2's omega (simulation.rs:2277-2288) therefore reads whatever
 * the last adaptivity pass left behind (Optimal if there never was one).  Reads params->maximum_surface_distance,
 * rest_density, sizing_function, particle_radius_fine / _base.  A particle without a level value (FluidInterior) is
 * `unreachable!()` in the reference: SPH_ERR_INVALID_ARGUMENT here. */
int  sph_classify(sph_ctx* ctx, const sph_params* params);

/* Message for the last non-zero status of this context (what the Rust shim puts in panic!). */
const char* sph_last_error(const sph_ctx* ctx);

/* ---- grid info (CellGrid of neighborhood_search.rs:355-410 as built by the last step) ------- */
typedef struct sph_grid_info {
    float   cell_size;           /* support radius of the largest particle */
    int32_t cells_min_x, cells_min_y;
    int32_t size_x, size_y;
} sph_grid_info;
int  sph_grid(const sph_ctx* ctx, sph_grid_info* out);

/* ---- measurement hooks (bench.py / rocprof cross-check; not part of the reference surface) -- */
typedef struct sph_kernel_time {
    char     name[48];
    uint64_t launches;
    double   total_ms;           /* HIP-event time on the context's stream */
    uint64_t working_launches;   /* launches longer than a quarter of the kernel's 90th-percentile duration: a speculatively
                                  * queued Jacobi iteration behind the stop decision returns at once and is not a sweep */
    double   working_ms;
} sph_kernel_time;
int  sph_profile_enable(sph_ctx* ctx, int enable);
int  sph_profile_reset(sph_ctx* ctx);
int  sph_profile_get(sph_ctx* ctx, sph_kernel_time* out, int capacity, int* n_out);
/* what a HIP-event pair adds to the duration of the kernel it brackets (microseconds), measured with empty kernels:
 * 2 x (pair around one) - (pair around two).  bench.py subtracts it to compare with rocprofv3's kernel durations. */
int  sph_profile_event_overhead(sph_ctx* ctx, double* microseconds);
/* a marker-event pair (what sph_profile_enable(ctx, 1) brackets every kernel with) around a one-wave kernel that spins `spin_us`
 * microseconds of the device clock, alone on the queue, mean over `reps` launches (diagnostic: the in-step calibration of bench.py is the
 * profiler's own "calibration_spin10" record) */
int sph_profile_dispatch_bracket(sph_ctx* ctx, uint32_t spin_us, int reps, double* mean_bracket_us);
/* bandwidth of a plain float4 copy kernel over `bytes` of device memory (read + write, GB/s, best of 5): the achievable
 * HBM rate on this device, reported next to the 8 TB/s spec peak */
int  sph_profile_copy_bandwidth(sph_ctx* ctx, uint64_t bytes, double* gb_per_s);
/* which form the neighbour lists of the LAST step were recorded in (counted over the list words the density sweep wrote:
 * owned particles and, on a slab, the first ghost ring): row masks (3 x 3 cells of the sorting grid, <= 32 candidates per row:
 * 16 B per particle, replayed without index loads), explicit index lists (particles next to larger ones, or crowded rows), or
 * neither (> 128 neighbours: the candidates are walked in every sweep) */
typedef struct sph_list_forms {
    uint64_t n_lists, n_mask, n_index, n_walk, n_wall;   /* n_wall: particles with boundary terms (either form) */
} sph_list_forms;
int  sph_profile_list_forms(sph_ctx* ctx, sph_list_forms* out);

/* Which form of the neighbour sweeps runs in uniform-h scenes (process-wide; both forms give bit-identical results, the
 * ablation harness scripts/variants/ times them): bit 0 = the density / list-building sweep, bit 1 = the list-replaying sweeps
 * through the LDS-staged form (the wave's three candidate rows loaded once, coalesced, into LDS) instead of per-lane gathers.
 * The environment variable SPH_TILE sets the initial value. */
int sph_set_sweep_variant(int mode);

/* ---- multi-GPU: 1-D slab decomposition along x, one process (= one context) per GPU --------
 * (the reference has no counterpart: its only parallelism is rayon inside one process, concurrency.rs:110-204)
 * sph_dist_configure makes `ctx` rank `rank` of `n_ranks`, owning the particles with cut_lo <= x < cut_hi
 * (the outermost ranks ignore their open side); sph_upload then takes this rank's particles only.
 * The 128-byte RCCL unique id is created on rank 0 (sph_comm_unique_id) and broadcast by the launcher
 * (torch.distributed / MPI / a socket); every rank then calls sph_comm_init.  After that sph_step hands
 * particles that left the slab to the x-neighbour, exchanges the ghost layer (two support radii: the ghosts of the first
 * ring compute their own pressure acceleration) with the x-neighbours over RCCL point-to-point after every sweep whose
 * output neighbours read -- once per Jacobi iteration --, and all-reduces the CFL minimum and the Jacobi residual
 * statistics so that every rank takes the same decisions.  A slab narrower than two ghost layers is refused.
 * With level_estimation_after_advection the ghost layer is widened by 4 x the previous step's largest displacement (at least
 * 2 h_max); a step in which the particles move further returns SPH_ERR_UNSUPPORTED on every rank.
 * Every rank makes the SAME sequence of calls with the same parameters (sph_step, sph_upload*, sph_dist_set_rebalance): which
 * collectives a step runs depends on that history (an ordinary step maintains the slabs in one round trip of counts, the first
 * step after an upload and re-balancing steps take two or more), not on anything a rank could decide alone.
 * A particle reaches its owner however far from its slab it is when a step begins (an edit set it down elsewhere, a diverged
 * solve threw it across a slab): the hand-over to the x-neighbour repeats until nobody is further past a cut than the narrowest slab.
 * On slabs of >= 786432 particles with a neighbour, the pressure-acceleration sweep of a Jacobi iteration runs in two launches
 * (particles without a ghost in reach on a second stream, beside the iteration's exchange and all-reduce; then the rest): same
 * results, SPH_OVERLAP=0 / 1 in the environment forces one form.
 * sph_group_step steps k contexts of ONE process as ranks 0..k-1 with plain copies as transport (one GPU or several), ordered by
 * events between the contexts' streams (no host wait inside an exchange): same algorithm, used to verify the decomposition
 * against a single context, and the form a single-process host drives several GPUs with. */
int  sph_dist_configure(sph_ctx* ctx, int rank, int n_ranks, float cut_lo, float cut_hi);
/* Slab re-balancing: every `every_n_steps` steps (0 = never, the default) the cuts move to the quantiles of the particles' x
 * (x range and a 4096-bin histogram, all-reduced) so that every rank owns the same number of particles again; particles follow
 * through the ordinary hand-over to the x-neighbour, repeated until nobody moves.  Cuts that would make a slab narrower than
 * two ghost layers are not applied.  Set the same value on every rank.  sph_dist_get_cuts reads the current cuts back. */
int  sph_dist_set_rebalance(sph_ctx* ctx, int every_n_steps);
int  sph_dist_get_cuts(sph_ctx* ctx, float* cut_lo, float* cut_hi, uint32_t* n_rebalances);
/* Communication counters of this rank since the last reset (measurement hook, bench.py): neighbour exchanges (one grouped
 * ncclSend/ncclRecv pair per x-neighbour), bytes through them, all-reduces, host waits on the device (also counted on a plain
 * context), and the sizes of the current decomposition. */
typedef struct sph_dist_stats {
    uint64_t steps, exchanges, bytes_sent, bytes_received, allreduces, host_waits;
    uint64_t n_owned;
    uint32_t n_halo[2];    /* owned particles copied to the left / right neighbour as its ghosts */
    uint32_t n_ghost[2];   /* ghosts received from the left / right neighbour */
    uint32_t transport;    /* what the NEXT step's exchanges run over: 0 no slab context, 1 none attached yet (sph_group_step's loopback
                            * copies), 2 RCCL, 3 threads, 4 shared memory (host-staged), 5 peer-mapped push (hipIpc) */
    uint32_t comm_ranks;   /* ranks of that transport AS THE LIBRARY SEES THEM: ncclCommCount of the communicator (RCCL), the ranks mapped
                            * (peer-mapped push), the group's size (threads, shared memory) -- the launcher checks it against its world size */
} sph_dist_stats;
int  sph_dist_get_stats(sph_ctx* ctx, sph_dist_stats* out, int reset);
int  sph_comm_unique_id(uint8_t id_out[128]);
int  sph_comm_init(sph_ctx* ctx, const uint8_t id[128], int rank, int n_ranks);
int  sph_group_step(sph_ctx** ctxs, int n, const sph_params* params, sph_step_stats* outs);
/* sph_share_particles (op 0) / sph_merge_particles (1) / sph_split_particles (2) for the k contexts sph_group_step steps, in their
 * slab form (see "adaptivity data path" above): merge_partner / merge_counter by global particle id, NULL for op 2 */
int  sph_group_adapt(sph_ctx** ctxs, int n, int op, const sph_params* params, const sph_adapt_params* ap, const uint32_t* merge_partner,
                     const uint16_t* merge_counter);
/* Verification transport for the PER-RANK driver code: all ranks in this process, one HOST THREAD per rank, each calling sph_step
 * on its own slab context -- a group of one member, its own view of the counts, its own branches, exactly as a rank of the RCCL
 * transport runs -- with the collectives as rendezvous in host memory.  What would hang RCCL is an error here: a collective
 * that not every rank enters (60 s), ranks in different collectives, a send without a receive of the same size on the other
 * side.  sph_thread_group_create(n) once, sph_comm_init_threads on every rank's context instead of sph_comm_init. */
/* Transport for the ranks of ONE NODE as separate processes without RCCL: rendezvous and staging through a POSIX shared-memory
 * segment `name` ("/..."), `bytes_per_side` of room for one message to one x-neighbour (ghost records: 28 B, migrants: 48 B each).
 * Rank 0 creates it (create = 1) before the others map it -- the launcher orders that.  Host-synchronous and PCIe-bound: for boxes
 * where RCCL cannot serve the launch (several ranks on one GPU: "Duplicate GPU detected") and for checking the launcher glue with
 * real processes; sph_comm_init is the fast path.  At most 16 ranks. */
int  sph_comm_init_shm(sph_ctx* ctx, const char* name, int rank, int n_ranks, uint64_t bytes_per_side, int create);
/* Peer-mapped PUSH transport on top of the shared-memory one (ranks as processes of one node, one per GPU or several per GPU): every
 * rank exports a device buffer (sph_comm_ipc_export: its hipIpcMemHandle, 64 bytes), the launcher all-gathers the handles, every rank
 * maps the others' (sph_comm_init_ipc).  From then on the ghost / migrant exchanges and the all-reduce of the Jacobi totals are pushed
 * device to device by small kernels -- the sender writes into the receiver's inbox and raises a flag there, the receiver's next kernel
 * waits on it -- with no collective-library launch and no host wait per exchange; the host-value collectives (counts, header) stay the
 * shared-memory transport's.  `bytes_per_side`: room for one message from one x-neighbour (as for sph_comm_init_shm). */
int  sph_comm_ipc_export(sph_ctx* ctx, uint64_t bytes_per_side, uint8_t handle_out[64]);
int  sph_comm_init_ipc(sph_ctx* ctx, const uint8_t* handles /* n_ranks x 64 bytes, rank order */, int n_ranks);
int  sph_thread_group_create(int n_ranks, void** group_out);
void sph_thread_group_destroy(void* group);
int  sph_comm_init_threads(sph_ctx* ctx, void* group, int rank, int n_ranks);

#ifdef __cplusplus
}
#endif

#endif /* SPH_FFI_H */
