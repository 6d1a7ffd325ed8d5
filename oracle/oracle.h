/*
 * TEST INFRASTRUCTURE -- CPU oracle for the SPH particle loop (internal header).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything in oracle/.
 * parity pinning: see DESIGN.md "Oracle" -- pinned against the reference's own known-answer tests
 * (lambda/dlambda Maxima values, kernel normalisation, gradient finite differences, radius/volume
 * round trip, scene particle counts); no trajectory golden data exists in the reference.
 */
#ifndef ORACLE_H
#define ORACLE_H

#include <stddef.h>
#include <stdint.h>

#include "../include/sph_ffi.h"

#define ORC_LUT_STEPS 10000
#define ORC_MAX_PLANES 8
#define ORC_MAX_NEIGHBOR_COUNT 20000 /* neighborhood_search.rs:3 */

typedef struct orc_lut {
    float min, max, len_inv;
    int steps;
    float data[ORC_LUT_STEPS + 1];
} orc_lut;

typedef struct oracle_ctx {
    uint64_t cap, n;
    int n_planes;
    sph_plane planes[ORC_MAX_PLANES];
    /* Sdf2D with one connected component (sdf/sdf2d.rs:4-16); poly_n > 0 replaces the planes */
    int poly_n;
    float poly_x[SPH_MAX_POLYGON_POINTS], poly_y[SPH_MAX_POLYGON_POINTS];       /* point */
    float poly_dx[SPH_MAX_POLYGON_POINTS], poly_dy[SPH_MAX_POLYGON_POINTS];     /* normalized_line_dir */
    float poly_nx[SPH_MAX_POLYGON_POINTS], poly_ny[SPH_MAX_POLYGON_POINTS];     /* point_pseudo_normal */
    orc_lut lambda_lut, dlambda_lut;

    /* ParticleVec (simulation.rs:284-334); VF<2> arrays are interleaved x,y */
    float *mass, *pos, *vel, *vel_tmp, *pacc;
    float *density, *source, *pressure, *pressure_next, *aii, *density_error;
    float *h2, *h2_next;
    float* omega; /* IISPH2 (simulation.rs:2262-2311) */
    float *level, *level_tmp, *level_old; /* LevelEstimationState: NaN = FluidInterior */
    float *constant_field, *stash;
    uint8_t *flag_surface, *flag_insufficient, *size_class, *flag_reduced;
    uint32_t* neighbor_count;

    /* BoundaryWinchenbach2020.lambda: Vec<Vec<(FT, VF<2>)>> (boundary_winchenbach2020.rs:27) */
    uint8_t* lam_n;
    float *lam, *lam_gx, *lam_gy; /* [n][ORC_MAX_PLANES] */

    /* NeighborhoodCache as CSR */
    uint64_t* nb_off;
    uint32_t* nb_idx;
    uint64_t nb_cap;
    uint32_t* nb_idx2; /* orc_filter_down's second buffer */
    uint64_t nb_cap2;

    sph_grid_info grid;
    uint32_t* cell_index;

    float time;
    uint64_t step_number;
    char err[256];
} oracle_ctx;

/* lambda.c */
double orc_lambda2(double d);
double orc_dlambda2(double d);
void orc_lut_build(orc_lut* lut, double (*f)(double));
float orc_lut_get(const orc_lut* lut, float x);

/* neigh.c */
int orc_build_neighbors(oracle_ctx* c, float k);      /* build_neighborhood_list (rstar semantics) */
void orc_filter_down(oracle_ctx* c, float k);         /* NeighborhoodCache::filter_down */
int orc_check_neighborhood(oracle_ctx* c);            /* check_correct_neighborhood, all i */
void orc_cell_indices(oracle_ctx* c);                 /* CellGrid convention, cell = 2*h_max */

/* step.c */
int orc_step(oracle_ctx* c, const sph_params* p, sph_step_stats* out);
int orc_classify_particles(oracle_ctx* c, const sph_params* p); /* adaptivity/mod.rs:50-59 */

int orc_fail(oracle_ctx* c, int code, const char* fmt, ...);

#endif
