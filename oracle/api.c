/*
 * TEST INFRASTRUCTURE -- CPU oracle (see oracle.h).
 *
 * api.c: the oracle behind the SAME C ABI as the product library (include/sph_ffi.h), with the
 * symbol prefix oracle_ instead of sph_, so one ctypes harness drives both.
 * Also exports the scalar primitives the reference unit-tests (sph_kernels.rs:88-163, 214-227;
 * plane_numerics.rs:180-299) and the scene initialiser (simulation.rs:2915-2983).
 */
#include "oracle.h"
#include "sphmath.h"

#include <math.h>
#include <omp.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int orc_fail(oracle_ctx* c, int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(c->err, sizeof c->err, fmt, ap);
    va_end(ap);
    return code;
}

#define ALLOC(ptr, count, type) ((ptr) = (type*)calloc((size_t)(count) + 1, sizeof(type)))

/* Sdf2D::new_boundary_box / Sdf2DConnectedComponents::from_points (sdf/sdf2d.rs:36-75, 167-179) */
int oracle_set_boundary_polygon(oracle_ctx* c, const float* pts, int n)
{
    if (!c || !pts || n < 3 || n > SPH_MAX_POLYGON_POINTS) return SPH_ERR_INVALID_ARGUMENT;
    for (int i = 0; i < n; i++) {
        c->poly_x[i] = pts[2 * i];
        c->poly_y[i] = pts[2 * i + 1];
    }
    for (int i = 0; i < n; i++) {
        float dx = c->poly_x[(i + 1) % n] - c->poly_x[i], dy = c->poly_y[(i + 1) % n] - c->poly_y[i];
        float n2 = dx * dx + dy * dy;
        if (!(n2 > 0.00001f)) return orc_fail(c, SPH_ERR_INVALID_ARGUMENT, "assertion failed: line_dir.norm_squared() > 0.00001");
        float nn = sqrtf(n2); /* normalize_mut */
        c->poly_dx[i] = dx / nn;
        c->poly_dy[i] = dy / nn;
    }
    for (int i = 0; i < n; i++) {
        int pi = i == 0 ? n - 1 : i - 1;
        /* rotate_left_90_degrees(v) = (-v.y, v.x) */
        float px = -c->poly_dy[pi] + -c->poly_dy[i], py = c->poly_dx[pi] + c->poly_dx[i];
        if (!(px * px + py * py > 0.00001f)) return orc_fail(c, SPH_ERR_INVALID_ARGUMENT, "assertion failed: pseudo_normal.norm_squared() > 0.00001");
        c->poly_nx[i] = px;
        c->poly_ny[i] = py;
    }
    c->poly_n = n;
    c->n_planes = 1; /* one Sdf */
    return SPH_OK;
}

int oracle_create(uint64_t n_capacity, int device_id, const sph_plane* planes, int n_planes, oracle_ctx** out)
{
    (void)device_id;
    if (!out || n_planes < 0 || n_planes > ORC_MAX_PLANES || (n_planes > 0 && !planes)) return SPH_ERR_INVALID_ARGUMENT;
    oracle_ctx* c = (oracle_ctx*)calloc(1, sizeof(oracle_ctx));
    if (!c) return SPH_ERR_DEVICE;
    c->cap = n_capacity;
    c->n_planes = n_planes;
    for (int k = 0; k < n_planes; k++) c->planes[k] = planes[k];
    /* boundary_winchenbach2020.rs:33-36 */
    orc_lut_build(&c->lambda_lut, orc_lambda2);
    orc_lut_build(&c->dlambda_lut, orc_dlambda2);
    const uint64_t n = n_capacity;
    ALLOC(c->mass, n, float); ALLOC(c->pos, 2 * n, float); ALLOC(c->vel, 2 * n, float); ALLOC(c->vel_tmp, 2 * n, float);
    ALLOC(c->pacc, 2 * n, float); ALLOC(c->density, n, float); ALLOC(c->source, n, float); ALLOC(c->pressure, n, float);
    ALLOC(c->pressure_next, n, float); ALLOC(c->aii, n, float); ALLOC(c->density_error, n, float);
    ALLOC(c->h2, n, float); ALLOC(c->h2_next, n, float); ALLOC(c->omega, n, float); ALLOC(c->level, n, float); ALLOC(c->level_tmp, n, float);
    ALLOC(c->level_old, n, float); ALLOC(c->constant_field, n, float); ALLOC(c->stash, n, float);
    ALLOC(c->flag_surface, n, uint8_t); ALLOC(c->flag_insufficient, n, uint8_t); ALLOC(c->size_class, n, uint8_t); ALLOC(c->flag_reduced, n, uint8_t);
    ALLOC(c->neighbor_count, n, uint32_t); ALLOC(c->lam_n, n, uint8_t);
    ALLOC(c->lam, n * ORC_MAX_PLANES, float); ALLOC(c->lam_gx, n * ORC_MAX_PLANES, float); ALLOC(c->lam_gy, n * ORC_MAX_PLANES, float);
    ALLOC(c->nb_off, n + 1, uint64_t); ALLOC(c->cell_index, n, uint32_t);
    c->nb_cap = 0;
    c->nb_idx = NULL;
    *out = c;
    return SPH_OK;
}

void oracle_destroy(oracle_ctx* c)
{
    if (!c) return;
    free(c->mass); free(c->pos); free(c->vel); free(c->vel_tmp); free(c->pacc); free(c->density); free(c->source);
    free(c->pressure); free(c->pressure_next); free(c->aii); free(c->density_error); free(c->h2); free(c->h2_next); free(c->omega);
    free(c->level); free(c->level_tmp); free(c->level_old); free(c->constant_field); free(c->stash);
    free(c->flag_surface); free(c->flag_insufficient); free(c->size_class); free(c->flag_reduced); free(c->neighbor_count); free(c->lam_n);
    free(c->lam); free(c->lam_gx); free(c->lam_gy); free(c->nb_off); free(c->nb_idx); free(c->nb_idx2); free(c->cell_index);
    free(c);
}

/* FluidSimulation::new (simulation.rs:487-533): defaults of ParticleVec (:284-334), h2_next from mass */
int oracle_upload(oracle_ctx* c, uint64_t n, const float* mass, const float* pos, const float* vel)
{
    if (!c || (n && (!mass || !pos || !vel))) return SPH_ERR_INVALID_ARGUMENT;
    if (n > c->cap) return orc_fail(c, SPH_ERR_CAPACITY, "n=%llu exceeds capacity %llu", (unsigned long long)n, (unsigned long long)c->cap);
    c->n = n;
    /* written by the threads that will read them: pages are placed on first touch, and a 128-core host has several NUMA nodes
     * (the buffers come from calloc, so nothing has been touched before the first upload) */
#pragma omp parallel for schedule(static)
    for (int64_t ii = 0; ii < (int64_t)n; ii++) {
        const uint64_t i = (uint64_t)ii;
        c->mass[i] = mass[i];
        for (int d = 0; d < 2; d++) {
            c->pos[2 * i + d] = pos[2 * i + d];
            c->vel[2 * i + d] = vel[2 * i + d];
            c->vel_tmp[2 * i + d] = 0.f;
            c->pacc[2 * i + d] = 0.f;
        }
        c->density[i] = c->source[i] = c->pressure[i] = c->pressure_next[i] = c->aii[i] = c->density_error[i] = 0.f;
        c->h2[i] = c->level_old[i] = c->constant_field[i] = c->stash[i] = c->omega[i] = 0.f;
        c->flag_surface[i] = c->flag_insufficient[i] = c->flag_reduced[i] = c->lam_n[i] = 0;
        c->neighbor_count[i] = c->cell_index[i] = 0;
        c->nb_off[i] = 0;
        for (int q = 0; q < ORC_MAX_PLANES; q++) c->lam[i * ORC_MAX_PLANES + q] = c->lam_gx[i * ORC_MAX_PLANES + q] = c->lam_gy[i * ORC_MAX_PLANES + q] = 0.f;
        c->h2_next[i] = orc_h_from_mass(mass[i], 1.f /* INIT_REST_DENSITY, simulation.rs:344 */);
        c->level[i] = NAN; /* LevelEstimationState::FluidInterior */
        c->level_tmp[i] = NAN;
        c->size_class[i] = 2; /* ParticleSizeClass::Optimal */
    }
    c->nb_off[n] = 0;
    return SPH_OK;
}

/* The edit script of sph_ffi.h applied the way the reference does it: element writes and ParticleVec::swap / truncate /
 * extend (simulation.rs:248-271; the same calls reach neighs and boundary_handler: particle_merging.rs:357-370,
 * splitting.rs:56-58), one op after the other on the host arrays. */
static void swapf(float* a, uint64_t i, uint64_t j, int w)
{
    for (int d = 0; d < w; d++) { float t = a[w * i + d]; a[w * i + d] = a[w * j + d]; a[w * j + d] = t; }
}

int oracle_apply_edits(oracle_ctx* c, const sph_edit_op* ops, uint64_t n_ops)
{
    if (!c || (n_ops && !ops)) return SPH_ERR_INVALID_ARGUMENT;
    for (uint64_t k = 0; k < n_ops; k++) {
        const sph_edit_op* o = &ops[k];
        switch (o->kind) {
        case SPH_EDIT_SET: {
            if (o->a >= c->n) return orc_fail(c, SPH_ERR_INVALID_ARGUMENT, "edit %llu: index out of bounds", (unsigned long long)k);
            const uint64_t i = o->a;
            if (o->fields & SPH_EDIT_F_MASS) c->mass[i] = o->mass;
            if (o->fields & SPH_EDIT_F_POSITION) { c->pos[2 * i] = o->position[0]; c->pos[2 * i + 1] = o->position[1]; }
            if (o->fields & SPH_EDIT_F_VELOCITY) { c->vel[2 * i] = o->velocity[0]; c->vel[2 * i + 1] = o->velocity[1]; }
            if (o->fields & SPH_EDIT_F_H2) c->h2[i] = o->h2;
            if (o->fields & SPH_EDIT_F_H2_NEXT) c->h2_next[i] = o->h2_next;
            if (o->fields & SPH_EDIT_F_LEVEL_ESTIMATION) c->level[i] = o->level_estimation;
            if (o->fields & SPH_EDIT_F_LEVEL_OLD) c->level_old[i] = o->level_old;
        } break;
        case SPH_EDIT_SWAP: {
            if (o->a >= c->n || o->b >= c->n) return orc_fail(c, SPH_ERR_INVALID_ARGUMENT, "edit %llu: swap out of bounds", (unsigned long long)k);
            const uint64_t i = o->a, j = o->b;
            swapf(c->mass, i, j, 1); swapf(c->pos, i, j, 2); swapf(c->vel, i, j, 2); swapf(c->h2, i, j, 1); swapf(c->h2_next, i, j, 1);
            swapf(c->level, i, j, 1); swapf(c->level_old, i, j, 1);
            swapf(c->lam, i, j, ORC_MAX_PLANES); swapf(c->lam_gx, i, j, ORC_MAX_PLANES); swapf(c->lam_gy, i, j, ORC_MAX_PLANES);
            uint8_t t = c->lam_n[i]; c->lam_n[i] = c->lam_n[j]; c->lam_n[j] = t;
        } break;
        case SPH_EDIT_TRUNCATE:
            if (o->a < c->n) c->n = o->a;
            break;
        case SPH_EDIT_EXTEND:
            if (c->n + o->a > c->cap) return orc_fail(c, SPH_ERR_CAPACITY, "edit %llu exceeds the capacity", (unsigned long long)k);
            for (uint64_t i = c->n; i < c->n + o->a; i++) {
                c->mass[i] = 0.f; c->pos[2 * i] = c->pos[2 * i + 1] = 0.f; c->vel[2 * i] = c->vel[2 * i + 1] = 0.f;
                c->h2[i] = 0.f; c->h2_next[i] = 0.f; c->level[i] = NAN; c->level_old[i] = 0.f; c->lam_n[i] = 0;
            }
            c->n += o->a;
            break;
        default: return orc_fail(c, SPH_ERR_INVALID_ARGUMENT, "edit %llu: unknown kind %d", (unsigned long long)k, o->kind);
        }
    }
    /* per-step outputs and the neighbour lists belong to the vector before the edit */
    memset(c->nb_off, 0, (c->n + 1) * sizeof(uint64_t));
    return SPH_OK;
}

typedef struct { void* ptr; size_t elem; int width; } field_ref;

static field_ref field_of(oracle_ctx* c, int field)
{
    field_ref r = {NULL, 0, 0};
    switch (field) {
    case SPH_F_MASS: r = (field_ref){c->mass, 4, 1}; break;
    case SPH_F_POSITION: r = (field_ref){c->pos, 4, 2}; break;
    case SPH_F_VELOCITY: r = (field_ref){c->vel, 4, 2}; break;
    case SPH_F_PRESSURE_ACCEL: r = (field_ref){c->pacc, 4, 2}; break;
    case SPH_F_DENSITY: r = (field_ref){c->density, 4, 1}; break;
    case SPH_F_PPE_SOURCE_TERM: r = (field_ref){c->source, 4, 1}; break;
    case SPH_F_PRESSURE: r = (field_ref){c->pressure, 4, 1}; break;
    case SPH_F_AII: r = (field_ref){c->aii, 4, 1}; break;
    case SPH_F_DENSITY_ERROR: r = (field_ref){c->density_error, 4, 1}; break;
    case SPH_F_H2: r = (field_ref){c->h2, 4, 1}; break;
    case SPH_F_H2_NEXT: r = (field_ref){c->h2_next, 4, 1}; break;
    case SPH_F_CONSTANT_FIELD: r = (field_ref){c->constant_field, 4, 1}; break;
    case SPH_F_NEIGHBOR_COUNT: r = (field_ref){c->neighbor_count, 4, 1}; break;
    case SPH_F_LEVEL_ESTIMATION: r = (field_ref){c->level, 4, 1}; break;
    case SPH_F_LEVEL_OLD: r = (field_ref){c->level_old, 4, 1}; break;
    case SPH_F_STASH: r = (field_ref){c->stash, 4, 1}; break;
    case SPH_F_FLAG_IS_FLUID_SURFACE: r = (field_ref){c->flag_surface, 1, 1}; break;
    case SPH_F_FLAG_INSUFFICIENT_NEIGHS: r = (field_ref){c->flag_insufficient, 1, 1}; break;
    case SPH_F_PARTICLE_SIZE_CLASS: r = (field_ref){c->size_class, 1, 1}; break;
    case SPH_F_FLAG_NEIGHBORHOOD_REDUCED: r = (field_ref){c->flag_reduced, 1, 1}; break;
    case SPH_F_CELL_INDEX: r = (field_ref){c->cell_index, 4, 1}; break;
    default: break;
    }
    return r;
}

int oracle_upload_field(oracle_ctx* c, int field, const void* src, uint64_t bytes)
{
    if (!c || !src) return SPH_ERR_INVALID_ARGUMENT;
    field_ref r = field_of(c, field);
    if (!r.ptr) return orc_fail(c, SPH_ERR_INVALID_ARGUMENT, "field %d cannot be uploaded", field);
    if (bytes != c->n * r.elem * r.width) return orc_fail(c, SPH_ERR_INVALID_ARGUMENT, "field %d: size mismatch", field);
    memcpy(r.ptr, src, bytes);
    return SPH_OK;
}

int oracle_download(oracle_ctx* c, int field, void* dst, uint64_t bytes)
{
    if (!c || !dst) return SPH_ERR_INVALID_ARGUMENT;
    if (field == SPH_F_LAMBDA_SUM || field == SPH_F_LAMBDA_GRAD_SUM) {
        int w = field == SPH_F_LAMBDA_SUM ? 1 : 2;
        if (bytes != c->n * 4 * w) return orc_fail(c, SPH_ERR_INVALID_ARGUMENT, "field %d: size mismatch", field);
        float* o = (float*)dst;
        for (uint64_t i = 0; i < c->n; i++) {
            float s = 0.f, gx = 0.f, gy = 0.f;
            for (int k = 0; k < c->lam_n[i]; k++) {
                s += c->lam[i * ORC_MAX_PLANES + k];
                gx += c->lam_gx[i * ORC_MAX_PLANES + k];
                gy += c->lam_gy[i * ORC_MAX_PLANES + k];
            }
            if (w == 1) o[i] = s;
            else { o[2 * i] = gx; o[2 * i + 1] = gy; }
        }
        return SPH_OK;
    }
    field_ref r = field_of(c, field);
    if (!r.ptr) return orc_fail(c, SPH_ERR_INVALID_ARGUMENT, "unknown field %d", field);
    if (bytes != c->n * r.elem * r.width) return orc_fail(c, SPH_ERR_INVALID_ARGUMENT, "field %d: size mismatch", field);
    memcpy(dst, r.ptr, bytes);
    return SPH_OK;
}

int oracle_download_neighbors(oracle_ctx* c, uint32_t* offsets, uint32_t* indices, uint64_t cap, uint64_t* n_indices)
{
    if (!c) return SPH_ERR_INVALID_ARGUMENT;
    uint64_t tot = c->nb_off[c->n];
    if (n_indices) *n_indices = tot;
    if (offsets) for (uint64_t i = 0; i <= c->n; i++) offsets[i] = (uint32_t)c->nb_off[i];
    if (indices) {
        if (cap < tot) return orc_fail(c, SPH_ERR_INVALID_ARGUMENT, "indices buffer too small");
        memcpy(indices, c->nb_idx, tot * sizeof(uint32_t));
    }
    return SPH_OK;
}

/* include/sph_ffi.h sph_set_math_policy: the oracle has ONE arithmetic -- the reference's IEEE operations in the reference's order
 * (what the device calls SPH_MATH_EXACT); the call is accepted so that the same host code drives both libraries. */
int oracle_set_math_policy(oracle_ctx* c, int policy)
{
    if (!c || (policy != SPH_MATH_FAST && policy != SPH_MATH_EXACT)) return SPH_ERR_INVALID_ARGUMENT;
    return SPH_OK;
}
int oracle_get_math_policy(const oracle_ctx* c) { return c ? SPH_MATH_EXACT : -1; }

uint64_t oracle_num_particles(const oracle_ctx* c) { return c ? c->n : 0; }
float oracle_time(const oracle_ctx* c) { return c ? c->time : 0.f; }
int oracle_set_time(oracle_ctx* c, float t, uint64_t step)
{
    if (!c) return SPH_ERR_INVALID_ARGUMENT;
    c->time = t;
    c->step_number = step;
    return SPH_OK;
}

int oracle_step(oracle_ctx* c, const sph_params* p, sph_step_stats* out)
{
    if (!c || !p) return SPH_ERR_INVALID_ARGUMENT;
    return orc_step(c, p, out);
}

int oracle_classify(oracle_ctx* c, const sph_params* p)
{
    if (!c || !p) return SPH_ERR_INVALID_ARGUMENT;
    return orc_classify_particles(c, p);
}

const char* oracle_last_error(const oracle_ctx* c) { return c ? c->err : "null context"; }

int oracle_grid(const oracle_ctx* c, sph_grid_info* out)
{
    if (!c || !out) return SPH_ERR_INVALID_ARGUMENT;
    *out = c->grid;
    return SPH_OK;
}

int oracle_num_threads(void) { return omp_get_max_threads(); }
void oracle_set_num_threads(int n) { omp_set_num_threads(n); }

/* ---- building blocks exposed for single-sweep parity tests ---------------------------------- */

/* build the k-radius neighbour lists for the CURRENT h2/positions (no step) */
int oracle_build_neighbors(oracle_ctx* c, float k)
{
    if (!c) return SPH_ERR_INVALID_ARGUMENT;
    int rc = orc_build_neighbors(c, k);
    if (!rc) orc_cell_indices(c);
    return rc;
}
int oracle_check_neighborhood(oracle_ctx* c) { return c ? orc_check_neighborhood(c) : SPH_ERR_INVALID_ARGUMENT; }

/* ---- scalar primitives (the reference's unit-test surface) ---------------------------------- */
float oracle_cubic_kernel_2d(float r, float h) { return orc_kernel2d(r, h); }
void oracle_cubic_kernel_2d_deriv(float dx, float dy, float h, float* gx, float* gy) { orc_kernel_derivh(dx, dy, h, gx, gy); }
float oracle_sphere_volume_to_radius(float a) { return orc_sphere_volume_to_radius(a); }
float oracle_radius_to_sphere_volume(float r) { return orc_radius_to_sphere_volume(r); }
float oracle_h_from_mass(float m, float rho0) { return orc_h_from_mass(m, rho0); }
double oracle_lambda2(double d) { return orc_lambda2(d); }
double oracle_dlambda2(double d) { return orc_dlambda2(d); }
void oracle_lambda_luts(const oracle_ctx* c, float* lambda_out, float* dlambda_out)
{
    memcpy(lambda_out, c->lambda_lut.data, sizeof c->lambda_lut.data);
    memcpy(dlambda_out, c->dlambda_lut.data, sizeof c->dlambda_lut.data);
}
float oracle_lut_get(const oracle_ctx* c, int which, float x) { return orc_lut_get(which ? &c->dlambda_lut : &c->lambda_lut, x); }

/* ---- scene init: add_fluid_block (simulation.rs:2915-2983), f32 arithmetic, x outer / y inner.
 * Returns the particle count; writes at most `cap` particles when the output pointers are non-NULL. */
uint64_t oracle_add_fluid_block(float min_x, float min_y, float size_x, float size_y, float spacing, float fill_ratio,
                                float vel_x, float vel_y, uint64_t cap, float* pos, float* mass, float* vel)
{
    /* init_fluid_sim (simulation.rs:3088-3099): max = pos + size, then box_size = max - min */
    float max_x = min_x + size_x, max_y = min_y + size_y;
    float particle_volume = spacing * spacing * fill_ratio;
    float particle_mass = particle_volume * 1.f;
    float bx = max_x - min_x, by = max_y - min_y;
    uint64_t nx = (uint64_t)floorf(bx / spacing), ny = (uint64_t)floorf(by / spacing);
    uint64_t k = 0;
    for (uint64_t x = 0; x < nx; x++)
        for (uint64_t y = 0; y < ny; y++, k++) {
            if (pos && k < cap) {
                pos[2 * k] = (float)x * spacing + min_x;
                pos[2 * k + 1] = (float)y * spacing + min_y;
                mass[k] = particle_mass;
                vel[2 * k] = vel_x;
                vel[2 * k + 1] = vel_y;
            }
        }
    return nx * ny;
}

/* SdfPlane::new_boundary_box (sdf_plane.rs:13-20) for a box centred at the origin
 * (init_fluid_sim, simulation.rs:3186-3199) */
void oracle_boundary_box(float width, float height, sph_plane out[4])
{
    float minx = 0.f - width / 2.f, miny = 0.f - height / 2.f;
    float maxx = 0.f + width / 2.f, maxy = 0.f + height / 2.f;
    out[0] = (sph_plane){1.f, 0.f, -minx};
    out[1] = (sph_plane){-1.f, 0.f, maxx};
    out[2] = (sph_plane){0.f, 1.f, -miny};
    out[3] = (sph_plane){0.f, -1.f, maxy};
}

/* ---- numerical integrals used by the reference's own unit tests, restated so the oracle can be
 * pinned against them at C speed ------------------------------------------------------------ */

/* sph_kernels.rs:88-114: midpoint-rule integral of the 2-D kernel over [-2h,2h]^2 (f32) */
float oracle_test_kernel_integral(float h, int grid_size)
{
    float support_radius = 2.0f * h;
    float square_len = 2.f * support_radius / (float)grid_size;
    float square_area = square_len * square_len;
    float integral = 0.f;
    for (int y = 0; y < grid_size; y++)
        for (int x = 0; x < grid_size; x++) {
            float px = ((float)x + 0.5f) * square_len - support_radius;
            float py = ((float)y + 0.5f) * square_len - support_radius;
            integral += orc_kernel2d(sqrtf(orc_norm_sq(px, py)), h) * square_area;
        }
    return integral;
}

/* plane_numerics.rs:251-299: integral of the kernel over the half plane y >= d (f64 accumulation
 * of f32 kernel values, 350x350 patches, partial patches by area fraction) */
double oracle_test_lambda2_integral(double h, double d)
{
    double support_radius = 2. * h;
    int grid_size = 350;
    double square_len = 2. * support_radius / (double)grid_size;
    double square_area = square_len * square_len;
    double integral = 0.;
    for (int y = 0; y < grid_size; y++)
        for (int x = 0; x < grid_size; x++) {
            double px = ((double)x + 0.5) * square_len - support_radius;
            double py = ((double)y + 0.5) * square_len - support_radius;
            double top = ((double)y + 1.0) * square_len - support_radius;
            double bottom = ((double)y + 0.0) * square_len - support_radius;
            double nrm = sqrt(px * px + py * py);
            if (bottom >= d) integral += (double)orc_kernel2d((float)nrm, (float)h) * square_area;
            else if (top > d) integral += (double)orc_kernel2d((float)nrm, (float)h) * ((top - d) / (top - bottom) * square_area);
        }
    return integral;
}
