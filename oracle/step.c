/*
 * TEST INFRASTRUCTURE -- CPU oracle (see oracle.h).
 *
 * step.c: scalar f32 restatement of FluidSimulation::single_step_without_adaptivity
 * (/root/reference/src/simulation/simulation.rs:1980-2730) and every sweep it launches.
 * Each function cites the lines it follows.  Arithmetic keeps the reference's operation order
 * (left-to-right, every op rounded; build with -ffp-contract=off).  The reference's rayon sweeps
 * are gather-only (each particle writes its own slot), so an OpenMP parallel-for over i is the
 * same computation; the only order-dependent pieces are the neighbour order (unpinned, see
 * neigh.c) and the residual tree-reduction (rayon order is nondeterministic; here: fixed
 * 1024-particle chunks, chunk sums added in index order).
 */
#include "oracle.h"
#include "sphmath.h"

#include <math.h>
#include <omp.h>
#include <stdlib.h>
#include <string.h>

#define NB_LOOP(c, i, j) \
    for (uint64_t q_ = (c)->nb_off[i], e_ = (c)->nb_off[(i) + 1], j = 0; q_ < e_ && ((j = (c)->nb_idx[q_]), 1); q_++)

#define PARFOR _Pragma("omp parallel for schedule(static)")

/* ------------------------------------------------------------------------------------------ */
/* boundary: BoundaryWinchenbach2020 (boundary_winchenbach2020.rs)                             */
/* ------------------------------------------------------------------------------------------ */

/* sdf_plane.rs:36-38 */
static inline float plane_probe(const sph_plane* pl, float x, float y) { return (pl->dir_x * x + pl->dir_y * y) + pl->delta; }

/* Sdf2D::probe = find_min_dist_object + to_dist_and_dir (sdf/sdf2d.rs:77-160, 196-228), one connected component */
static float polygon_probe(const oracle_ctx* c, float x, float y)
{
    const int n = c->poly_n;
    float min_dist_sq = INFINITY;
    int is_line = 0, point_idx = 0;
    float line_dist = 0.f, pdx = 0.f, pdy = 0.f, pt_dist_sq = INFINITY;
    for (int i = 0; i < n; i++) {
        const float sx = c->poly_x[i], sy = c->poly_y[i];
        const float ex = c->poly_x[(i + 1) % n], ey = c->poly_y[(i + 1) % n];
        const float lx = ex - sx, ly = ey - sy;
        const float line_len_sq = lx * lx + ly * ly;
        const float dx = c->poly_dx[i], dy = c->poly_dy[i];
        const float qx = x - sx, qy = y - sy;                 /* point_dir */
        const float leftx = -dy, lefty = dx;                   /* left_normalized_dir */
        const float projection_len = qx * dx + qy * dy;
        if (projection_len > 0.f && projection_len * projection_len < line_len_sq) {
            const float d = qx * leftx + qy * lefty;
            const float d2 = d * d;
            if (d2 < min_dist_sq) {
                is_line = 1;
                line_dist = d;
                min_dist_sq = d2;
            }
        }
        const float corner_dist_sq = qx * qx + qy * qy;
        if (corner_dist_sq < min_dist_sq) {
            is_line = 0;
            point_idx = i;
            pt_dist_sq = corner_dist_sq;
            pdx = qx;
            pdy = qy;
            min_dist_sq = corner_dist_sq;
        }
    }
    if (is_line) return line_dist;
    const float sign = (c->poly_nx[point_idx] * pdx + c->poly_ny[point_idx] * pdy) >= 0.f ? 1.0f : -1.0f;
    return sqrtf(pt_dist_sq) * sign;
}

/* Sdf::probe of SDF number k (sdf.rs): the polygon replaces the planes */
static inline float sdf_probe(const oracle_ctx* c, int k, float x, float y)
{
    return c->poly_n > 0 ? polygon_probe(c, x, y) : plane_probe(&c->planes[k], x, y);
}

/* boundary_winchenbach2020.rs:58-152 (+ sdf.rs:50-62 finite_diff_gradient) */
static void update_after_advect(oracle_ctx* c, const sph_params* p)
{
    const int np = c->n_planes;
    PARFOR
    for (int64_t ii = 0; ii < (int64_t)c->n; ii++) {
        uint64_t i = (uint64_t)ii;
        const float x = c->pos[2 * i], y = c->pos[2 * i + 1];
        const float sr_i = c->h2[i] * 2.f;
        int cnt = 0;
        for (int k = 0; k < np; k++) {
            float d = sdf_probe(c, k, x, y) / sr_i;
            if (!(d < 1.f)) continue;
            const float eps = p->sdf_gradient_eps;
            const float inv_2eps = 1.f / (2.f * eps);
            float gx = (sdf_probe(c, k, x + eps, y) - sdf_probe(c, k, x - eps, y)) * inv_2eps;
            float gy = (sdf_probe(c, k, x, y + eps) - sdf_probe(c, k, x, y - eps)) * inv_2eps;
            float gn = sqrtf(orc_norm_sq(gx, gy));
            if (!(gn >= 0.00001f)) continue;
            gx /= gn;
            gy /= gn;
            float penalty, dpenalty;
            switch (p->boundary_penalty_term) {
            case SPH_PENALTY_NONE: penalty = 1.f; dpenalty = 0.f; break;
            case SPH_PENALTY_LINEAR: penalty = 1.f - d; dpenalty = -1.f; break;
            case SPH_PENALTY_QUADRATIC1:
                if (d > 0.f) { penalty = 1.f; dpenalty = 0.f; }
                else if (d > -1.f) { penalty = 0.5f * d * d + 1.f; dpenalty = d; }
                else { penalty = 0.5f - d; dpenalty = -1.f; }
                break;
            default: /* Quadratic2 */
                if (d > 0.f) { penalty = 1.f; dpenalty = 0.f; }
                else if (d > -0.5f) { penalty = d * d + 1.f; dpenalty = 2.f * d; }
                else { penalty = 0.75f - d; dpenalty = -1.f; }
                break;
            }
            float lambda, dlambda;
            if (d <= -1.f) { lambda = 1.f; dlambda = 0.f; }
            else { lambda = orc_lut_get(&c->lambda_lut, d); dlambda = orc_lut_get(&c->dlambda_lut, d); }
            float s = dpenalty * lambda + penalty * dlambda;
            size_t o = (size_t)i * ORC_MAX_PLANES + (size_t)cnt;
            c->lam[o] = lambda * penalty;
            c->lam_gx[o] = gx / sr_i * s;
            c->lam_gy[o] = gy / sr_i * s;
            cnt++;
        }
        c->lam_n[i] = (uint8_t)cnt;
    }
}

/* boundary_winchenbach2020.rs:155-162: iterator sum from 0.0 */
static inline float density_boundary_term(const oracle_ctx* c, uint64_t i)
{
    float s = 0.f;
    for (int k = 0; k < c->lam_n[i]; k++) s += c->lam[i * ORC_MAX_PLANES + k];
    return s;
}

/* boundary_winchenbach2020.rs:164-194; pressure passed as p_i */
static inline void boundary_pressure_accel(const oracle_ctx* c, const sph_params* p, uint64_t i, float p_i, float* ax, float* ay)
{
    float rx = 0.f, ry = 0.f;
    for (int k = 0; k < c->lam_n[i]; k++) {
        size_t o = i * ORC_MAX_PLANES + k;
        float p_ib = (p->operator_discretization == SPH_OP_SYMMETRIC_GRADIENT) ? p_i : 0.f;
        float rho_i = c->density[i];
        float rho_b = p->rest_density;
        float f = -rho_b * (p_i / (rho_i * rho_i) + p_ib / (rho_b * rho_b));
        rx += f * c->lam_gx[o];
        ry += f * c->lam_gy[o];
    }
    *ax = rx;
    *ay = ry;
}

/* boundary_winchenbach2020.rs:196-223; quantity_b = 0 at every call site */
static inline float boundary_divergence(const oracle_ctx* c, const sph_params* p, uint64_t i, float qix, float qiy)
{
    float r = 0.f;
    for (int k = 0; k < c->lam_n[i]; k++) {
        size_t o = i * ORC_MAX_PLANES + k;
        float rho_i = c->density[i];
        float rho_b = p->rest_density;
        float dx = 0.f - qix, dy = 0.f - qiy;
        float dot = dx * c->lam_gx[o] + dy * c->lam_gy[o];
        if (p->operator_discretization == SPH_OP_WINCHENBACH2020) r += dot;
        else r += rho_b / rho_i * dot;
    }
    return r;
}

/* boundary_winchenbach2020.rs:225-306 */
static float iisph_aii(const oracle_ctx* c, const sph_params* p, uint64_t i)
{
    const float* pos = c->pos;
    const float mi = c->mass[i], rho_i = c->density[i], rho_0 = p->rest_density;
    const float rho_i_sq = rho_i * rho_i;
    const float rho_b = rho_0;
    if (p->operator_discretization == SPH_OP_WINCHENBACH2020) {
        float ax = 0.f, ay = 0.f, bx = 0.f, by = 0.f, b2 = 0.f;
        NB_LOOP(c, i, j) {
            float hij = orc_hij(c->h2[i], c->h2[j]);
            float gx, gy;
            orc_kernel_derivh(pos[2 * i] - pos[2 * j], pos[2 * i + 1] - pos[2 * j + 1], hij, &gx, &gy);
            ax += c->mass[j] * gx;
            ay += c->mass[j] * gy;
            float mr = c->mass[j] / c->density[j];
            bx += mr * gx;
            by += mr * gy;
            b2 += mr * orc_norm_sq(gx, gy);
        }
        float sgx = 0.f, sgy = 0.f, sbx = 0.f, sby = 0.f;
        for (int k = 0; k < c->lam_n[i]; k++) {
            size_t o = i * ORC_MAX_PLANES + k;
            sgx += c->lam_gx[o];
            sgy += c->lam_gy[o];
            float f = rho_b * (1.f / (rho_i * rho_i) + 0.f / (rho_b * rho_b));
            sbx += f * c->lam_gx[o];
            sby += f * c->lam_gy[o];
        }
        float lx = ax / rho_i_sq + sbx, ly = ay / rho_i_sq + sby;
        float rx = bx + sgx, ry = by + sgy;
        return (lx * rx + ly * ry) + (mi * b2 / rho_i_sq);
    } else {
        const float rho_i_cu = rho_i * rho_i * rho_i;
        float ax = 0.f, ay = 0.f, a2 = 0.f;
        NB_LOOP(c, i, j) {
            float hij = orc_hij(c->h2[i], c->h2[j]);
            float gx, gy;
            orc_kernel_derivh(pos[2 * i] - pos[2 * j], pos[2 * i + 1] - pos[2 * j + 1], hij, &gx, &gy);
            ax += c->mass[j] * gx;
            ay += c->mass[j] * gy;
            a2 += c->mass[j] * orc_norm_sq(gx, gy);
        }
        float rgx = 0.f, rgy = 0.f, sbx = 0.f, sby = 0.f;
        const float coeff = (p->operator_discretization == SPH_OP_SIMPLE_GRADIENT) ? 0.f : 1.f;
        for (int k = 0; k < c->lam_n[i]; k++) {
            size_t o = i * ORC_MAX_PLANES + k;
            rgx += rho_b * c->lam_gx[o];
            rgy += rho_b * c->lam_gy[o];
            float f = rho_b * (1.f / (rho_i * rho_i) + coeff / (rho_b * rho_b));
            sbx += f * c->lam_gx[o];
            sby += f * c->lam_gy[o];
        }
        float lx = ax / rho_i_sq + sbx, ly = ay / rho_i_sq + sby;
        float rx = ax / rho_i + rgx / rho_i, ry = ay / rho_i + rgy / rho_i;
        return (lx * rx + ly * ry) + (mi * a2) / rho_i_cu;
    }
}

/* boundary_winchenbach2020.rs:320-325 */
static inline float distance_to_boundary(const oracle_ctx* c, uint64_t i)
{
    float m = INFINITY;
    for (int k = 0; k < c->n_planes; k++) m = fminf(m, sdf_probe(c, k, c->pos[2 * i], c->pos[2 * i + 1]));
    return m;
}

/* ------------------------------------------------------------------------------------------ */
/* per-particle operators                                                                      */
/* ------------------------------------------------------------------------------------------ */

/* simulation.rs:1007-1028 */
static inline float particle_density(const oracle_ctx* c, uint64_t i)
{
    const float* pos = c->pos;
    float acc = 0.f;
    NB_LOOP(c, i, j) {
        float hij = orc_hij(c->h2[i], c->h2[j]);
        float w = orc_kernelh(pos[2 * i] - pos[2 * j], pos[2 * i + 1] - pos[2 * j + 1], hij);
        acc += c->mass[j] * w;
    }
    acc += density_boundary_term(c, i);
    return acc;
}

/* simulation.rs:931-1005 */
static inline void non_pressure_accel(const oracle_ctx* c, const sph_params* p, uint64_t i, float* ox, float* oy)
{
    const float* pos = c->pos;
    const float* vel = c->vel;
    const float speed_of_sound = 88.f;
    float vx = 0.f, vy = 0.f;
    if (p->viscosity_type == SPH_VISC_WCSPH) {
        NB_LOOP(c, i, j) {
            float xx = pos[2 * i] - pos[2 * j], xy = pos[2 * i + 1] - pos[2 * j + 1];
            float ux = vel[2 * i] - vel[2 * j], uy = vel[2 * i + 1] - vel[2 * j + 1];
            float hij = orc_hij(c->h2[i], c->h2[j]);
            float gx, gy;
            orc_kernel_derivh(xx, xy, hij, &gx, &gy);
            float est = ux * xx + uy * xy;
            if (est < 0.f) {
                float viscous_term = 2.f * p->viscosity * hij * speed_of_sound / (c->density[i] + c->density[j]);
                float pi_ab = -viscous_term * est / (orc_norm_sq(xx, xy) + 0.001f * hij * hij);
                float f = -c->mass[j] * pi_ab;
                vx += f * gx;
                vy += f * gy;
            }
        }
    } else if (p->viscosity_type == SPH_VISC_APPROX_LAPLACE) {
        NB_LOOP(c, i, j) {
            float xx = pos[2 * i] - pos[2 * j], xy = pos[2 * i + 1] - pos[2 * j + 1];
            float ux = vel[2 * i] - vel[2 * j], uy = vel[2 * i + 1] - vel[2 * j + 1];
            float xv = xx * ux + xy * uy;
            if (xv >= 0.f) continue;
            float hij = orc_hij(c->h2[i], c->h2[j]);
            float gx, gy;
            orc_kernel_derivh(xx, xy, hij, &gx, &gy);
            float rho_ij = (c->density[i] + c->density[j]) * 0.5f;
            float coeff = 2.f * 4.f * (c->mass[j] / rho_ij) * xv / (orc_norm_sq(xx, xy) + 0.01f * hij * hij);
            float f = p->viscosity * coeff;
            vx += f * gx;
            vy += f * gy;
        }
    }
    float px = 0.f, py = 0.f;
    if (p->has_pull_fluid_to) {
        float tx = p->pull_fluid_to[0] - pos[2 * i], ty = p->pull_fluid_to[1] - pos[2 * i + 1];
        float nn = sqrtf(orc_norm_sq(tx, ty));
        px = tx / nn * 13.f;
        py = ty / nn * 13.f;
    }
    /* gravity_vector (simulation_parameters.rs:133-146) = (0, gravity) */
    *ox = (vx + 0.f) + px;
    *oy = (vy + p->gravity) + py;
}

/* simulation.rs:1780-1808 + 1751-1777; pressure[] given, total = fluid + boundary */
static inline void pressure_accel_of(const oracle_ctx* c, const sph_params* p, const float* pressure, uint64_t i, float* ox, float* oy)
{
    const float* pos = c->pos;
    const float p1 = pressure[i] / (c->density[i] * c->density[i]);
    float ax = 0.f, ay = 0.f;
    NB_LOOP(c, i, j) {
        float hij = orc_hij(c->h2[i], c->h2[j]);
        float gx, gy;
        orc_kernel_derivh(pos[2 * i] - pos[2 * j], pos[2 * i + 1] - pos[2 * j + 1], hij, &gx, &gy);
        float p2 = pressure[j] / (c->density[j] * c->density[j]);
        float f = -c->mass[j] * (p1 + p2);
        ax += f * gx;
        ay += f * gy;
    }
    float bx, by;
    boundary_pressure_accel(c, p, i, pressure[i], &bx, &by);
    *ox = ax + bx;
    *oy = ay + by;
}

/* simulation.rs:1552-1592; Q is an interleaved VF<2> array */
static inline float divergence_iisph(const oracle_ctx* c, const sph_params* p, const float* Q, uint64_t i)
{
    const float* pos = c->pos;
    const float qix = Q[2 * i], qiy = Q[2 * i + 1];
    float sum = 0.f;
    NB_LOOP(c, i, j) {
        float hij = orc_hij(c->h2[i], c->h2[j]);
        float gx, gy;
        orc_kernel_derivh(pos[2 * i] - pos[2 * j], pos[2 * i + 1] - pos[2 * j + 1], hij, &gx, &gy);
        float dqx = Q[2 * j] - qix, dqy = Q[2 * j + 1] - qiy;
        float dot = dqx * gx + dqy * gy;
        if (p->operator_discretization == SPH_OP_WINCHENBACH2020) sum += c->mass[j] / c->density[j] * dot;
        else sum += c->mass[j] / c->density[i] * dot;
    }
    return sum + boundary_divergence(c, p, i, qix, qiy);
}

static inline float next_density_estimate(const oracle_ctx* c, const sph_params* p, uint64_t i)
{
    return p->operator_discretization == SPH_OP_WINCHENBACH2020 ? p->rest_density : c->density[i];
}

/* ------------------------------------------------------------------------------------------ */
/* sweeps                                                                                      */
/* ------------------------------------------------------------------------------------------ */

/* simulation.rs:1030-1049 */
static int all_densities(oracle_ctx* c)
{
    int bad_finite = 0, bad_small = 0;
#pragma omp parallel for schedule(static) reduction(| : bad_finite, bad_small)
    for (int64_t ii = 0; ii < (int64_t)c->n; ii++) {
        float d = particle_density(c, (uint64_t)ii);
        c->density[ii] = d;
        if (!isfinite(d)) bad_finite = 1;
        else if (!(d > 0.0001f)) bad_small = 1;
    }
    if (bad_finite) return orc_fail(c, SPH_ERR_DENSITY_NOT_FINITE, "assertion failed: p_density.is_finite()");
    if (bad_small) return orc_fail(c, SPH_ERR_DENSITY_TOO_SMALL, "assertion failed: *p_density > 0.0001");
    return SPH_OK;
}

/* simulation.rs:2235-2248 */
static void constant_field(oracle_ctx* c, const sph_params* p)
{
    PARFOR
    for (int64_t ii = 0; ii < (int64_t)c->n; ii++) {
        uint64_t i = (uint64_t)ii;
        float acc = 0.f;
        NB_LOOP(c, i, j) {
            float hij = orc_hij(c->h2[i], c->h2[j]);
            float w = orc_kernelh(c->pos[2 * i] - c->pos[2 * j], c->pos[2 * i + 1] - c->pos[2 * j + 1], hij);
            acc += c->mass[j] / c->density[j] * w;
        }
        acc += density_boundary_term(c, i) / p->rest_density;
        c->constant_field[i] = acc;
    }
}

/* calculate_aii_inefficiently (simulation.rs:1324-1345, 1594-1631): apply the operator to e_i */
static float aii_inefficient(const oracle_ctx* c, const sph_params* p, uint64_t i, float* scratch_p, float* scratch_a)
{
    /* pressure = e_i; pressure accel needed at i and its neighbours only */
    scratch_p[i] = 1.f;
    NB_LOOP(c, i, j) { pressure_accel_of(c, p, scratch_p, j, &scratch_a[2 * j], &scratch_a[2 * j + 1]); }
    float r = divergence_iisph(c, p, scratch_a, i);
    scratch_p[i] = 0.f;
    return r;
}

/* simulation.rs:1080-1125 */
static int compute_aii(oracle_ctx* c, const sph_params* p)
{
    int bad = 0;
#pragma omp parallel for schedule(static) reduction(| : bad)
    for (int64_t ii = 0; ii < (int64_t)c->n; ii++) {
        float a = iisph_aii(c, p, (uint64_t)ii);
        c->aii[ii] = a;
        if (!isfinite(a)) bad = 1;
    }
    if (bad) return orc_fail(c, SPH_ERR_AII_NOT_FINITE, "assertion failed: (*p_aii).is_finite()");
    if (p->check_aii) {
        /* check_aii (simulation.rs:1347-1375), tolerance 0.01 in f32 */
        float* sp = (float*)calloc(c->n, sizeof(float));
        float* sa = (float*)calloc(2 * c->n, sizeof(float));
        int mismatch = 0;
        for (uint64_t i = 0; i < c->n; i++) { /* serial: scratch arrays are shared */
            float real = aii_inefficient(c, p, i, sp, sa);
            float a = c->aii[i];
            if (!(real <= a + 0.01f && real >= a - 0.01f)) mismatch = 1;
        }
        free(sp);
        free(sa);
        if (mismatch) return orc_fail(c, SPH_ERR_CHECK_AII, "a_ii value not equal with a tolerance of 0.01");
    }
    return SPH_OK;
}

/* simulation.rs:1051-1077 */
static int update_velocity_with_non_pressure_accel(oracle_ctx* c, const sph_params* p, float dt)
{
    int bad = 0;
#pragma omp parallel for schedule(static) reduction(| : bad)
    for (int64_t ii = 0; ii < (int64_t)c->n; ii++) {
        uint64_t i = (uint64_t)ii;
        float ax, ay;
        non_pressure_accel(c, p, i, &ax, &ay);
        if (!isfinite(ax) || !isfinite(ay)) bad = 1;
        c->vel_tmp[2 * i] = c->vel[2 * i] + dt * ax;
        c->vel_tmp[2 * i + 1] = c->vel[2 * i + 1] + dt * ay;
    }
    float* t = c->vel;
    c->vel = c->vel_tmp;
    c->vel_tmp = t;
    if (bad) return orc_fail(c, SPH_ERR_VISCOSITY_NOT_FINITE, "Assertion 'viscosity_accel[d].is_finite()' failed!");
    return SPH_OK;
}

enum { SRC_DIVERGENCE, SRC_FULL, SRC_ONLY_DENSITY, SRC_FULL_WITH_OMEGA };

/* IISPH2's per-particle omega (simulation.rs:2263-2311; serial in the reference, independent per particle) */
static inline float iisph2_dwdh(float d, float H)
{
    float q = d / H;
    float cd = 40.f / (7.f * ORC_PI);
    float w = orc_cubic_unnormalized(q);
    float wd = orc_cubic_unnormalized_deriv(q);
    return cd * -(2.f) / (H * H * H) * w + cd / (H * H) * wd * (-d / (H * H));
}

static void iisph2_omega(oracle_ctx* c)
{
    PARFOR
    for (int64_t ii = 0; ii < (int64_t)c->n; ii++) {
        uint64_t i = (uint64_t)ii;
        float omega = 1.f;
        if (c->size_class[i] == 3 /* ParticleSizeClass::Large (adaptivity/mod.rs:12-23) */) {
            float h_ii = c->h2[i];
            float H_i = c->h2[i] * 2.f, H_ii = h_ii * 2.f;
            omega += H_i / (3.f * c->density[i]) * c->mass[i] * iisph2_dwdh(0.f, H_ii);
        } else {
            NB_LOOP(c, i, j) {
                float dx = c->pos[2 * i] - c->pos[2 * j], dy = c->pos[2 * i + 1] - c->pos[2 * j + 1];
                float h_ij = orc_hij(c->h2[i], c->h2[j]);
                float H_i = c->h2[i] * 2.f, H_ij = h_ij * 2.f;
                float d = sqrtf(orc_norm_sq(dx, dy));
                omega += H_i / (3.f * c->density[i]) * c->mass[j] * iisph2_dwdh(d, H_ij);
            }
        }
        omega = fminf(2.5f, fmaxf(omega, 0.125f));
        c->omega[i] = omega;
    }
}

/* prepare_ppe_divergence / prepare_full_ppe / prepare_only_density_part_ppe
 * (simulation.rs:1127-1204) with the source terms of :1633-1676, :1712-1748 */
static void prepare_ppe(oracle_ctx* c, const sph_params* p, float dt, int kind)
{
    PARFOR
    for (int64_t ii = 0; ii < (int64_t)c->n; ii++) {
        uint64_t i = (uint64_t)ii;
        c->pressure[i] = 0.f;
        float s;
        if (kind == SRC_DIVERGENCE) {
            float vdiv = divergence_iisph(c, p, c->vel, i);
            s = -vdiv / dt;
        } else if (kind == SRC_ONLY_DENSITY) {
            s = -(p->rest_density - c->density[i]) / (next_density_estimate(c, p, i) * dt * dt);
        } else if (kind == SRC_FULL_WITH_OMEGA) { /* calculate_source_term_full_with_omega, simulation.rs:1678-1710 */
            float vdiv = divergence_iisph(c, p, c->vel, i);
            s = -(p->rest_density - c->density[i]) / (p->rest_density * dt * dt) - vdiv / (dt * c->omega[i]);
        } else {
            float vdiv = divergence_iisph(c, p, c->vel, i);
            s = -(p->rest_density - c->density[i]) / (next_density_estimate(c, p, i) * dt * dt) - vdiv / dt;
        }
        c->source[i] = s;
    }
}

/* simulation.rs:1518-1543 */
static void all_pressure_accels(oracle_ctx* c, const sph_params* p)
{
    PARFOR
    for (int64_t ii = 0; ii < (int64_t)c->n; ii++) {
        uint64_t i = (uint64_t)ii;
        pressure_accel_of(c, p, c->pressure, i, &c->pacc[2 * i], &c->pacc[2 * i + 1]);
    }
}

typedef struct {
    uint64_t normal, singular, negative;
    float sum_err, max_err;
} solver_acc;

enum { RES_DENSITY, RES_DIVERGENCE };

/* iisph_single_pressure_iteration (simulation.rs:1207-1322) */
static int single_pressure_iteration(oracle_ctx* c, const sph_params* p, float dt, int residual, int clamp, solver_acc* out)
{
    const float w = p->jacobi_omega;
    all_pressure_accels(c, p);

    const uint64_t n = c->n;
    const uint64_t CH = 1024;
    const uint64_t nch = (n + CH - 1) / CH;
    solver_acc* part = (solver_acc*)calloc(nch ? nch : 1, sizeof(solver_acc));
    int bad_ap = 0, bad_p = 0;
#pragma omp parallel for schedule(static) reduction(| : bad_ap, bad_p)
    for (int64_t cc = 0; cc < (int64_t)nch; cc++) {
        solver_acc a = {0, 0, 0, 0.f, 0.f};
        uint64_t b = (uint64_t)cc * CH, e = b + CH < n ? b + CH : n;
        for (uint64_t i = b; i < e; i++) {
            if (fabsf(c->aii[i]) < 10e-4f) {
                c->pressure_next[i] = 0.f;
                a.singular++;
                continue;
            }
            float a_p = divergence_iisph(c, p, c->pacc, i);
            float s = c->source[i];
            if (!isfinite(a_p)) bad_ap = 1;
            float pn = c->pressure[i] + w * (s - a_p) / c->aii[i];
            if (!isfinite(pn)) bad_p = 1;
            float err;
            if (residual == RES_DENSITY) {
                err = c->density[i] * dt * dt * (s - a_p);
                c->density_error[i] = err;
            } else {
                err = dt * (s - a_p);
            }
            if (pn <= 0.f && clamp) {
                c->pressure_next[i] = 0.f;
                a.negative++;
            } else {
                c->pressure_next[i] = pn;
                a.normal++;
                a.sum_err += err;
                a.max_err = fmaxf(a.max_err, fabsf(err));
            }
        }
        part[cc] = a;
    }
    solver_acc t = {0, 0, 0, 0.f, 0.f};
    for (uint64_t cc = 0; cc < nch; cc++) {
        t.normal += part[cc].normal;
        t.singular += part[cc].singular;
        t.negative += part[cc].negative;
        t.sum_err += part[cc].sum_err;
        t.max_err = fmaxf(t.max_err, part[cc].max_err);
    }
    free(part);
    *out = t;
    if (bad_ap) return orc_fail(c, SPH_ERR_AP_NOT_FINITE, "'!a_p.is_finite()' failed. Pressure values probably have exploded!");
    if (bad_p) return orc_fail(c, SPH_ERR_PRESSURE_NOT_FINITE, "'!p_pressure_next_iter.is_finite()' failed.");
    return SPH_OK;
}

/* iisph_pressure_iterations (simulation.rs:1377-1516) */
static int pressure_iterations(oracle_ctx* c, const sph_params* p, float dt, float max_avg_error, int residual, int clamp,
                               sph_solver_stats* st)
{
    for (uint64_t i = 0; i < c->n; i++)
        if (c->aii[i] < 0.f) return orc_fail(c, SPH_ERR_AII_NEGATIVE, "AII should not be negative! i=%llu aii[i]=%g",
                                             (unsigned long long)i, (double)c->aii[i]);
    uint32_t iters = 0;
    solver_acc a;
    for (;;) {
        int rc = single_pressure_iteration(c, p, dt, residual, clamp, &a);
        if (rc) return rc;
        float* t = c->pressure;
        c->pressure = c->pressure_next;
        c->pressure_next = t;

        float avg = a.normal > 0 ? a.sum_err / (float)a.normal : NAN;
        if (residual == RES_DENSITY) {
            if (a.normal == 0 || (fabsf(avg / p->rest_density) < max_avg_error && iters > 1)) break;
        } else {
            if (a.normal == 0 || (fabsf(avg) < max_avg_error / dt && iters > 1)) break;
        }
        if (iters == p->max_iters) break; /* "Pressure sover not converged" -> break true */
        iters++;
    }
    all_pressure_accels(c, p);
    st->iters = iters;
    st->converged = 1;
    st->normal_count = (uint32_t)a.normal;
    st->singular_count = (uint32_t)a.singular;
    st->negative_count = (uint32_t)a.negative;
    st->avg_error = a.normal > 0 ? a.sum_err / (float)a.normal : NAN;
    st->max_error = a.max_err;
    return SPH_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* level estimation (simulation.rs:539-927)                                                    */
/* ------------------------------------------------------------------------------------------ */

static inline int lvl_is_surface(float v) { return !isnan(v); }

/* simulation.rs:697-723 */
static inline int in_level_estimation_range(const oracle_ctx* c, const sph_params* p, float particle_radius, uint64_t i, uint64_t j)
{
    if (p->support_length_estimation == SPH_H_FROM_DISTRIBUTION || p->support_length_estimation == SPH_H_FROM_DISTRIBUTION2) {
        float dx = c->pos[2 * j] - c->pos[2 * i], dy = c->pos[2 * j + 1] - c->pos[2 * i + 1];
        float r = particle_radius * p->maximum_range;
        if (orc_norm_sq(dx, dy) > r * r) return 0;
    }
    return 1;
}

/* simulation.rs:539-625 */
static void surface_detection_by_empty_angle(oracle_ctx* c, const sph_params* p)
{
    const float threshold = cosf(50.f * (ORC_PI / 180.f));
    PARFOR
    for (int64_t ii = 0; ii < (int64_t)c->n; ii++) {
        uint64_t i = (uint64_t)ii;
        const float* pos = c->pos;
        float particle_radius = orc_sphere_volume_to_radius(c->mass[i] / p->rest_density);
        float nx = 0.f, ny = 0.f;
        NB_LOOP(c, i, j) {
            float hij = orc_hij(c->h2[i], c->h2[j]);
            float gx, gy;
            orc_kernel_derivh(pos[2 * i] - pos[2 * j], pos[2 * i + 1] - pos[2 * j + 1], hij, &gx, &gy);
            float f = c->mass[i] / p->rest_density;
            nx -= f * gx;
            ny -= f * gy;
        }
        int interior;
        c->flag_insufficient[i] = 0;
        uint64_t cnt = c->nb_off[i + 1] - c->nb_off[i];
        if (cnt < 2 * 2 - 1) {
            interior = 0;
            c->flag_insufficient[i] = 1;
        } else if (orc_norm_sq(nx, ny) < 0.00001f) {
            interior = 1;
        } else if (!p->boundary_is_fluid_surface && distance_to_boundary(c, i) < c->h2[i] * 1.5f) {
            interior = 1;
        } else {
            interior = 0;
            float nn = sqrtf(orc_norm_sq(nx, ny));
            nx /= nn;
            ny /= nn;
            NB_LOOP(c, i, j) {
                if (!in_level_estimation_range(c, p, particle_radius, i, j)) continue;
                float dx = pos[2 * j] - pos[2 * i], dy = pos[2 * j + 1] - pos[2 * i + 1];
                float dn = sqrtf(orc_norm_sq(dx, dy)) + 0.000001f;
                dx /= dn;
                dy /= dn;
                if (dx * nx + dy * ny > threshold) {
                    interior = 1;
                    break;
                }
            }
        }
        if (interior) {
            c->level[i] = NAN;
            c->flag_surface[i] = 0;
        } else {
            c->level[i] = 0.0f;
            c->flag_surface[i] = 1;
        }
    }
}

/* simulation.rs:631-695 */
static void surface_detection_by_center_diff(oracle_ctx* c, const sph_params* p)
{
    PARFOR
    for (int64_t ii = 0; ii < (int64_t)c->n; ii++) {
        uint64_t i = (uint64_t)ii;
        const float* pos = c->pos;
        float wsum = 0.f, cxs = 0.f, cys = 0.f, rad = 0.f;
        int num = 0;
        NB_LOOP(c, i, j) {
            float vol = c->mass[j] / p->rest_density;
            float r = orc_sphere_volume_to_radius(vol);
            float hij = orc_hij(c->h2[i], c->h2[j]);
            float w = orc_kernelh(pos[2 * i] - pos[2 * j], pos[2 * i + 1] - pos[2 * j + 1], hij) * vol;
            cxs += pos[2 * j] * w;
            cys += pos[2 * j + 1] * w;
            rad += r * w;
            wsum += w;
            num++;
        }
        rad /= wsum;
        float surface_level = -0.85f * rad;
        float phi;
        if (num < 5) phi = surface_level;
        else {
            cxs /= wsum;
            cys /= wsum;
            phi = sqrtf(orc_norm_sq(pos[2 * i] - cxs, pos[2 * i + 1] - cys)) - rad;
        }
        if (phi >= surface_level) { c->level[i] = phi; c->flag_surface[i] = 1; }
        else { c->level[i] = NAN; c->flag_surface[i] = 0; }
    }
}

static void fill_stash_from_level(oracle_ctx* c, const sph_params* p)
{
    for (uint64_t i = 0; i < c->n; i++) c->stash[i] = lvl_is_surface(c->level[i]) ? c->level[i] : -p->maximum_surface_distance;
}

/* simulation.rs:729-801 */
static void propagate_level_set(oracle_ctx* c, const sph_params* p)
{
    memcpy(c->level_tmp, c->level, c->n * sizeof(float));
    int num_iter = 0;
    int changed = 1;
    while (changed) {
        changed = 0;
#pragma omp parallel for schedule(static) reduction(| : changed)
        for (int64_t ii = 0; ii < (int64_t)c->n; ii++) {
            uint64_t i = (uint64_t)ii;
            if (lvl_is_surface(c->level[i])) { c->level_tmp[i] = c->level[i]; continue; }
            float particle_radius = orc_sphere_volume_to_radius(c->mass[i] / p->rest_density);
            int have = 0;
            float best = 0.f;
            NB_LOOP(c, i, j) {
                float lj = c->level[j];
                if (!lvl_is_surface(lj)) continue;
                if (!in_level_estimation_range(c, p, particle_radius, i, j)) continue;
                float dx = c->pos[2 * j] - c->pos[2 * i], dy = c->pos[2 * j + 1] - c->pos[2 * i + 1];
                float est = lj - sqrtf(orc_norm_sq(dx, dy));
                if (have) best = fmaxf(best, est);
                else { best = est; have = 1; }
            }
            if (have) { c->level_tmp[i] = best; changed = 1; }
            else c->level_tmp[i] = NAN;
        }
        float* t = c->level;
        c->level = c->level_tmp;
        c->level_tmp = t;
        num_iter++;
        if (p->fill_stash_with == SPH_STASH_SURFACE_DISTANCE_MIDDLE && num_iter == 1) fill_stash_from_level(c, p);
    }
}

/* simulation.rs:862-927 */
static void perform_level_estimation(oracle_ctx* c, const sph_params* p)
{
    switch (p->level_estimation_method) {
    case SPH_LEVEL_NONE: return;
    case SPH_LEVEL_CENTER_DIFF: surface_detection_by_center_diff(c, p); break;
    default: surface_detection_by_empty_angle(c, p); break;
    }
    if (p->fill_stash_with == SPH_STASH_SURFACE_DISTANCE_FIRST) fill_stash_from_level(c, p);
    propagate_level_set(c, p);
}

/* simulation.rs:803-857 */
static int smooth_level_estimation_field(oracle_ctx* c, const sph_params* p)
{
    if (p->level_estimation_method == SPH_LEVEL_NONE) return SPH_OK;
    int bad = 0;
#pragma omp parallel for schedule(static) reduction(| : bad)
    for (int64_t ii = 0; ii < (int64_t)c->n; ii++) {
        uint64_t i = (uint64_t)ii;
        float level = 0.f, weight = 0.f;
        NB_LOOP(c, i, j) {
            float hij = orc_hij(c->h2[i], c->h2[j]);
            float w = orc_kernelh(c->pos[2 * i] - c->pos[2 * j], c->pos[2 * i + 1] - c->pos[2 * j + 1], hij);
            float lj = c->level[j];
            float dist = lvl_is_surface(lj) ? fmaxf(lj, -p->maximum_surface_distance) : -p->maximum_surface_distance;
            if (!(isfinite(c->density[j]) && c->density[j] > 0.f)) bad = 1;
            level += dist * c->mass[j] / c->density[j] * w;
            weight += c->mass[j] / c->density[j] * w;
        }
        if (!isfinite(weight) || weight <= 0.f) { bad = 1; continue; }
        level /= weight;
        if (!isfinite(level)) bad = 1;
        c->level_old[i] = level;
        c->level_tmp[i] = level;
    }
    float* t = c->level;
    c->level = c->level_tmp;
    c->level_tmp = t;
    if (bad) return orc_fail(c, SPH_ERR_LEVEL_WEIGHT, "weight is <=0 in smooth_level_estimation_field");
    return SPH_OK;
}

/* LevelEstimationState::target_mass + classify_particle (simulation.rs:213-237, adaptivity/mod.rs:32-59) */
int orc_classify_particles(oracle_ctx* c, const sph_params* p)
{
    for (uint64_t i = 0; i < c->n; i++) {
        if (!lvl_is_surface(c->level[i])) /* LevelEstimationState::level() of FluidInterior (simulation.rs:205-211) */
            return orc_fail(c, SPH_ERR_INVALID_ARGUMENT, "internal error: entered unreachable code (LevelEstimationState::level of FluidInterior), particle i=%llu",
                            (unsigned long long)i);
        float level = fmaxf(c->level[i], -p->maximum_surface_distance);
        float interp = level / -p->maximum_surface_distance;
        float target;
        float mass_fine = orc_radius_to_sphere_volume(p->particle_radius_fine) * p->rest_density;
        float mass_base = orc_radius_to_sphere_volume(p->particle_radius_base) * p->rest_density;
        if (p->sizing_function == SPH_SIZING_MASS) target = mass_fine * (1.f - interp) + mass_base * interp;
        else if (p->sizing_function == SPH_SIZING_RADIUS) {
            float r = p->particle_radius_fine * (1.f - interp) + p->particle_radius_base * interp;
            target = orc_radius_to_sphere_volume(r) * p->rest_density;
        } else {
            float e = 1.f / 2.f;
            float r = p->particle_radius_fine * (1.f - powf(interp, e)) + p->particle_radius_base * powf(interp, e);
            target = orc_radius_to_sphere_volume(r) * p->rest_density;
        }
        float mrel = c->mass[i] / target;
        uint8_t cls;
        if (mrel <= 0.5f) cls = 0;
        else if (mrel <= 1.f / 1.1f) cls = 1;
        else if (mrel < 1.1f) cls = 2;
        else if (mrel < 2.0f) cls = 3;
        else cls = 4;
        c->size_class[i] = cls;
    }
    return SPH_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* the step (simulation.rs:1980-2730)                                                          */
/* ------------------------------------------------------------------------------------------ */

static int integrate_v_then_x(oracle_ctx* c, float dt) /* simulation.rs:2433-2445, 2486-2499 */
{
    int bad = 0;
#pragma omp parallel for schedule(static) reduction(| : bad)
    for (int64_t ii = 0; ii < (int64_t)c->n; ii++) {
        uint64_t i = (uint64_t)ii;
        for (int d = 0; d < 2; d++) {
            c->vel[2 * i + d] += dt * c->pacc[2 * i + d];
            c->pos[2 * i + d] += dt * c->vel[2 * i + d];
            if (!isfinite(c->vel[2 * i + d])) bad = 1;
        }
    }
    if (bad) return orc_fail(c, SPH_ERR_VELOCITY_NOT_FINITE, "Assertion 'p_velocity[d].is_finite()' failed!");
    return SPH_OK;
}

/* estimate_h_next_from_distribution / _distribution2 (simulation.rs:1873-1971).  lambda_sum(i) is the boundary handler's
 * state of the PREVIOUS step (update_after_advect runs later in the step; empty lists = 0 before the first step). */
static int estimate_h_next_from_distribution(oracle_ctx* c, const sph_params* p)
{
    int bad = 0;
    const int mode = p->support_length_estimation;
#pragma omp parallel for schedule(static) reduction(| : bad)
    for (int64_t ii = 0; ii < (int64_t)c->n; ii++) {
        uint64_t i = (uint64_t)ii;
        const float* pos = c->pos;
        const float w = 0.5f;
        float volume_estimate;
        const float boundary_volume = density_boundary_term(c, i);   /* == lambda_sum(i), boundary_winchenbach2020.rs:47-49 */
        if (mode == SPH_H_FROM_DISTRIBUTION2) {
            float v_w_sum = 0.f;
            NB_LOOP(c, i, j) {
                float hij = orc_hij(c->h2[i], c->h2[j]);
                float vj = c->mass[j] / p->rest_density;
                v_w_sum += vj * orc_kernelh(pos[2 * i] - pos[2 * j], pos[2 * i + 1] - pos[2 * j + 1], hij);
            }
            float vi = c->mass[i] / p->rest_density;
            volume_estimate = vi / (v_w_sum + boundary_volume);
        } else {
            float w_sum = 0.f;
            NB_LOOP(c, i, j) {
                float hij = orc_hij(c->h2[i], c->h2[j]);
                w_sum += orc_kernelh(pos[2 * i] - pos[2 * j], pos[2 * i + 1] - pos[2 * j + 1], hij);
            }
            volume_estimate = (1.f - fminf(boundary_volume, 0.5f)) / w_sum;
        }
        if (!(volume_estimate >= 0.f)) { bad = 1; continue; }
        float h_new = ORC_ETA * orc_sphere_volume_to_radius(volume_estimate);
        float h_old = c->h2[i];
        float hn = w * h_new + (1.f - w) * h_old;
        if (mode == SPH_H_FROM_DISTRIBUTION_CLAMPED1) hn = fminf(hn, 1.f * orc_h_from_mass(c->mass[i], p->rest_density));
        if (mode == SPH_H_FROM_DISTRIBUTION_CLAMPED2) hn = fminf(hn, 2.f * orc_h_from_mass(c->mass[i], p->rest_density));
        c->h2_next[i] = hn;
    }
    if (bad) return orc_fail(c, SPH_ERR_VOLUME_ESTIMATE, "assertion failed: volume_estimate >= 0.");
    return SPH_OK;
}

/* constrain_neighborhood_count (simulation.rs:2145-2177): a particle with more than optimal_neighbor_number() + 5 neighbours
 * (simulation.rs:386-388: (ETA * 2)^2 = 14.44 -> 14 as usize, + 5 = 19; the particle itself is on its list) takes the
 * (count - target)-th largest of the "fringe" values 2 |x_ij| - 2 h_j as its smoothing length; everybody else keeps h2.
 * The values land in h2_next, then mem::swap(h2, h2_next).  The lists are NOT rebuilt (the TODO at :2174-2175). */
static int cmp_desc_f32(const void* a, const void* b)
{
    float x = *(const float*)a, y = *(const float*)b;
    return x > y ? -1 : (x < y ? 1 : 0);
}
static int constrain_neighborhood_count(oracle_ctx* c, const sph_params* p)
{
    const float onn = (ORC_ETA * 2.f) * (ORC_ETA * 2.f);   /* powi(2) */
    const uint64_t target = (uint64_t)onn + 5u;
    int bad_small = 0, bad_neg = 0;
#pragma omp parallel for schedule(dynamic, 256) reduction(| : bad_small, bad_neg)
    for (int64_t ii = 0; ii < (int64_t)c->n; ii++) {
        uint64_t i = (uint64_t)ii;
        const uint64_t cnt = c->nb_off[i + 1] - c->nb_off[i];
        if (cnt > target) {
            float* fringe = (float*)malloc(cnt * sizeof(float));
            uint64_t k = 0;
            NB_LOOP(c, i, j) {
                float dx = c->pos[2 * i] - c->pos[2 * j], dy = c->pos[2 * i + 1] - c->pos[2 * j + 1];
                float x_ij = sqrtf(dx * dx + dy * dy);
                float srj = c->h2[j] * 2.f;
                fringe[k++] = 2.f * x_ij - srj;
            }
            qsort(fringe, cnt, sizeof(float), cmp_desc_f32);
            float hn = fringe[cnt - target];
            free(fringe);
            c->h2_next[i] = hn;
            if (!(hn < c->h2[i])) bad_small = 1;
            c->flag_reduced[i] = 1;
            if (!(hn >= 0.f)) bad_neg = 1;
        } else {
            c->h2_next[i] = c->h2[i];
            c->flag_reduced[i] = 0;
        }
    }
    /* per particle the `<` assertion comes first; across particles the reference's order is the thread pool's: report the
     * smaller code when both kinds occurred */
    if (bad_small) return orc_fail(c, SPH_ERR_CONSTRAIN_NOT_SMALLER, "assertion failed: *p_h_next < smoothing_length_single(&particles.h2, i, simulation_params)");
    if (bad_neg) return orc_fail(c, SPH_ERR_CONSTRAIN_NEGATIVE, "assertion failed: *p_h_next >= 0.");
    float* t = c->h2;
    c->h2 = c->h2_next;
    c->h2_next = t;
    return SPH_OK;
}

int orc_step(oracle_ctx* c, const sph_params* p, sph_step_stats* out)
{
    double t_step0 = omp_get_wtime();
    double ms_neigh = 0, ms_level = 0, ms_div = 0, ms_dens = 0;
    int rc;
    sph_step_stats st;
    memset(&st, 0, sizeof st);
    st.n_particles = c->n;

    if (c->n_planes == 0) return orc_fail(c, SPH_ERR_NO_BOUNDARY, "not implemented: NoBoundaryHandler::iisph_aii");

    /* simulation.rs:1998-2016, 1865-1871 */
    if (p->support_length_estimation == SPH_H_FROM_MASS) {
        PARFOR
        for (int64_t i = 0; i < (int64_t)c->n; i++) c->h2[i] = orc_h_from_mass(c->mass[i], p->rest_density);
    } else {
        /* only apply the support length that was estimated in the last step: mem::swap(h2, h2_next) */
        float* t = c->h2;
        c->h2 = c->h2_next;
        c->h2_next = t;
    }

    orc_cell_indices(c);

    /* simulation.rs:2018-2070 */
    double t0 = omp_get_wtime();
    if (!p->level_estimation_after_advection) {
        if (!p->use_extended_range_for_level_estimation)
            return orc_fail(c, SPH_ERR_INVALID_ARGUMENT, "assertion failed: simulation_params.use_extended_range_for_level_estimation");
        if (p->level_estimation_method == SPH_LEVEL_CENTER_DIFF)
            return orc_fail(c, SPH_ERR_INVALID_ARGUMENT, "center diff level estimation method needs density values");
        if ((rc = orc_build_neighbors(c, p->level_estimation_range / ORC_ETA))) return rc;
        ms_neigh += (omp_get_wtime() - t0) * 1e3;
        t0 = omp_get_wtime();
        perform_level_estimation(c, p);
        ms_level += (omp_get_wtime() - t0) * 1e3;
        t0 = omp_get_wtime();
        orc_filter_down(c, 2.f);
        ms_neigh += (omp_get_wtime() - t0) * 1e3;
    } else {
        if ((rc = orc_build_neighbors(c, 2.f))) return rc;
        ms_neigh += (omp_get_wtime() - t0) * 1e3;
    }

    /* simulation.rs:2072-2088 */
    for (uint64_t i = 0; i < c->n; i++) c->neighbor_count[i] = (uint32_t)(c->nb_off[i + 1] - c->nb_off[i]);
    if (p->check_neighborhood && (rc = orc_check_neighborhood(c))) return rc;

    /* simulation.rs:2090-2143 */
    if (p->support_length_estimation != SPH_H_FROM_MASS && (rc = estimate_h_next_from_distribution(c, p))) return rc;

    /* simulation.rs:2145-2177 */
    if (p->constrain_neighborhood_count && (rc = constrain_neighborhood_count(c, p))) return rc;

    /* simulation.rs:2179-2180 */
    update_after_advect(c, p);

    /* simulation.rs:2182-2191 */
    float min_cfl = INFINITY;
    for (uint64_t i = 0; i < c->n; i++) {
        float sr = c->h2[i] * 2.f;
        float v = sr * sr / (orc_norm_sq(c->vel[2 * i], c->vel[2 * i + 1]) + 0.01f);
        if (v < min_cfl) min_cfl = v;
    }
    float cfl_dt = p->cfl_factor * sqrtf(min_cfl);
    float dt = fminf(p->max_dt, cfl_dt);
    st.dt = dt;

    /* simulation.rs:2204 */
    if ((rc = all_densities(c))) return rc;
    /* simulation.rs:2235-2248 */
    constant_field(c, p);
    /* simulation.rs:2250-2259 */
    if ((rc = compute_aii(c, p))) return rc;

    switch (p->pressure_solver_method) {
    case SPH_SOLVER_IISPH: /* simulation.rs:2389-2446 */
        if ((rc = update_velocity_with_non_pressure_accel(c, p, dt))) return rc;
        t0 = omp_get_wtime();
        prepare_ppe(c, p, dt, SRC_FULL);
        if ((rc = pressure_iterations(c, p, dt, p->iisph_max_avg_density_error, RES_DENSITY, 1, &st.density_solver))) return rc;
        ms_dens += (omp_get_wtime() - t0) * 1e3;
        if ((rc = integrate_v_then_x(c, dt))) return rc;
        break;
    case SPH_SOLVER_IISPH2: /* simulation.rs:2262-2387 */
        iisph2_omega(c);
        if ((rc = update_velocity_with_non_pressure_accel(c, p, dt))) return rc;
        t0 = omp_get_wtime();
        prepare_ppe(c, p, dt, SRC_FULL_WITH_OMEGA);
        if ((rc = pressure_iterations(c, p, dt, p->iisph_max_avg_density_error, RES_DENSITY, 1, &st.density_solver))) return rc;
        ms_dens += (omp_get_wtime() - t0) * 1e3;
        for (uint64_t i = 0; i < c->n; i++) c->pressure[i] /= sqrtf(c->omega[i]);
        all_pressure_accels(c, p);
        if ((rc = integrate_v_then_x(c, dt))) return rc;
        break;
    case SPH_SOLVER_ONLY_DIVERGENCE: /* simulation.rs:2448-2500 */
        if ((rc = update_velocity_with_non_pressure_accel(c, p, dt))) return rc;
        t0 = omp_get_wtime();
        prepare_ppe(c, p, dt, SRC_DIVERGENCE);
        if ((rc = pressure_iterations(c, p, dt, p->hybrid_dfsph_max_avg_divergence_error, RES_DIVERGENCE, 1, &st.div_solver))) return rc;
        ms_div += (omp_get_wtime() - t0) * 1e3;
        if ((rc = integrate_v_then_x(c, dt))) return rc;
        break;
    default: { /* HybridDFSPH, simulation.rs:2502-2670 */
        if (p->hybrid_dfsph_non_pressure_accel_before_divergence_free)
            if ((rc = update_velocity_with_non_pressure_accel(c, p, dt))) return rc;
        t0 = omp_get_wtime();
        prepare_ppe(c, p, dt, SRC_DIVERGENCE);
        if ((rc = pressure_iterations(c, p, dt, p->hybrid_dfsph_max_avg_divergence_error, RES_DIVERGENCE, 1, &st.div_solver))) return rc;
        ms_div += (omp_get_wtime() - t0) * 1e3;
        {
            int bad = 0;
#pragma omp parallel for schedule(static) reduction(| : bad)
            for (int64_t q = 0; q < (int64_t)(2 * c->n); q++) {
                c->vel[q] += dt * c->pacc[q];
                if (!isfinite(c->vel[q])) bad = 1;
            }
            if (bad) return orc_fail(c, SPH_ERR_VELOCITY_NOT_FINITE, "Assertion 'p_velocity[d].is_finite()' failed!");
        }
        if (!p->hybrid_dfsph_non_pressure_accel_before_divergence_free)
            if ((rc = update_velocity_with_non_pressure_accel(c, p, dt))) return rc;
        t0 = omp_get_wtime();
        prepare_ppe(c, p, dt, p->hybrid_dfsph_density_source_term == SPH_ONLY_DENSITY ? SRC_ONLY_DENSITY : SRC_FULL);
        if ((rc = pressure_iterations(c, p, dt, p->hybrid_dfsph_max_avg_density_error, RES_DENSITY, 1, &st.density_solver))) return rc;
        ms_dens += (omp_get_wtime() - t0) * 1e3;
        {
            /* simulation.rs:2644-2646 */
            const float vf = fminf(dt * p->hybrid_dfsph_factor, 1.f);
            int bad = 0;
#pragma omp parallel for schedule(static) reduction(| : bad)
            for (int64_t q = 0; q < (int64_t)(2 * c->n); q++) {
                c->pos[q] += dt * c->vel[q] + dt * dt * c->pacc[q];
                c->vel[q] += dt * c->pacc[q] * vf;
                if (!isfinite(c->pos[q])) bad = 1;
            }
            if (bad) return orc_fail(c, SPH_ERR_POSITION_NOT_FINITE, "Assertion 'p_position[d].is_finite()' failed!");
        }
        break;
    }
    }

    /* simulation.rs:2673-2676 */
    if (p->viscosity_type == SPH_VISC_XSPH) return orc_fail(c, SPH_ERR_XSPH_TODO, "not yet implemented (XSPH velocity smoothing)");

    /* simulation.rs:2678-2707 */
    if (p->level_estimation_after_advection) {
        if (p->use_extended_range_for_level_estimation) {
            /* orc_build_neighbors counts into c->neighbor_count; the FIELD keeps the counts of simulation.rs:2072-2074 */
            uint32_t* keep = (uint32_t*)malloc((c->n ? c->n : 1) * sizeof(uint32_t));
            memcpy(keep, c->neighbor_count, c->n * sizeof(uint32_t));
            rc = orc_build_neighbors(c, p->level_estimation_range / ORC_ETA);
            memcpy(c->neighbor_count, keep, c->n * sizeof(uint32_t));
            free(keep);
            if (rc) return rc;
        }
        t0 = omp_get_wtime();
        perform_level_estimation(c, p);
        ms_level += (omp_get_wtime() - t0) * 1e3;
    }
    /* simulation.rs:2709-2722 */
    t0 = omp_get_wtime();
    if ((rc = smooth_level_estimation_field(c, p))) return rc;
    ms_level += (omp_get_wtime() - t0) * 1e3;
    /* (classify_particles is NOT part of the step: only single_step_adaptivity calls it, simulation.rs:2749-2778) */

    c->time += dt;
    c->step_number += 1;

    st.time = c->time;
    st.step_number = c->step_number;
    st.ms_simulation_step = (omp_get_wtime() - t_step0) * 1e3;
    st.ms_neighborhood = ms_neigh;
    st.ms_level_estimation = ms_level;
    st.ms_div_solver = ms_div;
    st.ms_density_solver = ms_dens;
    if (out) *out = st;
    return SPH_OK;
}
