/*
 * TEST INFRASTRUCTURE -- CPU oracle for the SPH particle loop.  Not part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build/load/call this.
 *
 * sphmath.h: scalar f32 restatement of the reference's numerical primitives.  Every operation is
 * individually rounded (build with -ffp-contract=off, no -ffast-math): Rust never contracts, and
 * nalgebra's 2-vector dot/norm_squared is a0*b0 + a1*b1.
 *
 * Follows /root/reference/src/simulation/sph_kernels.rs:23-71 (cubic spline W and dW/dx),
 * :181-212 (DimensionUtils2d), :265-297 (smoothing_length symmetrisation) and
 * simulation.rs:369-380 (ETA, h from mass).
 */
#ifndef ORACLE_SPHMATH_H
#define ORACLE_SPHMATH_H

#include <math.h>

/* std::f32::consts::PI / FRAC_1_PI (mod.rs:23-27): the f64 constants rounded to f32 */
#define ORC_PI ((float)3.14159265358979323846264338327950288)
#define ORC_FRAC_1_PI ((float)0.318309886183790671537767526745028724)
/* simulation.rs:369 */
#define ORC_ETA 1.9f

/* sph_kernels.rs:23-32 */
static inline float orc_cubic_unnormalized(float q)
{
    if (q < 0.5f) {
        return 6.f * (q * q * q - q * q) + 1.f;
    } else if (q < 1.f) {
        float v = 1.f - q;
        return 2.f * (v * v * v);
    }
    return 0.f;
}

/* sph_kernels.rs:34-43 */
static inline float orc_cubic_unnormalized_deriv(float q)
{
    if (q < 0.5f) {
        return 18.f * q * q - 12.f * q;
    } else if (q < 1.f) {
        float v = 1.f - q;
        return -6.f * v * v;
    }
    return 0.f;
}

/* sph_kernels.rs:49-52: r = distance, h = smoothing length (support = 2h) */
static inline float orc_kernel2d(float r, float h)
{
    float norm_factor = 10.f / (7.f * ORC_PI * (h * h));
    return norm_factor * orc_cubic_unnormalized(r / (2.f * h));
}

/* nalgebra norm_squared of a 2-vector */
static inline float orc_norm_sq(float dx, float dy) { return dx * dx + dy * dy; }

/* DimensionUtils2d::kernelh (sph_kernels.rs:190-192) */
static inline float orc_kernelh(float dx, float dy, float h) { return orc_kernel2d(sqrtf(orc_norm_sq(dx, dy)), h); }

/* cubic_kernel_2d_deriv (sph_kernels.rs:61-71) == DimensionUtils2d::kernel_derivh (:194-196) */
static inline void orc_kernel_derivh(float dx, float dy, float h, float* gx, float* gy)
{
    float r = sqrtf(orc_norm_sq(dx, dy));
    float q = r / (2.f * h);
    if (q <= 1.0e-5f) {
        *gx = 0.f;
        *gy = 0.f;
        return;
    }
    dx = dx / r; /* unscale_mut */
    dy = dy / r;
    float norm_factor = 10.f / (7.f * ORC_PI * (h * h));
    float s = norm_factor * orc_cubic_unnormalized_deriv(q) / (2.f * h);
    *gx = s * dx;
    *gy = s * dy;
}

/* sph_kernels.rs:203-206 / :209-211 */
static inline float orc_sphere_volume_to_radius(float area) { return sqrtf(area * ORC_FRAC_1_PI); }
static inline float orc_radius_to_sphere_volume(float r) { return ORC_PI * r * r; }

/* simulation.rs:371-380 */
static inline float orc_h_from_mass(float mass, float rest_density)
{
    float volume = mass / rest_density;
    return ORC_ETA * orc_sphere_volume_to_radius(volume);
}

/* sph_kernels.rs:273-278, adaptive build */
static inline float orc_hij(float hi, float hj) { return (hi + hj) * 0.5f; }

#endif
