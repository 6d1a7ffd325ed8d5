/*
 * TEST INFRASTRUCTURE -- CPU oracle (see sphmath.h).
 *
 * lambda.c: semi-analytic plane boundary integrals of the 2-D cubic spline (support radius 1) in
 * f64, and the 10001-entry f32 lookup tables the reference samples them into.
 *
 * Follows /root/reference/src/simulation/boundary_handler/sdf_boundary_handler/plane_numerics.rs
 *   :19-63   lambda2 / lambda2_nonnegative   (Winchenbach et al. 2020, Eq. 57, Maxima-generated)
 *   :68-152  dlambda2 / dlambda2_nonnegative
 * and lookup_table.rs:12-48 (LookupTable1D::new / get), boundary_winchenbach2020.rs:33-36.
 *
 * The closed forms are restated with named sub-terms; pinned against the reference's Maxima
 * known answers (plane_numerics.rs:182-195, 229-241) by tests/test_oracle_golden.py.
 */
#include "oracle.h"

#include <math.h>
#include <stdlib.h>

static const double PI64 = 3.14159265358979323846264338327950288;

/* f64::powi -> compiler-rt __powidf2: square-and-multiply, LSB first */
static double ipow(double a, int b)
{
    double r = 1.0;
    for (;;) {
        if (b & 1) r *= a;
        b /= 2;
        if (b == 0) break;
        a *= a;
    }
    return r;
}

/* plane_numerics.rs:30-63 */
static double lambda2_nonneg(double d)
{
    if (d < 0.000000001) return 0.5;
    const double d3 = ipow(d, 3), d5 = ipow(d, 5);
    const double B = sqrt(1.0 - 1.0 * d) * sqrt(d + 1.0); /* sqrt(1-d)sqrt(1+d) */
    const double LB = log(B + 1.0);
    const double Ld = log(d);
    if (d < 0.5) {
        const double A = sqrt(1.0 - 2.0 * d) * sqrt(2.0 * d + 1.0);
        const double LA = log(A + 1.0);
        const double L2 = log(2.0);
        double num = ((-48.0 * d5) - 80.0 * d3) * LA
                   + (12.0 * d5 + 80.0 * d3) * LB
                   - 1.0 * acos(2.0 * d)
                   + 36.0 * Ld * d5
                   + 48.0 * L2 * d5
                   + A * (68.0 * d3 + 8.0 * d)
                   + 80.0 * L2 * d3
                   + B * ((-68.0 * d3) - 32.0 * d)
                   + 8.0 * acos(d);
        return num / (7. * PI64);
    } else if (d < 1.) {
        double num = ((-12.0 * d5) - 80.0 * d3) * LB
                   + Ld * (12.0 * d5 + 80.0 * d3)
                   + B * (68.0 * d3 + 32.0 * d)
                   - 8.0 * acos(d);
        return -num / (7. * PI64);
    }
    return 0.;
}

/* plane_numerics.rs:19-25 */
double orc_lambda2(double d) { return d >= 0. ? lambda2_nonneg(d) : 1. - lambda2_nonneg(-d); }

/* plane_numerics.rs:79-152 */
static double dlambda2_nonneg(double d)
{
    if (d < 0.0000000001) return -1.36418522650196;
    const double d2 = ipow(d, 2), d4 = ipow(d, 4), d6 = ipow(d, 6);
    const double s3 = sqrt(1.0 - 1.0 * d), s4 = sqrt(d + 1.0);
    const double LB = log(s3 * s4 + 1.0);
    const double Ld = log(d);
    if (d < 0.5) {
        const double d8 = ipow(d, 8);
        const double s1 = sqrt(2.0 * d + 1.0), s2 = sqrt(1.0 - 2.0 * d);
        const double LA = log(s2 * s1 + 1.0);
        const double L2 = log(2.0);
        const double T1 = (240.0 * d2 - 240.0 * d6) * LA
                        + (60.0 * d6 + 180.0 * d4 - 240.0 * d2) * LB
                        + Ld * (180.0 * d6 - 180.0 * d4)
                        + (240.0 * L2 - 1040.0) * d6
                        + 1000.0 * d4
                        + (10.0 - 240.0 * L2) * d2
                        + 30.0;
        const double T2 = (240.0 * d4 + 240.0 * d2) * LA
                        + ((-60.0 * d4) - 240.0 * d2) * LB
                        - 180.0 * Ld * d4
                        + (780.0 - 240.0 * L2) * d4
                        - 240.0 * L2 * d2
                        + 30.0;
        const double T3 = ((-960.0 * d6) - 720.0 * d4 + 240.0 * d2) * LA
                        + (240.0 * d6 + 900.0 * d4 - 240.0 * d2) * LB
                        + Ld * (720.0 * d6 - 180.0 * d4)
                        + (960.0 * L2 + 1040.0) * d6
                        + (720.0 * L2 - 100.0) * d4
                        + ((-240.0 * L2) - 160.0) * d2
                        + 30.0;
        const double num = s1 * (s2 * T1 + s2 * s3 * s4 * T2)
                         + s3 * s4 * T3
                         + (960.0 * d8 - 240.0 * d6 - 960.0 * d4 + 240.0 * d2) * LA
                         + ((-240.0 * d8) - 660.0 * d6 + 1140.0 * d4 - 240.0 * d2) * LB
                         - 960.0 * L2 * d8
                         + Ld * ((-720.0 * d8) + 900.0 * d6 - 180.0 * d4)
                         + 240.0 * L2 * d6
                         + (960.0 * L2 + 120.0) * d4
                         + ((-240.0 * L2) - 150.0) * d2
                         + 30.0;
        const double den = 28.0 * PI64 * d4
                         + s1 * (s2 * (7.0 * PI64 - 7.0 * PI64 * d2) + 7.0 * PI64 * s2 * s3 * s4)
                         + s3 * s4 * (7.0 * PI64 - 28.0 * PI64 * d2)
                         - 35.0 * PI64 * d2
                         + 7.0 * PI64;
        return -(1.0 * num) / den;
    } else if (d < 1.) {
        const double num = s3 * s4 * ((60.0 * d4 + 240.0 * d2) * LB
                                      + 260.0 * d4
                                      + Ld * ((-60.0 * d4) - 240.0 * d2)
                                      - 220.0 * d2
                                      - 40.0)
                         + ((-60.0 * d6) - 180.0 * d4 + 240.0 * d2) * LB
                         + Ld * (60.0 * d6 + 180.0 * d4 - 240.0 * d2)
                         + 260.0 * d4
                         - 220.0 * d2
                         - 40.0;
        const double den = (-7.0 * PI64 * d2) + 7.0 * PI64 * s3 * s4 + 7.0 * PI64;
        return num / den;
    }
    return 0.;
}

/* plane_numerics.rs:68-74 */
double orc_dlambda2(double d) { return d >= 0. ? dlambda2_nonneg(d) : dlambda2_nonneg(-d); }

/* lookup_table.rs:12-30 with (min,max,steps) = (-1,1,10000), boundary_winchenbach2020.rs:34-36 */
void orc_lut_build(orc_lut* lut, double (*f)(double))
{
    const float min = -1.f, max = 1.f;
    lut->min = min;
    lut->max = max;
    lut->len_inv = 1.f / (max - min);
    lut->steps = ORC_LUT_STEPS;
    for (int i = 0; i < ORC_LUT_STEPS + 1; i++) {
        float x = ((float)i / (float)ORC_LUT_STEPS) * (max - min) + min;
        lut->data[i] = (float)f((double)x);
    }
}

/* lookup_table.rs:32-48; caller guarantees min <= x < max */
float orc_lut_get(const orc_lut* lut, float x)
{
    float fidx = (x - lut->min) * lut->len_inv * (float)lut->steps;
    float fidx_floor = floorf(fidx);
    float interp = fidx - fidx_floor;
    int idx = (int)fidx_floor;
    if (idx + 1 >= lut->steps + 1) return lut->data[idx];
    return lut->data[idx] * (1.f - interp) + lut->data[idx + 1] * interp;
}
