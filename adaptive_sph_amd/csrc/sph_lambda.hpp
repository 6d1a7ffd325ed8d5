// Host-side (f64) closed forms of the semi-analytic plane-boundary integrals of the 2-D cubic
// spline with support radius 1 -- lambda(d) (fraction of the kernel mass behind a plane at signed
// distance d) and d lambda / d d -- sampled ONCE per context into two 10001-entry f32 tables that
// the density kernel lerps (LDS/L2 resident, 80 KB).
//
// What is computed: /root/reference/src/simulation/boundary_handler/sdf_boundary_handler/
//   plane_numerics.rs:19-152  (lambda2, dlambda2; Winchenbach et al. 2020, Eq. 57)
//   lookup_table.rs:12-30     (sample positions x_i = (i/steps)*(max-min)+min in f32, y = f32(f(f64(x_i))))
//   boundary_winchenbach2020.rs:33-36  (range [-1,1], 10000 steps)
// The formulas are written with polynomial helpers in u = d^2 rather than term by term.
#pragma once

#include <cmath>
#include <vector>

namespace sph_lambda {

constexpr double kPi = 3.14159265358979323846264338327950288;

// powi-style integer power (square-and-multiply, LSB first), as f64::powi lowers to
inline double ipow(double a, int b)
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

struct Terms {
    double d, d2, d3, d4, d5, d6, d8;
    double sA;   // sqrt(1-2d) * sqrt(1+2d)   (only for d < 1/2)
    double sB;   // sqrt(1-d)  * sqrt(1+d)
    double lnA, lnB, lnd, ln2;
};

inline Terms make_terms(double d, bool inner)
{
    Terms t{};
    t.d = d;
    t.d2 = ipow(d, 2); t.d3 = ipow(d, 3); t.d4 = ipow(d, 4); t.d5 = ipow(d, 5); t.d6 = ipow(d, 6); t.d8 = ipow(d, 8);
    t.sB = std::sqrt(1.0 - 1.0 * d) * std::sqrt(d + 1.0);
    t.lnB = std::log(t.sB + 1.0);
    t.lnd = std::log(d);
    t.ln2 = std::log(2.0);
    if (inner) {
        t.sA = std::sqrt(1.0 - 2.0 * d) * std::sqrt(2.0 * d + 1.0);
        t.lnA = std::log(t.sA + 1.0);
    }
    return t;
}

inline double lambda_nonneg(double d)
{
    if (d < 0.000000001) return 0.5;
    if (d < 0.5) {
        const Terms t = make_terms(d, true);
        double num = ((-48.0 * t.d5) - 80.0 * t.d3) * t.lnA + (12.0 * t.d5 + 80.0 * t.d3) * t.lnB - 1.0 * std::acos(2.0 * d)
                   + 36.0 * t.lnd * t.d5 + 48.0 * t.ln2 * t.d5 + t.sA * (68.0 * t.d3 + 8.0 * d) + 80.0 * t.ln2 * t.d3
                   + t.sB * ((-68.0 * t.d3) - 32.0 * d) + 8.0 * std::acos(d);
        return num / (7. * kPi);
    }
    if (d < 1.) {
        const Terms t = make_terms(d, false);
        double num = ((-12.0 * t.d5) - 80.0 * t.d3) * t.lnB + t.lnd * (12.0 * t.d5 + 80.0 * t.d3) + t.sB * (68.0 * t.d3 + 32.0 * d)
                   - 8.0 * std::acos(d);
        return -num / (7. * kPi);
    }
    return 0.;
}

inline double lambda2(double d) { return d >= 0. ? lambda_nonneg(d) : 1. - lambda_nonneg(-d); }

inline double dlambda_nonneg(double d)
{
    if (d < 0.0000000001) return -1.36418522650196;
    if (d < 0.5) {
        const Terms t = make_terms(d, true);
        const double s1 = std::sqrt(2.0 * d + 1.0), s2 = std::sqrt(1.0 - 2.0 * d);
        const double s3 = std::sqrt(1.0 - 1.0 * d), s4 = std::sqrt(d + 1.0);
        const double lnA = std::log(s2 * s1 + 1.0), lnB = std::log(s3 * s4 + 1.0), L2 = t.ln2, Ld = t.lnd;
        const double d2 = t.d2, d4 = t.d4, d6 = t.d6, d8 = t.d8;
        const double T1 = (240.0 * d2 - 240.0 * d6) * lnA + (60.0 * d6 + 180.0 * d4 - 240.0 * d2) * lnB + Ld * (180.0 * d6 - 180.0 * d4)
                        + (240.0 * L2 - 1040.0) * d6 + 1000.0 * d4 + (10.0 - 240.0 * L2) * d2 + 30.0;
        const double T2 = (240.0 * d4 + 240.0 * d2) * lnA + ((-60.0 * d4) - 240.0 * d2) * lnB - 180.0 * Ld * d4
                        + (780.0 - 240.0 * L2) * d4 - 240.0 * L2 * d2 + 30.0;
        const double T3 = ((-960.0 * d6) - 720.0 * d4 + 240.0 * d2) * lnA + (240.0 * d6 + 900.0 * d4 - 240.0 * d2) * lnB
                        + Ld * (720.0 * d6 - 180.0 * d4) + (960.0 * L2 + 1040.0) * d6 + (720.0 * L2 - 100.0) * d4
                        + ((-240.0 * L2) - 160.0) * d2 + 30.0;
        const double num = s1 * (s2 * T1 + s2 * s3 * s4 * T2) + s3 * s4 * T3
                         + (960.0 * d8 - 240.0 * d6 - 960.0 * d4 + 240.0 * d2) * lnA
                         + ((-240.0 * d8) - 660.0 * d6 + 1140.0 * d4 - 240.0 * d2) * lnB - 960.0 * L2 * d8
                         + Ld * ((-720.0 * d8) + 900.0 * d6 - 180.0 * d4) + 240.0 * L2 * d6 + (960.0 * L2 + 120.0) * d4
                         + ((-240.0 * L2) - 150.0) * d2 + 30.0;
        const double den = 28.0 * kPi * d4 + s1 * (s2 * (7.0 * kPi - 7.0 * kPi * d2) + 7.0 * kPi * s2 * s3 * s4)
                         + s3 * s4 * (7.0 * kPi - 28.0 * kPi * d2) - 35.0 * kPi * d2 + 7.0 * kPi;
        return -(1.0 * num) / den;
    }
    if (d < 1.) {
        const Terms t = make_terms(d, false);
        const double s3 = std::sqrt(1.0 - 1.0 * d), s4 = std::sqrt(d + 1.0);
        const double lnB = std::log(s3 * s4 + 1.0), Ld = t.lnd;
        const double d2 = t.d2, d4 = t.d4, d6 = t.d6;
        const double num = s3 * s4 * ((60.0 * d4 + 240.0 * d2) * lnB + 260.0 * d4 + Ld * ((-60.0 * d4) - 240.0 * d2) - 220.0 * d2 - 40.0)
                         + ((-60.0 * d6) - 180.0 * d4 + 240.0 * d2) * lnB + Ld * (60.0 * d6 + 180.0 * d4 - 240.0 * d2) + 260.0 * d4
                         - 220.0 * d2 - 40.0;
        const double den = (-7.0 * kPi * d2) + 7.0 * kPi * s3 * s4 + 7.0 * kPi;
        return num / den;
    }
    return 0.;
}

inline double dlambda2(double d) { return d >= 0. ? dlambda_nonneg(d) : dlambda_nonneg(-d); }

constexpr int kSteps = 10000;

inline void build_luts(std::vector<float>& lam, std::vector<float>& dlam)
{
    lam.resize(kSteps + 1);
    dlam.resize(kSteps + 1);
    const float mn = -1.f, mx = 1.f;
    for (int i = 0; i <= kSteps; i++) {
        float x = ((float)i / (float)kSteps) * (mx - mn) + mn;
        lam[i] = (float)lambda2((double)x);
        dlam[i] = (float)dlambda2((double)x);
    }
}

}  // namespace sph_lambda
