// Jacobi-sweep laboratory: the relaxed-Jacobi update (sweep B, OpJacobi of csrc/sph_sweeps.hip) of a uniform-h scene as a
// stand-alone program, in several STRUCTURES, timed on the same data and checked against the first one.  The product's sweep is
// reproduced as variant `gather4`; the others answer "what bounds it" by construction rather than by counters.
//
//   build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o jacobi_lab scripts/ubench/jacobi_lab.hip ; run: ./jacobi_lab [side=1024] [jitter=0.15] [reps=50]
//
// Data: side x side particles on a lattice of spacing d = 1/1024 (jittered), h = 1.9 sqrt(d^2 / pi), cell = 2 h, cell-sorted
// x-fastest exactly like the product (stable by lattice order); list word = three 32-bit row masks (bit b of row r = candidate b of
// the contiguous range of cell row cy + r - 1, cells cx - 1 .. cx + 1), own bit dropped from the middle row.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#define CHECK(x)                                                                                                   \
    do {                                                                                                           \
        hipError_t e_ = (x);                                                                                       \
        if (e_ != hipSuccess) {                                                                                    \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));                              \
            exit(1);                                                                                               \
        }                                                                                                          \
    } while (0)

struct Grid {
    float cs;
    int minx, miny, sx, sy;
};
struct Math {
    float h, nf, inv2h;
};
struct Args {
    uint32_t n, nblocks;
    Grid g;
    Math m;
    const uint32_t* __restrict__ cell_start;
    const uint4* __restrict__ nl;
    const float4* __restrict__ pm;     // x, y, m, h
    const float2* __restrict__ pacc;   // a^p
    const float4* __restrict__ comb;   // x, y, a^p.x, a^p.y
    const float* __restrict__ rho;
    const float* __restrict__ aii;
    const float* __restrict__ src;
    const float* __restrict__ p_in;
    float* __restrict__ p_out;
    float* __restrict__ pt_out;
    float omega, mass;
    // sweep A (pressure acceleration): p / rho^2 per particle, the same inside a combined record {x, y, p / rho^2, p}, the output
    const float* __restrict__ pt;
    const float4* __restrict__ rec;
    float4* __restrict__ pacc_out;
    // relative-offset lists (round 4): 24 slots of int16 (j - i) per particle, quad g of particle i at off16[g n + i]; slots behind the
    // count hold 0 (the particle itself: its pair term is exactly zero in the slim arithmetic); cnt8[i] = number of neighbours
    const uint4* __restrict__ off16;
    const uint8_t* __restrict__ cnt8;
    // the mask words with STORED row bases (20 B per particle instead of 36): rbd[i] = (rb0 - i) & 0xffff | (rb2 - i) << 16, and the
    // position of the particle's own bit in row 1 (i - rb1) in bits 8..12 of nl[i].w
    const uint32_t* __restrict__ rbd;
    // row-delta lists: one 32-bit word per group of up to four ROW-CONSECUTIVE neighbours -- 16-bit offset of the first (j0 - i), three
    // 5-bit deltas to the next ones (0 = no entry); group g of particle i at dlt[g n + i], dcnt[i] = number of groups (<= 6)
    const uint32_t* __restrict__ dlt;
    const uint8_t* __restrict__ dcnt;
};

__device__ __forceinline__ void grad_uniform(const Math& m, float dx, float dy, float r2, float& gx, float& gy)
{
    const float rinv = __builtin_amdgcn_rsqf(r2);
    const float q = (r2 * rinv) * m.inv2h;
    const float a = 18.f * q * q - 12.f * q;
    const float v = 1.f - q;
    const float b = -6.f * v * v;
    const float d = q < 0.5f ? a : (q < 1.f ? b : 0.f);
    float s = m.nf * d * m.inv2h * rinv;
    s = (q > 1.0e-5f) ? s : 0.f;
    gx = s * dx;
    gy = s * dy;
}

// LAB_CHECK=1: every gathered index is tested against its array's length; the first violation is recorded instead of faulting
#ifndef LAB_CHECK
#define LAB_CHECK 0
#endif
__device__ uint32_t g_bad[4];
__device__ __forceinline__ uint32_t ix(uint32_t j, uint32_t lim, uint32_t tag)
{
#if LAB_CHECK
    if (j >= lim) {
        if (atomicCAS(&g_bad[0], 0u, tag) == 0u) {
            g_bad[1] = j;
            g_bad[2] = lim;
            g_bad[3] = blockIdx.x * 256 + threadIdx.x;
        }
        return 0u;
    }
#endif
    return j;
}
__device__ __forceinline__ uint32_t __reduce_max_sync_lab(uint32_t v)
{
    for (int o = 32; o > 0; o >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, o, 64));
    return v;
}
struct Acc {
    float sum, qx, qy, inv_rho;
};
__device__ __forceinline__ void pair(const Args& A, Acc& a, float xi, float yi, float xj, float yj, float apx, float apy, bool on)
{
    const float dx = xi - xj, dy = yi - yj;
    const float r2 = dx * dx + dy * dy;
    if (on) {
        float gx, gy;
        grad_uniform(A.m, dx, dy, r2, gx, gy);
        const float dot = (apx - a.qx) * gx + (apy - a.qy) * gy;
        a.sum += A.mass * a.inv_rho * dot;
    }
}
// the same pair with the instruction mix of scripts/ubench/valu_issue.hip in mind (a compare or a select issues at half the rate of
// a multiply, a transcendental at a quarter): truncated-power spline W'(q) = 6 [4 (1/2 - q)+^2 - (1 - q)+^2] (two v_max instead of
// two compare + select pairs), r2 clamped away from 0 instead of the q > 1e-5 select (dx = dy = 0 then gives g = 0 by itself),
// the constant factors folded (nf6 = 6 nf / (2h)), the mass / rho_i factor applied once per particle.
__device__ __forceinline__ void pair_slim(const Args& A, Acc& a, float xi, float yi, float xj, float yj, float apx, float apy, bool on, float nf6)
{
    const float dx = xi - xj, dy = yi - yj;
    const float r2 = fmaxf(dx * dx + dy * dy, 1.0e-30f);
    if (on) {
        const float rinv = __builtin_amdgcn_rsqf(r2);
        const float q = (r2 * rinv) * A.m.inv2h;
        const float u = fmaxf(1.f - q, 0.f), t = fmaxf(0.5f - q, 0.f);
        const float d = fmaf(4.f * t, t, -(u * u));
        const float s = nf6 * d * rinv;
        a.sum = fmaf((apx - a.qx) * s, dx, fmaf((apy - a.qy) * s, dy, a.sum));
    }
}
__device__ __forceinline__ void finish(const Args& A, const Acc& a, uint32_t i, float rho_i)
{
    const float aii_i = A.aii[i];
    const float pn = A.p_in[i] + A.omega * (A.src[i] - a.sum) / aii_i;
    const bool pos = pn > 0.f;
    A.p_out[i] = pos ? pn : 0.f;
    A.pt_out[i] = pos ? pn / (rho_i * rho_i) : 0.f;
}
__device__ __forceinline__ uint32_t remap_block(uint32_t nblocks)
{
    const uint32_t per_xcd = (nblocks + 7) >> 3;
    return (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
}
__device__ __forceinline__ void row_bases(const Args& A, float x, float y, uint32_t (&rb)[3], int& cx, int& cy)
{
    cx = (int)floorf(x / A.g.cs) - A.g.minx;
    cy = (int)floorf(y / A.g.cs) - A.g.miny;
#pragma unroll
    for (int dr = 0; dr < 3; dr++) {
        const int yy = cy + dr - 1;
        const bool ok = yy >= 0 && yy < A.g.sy;
        rb[dr] = ok ? A.cell_start[ix((uint32_t)yy * (uint32_t)A.g.sx + (uint32_t)max(cx - 1, 0), (uint32_t)A.g.sx * (uint32_t)A.g.sy + 1u, 1u)] : 0u;
    }
}

// ---- variant: the product's form.  4 set bits of a row per trip: 4 record gathers (16 B) + 4 payload gathers (8 B) in flight ----
template <int COMB>   // COMB: one combined 16-B record {x, y, a^p} instead of record + payload
__global__ __launch_bounds__(256) void k_gather4(Args A)
{
    const uint32_t blk = remap_block(A.nblocks);
    if (blk >= A.nblocks) return;
    const uint32_t i = blk * 256 + threadIdx.x;
    if (i >= A.n) return;
    const float4 Ai = A.pm[i];
    const uint4 lw = A.nl[i];
    uint32_t rb[3];
    int cx, cy;
    row_bases(A, Ai.x, Ai.y, rb, cx, cy);
    Acc a;
    a.sum = 0.f;
    const float rho_i = A.rho[i];
    a.inv_rho = __builtin_amdgcn_rcpf(rho_i);
    const float2 q = A.pacc[i];
    a.qx = q.x;
    a.qy = q.y;
    const uint32_t masks[3] = {lw.x, lw.y, lw.z};
#pragma unroll
    for (int dr = 0; dr < 3; dr++) {
        uint32_t mk = masks[dr];
        const uint32_t base = rb[dr];
        while (mk) {
            const uint32_t b0 = __ffs(mk) - 1;
            mk &= mk - 1;
            const bool v1 = mk != 0;
            const uint32_t b1 = v1 ? __ffs(mk) - 1 : b0;
            mk &= mk - 1;
            const bool v2 = mk != 0;
            const uint32_t b2 = v2 ? __ffs(mk) - 1 : b0;
            mk &= mk - 1;
            const bool v3 = mk != 0;
            const uint32_t b3 = v3 ? __ffs(mk) - 1 : b0;
            mk &= mk - 1;
            if (COMB) {
                const float4 R0 = A.comb[ix(base + b0, A.n, 2u)], R1 = A.comb[ix(base + b1, A.n, 2u)], R2 = A.comb[ix(base + b2, A.n, 2u)], R3 = A.comb[ix(base + b3, A.n, 2u)];
                pair(A, a, Ai.x, Ai.y, R0.x, R0.y, R0.z, R0.w, true);
                pair(A, a, Ai.x, Ai.y, R1.x, R1.y, R1.z, R1.w, v1);
                pair(A, a, Ai.x, Ai.y, R2.x, R2.y, R2.z, R2.w, v2);
                pair(A, a, Ai.x, Ai.y, R3.x, R3.y, R3.z, R3.w, v3);
            } else {
                const float4 R0 = A.pm[ix(base + b0, A.n, 2u)], R1 = A.pm[ix(base + b1, A.n, 2u)], R2 = A.pm[ix(base + b2, A.n, 2u)], R3 = A.pm[ix(base + b3, A.n, 2u)];
                const float2 P0 = A.pacc[ix(base + b0, A.n, 2u)], P1 = A.pacc[ix(base + b1, A.n, 2u)], P2 = A.pacc[ix(base + b2, A.n, 2u)], P3 = A.pacc[ix(base + b3, A.n, 2u)];
                pair(A, a, Ai.x, Ai.y, R0.x, R0.y, P0.x, P0.y, true);
                pair(A, a, Ai.x, Ai.y, R1.x, R1.y, P1.x, P1.y, v1);
                pair(A, a, Ai.x, Ai.y, R2.x, R2.y, P2.x, P2.y, v2);
                pair(A, a, Ai.x, Ai.y, R3.x, R3.y, P3.x, P3.y, v3);
            }
        }
    }
    finish(A, a, i, rho_i);
}

// ---- variant: combined record, trips of 4, slim pair arithmetic; HOIST: the finish's own loads (a_ii, s, p) requested at the top ----
template <int HOIST>
__global__ __launch_bounds__(256) void k_gather4_slim(Args A)
{
    const uint32_t blk = remap_block(A.nblocks);
    if (blk >= A.nblocks) return;
    const uint32_t i = blk * 256 + threadIdx.x;
    if (i >= A.n) return;
    const float4 Ai = A.comb[i];
    const uint4 lw = A.nl[i];
    const float rho_i = A.rho[i];
    float aii_i = 0.f, src_i = 0.f, pin_i = 0.f;
    if (HOIST) {
        aii_i = A.aii[i];
        src_i = A.src[i];
        pin_i = A.p_in[i];
    }
    uint32_t rb[3];
    int cx, cy;
    row_bases(A, Ai.x, Ai.y, rb, cx, cy);
    Acc a;
    a.sum = 0.f;
    a.inv_rho = __builtin_amdgcn_rcpf(rho_i);
    a.qx = Ai.z;
    a.qy = Ai.w;
    const float nf6 = 6.f * A.m.nf * A.m.inv2h;
    const uint32_t masks[3] = {lw.x, lw.y, lw.z};
#pragma unroll
    for (int dr = 0; dr < 3; dr++) {
        uint32_t mk = masks[dr];
        const uint32_t base = rb[dr];
        while (mk) {
            const uint32_t b0 = __ffs(mk) - 1;
            mk &= mk - 1;
            const bool v1 = mk != 0;
            const uint32_t b1 = v1 ? __ffs(mk) - 1 : b0;
            mk &= mk - 1;
            const bool v2 = mk != 0;
            const uint32_t b2 = v2 ? __ffs(mk) - 1 : b0;
            mk &= mk - 1;
            const bool v3 = mk != 0;
            const uint32_t b3 = v3 ? __ffs(mk) - 1 : b0;
            mk &= mk - 1;
            const float4 R0 = A.comb[ix(base + b0, A.n, 2u)], R1 = A.comb[ix(base + b1, A.n, 2u)], R2 = A.comb[ix(base + b2, A.n, 2u)], R3 = A.comb[ix(base + b3, A.n, 2u)];
            pair_slim(A, a, Ai.x, Ai.y, R0.x, R0.y, R0.z, R0.w, true, nf6);
            pair_slim(A, a, Ai.x, Ai.y, R1.x, R1.y, R1.z, R1.w, v1, nf6);
            pair_slim(A, a, Ai.x, Ai.y, R2.x, R2.y, R2.z, R2.w, v2, nf6);
            pair_slim(A, a, Ai.x, Ai.y, R3.x, R3.y, R3.z, R3.w, v3, nf6);
        }
    }
    a.sum *= A.mass * a.inv_rho;
    if (HOIST) {
        const float pn = pin_i + A.omega * (src_i - a.sum) / aii_i;
        const bool pos = pn > 0.f;
        A.p_out[i] = pos ? pn : 0.f;
        A.pt_out[i] = pos ? pn / (rho_i * rho_i) : 0.f;
    } else {
        finish(A, a, i, rho_i);
    }
}

// ---- variant: sweep B, combined record, slim arithmetic, own loads first -- and NO branch around a pair: every slot of a trip is
// evaluated, an invalid slot's scale factor is selected to zero (the compiler otherwise sinks a slot's gather into the branch that
// uses it: one of a trip's four loads is then requested only after the first pair is done, a second latency per trip) ----
__global__ __launch_bounds__(256) void k_gather4_slim_nobranch(Args A)
{
    const uint32_t blk = remap_block(A.nblocks);
    if (blk >= A.nblocks) return;
    const uint32_t i = blk * 256 + threadIdx.x;
    if (i >= A.n) return;
    const float4 Ai = A.comb[i];
    const uint4 lw = A.nl[i];
    const float rho_i = A.rho[i];
    const float aii_i = A.aii[i], src_i = A.src[i], pin_i = A.p_in[i];
    uint32_t rb[3];
    int cx, cy;
    row_bases(A, Ai.x, Ai.y, rb, cx, cy);
    float sum = 0.f;
    const float inv_rho = __builtin_amdgcn_rcpf(rho_i);
    const float nf6 = 6.f * A.m.nf * A.m.inv2h;
    const uint32_t masks[3] = {lw.x, lw.y, lw.z};
#pragma unroll
    for (int dr = 0; dr < 3; dr++) {
        uint32_t mk = masks[dr];
        const uint32_t base = rb[dr];
        while (mk) {
            uint32_t b[4];
            bool v[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                v[k] = mk != 0u;
                b[k] = v[k] ? (uint32_t)__ffs(mk) - 1u : b[0];
                mk &= mk - 1u;
            }
            float4 R[4];
#pragma unroll
            for (int k = 0; k < 4; k++) R[k] = A.comb[ix(base + b[k], A.n, 30u)];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const float dx = Ai.x - R[k].x, dy = Ai.y - R[k].y;
                const float r2 = fmaxf(dx * dx + dy * dy, 1.0e-30f);
                const float rinv = __builtin_amdgcn_rsqf(r2);
                const float q = (r2 * rinv) * A.m.inv2h;
                const float u = fmaxf(1.f - q, 0.f), t = fmaxf(0.5f - q, 0.f);
                const float d = fmaf(4.f * t, t, -(u * u));
                const float sc = v[k] ? nf6 * d * rinv : 0.f;
                sum = fmaf((R[k].z - Ai.z) * sc, dx, fmaf((R[k].w - Ai.w) * sc, dy, sum));
            }
        }
    }
    sum *= A.mass * inv_rho;
    const float pn = pin_i + A.omega * (src_i - sum) / aii_i;
    const bool pos = pn > 0.f;
    A.p_out[i] = pos ? pn : 0.f;
    A.pt_out[i] = pos ? pn / (rho_i * rho_i) : 0.f;
}

// ---- variants of sweep B / sweep A on their combined records with a leaner slot (LEAN bits: 1 = r2 + 1e-30 instead of max(r2,
// 1e-30) (an add is full rate, a max half rate; same value for every r2 > 1e-23), 2 = no clamp of 1 - q (a listed neighbour is
// inside the support: q < 1 up to the rsq's rounding, where u^2 ~ 1e-14 is below the sum's resolution), 4 = no select on an empty
// slot's bit index (ffbl of 0 is -1: the slot gathers record base - 1, which exists -- the arrays carry one element in front) ----
template <int LEAN>
__global__ __launch_bounds__(256) void k_gather4_lean(Args A)
{
    const uint32_t blk = remap_block(A.nblocks);
    if (blk >= A.nblocks) return;
    const uint32_t i = blk * 256 + threadIdx.x;
    if (i >= A.n) return;
    const float4 Ai = A.comb[i];
    const uint4 lw = A.nl[i];
    const float rho_i = A.rho[i];
    const float aii_i = A.aii[i], src_i = A.src[i], pin_i = A.p_in[i];
    uint32_t rb[3];
    int cx, cy;
    row_bases(A, Ai.x, Ai.y, rb, cx, cy);
    float sum = 0.f;
    const float inv_rho = __builtin_amdgcn_rcpf(rho_i);
    const float nf6 = 6.f * A.m.nf * A.m.inv2h;
    const uint32_t masks[3] = {lw.x, lw.y, lw.z};
#pragma unroll
    for (int dr = 0; dr < 3; dr++) {
        uint32_t mk = masks[dr];
        const uint32_t base = rb[dr];
        while (mk) {
            uint32_t b[4];
            bool v[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                v[k] = mk != 0u;
                if (LEAN & 4) b[k] = (uint32_t)__ffs(mk) - 1u;
                else b[k] = v[k] ? (uint32_t)__ffs(mk) - 1u : b[0];
                mk &= mk - 1u;
            }
            float4 R[4];
#pragma unroll
            for (int k = 0; k < 4; k++) R[k] = A.comb[(int)(base + b[k])];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const float dx = Ai.x - R[k].x, dy = Ai.y - R[k].y;
                const float r2 = (LEAN & 1) ? (dx * dx + dy * dy) + 1.0e-30f : fmaxf(dx * dx + dy * dy, 1.0e-30f);
                if (v[k]) {
                    const float rinv = __builtin_amdgcn_rsqf(r2);
                    const float q = (r2 * rinv) * A.m.inv2h;
                    const float u = (LEAN & 2) ? 1.f - q : fmaxf(1.f - q, 0.f), t = fmaxf(0.5f - q, 0.f);
                    const float d = fmaf(4.f * t, t, -(u * u));
                    const float sc = nf6 * d * rinv;
                    sum = fmaf((R[k].z - Ai.z) * sc, dx, fmaf((R[k].w - Ai.w) * sc, dy, sum));
                }
            }
        }
    }
    sum *= A.mass * inv_rho;
    const float pn = pin_i + A.omega * (src_i - sum) / aii_i;
    const bool pos = pn > 0.f;
    A.p_out[i] = pos ? pn : 0.f;
    A.pt_out[i] = pos ? pn / (rho_i * rho_i) : 0.f;
}
template <int LEAN>
__global__ __launch_bounds__(256) void k_accel_lean(Args A)
{
    const uint32_t blk = remap_block(A.nblocks);
    if (blk >= A.nblocks) return;
    const uint32_t i = blk * 256 + threadIdx.x;
    if (i >= A.n) return;
    const float4 Ai = A.rec[i];
    const uint4 lw = A.nl[i];
    const float pti = Ai.z;
    uint32_t rb[3];
    int cx, cy;
    row_bases(A, Ai.x, Ai.y, rb, cx, cy);
    const float nf6 = 6.f * A.m.nf * A.m.inv2h;
    float ax = 0.f, ay = 0.f;
    const uint32_t masks[3] = {lw.x, lw.y, lw.z};
#pragma unroll
    for (int dr = 0; dr < 3; dr++) {
        uint32_t mk = masks[dr];
        const uint32_t base = rb[dr];
        while (mk) {
            uint32_t b[4];
            bool v[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                v[k] = mk != 0u;
                if (LEAN & 4) b[k] = (uint32_t)__ffs(mk) - 1u;
                else b[k] = v[k] ? (uint32_t)__ffs(mk) - 1u : b[0];
                mk &= mk - 1u;
            }
            float4 R[4];
#pragma unroll
            for (int k = 0; k < 4; k++) R[k] = A.rec[(int)(base + b[k])];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const float dx = Ai.x - R[k].x, dy = Ai.y - R[k].y;
                const float r2 = (LEAN & 1) ? (dx * dx + dy * dy) + 1.0e-30f : fmaxf(dx * dx + dy * dy, 1.0e-30f);
                if (v[k]) {
                    const float rinv = __builtin_amdgcn_rsqf(r2);
                    const float q = (r2 * rinv) * A.m.inv2h;
                    const float u = (LEAN & 2) ? 1.f - q : fmaxf(1.f - q, 0.f), t = fmaxf(0.5f - q, 0.f);
                    const float d = fmaf(4.f * t, t, -(u * u));
                    const float fs = (-A.mass * (pti + R[k].z)) * (nf6 * d * rinv);
                    ax = fmaf(fs, dx, ax);
                    ay = fmaf(fs, dy, ay);
                }
            }
        }
    }
    A.pacc_out[i] = make_float4(Ai.x, Ai.y, ax, ay);
}

// ---- variant (round 4): ROW-DELTA lists, 4 bytes per trip instead of 8: the rest lattice's 12 neighbours are 12 + 1 bytes per particle
// instead of 24 + 1.  An empty slot (delta 0) evaluates the particle itself.
__global__ __launch_bounds__(256) void k_delta_slim(Args A)
{
    const uint32_t blk = remap_block(A.nblocks);
    if (blk >= A.nblocks) return;
    const uint32_t i = blk * 256 + threadIdx.x;
    if (i >= A.n) return;
    const float4 Ai = A.comb[i];
    const uint32_t g0 = A.dlt[i], g1 = A.dlt[(size_t)A.n + i], g2 = A.dlt[2 * (size_t)A.n + i];
    const uint32_t ng = A.dcnt[i];
    const float rho_i = A.rho[i];
    const float aii_i = A.aii[i], src_i = A.src[i], pin_i = A.p_in[i];
    Acc a;
    a.sum = 0.f;
    a.inv_rho = __builtin_amdgcn_rcpf(rho_i);
    a.qx = Ai.z;
    a.qy = Ai.w;
    const float nf6 = 6.f * A.m.nf * A.m.inv2h;
#define DLT_IDX(G, J0, J1, J2, J3)                                                                  \
    const uint32_t J0 = i + (uint32_t)((int)((G) << 16) >> 16);                                     \
    const uint32_t e1 = ((G) >> 16) & 31u, e2 = ((G) >> 21) & 31u, e3 = ((G) >> 26) & 31u;          \
    const uint32_t J1 = e1 ? J0 + e1 : i, J2 = e2 ? J0 + e1 + e2 : i, J3 = e3 ? J0 + e1 + e2 + e3 : i;
#define DLT_TRIP(G)                                                                                 \
    {                                                                                               \
        DLT_IDX(G, j0, j1, j2, j3)                                                                  \
        const float4 R0 = A.comb[ix(j0, A.n, 10u)], R1 = A.comb[ix(j1, A.n, 10u)], R2 = A.comb[ix(j2, A.n, 10u)], R3 = A.comb[ix(j3, A.n, 10u)]; \
        pair_slim(A, a, Ai.x, Ai.y, R0.x, R0.y, R0.z, R0.w, true, nf6);                             \
        pair_slim(A, a, Ai.x, Ai.y, R1.x, R1.y, R1.z, R1.w, true, nf6);                             \
        pair_slim(A, a, Ai.x, Ai.y, R2.x, R2.y, R2.z, R2.w, true, nf6);                             \
        pair_slim(A, a, Ai.x, Ai.y, R3.x, R3.y, R3.z, R3.w, true, nf6);                             \
    }
    if (__any(ng > 2u)) {   // the usual wave: the gathers of three groups leave together
        float4 R[12];
        {
            DLT_IDX(g0, a0, a1, a2, a3)
            R[0] = A.comb[ix(a0, A.n, 10u)]; R[1] = A.comb[ix(a1, A.n, 10u)]; R[2] = A.comb[ix(a2, A.n, 10u)]; R[3] = A.comb[ix(a3, A.n, 10u)];
        }
        {
            DLT_IDX(g1, b0, b1, b2, b3)
            R[4] = A.comb[ix(b0, A.n, 10u)]; R[5] = A.comb[ix(b1, A.n, 10u)]; R[6] = A.comb[ix(b2, A.n, 10u)]; R[7] = A.comb[ix(b3, A.n, 10u)];
        }
        {
            DLT_IDX(g2, c0, c1, c2, c3)
            R[8] = A.comb[ix(c0, A.n, 10u)]; R[9] = A.comb[ix(c1, A.n, 10u)]; R[10] = A.comb[ix(c2, A.n, 10u)]; R[11] = A.comb[ix(c3, A.n, 10u)];
        }
#pragma unroll
        for (int k = 0; k < 12; k++) pair_slim(A, a, Ai.x, Ai.y, R[k].x, R[k].y, R[k].z, R[k].w, true, nf6);
    } else {
        DLT_TRIP(g0)
        if (__any(ng > 1u)) DLT_TRIP(g1)
    }
    for (uint32_t g = 3; g < 6u; g++) {
        if (!__any(ng > g)) break;
        const uint32_t gg = ng > g ? A.dlt[(size_t)g * A.n + i] : 0u;
        DLT_TRIP(gg)
    }
#undef DLT_TRIP
#undef DLT_IDX
    a.sum *= A.mass * a.inv_rho;
    const float pn = pin_i + A.omega * (src_i - a.sum) / aii_i;
    const bool pos = pn > 0.f;
    A.p_out[i] = pos ? pn : 0.f;
    A.pt_out[i] = pos ? pn / (rho_i * rho_i) : 0.f;
}

// ---- variant (round 4): the product's mask replay, but the three row bases come from 4 stored bytes (two 16-bit deltas and the
// position of the own bit) instead of two IEEE divisions and three dependent cell_start loads: which part of the offset lists' gain
// is the head of the sweep, which the per-slot decoding?
__global__ __launch_bounds__(256) void k_gather4_slim_rbd(Args A)
{
    const uint32_t blk = remap_block(A.nblocks);
    if (blk >= A.nblocks) return;
    const uint32_t i = blk * 256 + threadIdx.x;
    if (i >= A.n) return;
    const float4 Ai = A.comb[i];
    const uint4 lw = A.nl[i];
    const uint32_t dd = A.rbd[i];
    const float rho_i = A.rho[i];
    const float aii_i = A.aii[i], src_i = A.src[i], pin_i = A.p_in[i];
    const uint32_t rb[3] = {i + (uint32_t)((int)(dd << 16) >> 16), i - ((lw.w >> 8) & 31u), i + (uint32_t)((int)dd >> 16)};
    Acc a;
    a.sum = 0.f;
    a.inv_rho = __builtin_amdgcn_rcpf(rho_i);
    a.qx = Ai.z;
    a.qy = Ai.w;
    const float nf6 = 6.f * A.m.nf * A.m.inv2h;
    const uint32_t masks[3] = {lw.x, lw.y, lw.z};
#pragma unroll
    for (int dr = 0; dr < 3; dr++) {
        uint32_t mk = masks[dr];
        const uint32_t base = rb[dr];
        while (mk) {
            const uint32_t b0 = __ffs(mk) - 1;
            mk &= mk - 1;
            const bool v1 = mk != 0;
            const uint32_t b1 = v1 ? __ffs(mk) - 1 : b0;
            mk &= mk - 1;
            const bool v2 = mk != 0;
            const uint32_t b2 = v2 ? __ffs(mk) - 1 : b0;
            mk &= mk - 1;
            const bool v3 = mk != 0;
            const uint32_t b3 = v3 ? __ffs(mk) - 1 : b0;
            mk &= mk - 1;
            const float4 R0 = A.comb[ix(base + b0, A.n, 9u)], R1 = A.comb[ix(base + b1, A.n, 9u)], R2 = A.comb[ix(base + b2, A.n, 9u)], R3 = A.comb[ix(base + b3, A.n, 9u)];
            pair_slim(A, a, Ai.x, Ai.y, R0.x, R0.y, R0.z, R0.w, true, nf6);
            pair_slim(A, a, Ai.x, Ai.y, R1.x, R1.y, R1.z, R1.w, v1, nf6);
            pair_slim(A, a, Ai.x, Ai.y, R2.x, R2.y, R2.z, R2.w, v2, nf6);
            pair_slim(A, a, Ai.x, Ai.y, R3.x, R3.y, R3.z, R3.w, v3, nf6);
        }
    }
    a.sum *= A.mass * a.inv_rho;
    const float pn = pin_i + A.omega * (src_i - a.sum) / aii_i;
    const bool pos = pn > 0.f;
    A.p_out[i] = pos ? pn : 0.f;
    A.pt_out[i] = pos ? pn / (rho_i * rho_i) : 0.f;
}

// ---- variant (round 4): NO mask decoding, NO row bases.  The list is 16-bit offsets j - i (the neighbours of a
// particle of a cell-sorted array sit within +-(two cell rows) of it: a few thousand slots), flat and row-major like the masks'
// visiting order, padded with 0 = the particle itself, whose pair term is exactly zero -- so a slot costs one v_bfe_i32 / v_ashrrev and
// one add instead of ffs / and / compare / select / add, there is no predicate on a slot, and the sweep's head needs neither the two
// IEEE divisions of the cell index nor the three dependent cell_start loads.  PRED = 1: a slot behind the count is skipped by a branch.
template <int PRED>
__global__ __launch_bounds__(256) void k_off16_slim(Args A)
{
    const uint32_t blk = remap_block(A.nblocks);
    if (blk >= A.nblocks) return;
    const uint32_t i = blk * 256 + threadIdx.x;
    if (i >= A.n) return;
    const float4 Ai = A.comb[i];
    const uint4 q0 = A.off16[i], q1 = A.off16[(size_t)A.n + i];
    const uint32_t cnt = A.cnt8[i];
    const float rho_i = A.rho[i];
    const float aii_i = A.aii[i], src_i = A.src[i], pin_i = A.p_in[i];
    Acc a;
    a.sum = 0.f;
    a.inv_rho = __builtin_amdgcn_rcpf(rho_i);
    a.qx = Ai.z;
    a.qy = Ai.w;
    const float nf6 = 6.f * A.m.nf * A.m.inv2h;
#define OFF_TRIP(WA, WB, S0)                                                                                              \
    {                                                                                                                     \
        const uint32_t j0 = i + (uint32_t)((int)((WA) << 16) >> 16), j1 = i + (uint32_t)((int)(WA) >> 16);                \
        const uint32_t j2 = i + (uint32_t)((int)((WB) << 16) >> 16), j3 = i + (uint32_t)((int)(WB) >> 16);                \
        const float4 R0 = A.comb[ix(j0, A.n, 7u)], R1 = A.comb[ix(j1, A.n, 7u)], R2 = A.comb[ix(j2, A.n, 7u)], R3 = A.comb[ix(j3, A.n, 7u)]; \
        pair_slim(A, a, Ai.x, Ai.y, R0.x, R0.y, R0.z, R0.w, !PRED || (S0) < cnt, nf6);                                    \
        pair_slim(A, a, Ai.x, Ai.y, R1.x, R1.y, R1.z, R1.w, !PRED || (S0) + 1u < cnt, nf6);                               \
        pair_slim(A, a, Ai.x, Ai.y, R2.x, R2.y, R2.z, R2.w, !PRED || (S0) + 2u < cnt, nf6);                               \
        pair_slim(A, a, Ai.x, Ai.y, R3.x, R3.y, R3.z, R3.w, !PRED || (S0) + 3u < cnt, nf6);                               \
    }
    OFF_TRIP(q0.x, q0.y, 0u)
    if (__any(cnt > 4u)) OFF_TRIP(q0.z, q0.w, 4u)
    if (__any(cnt > 8u)) OFF_TRIP(q1.x, q1.y, 8u)
    if (__any(cnt > 12u)) OFF_TRIP(q1.z, q1.w, 12u)
    if (__any(cnt > 16u)) {
        const uint4 q2 = A.off16[2 * (size_t)A.n + i];
        OFF_TRIP(q2.x, q2.y, 16u)
        if (__any(cnt > 20u)) OFF_TRIP(q2.z, q2.w, 20u)
    }
#undef OFF_TRIP
    a.sum *= A.mass * a.inv_rho;
    const float pn = pin_i + A.omega * (src_i - a.sum) / aii_i;
    const bool pos = pn > 0.f;
    A.p_out[i] = pos ? pn : 0.f;
    A.pt_out[i] = pos ? pn / (rho_i * rho_i) : 0.f;
}

// ... the same list for sweep A on its combined record {x, y, p / rho^2, p}
__global__ __launch_bounds__(256) void k_accel_off16(Args A)
{
    const uint32_t blk = remap_block(A.nblocks);
    if (blk >= A.nblocks) return;
    const uint32_t i = blk * 256 + threadIdx.x;
    if (i >= A.n) return;
    const float4 Ri = A.rec[i];
    const uint4 q0 = A.off16[i], q1 = A.off16[(size_t)A.n + i];
    const uint32_t cnt = A.cnt8[i];
    const float nf6 = 6.f * A.m.nf * A.m.inv2h;
    float ax = 0.f, ay = 0.f;
#define OFF_SLOT(J)                                                                                                       \
    {                                                                                                                     \
        const float4 R = A.rec[ix(J, A.n, 8u)];                                                                           \
        const float dx = Ri.x - R.x, dy = Ri.y - R.y;                                                                     \
        const float r2 = fmaxf(dx * dx + dy * dy, 1.0e-30f);                                                              \
        const float rinv = __builtin_amdgcn_rsqf(r2);                                                                     \
        const float q = (r2 * rinv) * A.m.inv2h;                                                                          \
        const float u = fmaxf(1.f - q, 0.f), t = fmaxf(0.5f - q, 0.f);                                                    \
        const float s = (-A.mass * (Ri.z + R.z)) * (nf6 * fmaf(4.f * t, t, -(u * u)) * rinv);                             \
        ax = fmaf(s, dx, ax);                                                                                             \
        ay = fmaf(s, dy, ay);                                                                                             \
    }
#define OFF_TRIP(WA, WB)                                                                                                  \
    {                                                                                                                     \
        const uint32_t j0 = i + (uint32_t)((int)((WA) << 16) >> 16), j1 = i + (uint32_t)((int)(WA) >> 16);                \
        const uint32_t j2 = i + (uint32_t)((int)((WB) << 16) >> 16), j3 = i + (uint32_t)((int)(WB) >> 16);                \
        OFF_SLOT(j0) OFF_SLOT(j1) OFF_SLOT(j2) OFF_SLOT(j3)                                                               \
    }
    OFF_TRIP(q0.x, q0.y)
    if (__any(cnt > 4u)) OFF_TRIP(q0.z, q0.w)
    if (__any(cnt > 8u)) OFF_TRIP(q1.x, q1.y)
    if (__any(cnt > 12u)) OFF_TRIP(q1.z, q1.w)
    if (__any(cnt > 16u)) {
        const uint4 q2 = A.off16[2 * (size_t)A.n + i];
        OFF_TRIP(q2.x, q2.y)
        if (__any(cnt > 20u)) OFF_TRIP(q2.z, q2.w)
    }
#undef OFF_TRIP
#undef OFF_SLOT
    A.pacc_out[i] = make_float4(Ri.x, Ri.y, ax, ay);
}

// ---- round 5: STRUCTURAL variants of the offset-list sweep A -- what the two-batch structure costs (16 384 waves on 1 024 SIMDs at 8
// per SIMD: a sweep is two wave lives end to end, the waves of a batch move through their memory and compute phases in step)
#define ACC_SLOT(RI, R, AX, AY)                                                                                           \
    {                                                                                                                     \
        const float dx = (RI).x - (R).x, dy = (RI).y - (R).y;                                                             \
        const float r2 = fmaxf(dx * dx + dy * dy, 1.0e-30f);                                                              \
        const float rinv = __builtin_amdgcn_rsqf(r2);                                                                     \
        const float q = (r2 * rinv) * A.m.inv2h;                                                                          \
        const float u = __builtin_amdgcn_fmed3f(1.f - q, 0.f, 1.f), t = __builtin_amdgcn_fmed3f(0.5f - q, 0.f, 1.f);      \
        const float s = (-A.mass * ((RI).z + (R).z)) * (nf6 * (fmaf(4.f * t, t, -(u * u)) * rinv));                       \
        AX = fmaf(s, dx, AX);                                                                                             \
        AY = fmaf(s, dy, AY);                                                                                             \
    }
__device__ __forceinline__ float4 rec32(const float4* base, uint32_t j) { return *(const float4*)((const char*)base + (uint32_t)(j << 4)); }
#define LO16(W) ((uint32_t)((int)((W) << 16) >> 16))
#define HI16(W) ((uint32_t)((int)(W) >> 16))
// the product's form: the first twelve gathers leave together, compiled for WAVES waves per SIMD
template <int WAVES>
__global__ __launch_bounds__(256, WAVES) void k_accel_off16_wide(Args A)
{
    const uint32_t blk = remap_block(A.nblocks);
    if (blk >= A.nblocks) return;
    const uint32_t i = blk * 256 + threadIdx.x;
    if (i >= A.n) return;
    const float4 Ri = rec32(A.rec, i);
    const uint4 q0 = A.off16[i], q1 = A.off16[(size_t)A.n + i];
    const uint32_t cnt = A.cnt8[i];
    const float nf6 = 6.f * A.m.nf * A.m.inv2h;
    float ax = 0.f, ay = 0.f;
    const uint32_t w[6] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y};
    float4 R[12];
#pragma unroll
    for (int k = 0; k < 6; k++) {
        R[2 * k] = rec32(A.rec, i + LO16(w[k]));
        R[2 * k + 1] = rec32(A.rec, i + HI16(w[k]));
    }
#pragma unroll
    for (int k = 0; k < 12; k++) ACC_SLOT(Ri, R[k], ax, ay)
    if (__any(cnt > 12u)) {
        const uint32_t v[2] = {q1.z, q1.w};
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const float4 Ra = rec32(A.rec, i + LO16(v[k])), Rb = rec32(A.rec, i + HI16(v[k]));
            ACC_SLOT(Ri, Ra, ax, ay) ACC_SLOT(Ri, Rb, ax, ay)
        }
        if (__any(cnt > 16u)) {
            const uint4 q2 = A.off16[2 * (size_t)A.n + i];
            const uint32_t u4[4] = {q2.x, q2.y, q2.z, q2.w};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const float4 Ra = rec32(A.rec, i + LO16(u4[k])), Rb = rec32(A.rec, i + HI16(u4[k]));
                ACC_SLOT(Ri, Ra, ax, ay) ACC_SLOT(Ri, Rb, ax, ay)
            }
        }
    }
    A.pacc_out[i] = make_float4(Ri.x, Ri.y, ax, ay);
}
// TWO particles per lane (i and i + 256 of a 512-particle tile), half the waves: 24 gathers in flight per lane, one batch of waves
// (lists of more than 12 neighbours: not handled -- the rest lattice and its jitter have 12; the check column shows it)
template <int WAVES>
__global__ __launch_bounds__(256, WAVES) void k_accel_off16_x2(Args A)
{
    const uint32_t nb2 = (A.nblocks + 1) / 2;
    const uint32_t blk = remap_block(nb2);
    if (blk >= nb2) return;
    const uint32_t i0 = blk * 512 + threadIdx.x, i1 = i0 + 256;
    const bool v0 = i0 < A.n, v1 = i1 < A.n;
    const uint32_t c0 = v0 ? i0 : 0u, c1 = v1 ? i1 : 0u;
    const float4 Ra = rec32(A.rec, c0), Rb = rec32(A.rec, c1);
    const uint4 qa0 = A.off16[c0], qa1 = A.off16[(size_t)A.n + c0], qb0 = A.off16[c1], qb1 = A.off16[(size_t)A.n + c1];
    const float nf6 = 6.f * A.m.nf * A.m.inv2h;
    float ax = 0.f, ay = 0.f, bx = 0.f, by = 0.f;
    const uint32_t wa[6] = {qa0.x, qa0.y, qa0.z, qa0.w, qa1.x, qa1.y}, wb[6] = {qb0.x, qb0.y, qb0.z, qb0.w, qb1.x, qb1.y};
    float4 P[12], Q[12];
#pragma unroll
    for (int k = 0; k < 6; k++) {
        P[2 * k] = rec32(A.rec, c0 + LO16(wa[k]));
        P[2 * k + 1] = rec32(A.rec, c0 + HI16(wa[k]));
    }
#pragma unroll
    for (int k = 0; k < 6; k++) {
        Q[2 * k] = rec32(A.rec, c1 + LO16(wb[k]));
        Q[2 * k + 1] = rec32(A.rec, c1 + HI16(wb[k]));
    }
#pragma unroll
    for (int k = 0; k < 12; k++) ACC_SLOT(Ra, P[k], ax, ay)
#pragma unroll
    for (int k = 0; k < 12; k++) ACC_SLOT(Rb, Q[k], bx, by)
    if (v0) A.pacc_out[i0] = make_float4(Ra.x, Ra.y, ax, ay);
    if (v1) A.pacc_out[i1] = make_float4(Rb.x, Rb.y, bx, by);
}
// half the workgroups, each works TWO tiles one after the other; the second tile's own record and list are requested before the first
// tile's pairs are evaluated (its gathers leave as soon as the first tile's arithmetic has been issued)
template <int WAVES>
__global__ __launch_bounds__(256, WAVES) void k_accel_off16_loop2(Args A)
{
    const uint32_t per_xcd = (A.nblocks + 7) >> 3, half = (per_xcd + 1) >> 1;
    const uint32_t x = blockIdx.x & 7u, k0 = blockIdx.x >> 3;
    if (k0 >= half) return;
    const uint32_t t0 = x * per_xcd + k0, t1 = x * per_xcd + k0 + half;
    const bool has1 = k0 + half < per_xcd && t1 < A.nblocks;
    const float nf6 = 6.f * A.m.nf * A.m.inv2h;
    uint32_t i = t0 * 256 + threadIdx.x;
    bool valid = t0 < A.nblocks && i < A.n;
    uint32_t ic = valid ? i : 0u;
    float4 Ri = rec32(A.rec, ic);
    uint4 q0 = A.off16[ic], q1 = A.off16[(size_t)A.n + ic];
    for (int pass = 0; pass < 2; pass++) {
        const uint32_t w[6] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y};
        float4 R[12];
#pragma unroll
        for (int k = 0; k < 6; k++) {
            R[2 * k] = rec32(A.rec, ic + LO16(w[k]));
            R[2 * k + 1] = rec32(A.rec, ic + HI16(w[k]));
        }
        // the next tile's head
        const uint32_t ni = t1 * 256 + threadIdx.x;
        const bool nvalid = pass == 0 && has1 && ni < A.n;
        const uint32_t nc = nvalid ? ni : 0u;
        const float4 nRi = rec32(A.rec, nc);
        const uint4 nq0 = A.off16[nc], nq1 = A.off16[(size_t)A.n + nc];
        float ax = 0.f, ay = 0.f;
#pragma unroll
        for (int k = 0; k < 12; k++) ACC_SLOT(Ri, R[k], ax, ay)
        if (valid) A.pacc_out[i] = make_float4(Ri.x, Ri.y, ax, ay);
        if (pass == 1 || !has1) break;
        i = ni; valid = nvalid; ic = nc; Ri = nRi; q0 = nq0; q1 = nq1;
    }
}

// ---- variant: the three row masks decoded FIRST into up to 16 neighbour indices in registers (row-major, ascending: the same
// order), then trips of 4 over that flat sequence: no padding slot per row, 16 slots for up to 16 neighbours (more: a scalar tail) ----
__global__ __launch_bounds__(256) void k_flat16_slim(Args A)
{
    const uint32_t blk = remap_block(A.nblocks);
    if (blk >= A.nblocks) return;
    const uint32_t i = blk * 256 + threadIdx.x;
    if (i >= A.n) return;
    const float4 Ai = A.comb[i];
    const uint4 lw = A.nl[i];
    const float rho_i = A.rho[i];
    const float aii_i = A.aii[i], src_i = A.src[i], pin_i = A.p_in[i];
    uint32_t rb[3];
    int cx, cy;
    row_bases(A, Ai.x, Ai.y, rb, cx, cy);
    Acc a;
    a.sum = 0.f;
    a.inv_rho = __builtin_amdgcn_rcpf(rho_i);
    a.qx = Ai.z;
    a.qy = Ai.w;
    const float nf6 = 6.f * A.m.nf * A.m.inv2h;
    uint32_t m0 = lw.x, m1 = lw.y, m2 = lw.z;
    const uint32_t cnt = (uint32_t)(__popc(m0) + __popc(m1) + __popc(m2));
    uint32_t j[16];
#pragma unroll
    for (int s = 0; s < 16; s++) {
        const bool u0 = m0 != 0u, u1 = m1 != 0u;
        const uint32_t m = u0 ? m0 : (u1 ? m1 : m2);
        const uint32_t bs = u0 ? rb[0] : (u1 ? rb[1] : rb[2]);
        j[s] = m ? bs + (uint32_t)__ffs(m) - 1u : i;
        const uint32_t mm = m & (m - 1u);
        m0 = u0 ? mm : m0;
        m1 = (!u0 && u1) ? mm : m1;
        m2 = (!u0 && !u1) ? mm : m2;
    }
    const uint32_t wmax = min(16u, (uint32_t)__reduce_max_sync_lab(cnt));
#pragma unroll
    for (uint32_t s0 = 0; s0 < 16; s0 += 4) {   // (wave-uniform trip count: the largest list of the wave; unrolled, registers are not indexable)
        if (s0 >= wmax) break;
        const float4 R0 = A.comb[ix(j[s0], A.n, 7u)], R1 = A.comb[ix(j[s0 + 1], A.n, 7u)], R2 = A.comb[ix(j[s0 + 2], A.n, 7u)],
                     R3 = A.comb[ix(j[s0 + 3], A.n, 7u)];
        pair_slim(A, a, Ai.x, Ai.y, R0.x, R0.y, R0.z, R0.w, s0 < cnt, nf6);
        pair_slim(A, a, Ai.x, Ai.y, R1.x, R1.y, R1.z, R1.w, s0 + 1 < cnt, nf6);
        pair_slim(A, a, Ai.x, Ai.y, R2.x, R2.y, R2.z, R2.w, s0 + 2 < cnt, nf6);
        pair_slim(A, a, Ai.x, Ai.y, R3.x, R3.y, R3.z, R3.w, s0 + 3 < cnt, nf6);
    }
    // more than 16 neighbours: what is left in the masks, one at a time
    while (m0 | m1 | m2) {
        const bool u0 = m0 != 0u, u1 = m1 != 0u;
        const uint32_t m = u0 ? m0 : (u1 ? m1 : m2);
        const uint32_t bs = u0 ? rb[0] : (u1 ? rb[1] : rb[2]);
        const float4 R = A.comb[ix(bs + (uint32_t)__ffs(m) - 1u, A.n, 8u)];
        pair_slim(A, a, Ai.x, Ai.y, R.x, R.y, R.z, R.w, true, nf6);
        const uint32_t mm = m & (m - 1u);
        m0 = u0 ? mm : m0;
        m1 = (!u0 && u1) ? mm : m1;
        m2 = (!u0 && !u1) ? mm : m2;
    }
    a.sum *= A.mass * a.inv_rho;
    const float pn = pin_i + A.omega * (src_i - a.sum) / aii_i;
    const bool pos = pn > 0.f;
    A.p_out[i] = pos ? pn : 0.f;
    A.pt_out[i] = pos ? pn / (rho_i * rho_i) : 0.f;
}

// ---- variant: the masks decoded by three short per-row loops that append the neighbour's index to a per-lane column of LDS
// ([slot][lane]: bank = lane, no conflicts whatever the slot), then flat trips of 4 read back from that column ----
__global__ __launch_bounds__(256) void k_flat_lds_slim(Args A)
{
    __shared__ uint32_t jl[16][256];
    const uint32_t blk = remap_block(A.nblocks);
    if (blk >= A.nblocks) return;
    const uint32_t i = blk * 256 + threadIdx.x;
    if (i >= A.n) return;
    const float4 Ai = A.comb[i];
    const uint4 lw = A.nl[i];
    const float rho_i = A.rho[i];
    const float aii_i = A.aii[i], src_i = A.src[i], pin_i = A.p_in[i];
    uint32_t rb[3];
    int cx, cy;
    row_bases(A, Ai.x, Ai.y, rb, cx, cy);
    Acc a;
    a.sum = 0.f;
    a.inv_rho = __builtin_amdgcn_rcpf(rho_i);
    a.qx = Ai.z;
    a.qy = Ai.w;
    const float nf6 = 6.f * A.m.nf * A.m.inv2h;
    uint32_t k = 0;
#pragma unroll
    for (int r = 0; r < 3; r++) {
        uint32_t m = r == 0 ? lw.x : (r == 1 ? lw.y : lw.z);
        const uint32_t bs = rb[r];
        while (m) {
            const uint32_t j = bs + (uint32_t)__ffs(m) - 1u;
            if (k < 16u) jl[k][threadIdx.x] = j;
            else {
                const float4 R = A.comb[ix(j, A.n, 10u)];
                pair_slim(A, a, Ai.x, Ai.y, R.x, R.y, R.z, R.w, true, nf6);
            }
            k++;
            m &= m - 1u;
        }
    }
    const uint32_t cnt = min(k, 16u);
    const uint32_t wmax = __reduce_max_sync_lab(cnt);
#pragma unroll
    for (uint32_t s0 = 0; s0 < 16; s0 += 4) {
        if (s0 >= wmax) break;
        const uint32_t q0 = s0 < cnt ? jl[s0][threadIdx.x] : i, q1 = s0 + 1 < cnt ? jl[s0 + 1][threadIdx.x] : i,
                       q2 = s0 + 2 < cnt ? jl[s0 + 2][threadIdx.x] : i, q3 = s0 + 3 < cnt ? jl[s0 + 3][threadIdx.x] : i;
        const float4 R0 = A.comb[ix(q0, A.n, 11u)], R1 = A.comb[ix(q1, A.n, 11u)], R2 = A.comb[ix(q2, A.n, 11u)], R3 = A.comb[ix(q3, A.n, 11u)];
        pair_slim(A, a, Ai.x, Ai.y, R0.x, R0.y, R0.z, R0.w, s0 < cnt, nf6);
        pair_slim(A, a, Ai.x, Ai.y, R1.x, R1.y, R1.z, R1.w, s0 + 1 < cnt, nf6);
        pair_slim(A, a, Ai.x, Ai.y, R2.x, R2.y, R2.z, R2.w, s0 + 2 < cnt, nf6);
        pair_slim(A, a, Ai.x, Ai.y, R3.x, R3.y, R3.z, R3.w, s0 + 3 < cnt, nf6);
    }
    a.sum *= A.mass * a.inv_rho;
    const float pn = pin_i + A.omega * (src_i - a.sum) / aii_i;
    const bool pos = pn > 0.f;
    A.p_out[i] = pos ? pn : 0.f;
    A.pt_out[i] = pos ? pn / (rho_i * rho_i) : 0.f;
}

// ---- variant: trips of 4 per row (as the product), the gathers of trip k + 1 requested BEFORE the pairs of trip k are evaluated ----
__global__ __launch_bounds__(256) void k_gather4_slim_pipe(Args A)
{
    const uint32_t blk = remap_block(A.nblocks);
    if (blk >= A.nblocks) return;
    const uint32_t i = blk * 256 + threadIdx.x;
    if (i >= A.n) return;
    const float4 Ai = A.comb[i];
    const uint4 lw = A.nl[i];
    const float rho_i = A.rho[i];
    const float aii_i = A.aii[i], src_i = A.src[i], pin_i = A.p_in[i];
    uint32_t rb[3];
    int cx, cy;
    row_bases(A, Ai.x, Ai.y, rb, cx, cy);
    Acc a;
    a.sum = 0.f;
    a.inv_rho = __builtin_amdgcn_rcpf(rho_i);
    a.qx = Ai.z;
    a.qy = Ai.w;
    const float nf6 = 6.f * A.m.nf * A.m.inv2h;
    uint32_t m0 = lw.x, m1 = lw.y, m2 = lw.z;
    // one trip = the next (up to) 4 set bits of the first non-empty row; returns false when nothing is left
    float4 R0, R1, R2, R3;
    bool v0, v1, v2, v3;
    auto fetch = [&](float4& S0, float4& S1, float4& S2, float4& S3, bool& w0, bool& w1, bool& w2, bool& w3) {
        const bool u0 = m0 != 0u, u1 = m1 != 0u;
        uint32_t m = u0 ? m0 : (u1 ? m1 : m2);
        const uint32_t bs = u0 ? rb[0] : (u1 ? rb[1] : rb[2]);
        w0 = m != 0u;
        const uint32_t b0 = w0 ? (uint32_t)__ffs(m) - 1u : 0u;
        m &= m - 1u;
        w1 = m != 0u;
        const uint32_t b1 = w1 ? (uint32_t)__ffs(m) - 1u : b0;
        m &= m - 1u;
        w2 = m != 0u;
        const uint32_t b2 = w2 ? (uint32_t)__ffs(m) - 1u : b0;
        m &= m - 1u;
        w3 = m != 0u;
        const uint32_t b3 = w3 ? (uint32_t)__ffs(m) - 1u : b0;
        m &= m - 1u;
        m0 = u0 ? m : m0;
        m1 = (!u0 && u1) ? m : m1;
        m2 = (!u0 && !u1) ? m : m2;
        const uint32_t safe = w0 ? bs : i;
        S0 = A.comb[ix(safe + (w0 ? b0 : 0u), A.n, 9u)];
        S1 = A.comb[ix(safe + (w0 ? b1 : 0u), A.n, 9u)];
        S2 = A.comb[ix(safe + (w0 ? b2 : 0u), A.n, 9u)];
        S3 = A.comb[ix(safe + (w0 ? b3 : 0u), A.n, 9u)];
    };
    fetch(R0, R1, R2, R3, v0, v1, v2, v3);
    while (__any(v0)) {
        float4 N0, N1, N2, N3;
        bool n0, n1, n2, n3;
        fetch(N0, N1, N2, N3, n0, n1, n2, n3);
        pair_slim(A, a, Ai.x, Ai.y, R0.x, R0.y, R0.z, R0.w, v0, nf6);
        pair_slim(A, a, Ai.x, Ai.y, R1.x, R1.y, R1.z, R1.w, v1, nf6);
        pair_slim(A, a, Ai.x, Ai.y, R2.x, R2.y, R2.z, R2.w, v2, nf6);
        pair_slim(A, a, Ai.x, Ai.y, R3.x, R3.y, R3.z, R3.w, v3, nf6);
        R0 = N0; R1 = N1; R2 = N2; R3 = N3;
        v0 = n0; v1 = n1; v2 = n2; v3 = n3;
    }
    a.sum *= A.mass * a.inv_rho;
    const float pn = pin_i + A.omega * (src_i - a.sum) / aii_i;
    const bool pos = pn > 0.f;
    A.p_out[i] = pos ? pn : 0.f;
    A.pt_out[i] = pos ? pn / (rho_i * rho_i) : 0.f;
}

// ---- variant: no neighbour work at all (own loads, finish, stores): the floor of the launch ----
__global__ __launch_bounds__(256) void k_own_only(Args A)
{
    const uint32_t blk = remap_block(A.nblocks);
    if (blk >= A.nblocks) return;
    const uint32_t i = blk * 256 + threadIdx.x;
    if (i >= A.n) return;
    const float4 Ai = A.pm[i];
    const uint4 lw = A.nl[i];
    Acc a;
    a.sum = Ai.x * 1e-30f + (float)lw.x * 1e-30f;
    const float rho_i = A.rho[i];
    a.inv_rho = __builtin_amdgcn_rcpf(rho_i);
    const float2 q = A.pacc[i];
    a.sum += q.x * 1e-30f;
    finish(A, a, i, rho_i);
}

// ---- variants on LDS: every wave stages the three candidate windows of its 64 particles (contiguous index ranges: the cells
// cx_first - 1 .. cx_last + 1 of the rows cy - 1, cy, cy + 1) as compact {x, y, a^p} records, coalesced, into its own LDS region;
// the lanes then read neighbours from LDS.  A wave that straddles two cell rows, or whose window exceeds CAP, gathers from memory.
//   LOOP = 0: the product's trips (4 set bits of a row per trip, padding slots for lanes with fewer)
//   LOOP = 1: one set bit per iteration, no padding slots (the wave runs max-over-lanes(popcount) iterations per row)
//   LOOP = 2: two set bits per iteration
#ifndef CAP
#define CAP 96
#endif
template <int LOOP>
__global__ __launch_bounds__(256) void k_lds(Args A)
{
    __shared__ float4 s_rec[4][3 * CAP];
    const uint32_t blk = remap_block(A.nblocks);
    if (blk >= A.nblocks) return;
    const uint32_t w = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t wbase = blk * 256 + w * 64u;
    if (wbase >= A.n) return;
    const uint32_t nvalid = min(64u, A.n - wbase);
    const uint32_t i = wbase + min(lane, nvalid - 1u);
    const bool active = lane < nvalid;
    const float4 Ai = A.pm[i];
    const uint4 lw = A.nl[i];
    uint32_t rb[3];
    int cx, cy;
    row_bases(A, Ai.x, Ai.y, rb, cx, cy);
    const int cxf = __builtin_amdgcn_readfirstlane(cx), cyf = __builtin_amdgcn_readfirstlane(cy);
    const int cxl = __builtin_amdgcn_readlane(cx, (int)nvalid - 1), cyl = __builtin_amdgcn_readlane(cy, (int)nvalid - 1);
    bool ok = cyf == cyl;
    uint32_t sb[3], sl[3];
#pragma unroll
    for (int dr = 0; dr < 3; dr++) {
        const int yy = cyf + dr - 1;
        const bool in = yy >= 0 && yy < A.g.sy;
        const uint32_t row = (uint32_t)(in ? yy : 0) * (uint32_t)A.g.sx;
        sb[dr] = in ? A.cell_start[ix(row + (uint32_t)max(cxf - 1, 0), (uint32_t)A.g.sx * (uint32_t)A.g.sy + 1u, 5u)] : 0u;
        const uint32_t se = in ? A.cell_start[ix(row + (uint32_t)min(cxl + 2, A.g.sx), (uint32_t)A.g.sx * (uint32_t)A.g.sy + 1u, 6u)] : sb[dr];
        sl[dr] = se - sb[dr];
        ok = ok && sl[dr] <= (uint32_t)CAP;
    }
    Acc a;
    a.sum = 0.f;
    const float rho_i = A.rho[i];
    a.inv_rho = __builtin_amdgcn_rcpf(rho_i);
    const float2 q = A.pacc[i];
    a.qx = q.x;
    a.qy = q.y;
    const uint32_t masks[3] = {lw.x, lw.y, lw.z};
    if (!ok) {   // (wave-uniform) gather form
        if (active) {
#pragma unroll
            for (int dr = 0; dr < 3; dr++) {
                uint32_t mk = masks[dr];
                while (mk) {
                    const uint32_t b0 = __ffs(mk) - 1;
                    mk &= mk - 1;
                    const float4 R = A.pm[ix(rb[dr] + b0, A.n, 3u)];
                    const float2 P = A.pacc[ix(rb[dr] + b0, A.n, 3u)];
                    pair(A, a, Ai.x, Ai.y, R.x, R.y, P.x, P.y, true);
                }
            }
            finish(A, a, i, rho_i);
        }
        return;
    }
    // stage: all loads of the wave in flight, then the LDS writes
    float4 r_[3][2];
    float2 p_[3][2];
#pragma unroll
    for (int dr = 0; dr < 3; dr++) {
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const uint32_t k = lane + 64u * t;
            const uint32_t j = sl[dr] ? sb[dr] + min(k, sl[dr] - 1u) : 0u;   // (an empty window may start at n: cells behind the last particle)
            r_[dr][t] = A.pm[ix(j, A.n, 4u)];
            p_[dr][t] = A.pacc[ix(j, A.n, 4u)];
        }
    }
#pragma unroll
    for (int dr = 0; dr < 3; dr++) {
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const uint32_t k = lane + 64u * t;
            if (k < sl[dr]) s_rec[w][dr * CAP + k] = make_float4(r_[dr][t].x, r_[dr][t].y, p_[dr][t].x, p_[dr][t].y);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (!active) return;
#pragma unroll
    for (int dr = 0; dr < 3; dr++) {
        uint32_t mk = masks[dr];
        const float4* __restrict__ S = &s_rec[w][dr * CAP + (rb[dr] - sb[dr])];
        if (LOOP == 0) {
            while (mk) {
                const uint32_t b0 = __ffs(mk) - 1;
                mk &= mk - 1;
                const bool v1 = mk != 0;
                const uint32_t b1 = v1 ? __ffs(mk) - 1 : b0;
                mk &= mk - 1;
                const bool v2 = mk != 0;
                const uint32_t b2 = v2 ? __ffs(mk) - 1 : b0;
                mk &= mk - 1;
                const bool v3 = mk != 0;
                const uint32_t b3 = v3 ? __ffs(mk) - 1 : b0;
                mk &= mk - 1;
                const float4 R0 = S[b0], R1 = S[b1], R2 = S[b2], R3 = S[b3];
                pair(A, a, Ai.x, Ai.y, R0.x, R0.y, R0.z, R0.w, true);
                pair(A, a, Ai.x, Ai.y, R1.x, R1.y, R1.z, R1.w, v1);
                pair(A, a, Ai.x, Ai.y, R2.x, R2.y, R2.z, R2.w, v2);
                pair(A, a, Ai.x, Ai.y, R3.x, R3.y, R3.z, R3.w, v3);
            }
        } else if (LOOP == 1) {
            while (mk) {
                const uint32_t b0 = __ffs(mk) - 1;
                mk &= mk - 1;
                const float4 R0 = S[b0];
                pair(A, a, Ai.x, Ai.y, R0.x, R0.y, R0.z, R0.w, true);
            }
        } else {
            while (mk) {
                const uint32_t b0 = __ffs(mk) - 1;
                mk &= mk - 1;
                const bool v1 = mk != 0;
                const uint32_t b1 = v1 ? __ffs(mk) - 1 : b0;
                mk &= mk - 1;
                const float4 R0 = S[b0], R1 = S[b1];
                pair(A, a, Ai.x, Ai.y, R0.x, R0.y, R0.z, R0.w, true);
                pair(A, a, Ai.x, Ai.y, R1.x, R1.y, R1.z, R1.w, v1);
            }
        }
    }
    finish(A, a, i, rho_i);
}

// ---- variant: gathers, one / two set bits per iteration (no padding slots; fewer loads in flight) ----
template <int PER>
__global__ __launch_bounds__(256) void k_gather_loop(Args A)
{
    const uint32_t blk = remap_block(A.nblocks);
    if (blk >= A.nblocks) return;
    const uint32_t i = blk * 256 + threadIdx.x;
    if (i >= A.n) return;
    const float4 Ai = A.pm[i];
    const uint4 lw = A.nl[i];
    uint32_t rb[3];
    int cx, cy;
    row_bases(A, Ai.x, Ai.y, rb, cx, cy);
    Acc a;
    a.sum = 0.f;
    const float rho_i = A.rho[i];
    a.inv_rho = __builtin_amdgcn_rcpf(rho_i);
    const float2 q = A.pacc[i];
    a.qx = q.x;
    a.qy = q.y;
    const uint32_t masks[3] = {lw.x, lw.y, lw.z};
#pragma unroll
    for (int dr = 0; dr < 3; dr++) {
        uint32_t mk = masks[dr];
        const uint32_t base = rb[dr];
        while (mk) {
            const uint32_t b0 = __ffs(mk) - 1;
            mk &= mk - 1;
            if (PER == 1) {
                const float4 R0 = A.comb[ix(base + b0, A.n, 2u)];
                pair(A, a, Ai.x, Ai.y, R0.x, R0.y, R0.z, R0.w, true);
            } else {
                const bool v1 = mk != 0;
                const uint32_t b1 = v1 ? __ffs(mk) - 1 : b0;
                mk &= mk - 1;
                const float4 R0 = A.comb[ix(base + b0, A.n, 2u)], R1 = A.comb[ix(base + b1, A.n, 2u)];
                pair(A, a, Ai.x, Ai.y, R0.x, R0.y, R0.z, R0.w, true);
                pair(A, a, Ai.x, Ai.y, R1.x, R1.y, R1.z, R1.w, v1);
            }
        }
    }
    finish(A, a, i, rho_i);
}

// ---- variant: the pair arithmetic only -- every "neighbour" is a register value, no loads, no mask decoding: 20 pair slots per
// particle as the product executes them (what the VALU alone costs) ----
template <int SLOTS>
__global__ __launch_bounds__(256) void k_math_only(Args A)
{
    const uint32_t blk = remap_block(A.nblocks);
    if (blk >= A.nblocks) return;
    const uint32_t i = blk * 256 + threadIdx.x;
    if (i >= A.n) return;
    const float4 Ai = A.pm[i];
    Acc a;
    a.sum = 0.f;
    const float rho_i = A.rho[i];
    a.inv_rho = __builtin_amdgcn_rcpf(rho_i);
    const float2 q = A.pacc[i];
    a.qx = q.x;
    a.qy = q.y;
    float xj = Ai.x + 0.0007f, yj = Ai.y - 0.0004f;
#pragma unroll 4
    for (int s = 0; s < SLOTS; s++) {
        pair(A, a, Ai.x, Ai.y, xj, yj, q.y, q.x, true);
        xj += 1e-5f;
        yj -= 1e-5f;
    }
    finish(A, a, i, rho_i);
}

template <int SLOTS>
__global__ __launch_bounds__(256) void k_math_only_slim(Args A)
{
    const uint32_t blk = remap_block(A.nblocks);
    if (blk >= A.nblocks) return;
    const uint32_t i = blk * 256 + threadIdx.x;
    if (i >= A.n) return;
    const float4 Ai = A.pm[i];
    Acc a;
    a.sum = 0.f;
    const float rho_i = A.rho[i];
    a.inv_rho = __builtin_amdgcn_rcpf(rho_i);
    const float2 q = A.pacc[i];
    a.qx = q.x;
    a.qy = q.y;
    const float nf6 = 6.f * A.m.nf * A.m.inv2h;
    float xj = Ai.x + 0.0007f, yj = Ai.y - 0.0004f;
#pragma unroll 4
    for (int s = 0; s < SLOTS; s++) {
        pair_slim(A, a, Ai.x, Ai.y, xj, yj, q.y, q.x, true, nf6);
        xj += 1e-5f;
        yj -= 1e-5f;
    }
    finish(A, a, i, rho_i);
}

// ---- sweep A (OpPressureAccel, uniform h): a^p_i = - sum_j m (p_i / rho_i^2 + p_j / rho_j^2) grad W_ij, written as {x, y, a^p} ----
// COMB = 0: the product's form -- the neighbour's 16-B record {x, y, m, h} and its 4-B p / rho^2, two gathers per neighbour slot;
// COMB = 1: ONE 16-B gather of a combined record {x, y, p / rho^2, p} (what sweep B would have to store instead of p / rho^2 alone)
template <int COMB>
__global__ __launch_bounds__(256) void k_accel(Args A)
{
    const uint32_t blk = remap_block(A.nblocks);
    if (blk >= A.nblocks) return;
    const uint32_t i = blk * 256 + threadIdx.x;
    if (i >= A.n) return;
    const float4 Ai = COMB ? A.rec[i] : A.pm[i];
    const uint4 lw = A.nl[i];
    const float pti = COMB ? Ai.z : A.pt[i];
    uint32_t rb[3];
    int cx, cy;
    row_bases(A, Ai.x, Ai.y, rb, cx, cy);
    const float nf6 = 6.f * A.m.nf * A.m.inv2h;
    float ax = 0.f, ay = 0.f;
    const uint32_t masks[3] = {lw.x, lw.y, lw.z};
#pragma unroll
    for (int dr = 0; dr < 3; dr++) {
        uint32_t mk = masks[dr];
        const uint32_t base = rb[dr];
        while (mk) {
            uint32_t b[4];
            bool v[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                v[k] = mk != 0u;
                b[k] = v[k] ? (uint32_t)__ffs(mk) - 1u : b[0];
                mk &= mk - 1u;
            }
            float4 R[4];
            float T[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t j = ix(base + b[k], A.n, 20u);
                if (COMB) {
                    R[k] = A.rec[j];
                    T[k] = R[k].z;
                } else {
                    R[k] = A.pm[j];
                    T[k] = A.pt[j];
                }
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const float dx = Ai.x - R[k].x, dy = Ai.y - R[k].y;
                const float r2 = fmaxf(dx * dx + dy * dy, 1.0e-30f);
                if (v[k]) {
                    const float rinv = __builtin_amdgcn_rsqf(r2);
                    const float q = (r2 * rinv) * A.m.inv2h;
                    const float u = fmaxf(1.f - q, 0.f), t = fmaxf(0.5f - q, 0.f);
                    const float d = fmaf(4.f * t, t, -(u * u));
                    const float fs = (-A.mass * (pti + T[k])) * (nf6 * d * rinv);
                    ax = fmaf(fs, dx, ax);
                    ay = fmaf(fs, dy, ay);
                }
            }
        }
    }
    A.pacc_out[i] = make_float4(Ai.x, Ai.y, ax, ay);
}

// sweep A on the combined record without a branch around the pairs
__global__ __launch_bounds__(256) void k_accel_nobranch(Args A)
{
    const uint32_t blk = remap_block(A.nblocks);
    if (blk >= A.nblocks) return;
    const uint32_t i = blk * 256 + threadIdx.x;
    if (i >= A.n) return;
    const float4 Ai = A.rec[i];
    const uint4 lw = A.nl[i];
    const float pti = Ai.z;
    uint32_t rb[3];
    int cx, cy;
    row_bases(A, Ai.x, Ai.y, rb, cx, cy);
    const float nf6 = 6.f * A.m.nf * A.m.inv2h;
    float ax = 0.f, ay = 0.f;
    const uint32_t masks[3] = {lw.x, lw.y, lw.z};
#pragma unroll
    for (int dr = 0; dr < 3; dr++) {
        uint32_t mk = masks[dr];
        const uint32_t base = rb[dr];
        while (mk) {
            uint32_t b[4];
            bool v[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                v[k] = mk != 0u;
                b[k] = v[k] ? (uint32_t)__ffs(mk) - 1u : b[0];
                mk &= mk - 1u;
            }
            float4 R[4];
#pragma unroll
            for (int k = 0; k < 4; k++) R[k] = A.rec[ix(base + b[k], A.n, 31u)];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const float dx = Ai.x - R[k].x, dy = Ai.y - R[k].y;
                const float r2 = fmaxf(dx * dx + dy * dy, 1.0e-30f);
                const float rinv = __builtin_amdgcn_rsqf(r2);
                const float q = (r2 * rinv) * A.m.inv2h;
                const float u = fmaxf(1.f - q, 0.f), t = fmaxf(0.5f - q, 0.f);
                const float d = fmaf(4.f * t, t, -(u * u));
                const float fs = v[k] ? (-A.mass * (pti + R[k].z)) * (nf6 * d * rinv) : 0.f;
                ax = fmaf(fs, dx, ax);
                ay = fmaf(fs, dy, ay);
            }
        }
    }
    A.pacc_out[i] = make_float4(Ai.x, Ai.y, ax, ay);
}

// sweep B's stores with the combined record: p (4 B) and {x, y, p / rho^2, p} (16 B) instead of p and p / rho^2 (4 B + 4 B):
// variant `gather4_slim<1>` with that store, to price the 12 extra bytes per particle
__global__ __launch_bounds__(256) void k_gather4_slim_recstore(Args A)
{
    const uint32_t blk = remap_block(A.nblocks);
    if (blk >= A.nblocks) return;
    const uint32_t i = blk * 256 + threadIdx.x;
    if (i >= A.n) return;
    const float4 Ai = A.comb[i];
    const uint4 lw = A.nl[i];
    const float rho_i = A.rho[i];
    const float aii_i = A.aii[i], src_i = A.src[i], pin_i = A.p_in[i];
    uint32_t rb[3];
    int cx, cy;
    row_bases(A, Ai.x, Ai.y, rb, cx, cy);
    Acc a;
    a.sum = 0.f;
    a.inv_rho = __builtin_amdgcn_rcpf(rho_i);
    a.qx = Ai.z;
    a.qy = Ai.w;
    const float nf6 = 6.f * A.m.nf * A.m.inv2h;
    const uint32_t masks[3] = {lw.x, lw.y, lw.z};
#pragma unroll
    for (int dr = 0; dr < 3; dr++) {
        uint32_t mk = masks[dr];
        const uint32_t base = rb[dr];
        while (mk) {
            uint32_t b[4];
            bool v[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                v[k] = mk != 0u;
                b[k] = v[k] ? (uint32_t)__ffs(mk) - 1u : b[0];
                mk &= mk - 1u;
            }
            float4 R[4];
#pragma unroll
            for (int k = 0; k < 4; k++) R[k] = A.comb[ix(base + b[k], A.n, 21u)];
#pragma unroll
            for (int k = 0; k < 4; k++) pair_slim(A, a, Ai.x, Ai.y, R[k].x, R[k].y, R[k].z, R[k].w, v[k], nf6);
        }
    }
    a.sum *= A.mass * a.inv_rho;
    const float pn = pin_i + A.omega * (src_i - a.sum) / aii_i;
    const bool pos = pn > 0.f;
    const float po = pos ? pn : 0.f;
    A.p_out[i] = po;
    const_cast<float4*>(A.rec)[i] = make_float4(Ai.x, Ai.y, pos ? pn / (rho_i * rho_i) : 0.f, po);
}

int main(int argc, char** argv)
{
    const int side = argc > 1 ? atoi(argv[1]) : 1024;
    const float jitter = argc > 2 ? (float)atof(argv[2]) : 0.15f;
    const int reps = argc > 3 ? atoi(argv[3]) : 50;
    const uint32_t n = (uint32_t)side * (uint32_t)side;
    const float d = 1.f / 1024.f, rho0 = 1.f, mass = rho0 * d * d;
    const float h = 1.9f * sqrtf((mass / rho0) * 0.318309873342514038086f);
    const float cs = 2.f * h;
    std::mt19937 rng(1234);
    std::uniform_real_distribution<float> U(-jitter * d, jitter * d);
    std::vector<float> x(n), y(n);
    for (int r = 0; r < side; r++)
        for (int c = 0; c < side; c++) {
            x[(size_t)r * side + c] = -1.9995f + c * d + U(rng);
            y[(size_t)r * side + c] = -0.9995f + r * d + U(rng);
        }
    float mnx = 1e9f, mny = 1e9f, mxx = -1e9f, mxy = -1e9f;
    for (uint32_t i = 0; i < n; i++) {
        mnx = std::min(mnx, x[i]); mny = std::min(mny, y[i]); mxx = std::max(mxx, x[i]); mxy = std::max(mxy, y[i]);
    }
    Grid g;
    g.cs = cs;
    g.minx = (int)floorf(mnx / cs) - 1;
    g.miny = (int)floorf(mny / cs) - 1;
    g.sx = (int)floorf(mxx / cs) + 2 - g.minx;
    g.sy = (int)floorf(mxy / cs) + 2 - g.miny;
    const uint32_t ncells = (uint32_t)g.sx * (uint32_t)g.sy;
    std::vector<uint32_t> key(n), perm(n);
    for (uint32_t i = 0; i < n; i++) {
        const int cx = (int)floorf(x[i] / cs) - g.minx, cy = (int)floorf(y[i] / cs) - g.miny;
        key[i] = (uint32_t)cy * (uint32_t)g.sx + (uint32_t)cx;
        perm[i] = i;
    }
    std::stable_sort(perm.begin(), perm.end(), [&](uint32_t a, uint32_t b) { return key[a] < key[b]; });
    std::vector<float4> pm(n), comb(n);
    std::vector<float2> pacc(n);
    std::vector<uint32_t> cell_start(ncells + 1, 0), skey(n);
    std::uniform_real_distribution<float> V(-1.f, 1.f);
    for (uint32_t s = 0; s < n; s++) {
        const uint32_t i = perm[s];
        pm[s] = make_float4(x[i], y[i], mass, h);
        pacc[s] = make_float2(10.f * V(rng), 10.f * V(rng));
        comb[s] = make_float4(x[i], y[i], pacc[s].x, pacc[s].y);
        skey[s] = key[i];
        cell_start[key[i] + 1]++;
    }
    for (uint32_t c = 0; c < ncells; c++) cell_start[c + 1] += cell_start[c];
    // list words
    std::vector<uint4> nl(n);
    double sum_cnt = 0, sum_slots4 = 0;
    uint32_t overflow = 0;
    for (uint32_t s = 0; s < n; s++) {
        const int cx = (int)(skey[s] % (uint32_t)g.sx), cy = (int)(skey[s] / (uint32_t)g.sx);
        uint32_t mk[3] = {0, 0, 0}, cnt = 0;
        for (int dr = 0; dr < 3; dr++) {
            const int yy = cy + dr - 1;
            if (yy < 0 || yy >= g.sy) continue;
            const uint32_t b = cell_start[(uint32_t)yy * g.sx + std::max(cx - 1, 0)], e = cell_start[(uint32_t)yy * g.sx + std::min(cx + 2, g.sx)];
            if (e - b > 32) overflow++;
            for (uint32_t j = b; j < e && j - b < 32; j++) {
                const float dx = pm[s].x - pm[j].x, dy = pm[s].y - pm[j].y;
                const float sr = ((h + h) * 0.5f) * 2.f;
                if (dx * dx + dy * dy < sr * sr && j != s) {
                    mk[dr] |= 1u << (j - b);
                    cnt++;
                }
            }
            sum_slots4 += 4 * ((__builtin_popcount(mk[dr]) + 3) / 4);
        }
        nl[s] = make_uint4(mk[0], mk[1], mk[2], cnt);
        sum_cnt += cnt;
    }
    // relative-offset lists: the masks' visiting order (rows bottom to top, index ascending) as 16-bit j - i, 24 slots
    std::vector<uint16_t> off16((size_t)n * 24, 0);
    std::vector<uint8_t> cnt8(n);
    std::vector<uint32_t> rbd(n);
    std::vector<uint32_t> dlt((size_t)n * 6, 0);
    std::vector<uint8_t> dcnt(n);
    uint32_t dlt_overflow = 0;
    double sum_groups = 0;
    uint32_t off_overflow = 0, max_cnt = 0;
    for (uint32_t s = 0; s < n; s++) {
        const int cx = (int)(skey[s] % (uint32_t)g.sx), cy = (int)(skey[s] / (uint32_t)g.sx);
        uint32_t k = 0;
        const uint32_t mk[3] = {nl[s].x, nl[s].y, nl[s].z};
        for (int dr = 0; dr < 3; dr++) {
            const int yy = cy + dr - 1;
            if (yy < 0 || yy >= g.sy) continue;
            const uint32_t b = cell_start[(uint32_t)yy * g.sx + std::max(cx - 1, 0)];
            for (uint32_t bit = 0; bit < 32; bit++)
                if (mk[dr] & (1u << bit)) {
                    const long long d = (long long)(b + bit) - (long long)s;
                    if (d < -32768 || d > 32767 || k >= 24) { off_overflow++; continue; }
                    // quad g = k / 8 of particle s: 8 halfwords at off16[(g n + s) 8 ..]
                    off16[((size_t)(k / 8) * n + s) * 8 + (k % 8)] = (uint16_t)(int16_t)d;
                    k++;
                }
        }
        cnt8[s] = (uint8_t)k;
        max_cnt = std::max(max_cnt, k);
        {   // row-delta groups
            uint32_t ng = 0;
            for (int dr = 0; dr < 3; dr++) {
                const int yy = cy + dr - 1;
                if (yy < 0 || yy >= g.sy) continue;
                const uint32_t b = cell_start[(uint32_t)yy * g.sx + std::max(cx - 1, 0)];
                uint32_t word = 0, in_group = 0, last = 0;
                for (uint32_t bit = 0; bit < 32; bit++)
                    if (mk[dr] & (1u << bit)) {
                        const uint32_t j = b + bit;
                        if (in_group == 0) {
                            word = (uint32_t)((long long)j - (long long)s) & 0xffffu;
                        } else {
                            word |= (j - last) << (16 + 5 * (in_group - 1));
                        }
                        last = j;
                        if (++in_group == 4) {
                            if (ng < 6) dlt[(size_t)ng * n + s] = word; else dlt_overflow++;
                            ng++;
                            in_group = 0;
                        }
                    }
                if (in_group) {
                    if (ng < 6) dlt[(size_t)ng * n + s] = word; else dlt_overflow++;
                    ng++;
                }
            }
            dcnt[s] = (uint8_t)std::min(ng, 6u);
            sum_groups += ng;
        }
        {
            uint32_t rbv[3] = {0, 0, 0};
            for (int dr = 0; dr < 3; dr++) {
                const int yy = cy + dr - 1;
                if (yy >= 0 && yy < g.sy) rbv[dr] = cell_start[(uint32_t)yy * g.sx + std::max(cx - 1, 0)];
                else rbv[dr] = s;
            }
            const long long d0 = (long long)rbv[0] - s, d2 = (long long)rbv[2] - s, sb = (long long)s - rbv[1];
            if (d0 < -32768 || d0 > 32767 || d2 < -32768 || d2 > 32767 || sb < 0 || sb > 31) off_overflow++;
            rbd[s] = ((uint32_t)d0 & 0xffffu) | ((uint32_t)d2 << 16);
            nl[s].w |= ((uint32_t)sb & 31u) << 8;
        }
    }
    setvbuf(stdout, nullptr, _IOLBF, 0);
    printf("offset lists: largest count %u, entries that do not fit (24 slots, 16 bits): %u; row-delta lists: %.2f groups per particle, groups beyond 6: %u\n", max_cnt, off_overflow, sum_groups / n, dlt_overflow);
    printf("n = %u, cells %d x %d, %.2f neighbours per particle (self excluded), %.2f slots per particle in trips of 4 (per lane), rows > 32 candidates: %u\n", n, g.sx,
           g.sy, sum_cnt / n, sum_slots4 / n, overflow);
    std::vector<float> rho(n), aii(n), src(n), pin(n);
    for (uint32_t s = 0; s < n; s++) {
        rho[s] = 1.f + 0.01f * V(rng);
        aii[s] = 5.0e4f * (1.f + 0.1f * V(rng));
        src[s] = 100.f * V(rng);
        pin[s] = 1.0f + V(rng);
    }
    Args A{};
    A.n = n;
    A.nblocks = (n + 255) / 256;
    A.g = g;
    A.m = Math{h, 10.f / (7.f * 3.14159274101257324219f * (h * h)), 1.f / (2.f * h)};
    A.omega = 0.5f;
    A.mass = mass;
    auto up = [&](const void* src_, size_t bytes) {
        void* p;
        CHECK(hipMalloc(&p, bytes));
        CHECK(hipMemcpy(p, src_, bytes, hipMemcpyHostToDevice));
        return p;
    };
    A.cell_start = (const uint32_t*)up(cell_start.data(), cell_start.size() * 4);
    A.nl = (const uint4*)up(nl.data(), (size_t)n * 16);
    A.pm = (const float4*)up(pm.data(), (size_t)n * 16);
    A.pacc = (const float2*)up(pacc.data(), (size_t)n * 8);
    {   // (one element in front: the lean variants gather index base - 1 in an empty slot)
        std::vector<float4> padded(n + 1);
        padded[0] = comb[0];
        std::copy(comb.begin(), comb.end(), padded.begin() + 1);
        A.comb = (const float4*)up(padded.data(), (size_t)(n + 1) * 16) + 1;
    }
    A.off16 = (const uint4*)up(off16.data(), off16.size() * 2);
    A.cnt8 = (const uint8_t*)up(cnt8.data(), (size_t)n);
    A.rbd = (const uint32_t*)up(rbd.data(), (size_t)n * 4);
    A.dlt = (const uint32_t*)up(dlt.data(), dlt.size() * 4);
    A.dcnt = (const uint8_t*)up(dcnt.data(), (size_t)n);
    A.rho = (const float*)up(rho.data(), (size_t)n * 4);
    A.aii = (const float*)up(aii.data(), (size_t)n * 4);
    A.src = (const float*)up(src.data(), (size_t)n * 4);
    A.p_in = (const float*)up(pin.data(), (size_t)n * 4);
    {
        void *po, *pt;
        CHECK(hipMalloc(&po, (size_t)n * 4));
        CHECK(hipMalloc(&pt, (size_t)n * 4));
        A.p_out = (float*)po;
        A.pt_out = (float*)pt;
    }
    {
        std::vector<float> pt(n);
        std::vector<float4> rec(n);
        for (uint32_t s = 0; s < n; s++) {
            pt[s] = pin[s] / (rho[s] * rho[s]);
            rec[s] = make_float4(pm[s].x, pm[s].y, pt[s], pin[s]);
        }
        A.pt = (const float*)up(pt.data(), (size_t)n * 4);
        {
            std::vector<float4> padded(n + 1);
            padded[0] = rec[0];
            std::copy(rec.begin(), rec.end(), padded.begin() + 1);
            A.rec = (const float4*)up(padded.data(), (size_t)(n + 1) * 16) + 1;
        }
        void* po;
        CHECK(hipMalloc(&po, (size_t)n * 16));
        A.pacc_out = (float4*)po;
    }
    const uint32_t grid = ((A.nblocks + 7) / 8) * 8;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    std::vector<float> ref(n), out(n);
    struct V_ { const char* name; void (*k)(Args); bool check; };
    const V_ vs[] = {
        {"gather4 (product form: 16-B record + 8-B payload gathers, trips of 4)", k_gather4<0>, true},
        {"gather4, one combined 16-B record {x, y, a^p}", k_gather4<1>, true},
        {"gather4, combined record, slim pair arithmetic (v_max spline, clamped r2, folded constants, fma)", k_gather4_slim<0>, true},
        {"gather4, combined record, slim pair arithmetic, finish's loads requested at the top", k_gather4_slim<1>, true},
        {"gather4, combined record, slim, own loads first, NO branch around the pairs (invalid slots scaled by zero)", k_gather4_slim_nobranch, true},
        {"sweep B lean: r2 + floor", k_gather4_lean<1>, true},
        {"sweep B lean: + no clamp of 1 - q", k_gather4_lean<3>, true},
        {"sweep B lean: + no select on an empty slot's index", k_gather4_lean<7>, true},
        {"combined record, slim, 16-bit offset list j - i (no mask decoding, no row bases), padding slots = the particle itself, no predicate", k_off16_slim<0>, true},
        {"combined record, slim, 16-bit offset list, slots behind the count skipped by a branch", k_off16_slim<1>, true},
        {"combined record, slim, ROW-DELTA lists (4 B per group of four: 16-bit offset + three 5-bit deltas), three groups' gathers together", k_delta_slim, true},
        {"combined record, slim, mask replay with STORED row bases (20 B per particle; no cell index, no cell_start loads)", k_gather4_slim_rbd, true},
        {"combined record, slim, masks decoded into 16 indices first, then flat trips of 4 (no per-row padding)", k_flat16_slim, true},
        {"combined record, slim, masks decoded by per-row loops into an LDS column per lane, then flat trips of 4", k_flat_lds_slim, true},
        {"gather4, combined record, slim, next trip's gathers requested before this trip's pairs (rows merged into one trip sequence)", k_gather4_slim_pipe, true},
        {"gather, combined record, 2 bits per iteration (no padding beyond pairs)", k_gather_loop<2>, true},
        {"gather, combined record, 1 bit per iteration (no padding slots)", k_gather_loop<1>, true},
        {"LDS windows per wave, trips of 4", k_lds<0>, true},
        {"LDS windows per wave, 2 bits per iteration", k_lds<2>, true},
        {"LDS windows per wave, 1 bit per iteration (no padding slots)", k_lds<1>, true},
        {"own loads + finish only (no neighbours)", k_own_only, false},
        {"pair arithmetic only, 12 slots in registers (no loads, no decoding)", k_math_only<12>, false},
        {"pair arithmetic only, 20 slots in registers", k_math_only<20>, false},
        {"slim pair arithmetic only, 12 slots in registers", k_math_only_slim<12>, false},
        {"slim pair arithmetic only, 20 slots in registers", k_math_only_slim<20>, false},
    };
    printf("| variant | us per launch (HIP events, %d launches back to back) | max rel diff of p' vs variant 0 |\n|---|---|---|\n", reps);
    bool have_ref = false;
    for (auto& v : vs) {
        CHECK(hipMemset(A.p_out, 0, (size_t)n * 4));
        hipLaunchKernelGGL(v.k, dim3(grid), dim3(256), 0, 0, A);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(out.data(), A.p_out, (size_t)n * 4, hipMemcpyDeviceToHost));
#if LAB_CHECK
        {
            uint32_t bad[4] = {0, 0, 0, 0}, zero[4] = {0, 0, 0, 0};
            CHECK(hipMemcpyFromSymbol(bad, HIP_SYMBOL(g_bad), sizeof bad));
            if (bad[0]) printf("!! %s: index check %u failed: index %u >= %u at thread %u\n", v.name, bad[0], bad[1], bad[2], bad[3]);
            CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_bad), zero, sizeof zero));
        }
        fflush(stdout);
#endif
        double err = 0, mx = 0;
        if (!have_ref) {
            ref = out;
            have_ref = true;
        }
        if (v.check) {
            for (uint32_t s = 0; s < n; s++) {
                err = std::max(err, (double)fabsf(out[s] - ref[s]));
                mx = std::max(mx, (double)fabsf(ref[s]));
            }
        }
        CHECK(hipEventRecord(e0, 0));
        for (int r = 0; r < reps; r++) hipLaunchKernelGGL(v.k, dim3(grid), dim3(256), 0, 0, A);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipDeviceSynchronize());
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (v.check) printf("| %s | %.2f | %.2e |\n", v.name, ms * 1e3 / reps, mx > 0 ? err / mx : 0.0);
        else printf("| %s | %.2f | - |\n", v.name, ms * 1e3 / reps);
        fflush(stdout);
    }
    // ---- sweep A ----
    {
        struct VA { const char* name; void (*k)(Args); int grid_div = 1; };
        const VA va[] = {
            {"sweep A, product form: 16-B record {x, y, m, h} + 4-B p / rho^2, two gathers per slot", k_accel<0>},
            {"sweep A, ONE 16-B gather of a combined record {x, y, p / rho^2, p}", k_accel<1>},
            {"sweep A, combined record, NO branch around the pairs", k_accel_nobranch},
            {"sweep A, combined record, 16-bit offset list (no mask decoding, no row bases, no predicate)", k_accel_off16},
            {"sweep A, offset list, twelve gathers in flight, 32-bit record addressing, clamp modifier (the product's form), unbounded registers", k_accel_off16_wide<1>},
            {"... compiled for 8 waves per SIMD", k_accel_off16_wide<8>},
            {"... TWO particles per lane, half the waves (24 gathers in flight), unbounded", k_accel_off16_x2<1>, 2},
            {"... two particles per lane, compiled for 4 waves per SIMD", k_accel_off16_x2<4>, 2},
            {"... two particles per lane, compiled for 6 waves per SIMD", k_accel_off16_x2<6>, 2},
            {"... half the workgroups, two tiles each, the second tile's head requested before the first tile's pairs, 8 waves", k_accel_off16_loop2<8>, 2},
            {"... the same, unbounded registers", k_accel_off16_loop2<1>, 2},
            {"sweep A lean: r2 + floor", k_accel_lean<1>},
            {"sweep A lean: + no clamp of 1 - q", k_accel_lean<3>},
            {"sweep A lean: + no select on an empty slot's index", k_accel_lean<7>},
            {"sweep B (slim, own loads first) storing p and the 16-B combined record instead of p and p / rho^2", k_gather4_slim_recstore},
        };
        printf("| sweep A variant | us per launch | max rel diff of a^p vs the first |\n|---|---|---|\n");
        std::vector<float4> ref_a(n), out_a(n);
        bool have = false;
        for (auto& v : va) {
            CHECK(hipMemset(A.pacc_out, 0, (size_t)n * 16));
            hipLaunchKernelGGL(v.k, dim3(v.grid_div == 2 ? ((((A.nblocks + 7) / 8 + 1) / 2) * 8) : grid), dim3(256), 0, 0, A);
            CHECK(hipDeviceSynchronize());
            CHECK(hipMemcpy(out_a.data(), A.pacc_out, (size_t)n * 16, hipMemcpyDeviceToHost));
            double err = 0, mx = 0;
            if (!have) {
                ref_a = out_a;
                have = true;
            }
            const bool is_a = v.k != k_gather4_slim_recstore;
            if (is_a)
                for (uint32_t s2 = 0; s2 < n; s2++) {
                    err = std::max(err, (double)std::max(fabsf(out_a[s2].z - ref_a[s2].z), fabsf(out_a[s2].w - ref_a[s2].w)));
                    mx = std::max(mx, (double)std::max(fabsf(ref_a[s2].z), fabsf(ref_a[s2].w)));
                }
            CHECK(hipEventRecord(e0, 0));
            for (int r = 0; r < reps; r++) hipLaunchKernelGGL(v.k, dim3(v.grid_div == 2 ? ((((A.nblocks + 7) / 8 + 1) / 2) * 8) : grid), dim3(256), 0, 0, A);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipDeviceSynchronize());
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (is_a) printf("| %s | %.2f | %.2e |\n", v.name, ms * 1e3 / reps, mx > 0 ? err / mx : 0.0);
            else printf("| %s | %.2f | - |\n", v.name, ms * 1e3 / reps);
            fflush(stdout);
        }
    }
    return 0;
}
