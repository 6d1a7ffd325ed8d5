// Device-side numerical primitives of the SPH particle loop for gfx950 (wave64).
//
// What they compute is fixed by the reference:
//   cubic spline W / dW/dx            src/simulation/sph_kernels.rs:23-71
//   h_ij = (h_i + h_j) * 0.5          src/simulation/sph_kernels.rs:273-278
//   neighbour predicate               src/simulation/neighborhood_search.rs:143-146
//   LookupTable1D::get                src/simulation/boundary_handler/sdf_boundary_handler/lookup_table.rs:32-48
//
// The library is compiled with -ffp-contract=off, so `a*b + c` is two roundings like Rust;
// FMAs appear only where written as fmaf().  The neighbour predicate and the cell index use
// exactly the reference's operations (bit-exact index sets).  Pair VALUES use the hardware
// reciprocal / square root (v_rcp_f32, v_sqrt_f32, v_rsq_f32: 1 ulp) in FAST mode; the reference's
// own summation order is not reproducible (R*-tree traversal order + rayon reduce), so values
// are comparable to ~1e-6 relative per sweep either way.  EXACT mode (SPH_HIP_EXACT=1) keeps
// IEEE division and sqrt in the reference's operation order, for diagnosing parity.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sph_ffi.h"

#define SPH_PI_F 3.14159274101257324219f          // std::f32::consts::PI
#define SPH_FRAC_1_PI_F 0.318309873342514038086f  // std::f32::consts::FRAC_1_PI
#define SPH_ETA 1.9f                              // simulation.rs:369
#define SPH_SEVEN_PI (7.f * SPH_PI_F)

struct GridP {
    float cs;            // cell size = support radius of the largest particle
    int minx, miny;      // cells_min  (neighborhood_search.rs:273)
    int sx, sy;          // grid size  (cells_max - cells_min)
    uint32_t ncells;
};

// cell key of a position in grid g, clamped into it (the grid of a build queued ahead is a prediction: sph_sort.hip, cell_key_of).
// IEEE division, like `(particle_pos / kernel_support_radius).map(|x| x.floor() as i32)` (neighborhood_search.rs:253-255)
__device__ __forceinline__ uint32_t cell_key_clamped(const GridP& g, float x, float y)
{
    int cx = (int)floorf(x / g.cs) - g.minx;
    int cy = (int)floorf(y / g.cs) - g.miny;
    cx = min(max(cx, 0), g.sx - 1);
    cy = min(max(cy, 0), g.sy - 1);
    return (uint32_t)cx + (uint32_t)cy * (uint32_t)g.sx;
}

// Incremental cell sort (sph_sort.hip): what classifies a particle -- by k_inc_classify, or by the integrating tail of the step's
// last solve, which holds the new position in registers (sph_sweeps.hip: k_solver_tail).  head == nullptr: not wanted.
struct IncClassifyP {
    GridP cur, nxt;                   // the grid the array is sorted by; the grid of the keys to sort by (same cell size)
    const uint32_t* cxy_cur;          // cell of every particle in the current order (cx | cy << 16, grid `cur`)
    uint32_t* nk;                     // out: new key
    uint8_t* mv;                      // out: 1 = the particle changes its cell
    uint32_t* next;                   // out: list link of a mover
    unsigned long long* head;         // per-cell list heads of grid `nxt`, tagged with `epoch` (never cleared)
    uint32_t epoch;
};
// key of current-grid cell (cx, cy) in the next grid.  The next grid covers the bounding box the current cells were computed from
// (queue_ahead_build), so the cell lies inside it; if it ever did not, no key equals the value returned here and the particle counts
// as a mover -- it is then on exactly one list and in no cell's stayers, like every other mover: the slots still add up to n.
__device__ __forceinline__ uint32_t inc_key_of_cur_cell(const GridP& cur, const GridP& nxt, uint32_t cxy)
{
    const int cx = (int)(cxy & 0xffffu) + cur.minx - nxt.minx, cy = (int)(cxy >> 16) + cur.miny - nxt.miny;
    if (cx < 0 || cx >= nxt.sx || cy < 0 || cy >= nxt.sy) return 0xffffffffu;
    return (uint32_t)cx + (uint32_t)cy * (uint32_t)nxt.sx;
}
// new key and mover flag of particle i; a mover hangs itself into the list of the cell it enters
__device__ __forceinline__ void inc_register(const IncClassifyP& q, uint32_t i, uint32_t k, bool mover)
{
    q.nk[i] = k;
    q.mv[i] = mover ? 1 : 0;
    if (mover) {
        const unsigned long long prev = atomicExch(&q.head[k], ((unsigned long long)q.epoch << 32) | (unsigned long long)(i + 1u));
        q.next[i] = (uint32_t)(prev >> 32) == q.epoch ? (uint32_t)prev : 0u;
    }
}
// ... of particle i at (x, y), an element of the array that is sorted by q.cxy_cur
__device__ __forceinline__ void inc_classify_particle(const IncClassifyP& q, uint32_t i, float x, float y)
{
    const uint32_t k = cell_key_clamped(q.nxt, x, y);
    inc_register(q, i, k, k != inc_key_of_cur_cell(q.cur, q.nxt, q.cxy_cur[i]));
}

// Multi-resolution scenes sort by a grid whose cell is the support of the SMALLEST particle.  A tile is
// ts x ts cells with ts * cs >= the largest support, so every neighbour of a particle lives in the 3 x 3
// tiles around its own; hmax[tile] is the largest h found in those 3 x 3 tiles, i.e. an upper bound on h_j of
// any neighbour j.  ts == 0: uniform scene, every stencil is 3 x 3 cells.
struct TileP {
    int ts, tsx, tsy;
    const uint32_t* __restrict__ hmax;   // float bits (h > 0, so unsigned order == float order)
    float slack;   // added to every search range: the lists of the ADVECTED positions are gathered from the cells of the
                   // pre-step positions (level_estimation_after_advection), slack = 2 x the largest displacement; else 0
};

// per-step scalars the kernels read (subset of sph_params + dt)
struct StepP {
    float rest_density, viscosity, gravity, jacobi_omega, dt, sdf_eps;
    float pull_x, pull_y;
    float hyb_vfactor;  // min(dt * hybrid_dfsph_factor, 1)
    int viscosity_type, penalty, opdisc, has_pull, n_planes;
};

struct PlaneP {
    float dx, dy, delta;
};
#define SPH_MAX_PLANES 8

// The boundary handler's list of SDFs (BoundaryWinchenbach2020::sdf): up to SPH_MAX_PLANES SdfPlane, or -- poly_n > 0 --
// ONE Sdf2D connected component (sdf/sdf2d.rs:4-16) that replaces them.
struct BoundaryP {
    PlaneP planes[SPH_MAX_PLANES];
    int poly_n;
    float px[SPH_MAX_POLYGON_POINTS], py[SPH_MAX_POLYGON_POINTS];   // point
    float dx[SPH_MAX_POLYGON_POINTS], dy[SPH_MAX_POLYGON_POINTS];   // normalized_line_dir
    float nx[SPH_MAX_POLYGON_POINTS], ny[SPH_MAX_POLYGON_POINTS];   // point_pseudo_normal
};

struct SolverPartial {
    uint32_t normal, singular, negative;
    float sum_err, max_err;
};

// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float fast_rsq(float x) { return __builtin_amdgcn_rsqf(x); }

// sph_kernels.rs:23-32, written select-style (both branches are a handful of VALU ops; a divergent
// branch costs more than evaluating both)
__device__ __forceinline__ float cubic_unnorm(float q)
{
    const float a = 6.f * (q * q * q - q * q) + 1.f;
    const float v = 1.f - q;
    const float b = 2.f * (v * v * v);
    return q < 0.5f ? a : (q < 1.f ? b : 0.f);
}
// sph_kernels.rs:34-43
__device__ __forceinline__ float cubic_unnorm_deriv(float q)
{
    const float a = 18.f * q * q - 12.f * q;
    const float v = 1.f - q;
    const float b = -6.f * v * v;
    return q < 0.5f ? a : (q < 1.f ? b : 0.f);
}

// Math policies.  EXACT: IEEE division / sqrt in the reference's operation order.  FAST: hardware
// v_rcp_f32 / v_rsq_f32 / v_sqrt_f32 (1 ulp).  UNIFORM: FAST with every h_ij == h (all particles
// carry the bit-identical smoothing length), so the normalisation and 1/(2h) are launch constants.
struct MathExact {
    static constexpr bool EXACT = true, UNIFORM = false;
    float h;  // unused
    // The boundary handler's PER-SDF entries (BoundaryWinchenbach2020::lambda: Vec<Vec<(FT, VF)>>, boundary_winchenbach2020.rs:27): the
    // FAST policies fold a particle's entries into sum(lambda), sum(grad lambda) once (every use is linear in them); f (g1 + g2) and
    // f g1 + f g2 differ in the last bit for a particle in a CORNER of the box (two entries), so the EXACT policy keeps the gradient
    // entries apart -- entry k of particle i at wall_pl[k wall_n + i], wall_cnt[i] of them, written by the density sweep -- and adds
    // them up where and how the reference does (tests/test_gpu_bitexact.py).
    float2* wall_pl = nullptr;
    uint8_t* wall_cnt = nullptr;
    uint32_t wall_n = 0;
    __device__ __forceinline__ uint32_t wall_count(uint32_t i) const { return wall_cnt[i]; }
    __device__ __forceinline__ float2 wall_entry(uint32_t i, uint32_t k) const { return wall_pl[(size_t)k * wall_n + i]; }
    // calculate_divergence_iisph's boundary part (boundary_winchenbach2020.rs:196-223), quantity_b = 0
    __device__ __forceinline__ float wall_divergence(uint32_t i, float qx, float qy, float rho_i, float rho_b, bool by_volume) const
    {
        float r = 0.f;
        const uint32_t cnt = wall_cnt[i];
        for (uint32_t k = 0; k < cnt; k++) {
            const float2 g = wall_entry(i, k);
            const float dot = (0.f - qx) * g.x + (0.f - qy) * g.y;
            if (by_volume) r += dot;
            else r += rho_b / rho_i * dot;
        }
        return r;
    }
    __device__ __forceinline__ float w(float r2, float hij) const
    {
        float r = sqrtf(r2);
        float nf = 10.f / (SPH_SEVEN_PI * (hij * hij));
        return nf * cubic_unnorm(r / (2.f * hij));
    }
    __device__ __forceinline__ void grad(float dx, float dy, float r2, float hij, float& gx, float& gy) const
    {
        float r = sqrtf(r2);
        float q = r / (2.f * hij);
        float ux = dx / r, uy = dy / r;
        float nf = 10.f / (SPH_SEVEN_PI * (hij * hij));
        float s = nf * cubic_unnorm_deriv(q) / (2.f * hij);
        bool z = q <= 1.0e-5f;
        gx = z ? 0.f : s * ux;
        gy = z ? 0.f : s * uy;
    }
};
// FAST / UNIFORM pair values are written for the instruction mix of gfx950 as measured (scripts/ubench/valu_issue.hip,
// profiles/r3_variants.md): fma / add / mul issue in 2 clocks per wave, a compare, a select or a max in 4, a transcendental in 8.
//   * the spline in truncated-power form -- W(q) = 2 [(1-q)+^3 - 4 (1/2-q)+^3], W'(q) = 6 [4 (1/2-q)+^2 - (1-q)+^2] -- two v_max
//     instead of two compare + select pairs;
//   * r^2 clamped away from zero instead of the reference's `q > 1e-5 ? .. : 0` select: with dx = dy = 0 the gradient s (dx, dy)
//     is zero by itself, and W'(q) / r stays finite (-> -2 / (2h) as q -> 0); a pair closer than 1e-5 of the support -- below the
//     resolution of f32 positions at these scales -- gets a gradient of relative size 1e-5 instead of exactly zero;
//   * constant factors folded, explicit fma.  `gscale` returns s with grad W_ij = s (dx, dy), so that a sweep whose pair term is
//     (scalar) x grad W multiplies scalars first and needs two fma for the vector.
// Stand-alone effect on the Jacobi sweep: 21.9 -> 20.0 us (profiles/r3_jacobi_lab.md).  EXACT mode keeps the reference's operations.
#define SPH_R2_FLOOR 1.0e-30f
// max(x, 0) for an x that is never above 1 -- (1 - q) and (1/2 - q) with q >= 0 -- as med3(x, 0, 1): the backend folds that into the CLAMP
// modifier of the subtraction that produced x, one instruction instead of two, the same value bit for bit (round 5: two of the ~23
// instructions a pair costs in every FAST / UNIFORM sweep)
__device__ __forceinline__ float pos_part(float x) { return __builtin_amdgcn_fmed3f(x, 0.f, 1.f); }
struct MathFast {
    static constexpr bool EXACT = false, UNIFORM = false;
    float h;  // unused
    // one reciprocal per pair: 1 / (2 h_ij), and the normalisation 10 / (7 pi h_ij^2) = (40 / (7 pi)) (1 / (2 h_ij))^2 from it
    __device__ __forceinline__ float w(float r2, float hij) const
    {
        const float inv2h = fast_rcp(hij + hij);
        const float q = fast_sqrt(r2) * inv2h;
        const float u = pos_part(1.f - q), t = pos_part(0.5f - q);
        return ((80.f / SPH_SEVEN_PI) * (inv2h * inv2h)) * fmaf(-4.f * t, t * t, u * (u * u));
    }
    __device__ __forceinline__ float gscale(float r2, float hij) const
    {
        const float r2c = fmaxf(r2, SPH_R2_FLOOR);
        const float rinv = fast_rsq(r2c);
        const float inv2h = fast_rcp(hij + hij);
        const float q = (r2c * rinv) * inv2h;
        const float u = pos_part(1.f - q), t = pos_part(0.5f - q);
        const float d = fmaf(4.f * t, t, -(u * u));
        return (((240.f / SPH_SEVEN_PI) * inv2h) * (inv2h * inv2h)) * (d * rinv);
    }
    __device__ __forceinline__ void grad(float dx, float dy, float r2, float hij, float& gx, float& gy) const
    {
        const float s = gscale(r2, hij);
        gx = s * dx;
        gy = s * dy;
    }
    // W and the gradient's scale of one pair from ONE reciprocal square root and one set of truncated powers (a sweep that needs
    // both -- a_ii + constant field -- otherwise pays a v_sqrt, a v_rsq and the two v_max twice): q from the rsq path for both
    __device__ __forceinline__ void wg(float r2, float hij, float& wv, float& s) const
    {
        const float r2c = fmaxf(r2, SPH_R2_FLOOR);
        const float rinv = fast_rsq(r2c);
        const float inv2h = fast_rcp(hij + hij);
        const float q = (r2c * rinv) * inv2h;
        const float u = pos_part(1.f - q), t = pos_part(0.5f - q);
        const float i2 = inv2h * inv2h;
        wv = ((80.f / SPH_SEVEN_PI) * i2) * fmaf(-4.f * t, t * t, u * (u * u));
        s = (((240.f / SPH_SEVEN_PI) * inv2h) * i2) * (fmaf(4.f * t, t, -(u * u)) * rinv);
    }
};
struct MathUniform {
    static constexpr bool EXACT = false, UNIFORM = true;
    float h, nf, inv2h;
    float nf2, nf6;   // 2 nf (value), 6 nf / (2h) (gradient)
    __device__ __forceinline__ float w(float r2, float) const
    {
        const float q = fast_sqrt(r2) * inv2h;
        const float u = pos_part(1.f - q), t = pos_part(0.5f - q);
        return nf2 * fmaf(-4.f * t, t * t, u * (u * u));
    }
    __device__ __forceinline__ float gscale(float r2, float) const
    {
        const float r2c = fmaxf(r2, SPH_R2_FLOOR);
        const float rinv = fast_rsq(r2c);
        const float q = (r2c * rinv) * inv2h;
        const float u = pos_part(1.f - q), t = pos_part(0.5f - q);
        const float d = fmaf(4.f * t, t, -(u * u));
        return nf6 * (d * rinv);
    }
    __device__ __forceinline__ void grad(float dx, float dy, float r2, float, float& gx, float& gy) const
    {
        const float s = gscale(r2, 0.f);
        gx = s * dx;
        gy = s * dy;
    }
    __device__ __forceinline__ void wg(float r2, float, float& wv, float& s) const   // see MathFast::wg
    {
        const float r2c = fmaxf(r2, SPH_R2_FLOOR);
        const float rinv = fast_rsq(r2c);
        const float q = (r2c * rinv) * inv2h;
        const float u = pos_part(1.f - q), t = pos_part(0.5f - q);
        wv = nf2 * fmaf(-4.f * t, t * t, u * (u * u));
        s = nf6 * (fmaf(4.f * t, t, -(u * u)) * rinv);
    }
};

// DimensionUtils2d::sphere_volume_to_radius, local_smoothing_length_from_mass
// (sph_kernels.rs:203-206, simulation.rs:371-380): IEEE ops (bit-exact h => bit-exact neighbour sets)
__device__ __forceinline__ float h_from_mass(float mass, float rest_density)
{
    float volume = mass / rest_density;
    return SPH_ETA * sqrtf(volume * SPH_FRAC_1_PI_F);
}

// stencil half-width in cells that is guaranteed to cover every j with |x_ij| < (h_i + h_j) * 0.5 * k (k = 2: the
// SPH support; k = level_estimation_range / ETA: the extended lists of the level estimation): such a j is STRICTLY closer
// than S = (h_i + hmax) * 0.5 * k, hence at most ceil(S / cs) cells away on either axis -- floor(S / cs) + 1, less one
// when S is a whole number of cells.  That case is the common one, not a corner: the sorting grid's cell IS the support of
// the smallest particle (cs = 2 h_min), so a fine particle among fine ones has S == cs and a 3 x 3 stencil, exactly what the
// uniform path (and the reference's own CellGrid, cell = support of the largest particle, 3 x 3 cells) relies on.
__device__ __forceinline__ int stencil_radius(const GridP& g, const TileP& t, float h_i, int cx, int cy, float k)
{
    float hn = h_i;   // uniform scene: every h is h_i
    if (t.ts > 0) hn = __uint_as_float(t.hmax[(uint32_t)(cy / t.ts) * (uint32_t)t.tsx + (uint32_t)(cx / t.ts)]);
    const float S = (h_i + hn) * 0.5f * k + t.slack;
    int R = (int)floorf(S / g.cs) + 1;
    if (R > 1 && (float)(R - 1) * g.cs >= S) R--;
    return R;
}

// Sdf2D::probe = find_min_dist_object + to_dist_and_dir (sdf/sdf2d.rs:77-160, 196-228), IEEE operations in the
// reference's order: the nearest edge (if the point projects inside it) or corner; positive on the air side
__device__ __forceinline__ float polygon_probe(const BoundaryP* __restrict__ b, float x, float y)
{
    const int n = b->poly_n;
    float min_dist_sq = __uint_as_float(0x7f800000u);
    bool is_line = false;
    int point_idx = 0;
    float line_dist = 0.f, pdx = 0.f, pdy = 0.f, pt_dist_sq = min_dist_sq;
    for (int i = 0; i < n; i++) {
        const int i1 = i + 1 == n ? 0 : i + 1;
        const float sx = b->px[i], sy = b->py[i];
        const float lx = b->px[i1] - sx, ly = b->py[i1] - sy;
        const float line_len_sq = lx * lx + ly * ly;
        const float dx = b->dx[i], dy = b->dy[i];
        const float qx = x - sx, qy = y - sy;   // point_dir
        const float projection_len = qx * dx + qy * dy;
        if (projection_len > 0.f && projection_len * projection_len < line_len_sq) {
            const float d = qx * -dy + qy * dx;   // dot(point_dir, rotate_left_90_degrees(line_dir))
            const float d2 = d * d;
            if (d2 < min_dist_sq) {
                is_line = true;
                line_dist = d;
                min_dist_sq = d2;
            }
        }
        const float corner_dist_sq = qx * qx + qy * qy;
        if (corner_dist_sq < min_dist_sq) {
            is_line = false;
            point_idx = i;
            pt_dist_sq = corner_dist_sq;
            pdx = qx;
            pdy = qy;
            min_dist_sq = corner_dist_sq;
        }
    }
    if (is_line) return line_dist;
    const float sign = (b->nx[point_idx] * pdx + b->ny[point_idx] * pdy) >= 0.f ? 1.0f : -1.0f;
    return sqrtf(pt_dist_sq) * sign;
}
// Sdf::probe of SDF number k (SdfPlane::probe, sdf_plane.rs:36-38, or the polygon)
__device__ __forceinline__ float sdf_probe(const BoundaryP* __restrict__ b, int k, float x, float y)
{
    if (b->poly_n > 0) return polygon_probe(b, x, y);
    const PlaneP pl = b->planes[k];
    return (pl.dx * x + pl.dy * y) + pl.delta;
}

// LookupTable1D::get with (min,max,steps) = (-1,1,10000); caller guarantees -1 <= x < 1
__device__ __forceinline__ float lut_get(const float* __restrict__ data, float x)
{
    float fidx = (x - (-1.f)) * 0.5f * 10000.f;
    float fl = floorf(fidx);
    float interp = fidx - fl;
    int idx = (int)fl;
    if (idx + 1 >= 10001) return data[idx];
    return data[idx] * (1.f - interp) + data[idx + 1] * interp;
}

// wave64 / block reductions (fixed order => deterministic)
__device__ __forceinline__ float wave_sum(float v)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v)
{
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_down(v, o, 64));
    return v;
}
// wave64 reductions on the DPP network (row_shr 1, 2, 4, 8 inside the rows of 16, then row_bcast:15 and row_bcast:31): seven 4-clock
// VALU instructions and no LDS round trip, against six ds_bpermute_b32 (what __shfl_down compiles to on gfx9: ~24 clocks of issue
// and an LDS latency each, as a dependent chain).  The total arrives in lane 63 and is broadcast from there.  A lane without a
// source in a step receives `old` = 0: the sum's neutral element, and the maximum's as long as the values are not negative.
#define SPH_DPP(V, CTRL, ROWMASK) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(V), CTRL, ROWMASK, 0xf, false))
__device__ __forceinline__ float wave_sum_dpp(float v)
{
    v += SPH_DPP(v, 0x111, 0xf);   // row_shr:1
    v += SPH_DPP(v, 0x112, 0xf);   // row_shr:2
    v += SPH_DPP(v, 0x114, 0xf);   // row_shr:4
    v += SPH_DPP(v, 0x118, 0xf);   // row_shr:8   -> lane 15 of every row: the row's sum
    v += SPH_DPP(v, 0x142, 0xa);   // row_bcast:15 into rows 1 and 3
    v += SPH_DPP(v, 0x143, 0xc);   // row_bcast:31 into rows 2 and 3
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_max_nonneg_dpp(float v)   // v >= 0 on every lane
{
    v = fmaxf(v, SPH_DPP(v, 0x111, 0xf));
    v = fmaxf(v, SPH_DPP(v, 0x112, 0xf));
    v = fmaxf(v, SPH_DPP(v, 0x114, 0xf));
    v = fmaxf(v, SPH_DPP(v, 0x118, 0xf));
    v = fmaxf(v, SPH_DPP(v, 0x142, 0xa));
    v = fmaxf(v, SPH_DPP(v, 0x143, 0xc));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
#undef SPH_DPP
// minimum / maximum of any floats the same way (a lane without a source receives +inf / -inf); exact, so bit-identical to any other order
#define SPH_DPPO(V, OLD, CTRL, ROWMASK) __int_as_float(__builtin_amdgcn_update_dpp((int)(OLD), __float_as_int(V), CTRL, ROWMASK, 0xf, false))
__device__ __forceinline__ float wave_min_dpp(float v)
{
    v = fminf(v, SPH_DPPO(v, 0x7f800000u, 0x111, 0xf));
    v = fminf(v, SPH_DPPO(v, 0x7f800000u, 0x112, 0xf));
    v = fminf(v, SPH_DPPO(v, 0x7f800000u, 0x114, 0xf));
    v = fminf(v, SPH_DPPO(v, 0x7f800000u, 0x118, 0xf));
    v = fminf(v, SPH_DPPO(v, 0x7f800000u, 0x142, 0xa));
    v = fminf(v, SPH_DPPO(v, 0x7f800000u, 0x143, 0xc));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_max_dpp(float v)
{
    v = fmaxf(v, SPH_DPPO(v, 0xff800000u, 0x111, 0xf));
    v = fmaxf(v, SPH_DPPO(v, 0xff800000u, 0x112, 0xf));
    v = fmaxf(v, SPH_DPPO(v, 0xff800000u, 0x114, 0xf));
    v = fmaxf(v, SPH_DPPO(v, 0xff800000u, 0x118, 0xf));
    v = fmaxf(v, SPH_DPPO(v, 0xff800000u, 0x142, 0xa));
    v = fmaxf(v, SPH_DPPO(v, 0xff800000u, 0x143, 0xc));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
#undef SPH_DPPO
__device__ __forceinline__ float wave_min(float v)
{
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_down(v, o, 64));
    return v;
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// A rank's solver totals of one Jacobi iteration from sweep B's per-block partials, by ONE workgroup of 1024 lanes in a fixed order
// (lane t takes partials t, t + 1024, ...; lanes, then waves in index order): deterministic.  Thread 0 stores the six doubles the ranks
// all-reduce (solver_decide_multi): normal, singular, negative, sum of errors, largest error (informational when summed), "a guard
// fired on this rank".  Shared by the slab decomposition's packing launch (sph_sweeps.hip: k_pack_totals) and the push transport's fused
// pack + push (sph_transport.hip: k_ipc_pack_push); every lane of the workgroup must call it, a __syncthreads() sits inside.
#define RANK_TOTALS_THREADS 1024
struct DeviceStatus;
__device__ __forceinline__ void rank_totals_block(const SolverPartial* __restrict__ partials, uint32_t nparts, double* __restrict__ tot, const uint32_t* __restrict__ status_error)
{
    __shared__ SolverPartial s_r[RANK_TOTALS_THREADS / 64];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    SolverPartial t{0, 0, 0, 0.f, 0.f};
    for (uint32_t k0 = tid; k0 < nparts; k0 += 8u * RANK_TOTALS_THREADS) {
        SolverPartial v[8];
#pragma unroll
        for (uint32_t u = 0; u < 8u; u++) {
            const uint32_t k = k0 + u * RANK_TOTALS_THREADS;
            v[u] = k < nparts ? partials[k] : SolverPartial{0, 0, 0, 0.f, 0.f};
        }
#pragma unroll
        for (uint32_t u = 0; u < 8u; u++) {
            t.normal += v[u].normal;
            t.singular += v[u].singular;
            t.negative += v[u].negative;
            t.sum_err += v[u].sum_err;
            t.max_err = fmaxf(t.max_err, v[u].max_err);
        }
    }
    t.normal = wave_sum_u32(t.normal);
    t.singular = wave_sum_u32(t.singular);
    t.negative = wave_sum_u32(t.negative);
    t.sum_err = wave_sum(t.sum_err);
    t.max_err = wave_max(t.max_err);
    if (lane == 0) s_r[w] = t;
    __syncthreads();
    if (tid != 0) return;
    t = s_r[0];
    for (int k = 1; k < RANK_TOTALS_THREADS / 64; k++) {
        t.normal += s_r[k].normal;
        t.singular += s_r[k].singular;
        t.negative += s_r[k].negative;
        t.sum_err += s_r[k].sum_err;
        t.max_err = fmaxf(t.max_err, s_r[k].max_err);
    }
    tot[0] = (double)t.normal;
    tot[1] = (double)t.singular;
    tot[2] = (double)t.negative;
    tot[3] = (double)t.sum_err;
    tot[4] = (double)t.max_err;   // summed over the ranks: informational
    tot[5] = *status_error != 0u ? 1.0 : 0.0;   // a guard fired on this rank: every rank ends the solve (solver_decide_multi)
}
