// Neighbour sweeps of the SPH step as hand-written HIP kernels for gfx950 (MI355X).
//
// Execution model (MI355X-first, not a translation of the reference's rayon loops over
// Vec<Vec<u32>> neighbour lists):
//   * particles live in a cell-sorted SoA (x,y,m,h packed as one float4 => one 16-B load per
//     particle, coalesced);
//   * a 256-thread workgroup owns one TILE of TX x TY cells.  It stages the tile plus a one-cell
//     halo -- TY+2 contiguous runs of the sorted arrays, because cells are ordered x-fastest --
//     into LDS once, together with the sweep's per-neighbour payload (so p_j/rho_j^2, m_j/rho_j ...
//     are computed once per particle, not once per pair);
//   * each thread then owns one particle of the tile and walks the 3 rows x 3 cells of its
//     neighbourhood as three contiguous LDS ranges (ds_read_b128 per candidate, lanes of one cell
//     read the same address => LDS broadcast), applying the reference's neighbour predicate
//         |x_ij|^2 < ((h_i + h_j) * 0.5 * 2)^2        (neighborhood_search.rs:143-146)
//     with exactly the reference's operations, so the visited set IS the reference's list;
//   * every particle writes only its own outputs (gather-only, no atomics, deterministic order:
//     rows bottom-to-top, sorted index ascending).
// Tiles whose halo does not fit the LDS budget (extreme size ratios) fall back to reading the
// payload straight from L2/HBM with the same code path.
//
// Each Op below cites the reference sweep it implements (src/simulation/simulation.rs and
// src/simulation/boundary_handler/sdf_boundary_handler/boundary_winchenbach2020.rs).
#include "sph_internal.hpp"

#define TILE_THREADS 256
#define TILE_TX 8
#define TILE_TY 7
#define TILE_CAP 1024

struct SweepCommon {
    GridP g;
    const uint32_t* __restrict__ cell_start;
    const uint32_t* __restrict__ tiles;
    const uint32_t* __restrict__ n_tiles;
    const uint32_t* __restrict__ cxy;
};

__device__ __forceinline__ void raise_error(DeviceStatus* st, uint32_t code, uint32_t info)
{
    if (atomicCAS(&st->error, 0u, code) == 0u) st->info = info;
}

// ------------------------------------------------------------------------------------------------
// generic tile sweep
// ------------------------------------------------------------------------------------------------
template <class Op, int TX, int TY, int CAP>
__global__ __launch_bounds__(TILE_THREADS) void k_sweep(Op op, SweepCommon c)
{
    __shared__ uint32_t s_cs[TY + 2][TX + 3];
    __shared__ uint32_t s_off[TY + 3];
    __shared__ uint32_t s_ioff[TY + 1];
    __shared__ float4 s_A[CAP];
    __shared__ float4 s_B[Op::HAS_B ? CAP : 1];

    if (op.skip()) return;

    const int tid = threadIdx.x;
    const uint32_t n_tiles = *c.n_tiles;
    const GridP g = c.g;

    for (uint32_t slot = blockIdx.x; slot < n_tiles; slot += gridDim.x) {
        const uint32_t t = c.tiles[slot];
        const int tx = t % g.ntx, ty = t / g.ntx;
        const int cx0 = tx * TX, cy0 = ty * TY;
        const int cA = max(cx0 - 1, 0);
        const int cE = min(cx0 + TX + 1, g.sx);
        const int ncol = cE - cA + 1;  // cell_start entries per row
        const int cxi_end = min(cx0 + TX, g.sx);

        for (int idx = tid; idx < (TY + 2) * (TX + 3); idx += TILE_THREADS) {
            int r = idx / (TX + 3), q = idx - r * (TX + 3);
            int cy = cy0 - 1 + r;
            uint32_t v = 0;
            if (cy >= 0 && cy < g.sy) v = c.cell_start[(uint32_t)cy * g.sx + cA + min(q, ncol - 1)];
            s_cs[r][q] = v;
        }
        __syncthreads();
        if (tid == 0) {
            uint32_t o = 0;
            for (int r = 0; r < TY + 2; r++) {
                s_off[r] = o;
                o += s_cs[r][ncol - 1] - s_cs[r][0];
            }
            s_off[TY + 2] = o;
            uint32_t io = 0;
            s_ioff[0] = 0;
            for (int r = 1; r <= TY; r++) {
                io += s_cs[r][cxi_end - cA] - s_cs[r][cx0 - cA];
                s_ioff[r] = io;
            }
        }
        __syncthreads();
        const uint32_t total = s_off[TY + 2];
        const uint32_t n_int = s_ioff[TY];
        const bool lds_mode = total <= (uint32_t)CAP;

        if (lds_mode) {
            for (uint32_t k = tid; k < total; k += TILE_THREADS) {
                int r = 0;
#pragma unroll
                for (int q = 1; q < TY + 2; q++) r += (k >= s_off[q]) ? 1 : 0;
                uint32_t gidx = s_cs[r][0] + (k - s_off[r]);
                s_A[k] = op.loadA(gidx);
                if (Op::HAS_B) s_B[k] = op.loadB(gidx);
            }
        }
        __syncthreads();

        for (uint32_t k = tid; k < n_int; k += TILE_THREADS) {
            int r = 1;
#pragma unroll
            for (int q = 1; q < TY; q++) r += (k >= s_ioff[q]) ? 1 : 0;
            const uint32_t gi = s_cs[r][cx0 - cA] + (k - s_ioff[r - 1]);
            const int lc = (int)(c.cxy[gi] & 0xffffu) - cA;
            float4 Ai, Bi = make_float4(0.f, 0.f, 0.f, 0.f);
            if (lds_mode) {
                uint32_t li = s_off[r] + (gi - s_cs[r][0]);
                Ai = s_A[li];
                if (Op::HAS_B) Bi = s_B[li];
            } else {
                Ai = op.loadA(gi);
                if (Op::HAS_B) Bi = op.loadB(gi);
            }
            typename Op::Acc acc;
            op.begin(acc, gi, Ai, Bi);
#pragma unroll
            for (int dr = -1; dr <= 1; dr++) {
                const int rr = r + dr;
                const uint32_t b = s_cs[rr][lc - 1], e = s_cs[rr][lc + 2];
                const uint32_t lbase = s_off[rr] - s_cs[rr][0];
                for (uint32_t j = b; j < e; j++) {
                    float4 Aj, Bj = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (lds_mode) {
                        Aj = s_A[lbase + j];
                        if (Op::HAS_B) Bj = s_B[lbase + j];
                    } else {
                        Aj = op.loadA(j);
                        if (Op::HAS_B) Bj = op.loadB(j);
                    }
                    // neighbour predicate, exactly the reference's operations (no FMA, strict <)
                    const float dx = Ai.x - Aj.x, dy = Ai.y - Aj.y;
                    const float r2 = dx * dx + dy * dy;
                    const float hij = (Ai.w + Aj.w) * 0.5f;
                    const float s = hij * 2.f;
                    if (r2 < s * s) op.pair(acc, Aj, Bj, dx, dy, r2, hij);
                }
            }
            op.finish(acc, gi, Ai, Bi);
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// Op: density  (+ boundary lambda terms, + neighbour count)
//   calculate_particle_density               simulation.rs:1007-1028, asserts :1046-1047
//   BoundaryWinchenbach2020::update_after_advect   boundary_winchenbach2020.rs:58-152
//   neighbor_count                           simulation.rs:2072-2074
// ------------------------------------------------------------------------------------------------
template <bool EXACT>
struct OpDensity {
    static constexpr bool HAS_B = false;
    const float4* __restrict__ pm;
    const uint32_t* __restrict__ orig;
    float* __restrict__ rho;
    float* __restrict__ lam_sum;
    float2* __restrict__ lam_grad;
    uint32_t* __restrict__ ncount;
    const PlaneP* __restrict__ planes;
    const float* __restrict__ lam_lut;
    const float* __restrict__ dlam_lut;
    DeviceStatus* status;
    StepP sp;
    struct Acc {
        float sum, lam;
        uint32_t cnt;
    };
    __device__ bool skip() const { return false; }
    __device__ float4 loadA(uint32_t g) const { return pm[g]; }
    __device__ float4 loadB(uint32_t) const { return make_float4(0.f, 0.f, 0.f, 0.f); }

    __device__ static float probe(const PlaneP& pl, float x, float y) { return (pl.dx * x + pl.dy * y) + pl.delta; }

    __device__ void begin(Acc& a, uint32_t gi, float4 Ai, float4) const
    {
        a.sum = 0.f;
        a.cnt = 0;
        // semi-analytic boundary: every IEEE op as in the reference (same values as the oracle)
        const float x = Ai.x, y = Ai.y;
        const float sr_i = Ai.w * 2.f;
        float ls = 0.f, gxs = 0.f, gys = 0.f;
        for (int k = 0; k < sp.n_planes; k++) {
            const PlaneP pl = planes[k];
            float d = probe(pl, x, y) / sr_i;
            if (!(d < 1.f)) continue;
            const float eps = sp.sdf_eps;
            const float inv_2eps = 1.f / (2.f * eps);
            float gx = (probe(pl, x + eps, y) - probe(pl, x - eps, y)) * inv_2eps;
            float gy = (probe(pl, x, y + eps) - probe(pl, x, y - eps)) * inv_2eps;
            float gn = sqrtf(gx * gx + gy * gy);
            if (!(gn >= 0.00001f)) continue;
            gx /= gn;
            gy /= gn;
            float penalty, dpenalty;
            if (sp.penalty == SPH_PENALTY_NONE) {
                penalty = 1.f;
                dpenalty = 0.f;
            } else if (sp.penalty == SPH_PENALTY_LINEAR) {
                penalty = 1.f - d;
                dpenalty = -1.f;
            } else if (sp.penalty == SPH_PENALTY_QUADRATIC1) {
                if (d > 0.f) { penalty = 1.f; dpenalty = 0.f; }
                else if (d > -1.f) { penalty = 0.5f * d * d + 1.f; dpenalty = d; }
                else { penalty = 0.5f - d; dpenalty = -1.f; }
            } else {
                if (d > 0.f) { penalty = 1.f; dpenalty = 0.f; }
                else if (d > -0.5f) { penalty = d * d + 1.f; dpenalty = 2.f * d; }
                else { penalty = 0.75f - d; dpenalty = -1.f; }
            }
            float lambda, dlambda;
            if (d <= -1.f) { lambda = 1.f; dlambda = 0.f; }
            else { lambda = lut_get(lam_lut, d); dlambda = lut_get(dlam_lut, d); }
            float s = dpenalty * lambda + penalty * dlambda;
            ls += lambda * penalty;
            gxs += gx / sr_i * s;
            gys += gy / sr_i * s;
        }
        a.lam = ls;
        lam_sum[gi] = ls;
        lam_grad[gi] = make_float2(gxs, gys);
    }
    __device__ void pair(Acc& a, float4 Aj, float4, float, float, float r2, float hij) const
    {
        a.sum += Aj.z * kernel_w<EXACT>(r2, hij);
        a.cnt++;
    }
    __device__ void finish(Acc& a, uint32_t gi, float4, float4) const
    {
        float d = a.sum + a.lam;
        rho[gi] = d;
        ncount[gi] = a.cnt;
        if (!isfinite(d)) raise_error(status, SPH_ERR_DENSITY_NOT_FINITE, orig[gi]);
        else if (!(d > 0.0001f)) raise_error(status, SPH_ERR_DENSITY_TOO_SMALL, orig[gi]);
        if (a.cnt > 20000u) raise_error(status, SPH_ERR_TOO_MANY_NEIGHBORS, orig[gi]);
    }
};

// ------------------------------------------------------------------------------------------------
// Op: a_ii + constant_field
//   compute_aii -> BoundaryWinchenbach2020::iisph_aii   simulation.rs:1080-1125, boundary_winchenbach2020.rs:225-306
//   constant_field                                       simulation.rs:2235-2248
// ------------------------------------------------------------------------------------------------
template <bool EXACT>
struct OpAiiConst {
    static constexpr bool HAS_B = true;
    const float4* __restrict__ pm;
    const uint32_t* __restrict__ orig;
    const float* __restrict__ rho;
    const float* __restrict__ lam_sum;
    const float2* __restrict__ lam_grad;
    float* __restrict__ aii;
    float* __restrict__ constf;
    DeviceStatus* status;
    StepP sp;
    struct Acc {
        float cf, ax, ay, a2, bx, by;
    };
    __device__ bool skip() const { return false; }
    __device__ float4 loadA(uint32_t g) const { return pm[g]; }
    __device__ float4 loadB(uint32_t g) const
    {
        float r = rho[g];
        return make_float4(r, pm[g].z / r, 0.f, 0.f);  // rho_j, m_j / rho_j
    }
    __device__ void begin(Acc& a, uint32_t, float4, float4) const { a.cf = a.ax = a.ay = a.a2 = a.bx = a.by = 0.f; }
    __device__ void pair(Acc& a, float4 Aj, float4 Bj, float dx, float dy, float r2, float hij) const
    {
        a.cf += Bj.y * kernel_w<EXACT>(r2, hij);
        float gx, gy;
        kernel_grad<EXACT>(dx, dy, r2, hij, gx, gy);
        a.ax += Aj.z * gx;
        a.ay += Aj.z * gy;
        if (sp.opdisc == SPH_OP_WINCHENBACH2020) {
            a.bx += Bj.y * gx;
            a.by += Bj.y * gy;
            a.a2 += Bj.y * (gx * gx + gy * gy);
        } else {
            a.a2 += Aj.z * (gx * gx + gy * gy);
        }
    }
    __device__ void finish(Acc& a, uint32_t gi, float4 Ai, float4 Bi) const
    {
        constf[gi] = a.cf + lam_sum[gi] / sp.rest_density;
        const float mi = Ai.z, rho_i = Bi.x, rho_b = sp.rest_density;
        const float rho_i_sq = rho_i * rho_i;
        const float2 gl = lam_grad[gi];
        float v;
        if (sp.opdisc == SPH_OP_WINCHENBACH2020) {
            float f = rho_b * (1.f / (rho_i * rho_i) + 0.f / (rho_b * rho_b));
            float lx = a.ax / rho_i_sq + f * gl.x, ly = a.ay / rho_i_sq + f * gl.y;
            float rx = a.bx + gl.x, ry = a.by + gl.y;
            v = (lx * rx + ly * ry) + (mi * a.a2 / rho_i_sq);
        } else {
            float coeff = sp.opdisc == SPH_OP_SIMPLE_GRADIENT ? 0.f : 1.f;
            float f = rho_b * (1.f / (rho_i * rho_i) + coeff / (rho_b * rho_b));
            float rgx = rho_b * gl.x, rgy = rho_b * gl.y;
            float lx = a.ax / rho_i_sq + f * gl.x, ly = a.ay / rho_i_sq + f * gl.y;
            float rx = a.ax / rho_i + rgx / rho_i, ry = a.ay / rho_i + rgy / rho_i;
            v = (lx * rx + ly * ry) + (mi * a.a2) / (rho_i * rho_i * rho_i);
        }
        aii[gi] = v;
        if (!isfinite(v)) raise_error(status, SPH_ERR_AII_NOT_FINITE, orig[gi]);
    }
};

// ------------------------------------------------------------------------------------------------
// Op: non-pressure acceleration -> velocity_temp
//   update_velocity_with_non_pressure_accel / calculate_particle_non_pressure_accel
//   simulation.rs:1051-1077, 931-1005
// ------------------------------------------------------------------------------------------------
// pair() needs the particle's own (rho_i, v_i): the skeleton passes only the neighbour payload, so
// the op keeps a copy of Bi in its accumulator.
template <bool EXACT>
struct OpNonPressure {
    static constexpr bool HAS_B = true;
    const float4* __restrict__ pm;
    const uint32_t* __restrict__ orig;
    const float* __restrict__ rho;
    const float2* __restrict__ vel;
    float2* __restrict__ vel_out;
    DeviceStatus* status;
    StepP sp;
    struct Acc {
        float vx, vy, rho_i, vix, viy;
    };
    __device__ bool skip() const { return false; }
    __device__ float4 loadA(uint32_t g) const { return pm[g]; }
    __device__ float4 loadB(uint32_t g) const
    {
        float2 v = vel[g];
        return make_float4(rho[g], v.x, v.y, 0.f);
    }
    __device__ void begin(Acc& a, uint32_t, float4, float4 Bi) const
    {
        a.vx = a.vy = 0.f;
        a.rho_i = Bi.x;
        a.vix = Bi.y;
        a.viy = Bi.z;
    }
    __device__ void pair(Acc& a, float4 Aj, float4 Bj, float dx, float dy, float r2, float hij) const
    {
        const float ux = a.vix - Bj.y, uy = a.viy - Bj.z;
        if (sp.viscosity_type == SPH_VISC_APPROX_LAPLACE) {
            const float xv = dx * ux + dy * uy;
            if (xv >= 0.f) return;
            float gx, gy;
            kernel_grad<EXACT>(dx, dy, r2, hij, gx, gy);
            const float rho_ij = (a.rho_i + Bj.x) * 0.5f;
            const float den = r2 + 0.01f * hij * hij;
            float coeff;
            if (EXACT) coeff = 2.f * 4.f * (Aj.z / rho_ij) * xv / den;
            else coeff = 8.f * (Aj.z * fast_rcp(rho_ij)) * xv * fast_rcp(den);
            const float f = sp.viscosity * coeff;
            a.vx += f * gx;
            a.vy += f * gy;
        } else if (sp.viscosity_type == SPH_VISC_WCSPH) {
            const float est = ux * dx + uy * dy;
            if (est < 0.f) {
                float gx, gy;
                kernel_grad<EXACT>(dx, dy, r2, hij, gx, gy);
                const float den = r2 + 0.001f * hij * hij;
                float viscous_term, pi_ab;
                if (EXACT) {
                    viscous_term = 2.f * sp.viscosity * hij * 88.f / (a.rho_i + Bj.x);
                    pi_ab = -viscous_term * est / den;
                } else {
                    viscous_term = 2.f * sp.viscosity * hij * 88.f * fast_rcp(a.rho_i + Bj.x);
                    pi_ab = -viscous_term * est * fast_rcp(den);
                }
                const float f = -Aj.z * pi_ab;
                a.vx += f * gx;
                a.vy += f * gy;
            }
        }
    }
    __device__ void finish(Acc& a, uint32_t gi, float4 Ai, float4) const
    {
        float px = 0.f, py = 0.f;
        if (sp.has_pull) {
            float tx = sp.pull_x - Ai.x, ty = sp.pull_y - Ai.y;
            float nn = sqrtf(tx * tx + ty * ty);
            px = tx / nn * 13.f;
            py = ty / nn * 13.f;
        }
        const float ax = (a.vx + 0.f) + px;
        const float ay = (a.vy + sp.gravity) + py;
        if (!isfinite(a.vx) || !isfinite(a.vy)) raise_error(status, SPH_ERR_VISCOSITY_NOT_FINITE, orig[gi]);
        vel_out[gi] = make_float2(a.vix + sp.dt * ax, a.viy + sp.dt * ay);
    }
};

// ------------------------------------------------------------------------------------------------
// Op: PPE source term (+ pressure := 0)
//   prepare_ppe_divergence / prepare_full_ppe / prepare_only_density_part_ppe   simulation.rs:1127-1204
//   calculate_source_term_divergence/_full/_only_density_part                   simulation.rs:1633-1676, 1712-1748
//   calculate_divergence_iisph (+ boundary part)    simulation.rs:1552-1592, boundary_winchenbach2020.rs:196-223
// ------------------------------------------------------------------------------------------------
template <bool EXACT>
struct OpSource {
    static constexpr bool HAS_B = true;
    const float4* __restrict__ pm;
    const float* __restrict__ rho;
    const float2* __restrict__ vel;
    const float2* __restrict__ lam_grad;
    float* __restrict__ src;
    float* __restrict__ p_zero;
    StepP sp;
    int kind;  // 0 divergence, 1 full, 2 only density
    struct Acc {
        float sum, rho_i, inv_rho_i, qx, qy;
    };
    __device__ bool skip() const { return false; }
    __device__ float4 loadA(uint32_t g) const { return pm[g]; }
    __device__ float4 loadB(uint32_t g) const
    {
        float2 v = vel[g];
        float r = rho[g];
        return make_float4(v.x, v.y, pm[g].z / r, r);  // v_j, m_j/rho_j, rho_j
    }
    __device__ void begin(Acc& a, uint32_t, float4, float4 Bi) const
    {
        a.sum = 0.f;
        a.rho_i = Bi.w;
        a.inv_rho_i = fast_rcp(Bi.w);
        a.qx = Bi.x;
        a.qy = Bi.y;
    }
    __device__ void pair(Acc& a, float4 Aj, float4 Bj, float dx, float dy, float r2, float hij) const
    {
        if (kind == 2) return;
        float gx, gy;
        kernel_grad<EXACT>(dx, dy, r2, hij, gx, gy);
        const float dot = (Bj.x - a.qx) * gx + (Bj.y - a.qy) * gy;
        if (sp.opdisc == SPH_OP_WINCHENBACH2020) a.sum += Bj.z * dot;
        else if (EXACT) a.sum += Aj.z / a.rho_i * dot;
        else a.sum += Aj.z * a.inv_rho_i * dot;
    }
    __device__ void finish(Acc& a, uint32_t gi, float4, float4) const
    {
        const float rho_i = a.rho_i, rho_b = sp.rest_density, dt = sp.dt;
        float s;
        const float nde = sp.opdisc == SPH_OP_WINCHENBACH2020 ? sp.rest_density : rho_i;
        if (kind == 2) {
            s = -(sp.rest_density - rho_i) / (nde * dt * dt);
        } else {
            const float2 gl = lam_grad[gi];
            const float bdot = (0.f - a.qx) * gl.x + (0.f - a.qy) * gl.y;
            const float bdiv = sp.opdisc == SPH_OP_WINCHENBACH2020 ? bdot : rho_b / rho_i * bdot;
            const float vdiv = a.sum + (sp.n_planes ? bdiv : 0.f);
            if (kind == 0) s = -vdiv / dt;
            else s = -(sp.rest_density - rho_i) / (nde * dt * dt) - vdiv / dt;
        }
        src[gi] = s;
        p_zero[gi] = 0.f;
    }
};

// ------------------------------------------------------------------------------------------------
// Op: pressure acceleration (Jacobi sweep A)
//   calculate_particle_pressure_accel(s) / calculate_fluid_fluid_pressure_accel   simulation.rs:1518-1543, 1750-1808
//   iisph_boundary_pressure_accel                                  boundary_winchenbach2020.rs:164-194
// ------------------------------------------------------------------------------------------------
template <bool EXACT>
struct OpPressureAccel {
    static constexpr bool HAS_B = true;
    const float4* __restrict__ pm;
    const float* __restrict__ rho;
    const float* __restrict__ p0;
    const float* __restrict__ p1;
    const float2* __restrict__ lam_grad;
    float2* __restrict__ pacc;
    const SolverCtrl* __restrict__ ctrl;
    StepP sp;
    int iter;  // >= 0: Jacobi iteration `iter` (reads buffer iter&1, skipped when the solve is done); < 0: final sweep
    struct Acc {
        float ax, ay, p1t;
    };
    __device__ bool skip() const { return iter >= 0 && ctrl->done != 0u; }
    __device__ const float* pbuf() const
    {
        uint32_t cur = iter >= 0 ? (uint32_t)(iter & 1) : ctrl->cur;
        return cur ? p1 : p0;
    }
    __device__ float4 loadA(uint32_t g) const { return pm[g]; }
    __device__ float4 loadB(uint32_t g) const
    {
        float r = rho[g], p = pbuf()[g];
        return make_float4(p / (r * r), p, r, 0.f);  // p_j / rho_j^2, p_j, rho_j
    }
    __device__ void begin(Acc& a, uint32_t, float4, float4 Bi) const
    {
        a.ax = a.ay = 0.f;
        a.p1t = Bi.x;
    }
    __device__ void pair(Acc& a, float4 Aj, float4 Bj, float dx, float dy, float r2, float hij) const
    {
        float gx, gy;
        kernel_grad<EXACT>(dx, dy, r2, hij, gx, gy);
        const float f = -Aj.z * (a.p1t + Bj.x);
        a.ax += f * gx;
        a.ay += f * gy;
    }
    __device__ void finish(Acc& a, uint32_t gi, float4, float4 Bi) const
    {
        float bx = 0.f, by = 0.f;
        if (sp.n_planes) {
            const float p_i = Bi.y, rho_i = Bi.z, rho_b = sp.rest_density;
            const float p_ib = sp.opdisc == SPH_OP_SYMMETRIC_GRADIENT ? p_i : 0.f;
            const float f = -rho_b * (p_i / (rho_i * rho_i) + p_ib / (rho_b * rho_b));
            const float2 gl = lam_grad[gi];
            bx = f * gl.x;
            by = f * gl.y;
        }
        pacc[gi] = make_float2(a.ax + bx, a.ay + by);
    }
};

// ------------------------------------------------------------------------------------------------
// Op: relaxed-Jacobi pressure update (sweep B)
//   iisph_single_pressure_iteration    simulation.rs:1241-1319
//   a_ii >= 0 pre-check                simulation.rs:1390-1403 (done here at iteration 0)
// Per-particle residual class goes to stat[] (normal: the error; singular / negative: tagged NaNs),
// reduced in fixed particle order by k_solver_reduce.
// ------------------------------------------------------------------------------------------------
#define STAT_SINGULAR 0x7fc00001u
#define STAT_NEGATIVE 0x7fc00002u

template <bool EXACT>
struct OpJacobi {
    static constexpr bool HAS_B = true;
    const float4* __restrict__ pm;
    const uint32_t* __restrict__ orig;
    const float* __restrict__ rho;
    const float2* __restrict__ pacc;
    const float2* __restrict__ lam_grad;
    const float* __restrict__ aii;
    const float* __restrict__ src;
    const float* __restrict__ p_in;
    float* __restrict__ p_out;
    float* __restrict__ dens_err;
    float* __restrict__ stat;
    const SolverCtrl* __restrict__ ctrl;
    DeviceStatus* status;
    StepP sp;
    int iter;
    int residual_density;
    struct Acc {
        float sum, rho_i, inv_rho_i, qx, qy;
    };
    __device__ bool skip() const { return ctrl->done != 0u; }
    __device__ float4 loadA(uint32_t g) const { return pm[g]; }
    __device__ float4 loadB(uint32_t g) const
    {
        float2 a = pacc[g];
        float r = rho[g];
        return make_float4(a.x, a.y, pm[g].z / r, r);
    }
    __device__ void begin(Acc& a, uint32_t, float4, float4 Bi) const
    {
        a.sum = 0.f;
        a.rho_i = Bi.w;
        a.inv_rho_i = fast_rcp(Bi.w);
        a.qx = Bi.x;
        a.qy = Bi.y;
    }
    __device__ void pair(Acc& a, float4 Aj, float4 Bj, float dx, float dy, float r2, float hij) const
    {
        float gx, gy;
        kernel_grad<EXACT>(dx, dy, r2, hij, gx, gy);
        const float dot = (Bj.x - a.qx) * gx + (Bj.y - a.qy) * gy;
        if (sp.opdisc == SPH_OP_WINCHENBACH2020) a.sum += Bj.z * dot;
        else if (EXACT) a.sum += Aj.z / a.rho_i * dot;
        else a.sum += Aj.z * a.inv_rho_i * dot;
    }
    __device__ void finish(Acc& a, uint32_t gi, float4, float4) const
    {
        const float aii_i = aii[gi];
        if (iter == 0 && aii_i < 0.f) raise_error(status, SPH_ERR_AII_NEGATIVE, orig[gi]);
        if (fabsf(aii_i) < 10e-4f) {
            p_out[gi] = 0.f;
            stat[gi] = __uint_as_float(STAT_SINGULAR);
            return;
        }
        const float rho_i = a.rho_i, rho_b = sp.rest_density, dt = sp.dt;
        float bdiv = 0.f;
        if (sp.n_planes) {
            const float2 gl = lam_grad[gi];
            const float bdot = (0.f - a.qx) * gl.x + (0.f - a.qy) * gl.y;
            bdiv = sp.opdisc == SPH_OP_WINCHENBACH2020 ? bdot : rho_b / rho_i * bdot;
        }
        const float a_p = a.sum + bdiv;
        const float s = src[gi];
        if (!isfinite(a_p)) raise_error(status, SPH_ERR_AP_NOT_FINITE, orig[gi]);
        float pn = p_in[gi] + sp.jacobi_omega * (s - a_p) / aii_i;
        if (!isfinite(pn)) raise_error(status, SPH_ERR_PRESSURE_NOT_FINITE, orig[gi]);
        float err;
        if (residual_density) {
            err = rho_i * dt * dt * (s - a_p);
            dens_err[gi] = err;
        } else {
            err = dt * (s - a_p);
        }
        if (pn <= 0.f) {  // clamp_negative_pressures is true at every call site
            p_out[gi] = 0.f;
            stat[gi] = __uint_as_float(STAT_NEGATIVE);
        } else {
            p_out[gi] = pn;
            stat[gi] = err;
        }
    }
};

// ------------------------------------------------------------------------------------------------
// residual reduction + stop decision    (PressureSolverStatistics simulation.rs:397-469,
// stopping rule of iisph_pressure_iterations simulation.rs:1453-1479)
// Fixed particle-order chunks -> block partials -> the last block to arrive (agent-scope
// release/acquire around the ticket) adds them in index order: deterministic.
// ------------------------------------------------------------------------------------------------
#define REDUCE_BLOCKS 128

__global__ __launch_bounds__(256) void k_solver_reduce(const float* __restrict__ stat, uint32_t n, SolverCtrl* ctrl,
                                                        SolverPartial* partials, int iter, int residual_density, float max_avg_error,
                                                        uint32_t max_iters, float rest_density, float dt)
{
    if (ctrl->done) return;
    __shared__ SolverPartial s_w[4];
    __shared__ bool s_last;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint32_t chunk = (n + REDUCE_BLOCKS - 1) / REDUCE_BLOCKS;
    const uint32_t b0 = blockIdx.x * chunk, b1 = min(b0 + chunk, n);
    uint32_t normal = 0, singular = 0, negative = 0;
    float sum = 0.f, mx = 0.f;
    for (uint32_t i = b0 + tid; i < b1; i += 256) {
        float v = stat[i];
        uint32_t bits = __float_as_uint(v);
        if (bits == STAT_SINGULAR) singular++;
        else if (bits == STAT_NEGATIVE) negative++;
        else {
            normal++;
            sum += v;
            mx = fmaxf(mx, fabsf(v));
        }
    }
    normal = wave_sum_u32(normal);
    singular = wave_sum_u32(singular);
    negative = wave_sum_u32(negative);
    sum = wave_sum(sum);
    mx = wave_max(mx);
    if (lane == 0) s_w[w] = SolverPartial{normal, singular, negative, sum, mx};
    __syncthreads();
    if (tid == 0) {
        SolverPartial t = s_w[0];
        for (int k = 1; k < 4; k++) {
            t.normal += s_w[k].normal;
            t.singular += s_w[k].singular;
            t.negative += s_w[k].negative;
            t.sum_err += s_w[k].sum_err;
            t.max_err = fmaxf(t.max_err, s_w[k].max_err);
        }
        partials[blockIdx.x] = t;
        __threadfence();  // agent-scope release of the partial before the ticket
        uint32_t prev = atomicAdd(&ctrl->ticket, 1u);
        s_last = (prev == gridDim.x - 1);
    }
    __syncthreads();
    if (!s_last) return;
    if (tid == 0) {
        __threadfence();  // agent-scope acquire: drop stale L1 lines before reading the other blocks' partials
        SolverPartial t{0, 0, 0, 0.f, 0.f};
        for (uint32_t k = 0; k < gridDim.x; k++) {
            const volatile uint32_t* q = (const volatile uint32_t*)&partials[k];
            t.normal += q[0];
            t.singular += q[1];
            t.negative += q[2];
            t.sum_err += __uint_as_float(q[3]);
            t.max_err = fmaxf(t.max_err, __uint_as_float(q[4]));
        }
        const float avg = t.normal > 0 ? t.sum_err / (float)t.normal : __uint_as_float(0x7fc00000u);
        bool stop;
        if (residual_density) stop = t.normal == 0 || (fabsf(avg / rest_density) < max_avg_error && iter > 1);
        else stop = t.normal == 0 || (fabsf(avg) < max_avg_error / dt && iter > 1);
        if (!stop && (uint32_t)iter == max_iters) stop = true;
        ctrl->normal = t.normal;
        ctrl->singular = t.singular;
        ctrl->negative = t.negative;
        ctrl->sum_err = t.sum_err;
        ctrl->max_err = t.max_err;
        ctrl->iters = (uint32_t)iter;
        ctrl->cur = (uint32_t)((iter + 1) & 1);  // mem::swap(pressure, pressure_next_iter)
        ctrl->ticket = 0;
        if (stop) ctrl->done = 1u;
    }
}

// ------------------------------------------------------------------------------------------------
// per-particle maps
// ------------------------------------------------------------------------------------------------
// HybridDFSPH after the divergence solve: v += dt * a^p   (simulation.rs:2547-2560)
__global__ __launch_bounds__(256) void k_vel_add_pacc(uint32_t n, float dt, float2* __restrict__ vel, const float2* __restrict__ pacc,
                                                       const uint32_t* __restrict__ orig, DeviceStatus* status)
{
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float2 v = vel[i], a = pacc[i];
    v.x += dt * a.x;
    v.y += dt * a.y;
    vel[i] = v;
    if (!isfinite(v.x) || !isfinite(v.y)) raise_error(status, SPH_ERR_VELOCITY_NOT_FINITE, orig[i]);
}

// mode 0: v += dt a^p ; x += dt v                     (IISPH / OnlyDivergence, simulation.rs:2433-2445, 2486-2499)
// mode 1: x += dt v + dt^2 a^p ; v += dt a^p * min(dt*factor, 1)   (HybridDFSPH, simulation.rs:2644-2646)
__global__ __launch_bounds__(256) void k_integrate(uint32_t n, float dt, float vfactor, int mode, const float4* __restrict__ pm,
                                                    float4* __restrict__ pm_out, float2* __restrict__ vel, const float2* __restrict__ pacc,
                                                    const uint32_t* __restrict__ orig, DeviceStatus* status)
{
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float4 p = pm[i];
    float2 v = vel[i], a = pacc[i];
    if (mode == 0) {
        v.x += dt * a.x;
        v.y += dt * a.y;
        p.x += dt * v.x;
        p.y += dt * v.y;
        if (!isfinite(v.x) || !isfinite(v.y)) raise_error(status, SPH_ERR_VELOCITY_NOT_FINITE, orig[i]);
    } else {
        p.x += dt * v.x + dt * dt * a.x;
        p.y += dt * v.y + dt * dt * a.y;
        v.x += dt * a.x * vfactor;
        v.y += dt * a.y * vfactor;
        if (!isfinite(p.x) || !isfinite(p.y)) raise_error(status, SPH_ERR_POSITION_NOT_FINITE, orig[i]);
    }
    pm_out[i] = p;
    vel[i] = v;
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
static SweepCommon common_of(const SweepArgs& a) { return SweepCommon{a.g, a.cell_start, a.tiles, a.n_tiles, a.cxy}; }

template <class Op>
static void launch_sweep(hipStream_t s, const SweepArgs& a, const Op& op)
{
    hipLaunchKernelGGL((k_sweep<Op, TILE_TX, TILE_TY, TILE_CAP>), dim3(a.grid_blocks), dim3(TILE_THREADS), 0, s, op, common_of(a));
}

void sweep_tile_dims(int* tx, int* ty)
{
    *tx = TILE_TX;
    *ty = TILE_TY;
}

void launch_density(hipStream_t s, Profiler* prof, const SweepArgs& a)
{
    ProfScope ps(prof, "density", s);
    if (a.exact) {
        OpDensity<true> op{a.pm, a.orig, a.rho, a.lam_sum, a.lam_grad, a.ncount, a.planes, a.lam_lut, a.dlam_lut, a.status, a.sp};
        launch_sweep(s, a, op);
    } else {
        OpDensity<false> op{a.pm, a.orig, a.rho, a.lam_sum, a.lam_grad, a.ncount, a.planes, a.lam_lut, a.dlam_lut, a.status, a.sp};
        launch_sweep(s, a, op);
    }
}

void launch_aii_const(hipStream_t s, Profiler* prof, const SweepArgs& a)
{
    ProfScope ps(prof, "aii_constfield", s);
    if (a.exact) {
        OpAiiConst<true> op{a.pm, a.orig, a.rho, a.lam_sum, a.lam_grad, a.aii, a.constf, a.status, a.sp};
        launch_sweep(s, a, op);
    } else {
        OpAiiConst<false> op{a.pm, a.orig, a.rho, a.lam_sum, a.lam_grad, a.aii, a.constf, a.status, a.sp};
        launch_sweep(s, a, op);
    }
}

void launch_non_pressure(hipStream_t s, Profiler* prof, const SweepArgs& a)
{
    ProfScope ps(prof, "non_pressure_accel", s);
    if (a.exact) {
        OpNonPressure<true> op{a.pm, a.orig, a.rho, a.vel, a.vel_tmp, a.status, a.sp};
        launch_sweep(s, a, op);
    } else {
        OpNonPressure<false> op{a.pm, a.orig, a.rho, a.vel, a.vel_tmp, a.status, a.sp};
        launch_sweep(s, a, op);
    }
}

void launch_source_term(hipStream_t s, Profiler* prof, const SweepArgs& a, int kind)
{
    ProfScope ps(prof, "source_term", s);
    if (a.exact) {
        OpSource<true> op{a.pm, a.rho, a.vel, a.lam_grad, a.src, a.p0, a.sp, kind};
        launch_sweep(s, a, op);
    } else {
        OpSource<false> op{a.pm, a.rho, a.vel, a.lam_grad, a.src, a.p0, a.sp, kind};
        launch_sweep(s, a, op);
    }
}

void launch_pressure_accel(hipStream_t s, Profiler* prof, const SweepArgs& a, int iter)
{
    ProfScope ps(prof, "pressure_accel", s);
    if (a.exact) {
        OpPressureAccel<true> op{a.pm, a.rho, a.p0, a.p1, a.lam_grad, a.pacc, a.ctrl, a.sp, iter};
        launch_sweep(s, a, op);
    } else {
        OpPressureAccel<false> op{a.pm, a.rho, a.p0, a.p1, a.lam_grad, a.pacc, a.ctrl, a.sp, iter};
        launch_sweep(s, a, op);
    }
}

void launch_jacobi_update(hipStream_t s, Profiler* prof, const SweepArgs& a, int iter, int residual_density)
{
    ProfScope ps(prof, "jacobi_update", s);
    const float* pin = (iter & 1) ? a.p1 : a.p0;
    float* pout = (iter & 1) ? a.p0 : a.p1;
    if (a.exact) {
        OpJacobi<true> op{a.pm, a.orig, a.rho, a.pacc, a.lam_grad, a.aii, a.src, pin, pout, a.dens_err, a.stat, a.ctrl, a.status, a.sp, iter, residual_density};
        launch_sweep(s, a, op);
    } else {
        OpJacobi<false> op{a.pm, a.orig, a.rho, a.pacc, a.lam_grad, a.aii, a.src, pin, pout, a.dens_err, a.stat, a.ctrl, a.status, a.sp, iter, residual_density};
        launch_sweep(s, a, op);
    }
}

void launch_solver_reduce(hipStream_t s, Profiler* prof, const SweepArgs& a, int iter, int residual_density, float max_avg_error,
                          uint32_t max_iters, float* block_partials)
{
    ProfScope ps(prof, "solver_reduce", s);
    hipLaunchKernelGGL(k_solver_reduce, dim3(REDUCE_BLOCKS), dim3(256), 0, s, a.stat, a.n, a.ctrl, (SolverPartial*)block_partials, iter,
                       residual_density, max_avg_error, max_iters, a.sp.rest_density, a.sp.dt);
}

void launch_vel_add_pacc(hipStream_t s, Profiler* prof, const SweepArgs& a)
{
    ProfScope ps(prof, "vel_add_pacc", s);
    hipLaunchKernelGGL(k_vel_add_pacc, dim3((a.n + 255) / 256), dim3(256), 0, s, a.n, a.sp.dt, a.vel, a.pacc, a.orig, a.status);
}

void launch_integrate(hipStream_t s, Profiler* prof, const SweepArgs& a, float4* pm_rw, int mode)
{
    ProfScope ps(prof, "integrate", s);
    hipLaunchKernelGGL(k_integrate, dim3((a.n + 255) / 256), dim3(256), 0, s, a.n, a.sp.dt, a.sp.hyb_vfactor, mode, a.pm, pm_rw, a.vel, a.pacc,
                       a.orig, a.status);
}
