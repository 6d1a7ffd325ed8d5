// Slab maintenance of the 1-D decomposition (the reference has no counterpart): which particles a rank owns,
// hands over and shows its x-neighbours as ghosts, step by step.
//   ordinary steps   slab_refresh_fused: ONE pass classes every slot of the previous step's arrays, one collective round carries the
//                    counts, migrants and ghost records are appended, the cell sort drops what left
//   general path     partition_and_migrate + build_ghost_layer (first step, re-balancing, a migrant deeper than one ghost width, ...)
//   rebalance_cuts   the cuts follow the x quantiles of the particles
//   refresh_ghosts   one field of the ghosts from their owners, after every sweep whose output neighbours read
// The step driver (sph_step.hip) calls these between its phases; the transports are in sph_transport.hip.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "sph_dist.hpp"


// class counts of a 256-thread block -> at most one atomic per class per block (one per wave on the same four words cost
// ~100 us per launch at 512k particles: 32k same-address atomics)
__device__ __forceinline__ void block_class_counts(uint32_t cls, uint32_t* __restrict__ counts)
{
    __shared__ uint32_t s_cnt[4];
    if (threadIdx.x < 4) s_cnt[threadIdx.x] = 0u;
    __syncthreads();
    for (uint32_t k = 0; k < 4; k++) {
        const uint64_t m = __ballot(cls == k);
        if (m && (threadIdx.x & 63) == (uint32_t)(__ffsll((unsigned long long)m) - 1)) atomicAdd(&s_cnt[k], (uint32_t)__popcll(m));
    }
    __syncthreads();
    // class 0 (stay / no halo) is nearly everybody: 4096 adds on ONE word cost ~40 us.  It is not counted -- the host derives it
    // from the total (classes 1..3 are the few particles near a cut)
    if (threadIdx.x >= 1 && threadIdx.x < 4 && s_cnt[threadIdx.x]) atomicAdd(&counts[threadIdx.x], s_cnt[threadIdx.x]);
}

// ------------------------------------------------------------------------------------------------
// kernels of the slab decomposition
// ------------------------------------------------------------------------------------------------
// class of every slot of the previous step's arrays: 0 stay, 1 migrate left, 2 migrate right, 3 drop (ghost)
__global__ __launch_bounds__(256) void k_classify_migrate(uint32_t n, const float4* __restrict__ pm, const uint8_t* __restrict__ owned,
                                                           float cut_lo, float cut_hi, int has_left, int has_right, uint32_t* __restrict__ key,
                                                           uint32_t* __restrict__ val, uint32_t* __restrict__ counts, float far_l, float far_r)
{
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    uint32_t cls = 4;
    bool far = false;   // further past the cut than the narrowest slab allowed: the x-neighbour may not be its owner either
    if (i < n) {
        const float x = pm[i].x;
        if (owned && !owned[i]) cls = 3;
        else if (has_left && x < cut_lo) {
            cls = 1;
            far = x < cut_lo - far_l;
        } else if (has_right && !(x < cut_hi)) {
            cls = 2;
            far = !(x < cut_hi + far_r);
        } else cls = 0;
        key[i] = cls;
        val[i] = i;
    }
    if (__ballot(far) != 0ull && (threadIdx.x & 63u) == 0u) atomicOr(&counts[RC_FAR], 1u);
    block_class_counts(cls, counts);
}

// ---- slab re-balancing: x range and x histogram of the owned particles --------------------------------------
__device__ __forceinline__ uint32_t f32_ordered(float f)   // monotone map float -> uint (for atomicMin / atomicMax)
{
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
static float f32_from_ordered(uint32_t u)
{
    const uint32_t b = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    float f;
    memcpy(&f, &b, 4);
    return f;
}
__global__ __launch_bounds__(256) void k_minmax_x(uint32_t n, const float4* __restrict__ pm, const uint8_t* __restrict__ owned, uint32_t* __restrict__ out)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n || (owned && !owned[i])) return;
    const uint32_t u = f32_ordered(pm[i].x);
    atomicMin(&out[0], u);
    atomicMax(&out[1], u);
}
#define REBALANCE_BINS 4096
__global__ __launch_bounds__(256) void k_hist_x(uint32_t n, const float4* __restrict__ pm, const uint8_t* __restrict__ owned, float gmin, float binw,
                                                 uint32_t* __restrict__ hist)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n || (owned && !owned[i])) return;
    int b = (int)floorf((pm[i].x - gmin) / binw);
    b = b < 0 ? 0 : (b >= REBALANCE_BINS ? REBALANCE_BINS - 1 : b);
    atomicAdd(&hist[b], 1u);
}

// largest smoothing length among the owned particles within `region` of the left / right cut (float bits: h >= 0, so unsigned order)
__global__ __launch_bounds__(256) void k_region_hmax(uint32_t n, const float4* __restrict__ pm, float lo_edge, float hi_edge, int has_left, int has_right,
                                                      uint32_t* __restrict__ out /* [2] */)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    float hl = 0.f, hr = 0.f;
    if (i < n) {
        const float4 p = pm[i];
        if (has_left && p.x < lo_edge) hl = p.w;
        if (has_right && !(p.x < hi_edge)) hr = p.w;
    }
    hl = wave_max(hl);
    hr = wave_max(hr);
    if ((threadIdx.x & 63u) == 0u) {
        if (hl > 0.f) atomicMax(&out[0], __float_as_uint(hl));
        if (hr > 0.f) atomicMax(&out[1], __float_as_uint(hr));
    }
}

// owned particles within the layer width of a cut are ghosts of that neighbour: 1 left halo, 2 right halo, 0 none
__global__ __launch_bounds__(256) void k_classify_halo(uint32_t n, const float4* __restrict__ pm, float lo_edge, float hi_edge, int has_left,
                                                        int has_right, uint32_t* __restrict__ key, uint32_t* __restrict__ val,
                                                        uint32_t* __restrict__ counts)
{
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    uint32_t cls = 4;
    if (i < n) {
        const float x = pm[i].x;
        const bool l = has_left && x < lo_edge, r = has_right && !(x < hi_edge);
        cls = (l && r) ? 3u : (l ? 1u : (r ? 2u : 0u));
        key[i] = cls;
        val[i] = i;
    }
    block_class_counts(cls, counts + 4);
}

// migrant record: x, y, m, h, vx, vy, id, level, level_old
#define MIG_WORDS 12
__global__ __launch_bounds__(256) void k_pack_migrants(uint32_t base, uint32_t cnt, const float4* __restrict__ pm, const float2* __restrict__ vel,
                                                        const uint32_t* __restrict__ orig, const float* __restrict__ lvl,
                                                        const float* __restrict__ lvlold, const float* __restrict__ h2n,
                                                        const float* __restrict__ lam_sum, const uint8_t* __restrict__ szc, float* __restrict__ rec)
{
    uint32_t k = blockIdx.x * 256 + threadIdx.x;
    if (k >= cnt) return;
    const uint32_t i = base + k;
    const float4 p = pm[i];
    const float2 v = vel[i];
    float* r = rec + (size_t)k * MIG_WORDS;
    r[0] = p.x; r[1] = p.y; r[2] = p.z; r[3] = p.w; r[4] = v.x; r[5] = v.y;
    r[6] = __uint_as_float(orig[i]);
    r[7] = lvl[i];
    r[8] = lvlold[i];
    r[9] = h2n[i];        // h2_next and the previous step's lambda sum: FromDistribution* support lengths
    r[10] = lam_sum[i];
    r[11] = __uint_as_float((uint32_t)szc[i]);
}
__global__ __launch_bounds__(256) void k_unpack_migrants(uint32_t base, uint32_t cnt, const float* __restrict__ rec, float4* __restrict__ pm,
                                                          float2* __restrict__ vel, uint32_t* __restrict__ orig, float* __restrict__ lvl,
                                                          float* __restrict__ lvlold, float* __restrict__ h2n, float* __restrict__ lam_sum,
                                                          uint8_t* __restrict__ szc)
{
    uint32_t k = blockIdx.x * 256 + threadIdx.x;
    if (k >= cnt) return;
    const uint32_t i = base + k;
    const float* r = rec + (size_t)k * MIG_WORDS;
    pm[i] = make_float4(r[0], r[1], r[2], r[3]);
    vel[i] = make_float2(r[4], r[5]);
    orig[i] = __float_as_uint(r[6]);
    lvl[i] = r[7];
    lvlold[i] = r[8];
    h2n[i] = r[9];
    lam_sum[i] = r[10];
    szc[i] = (uint8_t)__float_as_uint(r[11]);
}

// ghost record (static per step): x, y, m, h, vx, vy, id (the neighbour-list export of a slab names ghosts by their global id)
#define GHOST_WORDS 7
__global__ __launch_bounds__(256) void k_pack_ghosts(const uint32_t* __restrict__ idx, uint32_t cnt, const float4* __restrict__ pm,
                                                      const float2* __restrict__ vel, const uint32_t* __restrict__ orig, float* __restrict__ rec)
{
    uint32_t k = blockIdx.x * 256 + threadIdx.x;
    if (k >= cnt) return;
    const uint32_t i = idx[k];
    const float4 p = pm[i];
    const float2 v = vel[i];
    float* r = rec + (size_t)k * GHOST_WORDS;
    r[0] = p.x; r[1] = p.y; r[2] = p.z; r[3] = p.w; r[4] = v.x; r[5] = v.y;
    r[6] = __uint_as_float(orig[i]);
}
__global__ __launch_bounds__(256) void k_unpack_ghosts(uint32_t base, uint32_t cnt, const float* __restrict__ rec, float4* __restrict__ pm,
                                                        float2* __restrict__ vel, uint32_t* __restrict__ orig, float* __restrict__ lvl,
                                                        float* __restrict__ lvlold, uint8_t* __restrict__ ring1_src, uint32_t ord_base, float ring_edge,
                                                        int side)
{
    uint32_t k = blockIdx.x * 256 + threadIdx.x;
    if (k >= cnt) return;
    const uint32_t i = base + k;
    const float* r = rec + (size_t)k * GHOST_WORDS;
    // first ring = within one support radius of the cut: its pressure acceleration is computed here, not fetched
    ring1_src[ord_base + k] = (side == 0 ? r[0] >= ring_edge : r[0] < ring_edge) ? 1 : 0;
    pm[i] = make_float4(r[0], r[1], r[2], r[3]);
    vel[i] = make_float2(r[4], r[5]);
    orig[i] = __float_as_uint(r[6]);
    lvl[i] = __uint_as_float(0x7fc00000u);
    lvlold[i] = 0.f;
}

__global__ __launch_bounds__(256) void k_halo_pos(const uint32_t* __restrict__ halo_idx, uint32_t cnt, uint32_t* __restrict__ halo_pos)
{
    uint32_t k = blockIdx.x * 256 + threadIdx.x;
    if (k < cnt) halo_pos[halo_idx[k]] = k;
}

// after the cell sort: where did my halo particles and my ghosts end up?  perm[s] = pre-sort index of slot s
// `cls` (fused refresh, else nullptr): halo_pos is only defined for the halo members then -- slots whose class byte says so and the
// arrivals behind the n_cls previous slots -- and nothing had to clear the rest of it
// pacc[slot] = {x, y, 0, 0} for every ghost slot (see setup_member)
// ... and the same for the two pressure records {x, y, p / rho^2, p} sweep A gathers (OpPressureAccelU): the ghosts' p / rho^2 arrives
// with every iteration's exchange, their positions are seeded here
__global__ __launch_bounds__(256) void k_seed_ghost_records(const uint32_t* __restrict__ ghost_dst, uint32_t ng, const float4* __restrict__ pm, float4* __restrict__ pacc,
                                                             float4* __restrict__ rec0, float4* __restrict__ rec1)
{
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    if (k >= ng) return;
    const uint32_t i = ghost_dst[k];
    const float4 p = pm[i];
    const float4 r = make_float4(p.x, p.y, 0.f, 0.f);
    pacc[i] = r;
    rec0[i] = r;
    rec1[i] = r;
}
__global__ void k_edge_mark(const uint32_t* __restrict__ halo_src, uint32_t nh, const uint32_t* __restrict__ ghost_dst, uint32_t ng, uint8_t* __restrict__ edge)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < nh) edge[halo_src[t]] = 1;
    else if (t < nh + ng) edge[ghost_dst[t - nh]] = 1;
}

__global__ __launch_bounds__(256) void k_build_maps(uint32_t n_tot, uint32_t n_own, const uint32_t* __restrict__ perm,
                                                     const uint32_t* __restrict__ halo_pos, uint32_t* __restrict__ halo_src,
                                                     uint32_t* __restrict__ ghost_dst, uint8_t* __restrict__ owned, const uint8_t* __restrict__ ring1_src,
                                                     uint8_t* __restrict__ ring1, const uint8_t* __restrict__ cls, uint32_t n_cls)
{
    uint32_t s = blockIdx.x * 256 + threadIdx.x;
    if (s >= n_tot) return;
    const uint32_t old = perm[s];
    const bool own = old < n_own;
    owned[s] = own ? 1 : 0;
    ring1[s] = own ? 0 : ring1_src[old - n_own];
    if (own) {
        if (cls && old < n_cls && cls[old] != 1 /* SC_HALO_L */ && cls[old] != 2 /* SC_HALO_R */) return;
        const uint32_t k = halo_pos[old];
        if (k != 0xffffffffu) halo_src[k] = s;
    } else {
        ghost_dst[old - n_own] = s;
    }
}

// refresh one field of the ghosts: gather my halo particles' values / scatter the received ones
// both sides in one launch: entries [0, cnt0) belong to the left neighbour's staging buffer, [cnt0, cnt0 + cnt1) to the right one's
// (`stride`, `off`: the field's `words` floats sit at field[i * stride + off ..) -- a plain array has stride == words, off == 0; p / rho^2
//  inside the 16-byte pressure records {x, y, p / rho^2, p} has words 1, stride 4, off 2)
__global__ __launch_bounds__(256) void k_pack_field(const uint32_t* __restrict__ src_idx, uint32_t cnt0, uint32_t cnt1, int words, int stride, int off,
                                                     const float* __restrict__ field, float* __restrict__ out0, float* __restrict__ out1)
{
    uint32_t k = blockIdx.x * 256 + threadIdx.x;
    if (k >= cnt0 + cnt1) return;
    const uint32_t i = src_idx[k];
    float* out = k < cnt0 ? out0 + (size_t)k * words : out1 + (size_t)(k - cnt0) * words;
    for (int w = 0; w < words; w++) out[w] = field[(size_t)i * stride + off + w];
}
__global__ __launch_bounds__(256) void k_unpack_field(const uint32_t* __restrict__ dst_idx, uint32_t cnt0, uint32_t cnt1, int words, int stride, int off,
                                                       const float* __restrict__ in0, const float* __restrict__ in1, float* __restrict__ field)
{
    uint32_t k = blockIdx.x * 256 + threadIdx.x;
    if (k >= cnt0 + cnt1) return;
    const uint32_t i = dst_idx[k];
    const float* in = k < cnt0 ? in0 + (size_t)k * words : in1 + (size_t)(k - cnt0) * words;
    for (int w = 0; w < words; w++) field[(size_t)i * stride + off + w] = in[w];
}

// m / rho of the ghosts from the refreshed rho (the same expression the owner evaluated: bit-identical, no second exchange)
__global__ __launch_bounds__(256) void k_ghost_mrho(const uint32_t* __restrict__ dst_idx, uint32_t cnt, const float4* __restrict__ pm,
                                                     const float* __restrict__ rho, float* __restrict__ mrho)
{
    uint32_t k = blockIdx.x * 256 + threadIdx.x;
    if (k >= cnt) return;
    const uint32_t i = dst_idx[k];
    mrho[i] = pm[i].z / rho[i];
}

// ------------------------------------------------------------------------------------------------
// The fused slab refresh (ordinary steps; slab_refresh_fused below): ONE pass classifies every slot of the previous step's
// arrays, ONE round trip carries the four counts, nothing is reordered -- the slots that left are dropped by the cell sort.
//   0 stays, interior   1 stays, within halo_w of the left cut   2 stays, within halo_w of the right cut
//   3 migrates left     4 migrates right                          5 ghost of the previous step
// A migrant is by construction a member of its receiver's halo towards the sender (it is within one step's displacement of the
// cut), so the receiver's ghost-record count is known without a second round trip: `bad` = a migrant that the receiver's own
// halo test would reject, or a particle in both halos (slab narrower than two ghost layers) -> the general path takes over.
// ------------------------------------------------------------------------------------------------
// what the host already knows when it queues the classification: the step's header values (to be min-reduced over the ranks)
// and its status / "take the general path" words -- staged for the collective round by the kernel itself
struct RefreshStage {
    float red[8];
    uint32_t status, fallback;
    float hpred[2];   // the H per cut the classification's layer widths were predicted from: a larger one in this step's regions -> the general path
};
// launch 1: class byte per slot + per-block class counts
// (a last-block-done ticket that would fold launch 2 into this one costs 230 us at N = 1M: 4096 device-scope fences + 4096 adds
//  on one word -- measured, forced one-rank slab step 0.72 -> 0.95 ms)
// (w_l / w_r: the layer widths of the left / right cut as PREDICTED from the previous step's H_cut; `region`: the distance from a cut
//  within which a particle's h counts towards this step's H_cut -- reduced here, compared with the prediction after the round)
__global__ __launch_bounds__(256) void k_slab_classify(uint32_t n, const float4* __restrict__ pm, const uint8_t* __restrict__ owned, float cut_lo,
                                                        float cut_hi, float w_l, float w_r, float region, int has_left, int has_right, uint8_t* __restrict__ cls,
                                                        uint32_t* __restrict__ blk_cnt, uint32_t* __restrict__ counts)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    uint32_t c = 7u;
    bool bad = false;
    float hl = 0.f, hr = 0.f;
    if (i < n) {
        const float4 P = pm[i];
        const float x = P.x;
        if (owned && !owned[i]) c = SC_GHOST;
        else if (has_left && x < cut_lo) {
            c = SC_MIG_L;
            bad = x < cut_lo - w_l;             // the left rank's halo test is !(x < its cut_hi - w), its cut_hi == my cut_lo, the same w
        } else if (has_right && !(x < cut_hi)) {
            c = SC_MIG_R;
            bad = !(x < cut_hi + w_r);          // the right rank's: x < its cut_lo + w
        } else {
            const bool l = has_left && x < cut_lo + w_l, r = has_right && !(x < cut_hi - w_r);
            bad = l && r;
            c = l ? SC_HALO_L : (r ? SC_HALO_R : SC_STAY);
        }
        cls[i] = (uint8_t)c;
        if (c != SC_GHOST) {
            if (has_left && x < cut_lo + region) hl = P.w;
            if (has_right && !(x < cut_hi - region)) hr = P.w;
        }
    }
    hl = wave_max(hl);
    hr = wave_max(hr);
    if ((threadIdx.x & 63u) == 0u) {
        if (hl > 0.f) atomicMax(&counts[RC_HL], __float_as_uint(hl));
        if (hr > 0.f) atomicMax(&counts[RC_HR], __float_as_uint(hr));
    }
    __shared__ uint32_t s_cnt[4][4];   // [wave][class - 1]
    const uint32_t t = threadIdx.x, wave = t >> 6, lane = t & 63u;
#pragma unroll
    for (uint32_t k = 1; k <= 4; k++) {
        const uint64_t m = __ballot(c == k);
        if (lane == 0) s_cnt[wave][k - 1] = (uint32_t)__popcll(m);
    }
    if (__ballot(bad) != 0ull && lane == 0) atomicOr(&counts[RC_BAD], 1u);
    __syncthreads();
    if (t < 4) blk_cnt[blockIdx.x * 4 + t] = s_cnt[0][t] + s_cnt[1][t] + s_cnt[2][t] + s_cnt[3][t];
}

// launch 2, one 1024-thread block: exclusive scan of the block counts over the blocks (thread t owns a contiguous run of blocks),
// the totals (counts[0 .. 3] = classes 1 .. 4, counts[4] = bad) and the words of the collective round, staged behind the counters:
//   stage[0 .. 7] header values, stage[8] = -status, stage[9] = -"general path" (floats: ONE min all-reduce takes all ten),
//   stage[10 .. 12] = (migrants, halo members, largest h in the cut's region) for the left neighbour, stage[13 .. 15] for the right
//   one, stage[16 .. 21] = 0 (received)
__global__ __launch_bounds__(1024) void k_slab_scan(uint32_t nb, const uint32_t* __restrict__ blk_cnt, uint32_t* __restrict__ blk_off,
                                                     uint32_t* __restrict__ counts, uint32_t* __restrict__ stage, RefreshStage rs)
{
    __shared__ uint32_t s_wave[16][4];
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    const uint32_t per = (nb + 1023u) / 1024u;
    const uint32_t b0 = min(nb, t * per), b1 = min(nb, b0 + per);
    uint32_t sum[4] = {0u, 0u, 0u, 0u};
    const uint4* __restrict__ cnt4 = reinterpret_cast<const uint4*>(blk_cnt);   // (one 16-byte load per block: 32768 blocks at 8M particles, one thread block)
    for (uint32_t b = b0; b < b1; b++) {
        const uint4 v = cnt4[b];
        sum[0] += v.x; sum[1] += v.y; sum[2] += v.z; sum[3] += v.w;
    }
    uint32_t inc[4];
    for (int k = 0; k < 4; k++) {
        uint32_t v = sum[k];
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t u = __shfl_up(v, d);
            if ((int)lane >= d) v += u;
        }
        inc[k] = v;
        if (lane == 63u) s_wave[wave][k] = v;
    }
    __syncthreads();
    uint32_t run[4];
    for (int k = 0; k < 4; k++) {
        uint32_t base = 0;
        for (uint32_t w = 0; w < wave; w++) base += s_wave[w][k];
        run[k] = base + inc[k] - sum[k];
    }
    uint4* __restrict__ off4 = reinterpret_cast<uint4*>(blk_off);
    for (uint32_t b = b0; b < b1; b++) {
        const uint4 v = cnt4[b];
        off4[b] = make_uint4(run[0], run[1], run[2], run[3]);
        run[0] += v.x; run[1] += v.y; run[2] += v.z; run[3] += v.w;
    }
    if (t == 1023u) {
        uint32_t is_bad = counts[RC_BAD];
        counts[RC_BAD] = 0u;
        // (h >= 0: the float bits order like the values.  The neighbour holds the same prediction for the shared cut and tests its own
        //  particles; the verdicts meet in the round's max-reduced fallback flag)
        if (counts[RC_HL] > __float_as_uint(rs.hpred[0]) || counts[RC_HR] > __float_as_uint(rs.hpred[1])) is_bad = 1u;
        for (int k = 0; k < 4; k++) counts[k] = run[k];
        counts[4] = is_bad;
        for (int k = 0; k < 8; k++) stage[k] = __float_as_uint(rs.red[k]);
        stage[8] = __float_as_uint(-(float)rs.status);
        stage[9] = __float_as_uint((rs.fallback || is_bad) ? -1.f : 0.f);
        stage[10] = run[2];   // SC_MIG_L
        stage[11] = run[0];   // SC_HALO_L
        stage[12] = counts[RC_HL];   // largest h in the left cut's region (float bits)
        stage[13] = run[3];   // SC_MIG_R
        stage[14] = run[1];   // SC_HALO_R
        stage[15] = counts[RC_HR];
        counts[5] = counts[RC_HL];   // (beside the class totals: what the transports without device staging read)
        counts[6] = counts[RC_HR];
        counts[RC_HL] = counts[RC_HR] = 0u;
        for (int k = 16; k < 22; k++) stage[k] = 0u;
    }
}

// pass 2: migrant records and the halo index lists, in slot order (deterministic: the same arrays give the same order)
__global__ __launch_bounds__(256) void k_slab_pack(uint32_t n, const uint8_t* __restrict__ cls, const uint32_t* __restrict__ blk_off,
                                                    const float4* __restrict__ pm, const float2* __restrict__ vel, const uint32_t* __restrict__ orig,
                                                    const float* __restrict__ lvl, const float* __restrict__ lvlold, const float* __restrict__ h2n,
                                                    const float* __restrict__ lam_sum, const uint8_t* __restrict__ szc, float* __restrict__ send_l,
                                                    float* __restrict__ send_r, uint32_t* __restrict__ halo_idx, uint32_t halo_r_base)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    const uint32_t c = i < n ? (uint32_t)cls[i] : 7u;
    __shared__ uint32_t s_cnt[4][4];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    uint32_t my_rank = 0;
#pragma unroll
    for (uint32_t k = 1; k <= 4; k++) {
        const uint64_t m = __ballot(c == k);
        if (lane == 0) s_cnt[wave][k - 1] = (uint32_t)__popcll(m);
        if (c == k) my_rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    }
    __syncthreads();
    if (c < 1u || c > 4u) return;
    uint32_t pos = blk_off[blockIdx.x * 4 + (c - 1u)] + my_rank;
    for (uint32_t w = 0; w < wave; w++) pos += s_cnt[w][c - 1u];
    if (c == SC_HALO_L) halo_idx[pos] = i;
    else if (c == SC_HALO_R) halo_idx[halo_r_base + pos] = i;
    else {
        const float4 p = pm[i];
        const float2 v = vel[i];
        float* r = (c == SC_MIG_L ? send_l : send_r) + (size_t)pos * MIG_WORDS;
        r[0] = p.x; r[1] = p.y; r[2] = p.z; r[3] = p.w; r[4] = v.x; r[5] = v.y;
        r[6] = __uint_as_float(orig[i]);
        r[7] = lvl[i];
        r[8] = lvlold[i];
        r[9] = h2n[i];
        r[10] = lam_sum[i];
        r[11] = __uint_as_float((uint32_t)szc[i]);
    }
}
__global__ __launch_bounds__(256) void k_iota_u32(uint32_t* __restrict__ out, uint32_t cnt, uint32_t first)
{
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    if (k < cnt) out[k] = first + k;
}
// A rank that cannot take what its neighbours hand it must not leave the step alone (the others would wait in their next
// collective): it raises the guard word, goes on WITHOUT the arrivals / ghosts -- every exchange keeps the sizes that were agreed --
// and the step ends on every rank through the all-reduced guards (solver totals, agree_guards_queued) with SPH_ERR_CAPACITY.
__global__ void k_raise(DeviceStatus* st, uint32_t code, uint32_t info)
{
    if (threadIdx.x == 0 && blockIdx.x == 0 && atomicCAS(&st->error, 0u, code) == 0u) st->info = info;
}
__global__ __launch_bounds__(256) void k_fill_u32(uint32_t* __restrict__ out, uint32_t cnt, uint32_t v)
{
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    if (k < cnt) out[k] = v;
}
static void slab_out_of_room(sph_ctx* c, unsigned long long need)
{
    (void)c->fail(SPH_ERR_CAPACITY, "slab of rank %d needs %llu slots, capacity %llu", c->dist.rank, need, (unsigned long long)c->cap);
    hipLaunchKernelGGL(k_raise, dim3(1), dim3(64), 0, c->stream, c->status.as<DeviceStatus>(), (uint32_t)SPH_ERR_CAPACITY, (uint32_t)c->dist.rank);
}

__global__ void k_fill_u8(uint8_t* p, uint32_t n, uint8_t v)
{
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = v;
}


// refresh `field` (words floats per particle) of every member's ghosts from their owners
// `tot_slot` >= 0: the all-reduce of the solver totals of that slot rides in the same call (Comm::exchange_and_allreduce_solver)
int refresh_ghosts(Group& G, std::vector<Member>& M, float* (*sel)(Member&), int words, const char* what, int tot_slot, float* (*sel2)(Member&), int stride, int off,
                   const TotalsJob* totals)
{
    if (!G.multi()) return SPH_OK;
    if (stride <= 0) stride = words;
    const int nf = sel2 ? 2 : 1;   // fields per exchange: the second one is packed behind the first in every buffer
    if (totals && nf == 1 && words == 1 && tot_slot >= 0 && M.size() == 1 && M[0].n) {
        // a Jacobi iteration's exchange on a transport with launches of its own (the push transport): pack + totals + push in one launch,
        // wait + unpack in one (Comm::exchange_fused) -- no k_pack_totals, no k_unpack_field
        sph_ctx* c = M[0].c;
        auto& d = c->dist;
        const size_t bytes = 4 * (size_t)std::max(std::max(d.n_halo[0], d.n_halo[1]), std::max(d.n_ghost[0], d.n_ghost[1]));
        if (G.comm->can_fuse_iteration(G, bytes)) {
            const FusedField f{d.halo_src.as<uint32_t>(), {d.n_halo[0], d.n_halo[1]}, d.ghost_dst.as<uint32_t>(), {d.n_ghost[0], d.n_ghost[1]}, d.ghosts_ok,
                               sel(M[0]), stride, off, M[0].a.partials, solver_reduce_blocks(M[0].a.n), M[0].a.ctrl, M[0].a.gate, totals->iter};
            return G.comm->exchange_fused(G, f, tot_slot);
        }
    }
    std::vector<Xfer> x(M.size());
    for (size_t i = 0; i < M.size(); i++) {
        sph_ctx* c = M[i].c;
        (void)hipSetDevice(c->device);
        ProfScope ps(&c->prof, "ghost_pack", c->stream);
        const uint32_t nh = c->dist.n_halo[0] + c->dist.n_halo[1];
        if (totals && M[i].n && nf == 1) {   // the pack launch also adds up the rank's solver totals (its block 0): one launch, neighbours or not
            launch_pack_and_totals(c->stream, nullptr, M[i].a, totals->iter, totals->residual_density, totals->max_avg_error, totals->max_iters,
                                   c->dist.halo_src.as<uint32_t>(), c->dist.n_halo[0], c->dist.n_halo[1], words, stride, off, sel(M[i]), c->dist.send[0].as<float>(),
                                   c->dist.send[1].as<float>());
        } else
        for (int f = 0; f < nf && nh; f++)
            hipLaunchKernelGGL(k_pack_field, dim3((nh + 255) / 256), dim3(256), 0, c->stream, c->dist.halo_src.as<uint32_t>(), c->dist.n_halo[0],
                               c->dist.n_halo[1], words, stride, off, f ? sel2(M[i]) : sel(M[i]), c->dist.send[0].as<float>() + (size_t)f * c->dist.n_halo[0] * words,
                               c->dist.send[1].as<float>() + (size_t)f * c->dist.n_halo[1] * words);
        for (int side = 0; side < 2; side++) {
            const uint32_t cnt = c->dist.n_halo[side];
            x[i].send[side] = c->dist.send[side].p;
            x[i].send_bytes[side] = (size_t)cnt * words * 4 * nf;
            x[i].recv[side] = c->dist.recv[side].p;
            x[i].recv_bytes[side] = (size_t)c->dist.n_ghost[side] * words * 4 * nf;
        }
    }
    int rc = tot_slot >= 0 ? G.comm->exchange_and_allreduce_solver(G, x, tot_slot) : G.comm->exchange(G, x);
    if (rc) return rc;
    for (size_t i = 0; i < M.size(); i++) {
        sph_ctx* c = M[i].c;
        (void)hipSetDevice(c->device);
        ProfScope ps(&c->prof, "ghost_unpack", c->stream);
        const uint32_t ng = c->dist.ghosts_ok ? c->dist.n_ghost[0] + c->dist.n_ghost[1] : 0u;   // (no ghost slots: received, dropped)
        for (int f = 0; f < nf && ng; f++)
            hipLaunchKernelGGL(k_unpack_field, dim3((ng + 255) / 256), dim3(256), 0, c->stream, c->dist.ghost_dst.as<uint32_t>(), c->dist.n_ghost[0],
                               c->dist.n_ghost[1], words, stride, off, (const float*)x[i].recv[0] + (size_t)f * c->dist.n_ghost[0] * words,
                               (const float*)x[i].recv[1] + (size_t)f * c->dist.n_ghost[1] * words, f ? sel2(M[i]) : sel(M[i]));
    }
    (void)what;
    return SPH_OK;
}


int ensure_dist_buffers(sph_ctx* c, uint32_t n)
{
    auto& d = c->dist;
    const size_t cap = c->cap ? c->cap : 1;
    (void)n;
    HIPCHK(c, d.owned.ensure(cap));
    HIPCHK(c, d.ring1.ensure(cap));
    HIPCHK(c, d.ring1_src.ensure(cap));
    HIPCHK(c, d.halo_idx.ensure(cap * 4));
    HIPCHK(c, d.halo_pos.ensure(cap * 4));
    HIPCHK(c, d.halo_src.ensure(cap * 4));
    HIPCHK(c, d.ghost_dst.ensure(cap * 4));
    for (int s = 0; s < 2; s++) {
        HIPCHK(c, d.send[s].ensure(cap * MIG_WORDS * 4 / 2 + 1024));
        HIPCHK(c, d.recv[s].ensure(cap * MIG_WORDS * 4 / 2 + 1024));
    }
    if (!d.counts.p) {
        HIPCHK(c, d.counts.ensure(256));
        HIPCHK(c, hipMemset(d.counts.p, 0, 256));   // (the bad-flag word of the fused refresh is taken and cleared by k_slab_scan)
    }
    HIPCHK(c, d.solver_tot.ensure(128));   // two slots of 6 doubles (chained solves)
    HIPCHK(c, d.cls.ensure(cap));
    HIPCHK(c, d.edge.ensure(cap));
    if (!d.xstream) {
        HIPCHK(c, hipStreamCreateWithFlags(&d.xstream, hipStreamNonBlocking));
        for (auto& e : d.ev_x) HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        HIPCHK(c, hipEventCreateWithFlags(&d.ev_pack, hipEventDisableTiming));
        HIPCHK(c, hipEventCreateWithFlags(&d.ev_copied, hipEventDisableTiming));
        HIPCHK(c, hipEventCreateWithFlags(&d.ev_tot, hipEventDisableTiming));
    }
    HIPCHK(c, d.blk.ensure(((cap + 255) / 256) * 8 * sizeof(uint32_t)));
    if (!d.counts_host) {
        HIPCHK(c, hipHostMalloc((void**)&d.counts_host, 256, hipHostMallocMapped));   // 64 B of counters + 192 B of staging
        HIPCHK(c, hipHostGetDevicePointer((void**)&d.counts_host_dev, d.counts_host, 0));
        memset(d.counts_host, 0, 256);
    }
    return SPH_OK;
}

// ---- slab maintenance (multi-rank only) ---------------------------------------------------------------
// part 1: drop last step's ghosts, hand over particles that left the slab
// `red`: the step's header values, min-reduced over all ranks in the SAME round trip that brings the partition counts to the
// host and exchanges them with the x-neighbours (Comm::counts_round)
int partition_and_migrate(Group& G, std::vector<Member>& M, std::vector<int>* moved, std::vector<std::vector<float>>* red)
{
    const size_t nm = M.size();
    int rc = SPH_OK;
    std::vector<uint32_t> n_prev_of;
    // (1) classify the previous arrays: stay / migrate left / migrate right / drop (ghost); stable partition by a
    //     1-pass radix sort on the 2-bit class (reuses the neighbour-build sort: deterministic order)
    for (auto& m : M) {
        sph_ctx* c = m.c;
        auto& d = c->dist;
        (void)hipSetDevice(c->device);
        if ((rc = ensure_dist_buffers(c, m.n))) return rc;
        const uint32_t n_prev = d.have_flags ? d.n_tot : (uint32_t)c->n;
        n_prev_of.push_back(n_prev);
        (void)hipMemsetAsync(d.counts.p, 0, 64, c->stream);
        if (n_prev) {
            ProfScope ps(&c->prof, "slab_partition", c->stream);
            hipLaunchKernelGGL(k_classify_migrate, dim3((n_prev + 255) / 256), dim3(256), 0, c->stream, n_prev, c->pm[c->pcur].as<float4>(),
                               d.have_flags ? d.owned.as<uint8_t>() : (const uint8_t*)nullptr, d.cut_lo, d.cut_hi, d.rank > 0 ? 1 : 0,
                               d.rank + 1 < d.nranks ? 1 : 0, c->key[0].as<uint32_t>(), c->val[0].as<uint32_t>(), d.counts.as<uint32_t>(),
                               d.halo_w[0], d.halo_w[1]);   // the x-neighbour's slab is at least as wide as its ghost layer at the shared cut -- the previous layer's width (0: unknown, every migrant may need another hand-over)
            int res = radix_sort_pairs(c->stream, &c->prof, c->key[0].as<uint32_t>(), c->val[0].as<uint32_t>(), c->key[1].as<uint32_t>(),
                                       c->val[1].as<uint32_t>(), n_prev, 2, c->sort_scratch.as<uint32_t>());
            if (res == 1) {
                std::swap(c->key[0], c->key[1]);
                std::swap(c->val[0], c->val[1]);
            }
            GridP g1{};
            g1.sx = 1;
            const int k = c->cur;
            launch_reorder(c->stream, &c->prof, n_prev, g1, c->key[0].as<uint32_t>(), c->val[0].as<uint32_t>(), c->pm[c->pcur].as<float4>(),
                           c->vel[k].as<float2>(), c->orig[k].as<uint32_t>(), c->lvl[k].as<float>(), c->lvlold[k].as<float>(),
                           c->pm[c->pcur ^ 1].as<float4>(), c->vel[k ^ 1].as<float2>(), c->orig[k ^ 1].as<uint32_t>(), c->lvl[k ^ 1].as<float>(),
                           c->lvlold[k ^ 1].as<float>(), c->cxy.as<uint32_t>(), c->h2n[k].as<float>(), c->h2n[k ^ 1].as<float>(),
                           c->lam_sum.as<float>(), c->lam_prev.as<float>(), nullptr, c->szc[k].as<uint8_t>(), c->szc[k ^ 1].as<uint8_t>());
            c->cur = k ^ 1;
            c->pcur ^= 1;
            std::swap(c->lam_sum, c->lam_prev);   // the permuted lambda sums are the CURRENT ones again (the cell sort moves them on)
        }
    }
    // (2) migrants: counts -> host and x-neighbours in one round trip, then the records
    std::vector<uint32_t> tl(nm), tr(nm), fl(nm), fr(nm);
    std::vector<char> over(nm, 0);
    for (auto& m : M) m.c->hint_word = nullptr;
    if ((rc = G.comm->counts_round(G, 0, red, nullptr, tl, tr, fl, fr))) return rc;   // (a device failure here is fatal for the whole job)
    for (size_t i = 0; i < nm; i++) {
        auto& d = M[i].c->dist;
        // class 0 = everybody else (see block_class_counts); class 3 = ghosts of the previous step, dropped
        d.counts_host[0] = n_prev_of[i] - d.counts_host[1] - d.counts_host[2] - d.counts_host[3];
    }
    if (moved)   // somebody may have to be handed on once more
        for (size_t i = 0; i < nm; i++) (*moved)[i] = (tl[i] + tr[i]) && M[i].c->dist.counts_host[RC_FAR] ? 1 : 0;
    std::vector<Xfer> x(nm);
    for (size_t i = 0; i < nm; i++) {
        sph_ctx* c = M[i].c;
        auto& d = c->dist;
        (void)hipSetDevice(c->device);
        const uint32_t n_stay = d.counts_host[0];
        if ((uint64_t)n_stay + fl[i] + fr[i] > c->cap) {   // the arrivals are received (the sizes are agreed) and dropped
            slab_out_of_room(c, (unsigned long long)n_stay + fl[i] + fr[i]);
            over[i] = 1;
        }
        const int k = c->cur;
        const uint32_t base[2] = {n_stay, n_stay + tl[i]};
        const uint32_t cnt[2] = {tl[i], tr[i]};
        for (int side = 0; side < 2; side++) {
            if (cnt[side])
                hipLaunchKernelGGL(k_pack_migrants, dim3((cnt[side] + 255) / 256), dim3(256), 0, c->stream, base[side], cnt[side],
                                   c->pm[c->pcur].as<float4>(), c->vel[k].as<float2>(), c->orig[k].as<uint32_t>(), c->lvl[k].as<float>(),
                                   c->lvlold[k].as<float>(), c->h2n[k].as<float>(), c->lam_sum.as<float>(), c->szc[k].as<uint8_t>(),
                                   d.send[side].as<float>());
            x[i].send[side] = d.send[side].p;
            x[i].send_bytes[side] = (size_t)cnt[side] * MIG_WORDS * 4;
            x[i].recv[side] = d.recv[side].p;
        }
        x[i].recv_bytes[0] = (size_t)fl[i] * MIG_WORDS * 4;
        x[i].recv_bytes[1] = (size_t)fr[i] * MIG_WORDS * 4;
    }
    if ((rc = G.comm->exchange(G, x))) return rc;
    for (size_t i = 0; i < nm; i++) {
        sph_ctx* c = M[i].c;
        auto& d = c->dist;
        (void)hipSetDevice(c->device);
        const int k = c->cur;
        const uint32_t n_stay = d.counts_host[0];
        const uint32_t cnt[2] = {over[i] ? 0u : fl[i], over[i] ? 0u : fr[i]};
        const uint32_t base[2] = {n_stay, n_stay + cnt[0]};
        for (int side = 0; side < 2; side++)
            if (cnt[side])
                hipLaunchKernelGGL(k_unpack_migrants, dim3((cnt[side] + 255) / 256), dim3(256), 0, c->stream, base[side], cnt[side],
                                   (const float*)x[i].recv[side], c->pm[c->pcur].as<float4>(), c->vel[k].as<float2>(), c->orig[k].as<uint32_t>(),
                                   c->lvl[k].as<float>(), c->lvlold[k].as<float>(), c->h2n[k].as<float>(), c->lam_sum.as<float>(),
                                   c->szc[k].as<uint8_t>());
        c->n = n_stay + cnt[0] + cnt[1];
        d.have_flags = false;
        d.n_tot = (uint32_t)c->n;
        M[i].n = (uint32_t)c->n;
    }
    return SPH_OK;
}

// Move the cuts so that every rank owns the same number of particles ("rebalance every M steps by
// shifting column cuts"): global x range (all-reduce min) -> histogram of the owned x over REBALANCE_BINS bins (all-reduce sum)
// -> cuts at the quantiles, identical on every rank (same integers, same floats).  Cuts that would make a slab narrower than
// the ghost exchange allows are not applied.  The particles follow in the migration rounds of the caller.
int rebalance_cuts(Group& G, std::vector<Member>& M, bool* applied)
{
    const size_t nm = M.size();
    int rc;
    *applied = false;
    for (auto& m : M) {
        sph_ctx* c = m.c;
        auto& d = c->dist;
        (void)hipSetDevice(c->device);
        if ((rc = ensure_dist_buffers(c, m.n))) return rc;
        HIPCHK(c, d.hist.ensure(REBALANCE_BINS * 4));
        const uint32_t n_prev = d.have_flags ? d.n_tot : (uint32_t)c->n;
        uint32_t* mm = d.counts.as<uint32_t>() + 8;
        const uint32_t init[2] = {0xffffffffu, 0u};
        HIPCHK(c, hipMemcpyAsync(mm, init, 8, hipMemcpyHostToDevice, c->stream));
        if (n_prev)
            hipLaunchKernelGGL(k_minmax_x, dim3((n_prev + 255) / 256), dim3(256), 0, c->stream, n_prev, c->pm[c->pcur].as<float4>(),
                               d.have_flags ? d.owned.as<uint8_t>() : (const uint8_t*)nullptr, mm);
        HIPCHK(c, hipMemcpyAsync(d.counts_host + 8, mm, 8, hipMemcpyDeviceToHost, c->stream));
    }
    if ((rc = agree(G, wait_all(G)))) return rc;
    std::vector<std::vector<float>> mmr(nm, std::vector<float>(2));
    for (size_t i = 0; i < nm; i++) {
        const uint32_t lo = M[i].c->dist.counts_host[8], hi = M[i].c->dist.counts_host[9];
        const bool any = lo <= hi;
        mmr[i][0] = any ? f32_from_ordered(lo) : INFINITY;
        mmr[i][1] = any ? -f32_from_ordered(hi) : INFINITY;
    }
    if ((rc = G.comm->allreduce_min_f32(G, mmr))) return rc;
    const float gmin = mmr[0][0], gmax = -mmr[0][1];
    if (!(gmax > gmin) || !std::isfinite(gmin) || !std::isfinite(gmax)) return SPH_OK;
    const float binw = (gmax - gmin) / (float)REBALANCE_BINS;
    if (!(binw > 0.f)) return SPH_OK;
    std::vector<std::vector<uint32_t>> hist(nm, std::vector<uint32_t>(REBALANCE_BINS));
    for (size_t i = 0; i < nm; i++) {
        sph_ctx* c = M[i].c;
        auto& d = c->dist;
        (void)hipSetDevice(c->device);
        ProfScope ps(&c->prof, "slab_rebalance", c->stream);
        const uint32_t n_prev = d.have_flags ? d.n_tot : (uint32_t)c->n;
        (void)hipMemsetAsync(d.hist.p, 0, REBALANCE_BINS * 4, c->stream);
        if (n_prev)
            hipLaunchKernelGGL(k_hist_x, dim3((n_prev + 255) / 256), dim3(256), 0, c->stream, n_prev, c->pm[c->pcur].as<float4>(),
                               d.have_flags ? d.owned.as<uint8_t>() : (const uint8_t*)nullptr, gmin, binw, d.hist.as<uint32_t>());
        HIPCHK(c, hipMemcpyAsync(hist[i].data(), d.hist.p, REBALANCE_BINS * 4, hipMemcpyDeviceToHost, c->stream));
    }
    if ((rc = agree(G, wait_all(G)))) return rc;
    if ((rc = G.comm->allreduce_sum_u32(G, hist))) return rc;
    const int nr = M[0].c->dist.nranks;
    uint64_t total = 0;
    for (uint32_t v : hist[0]) total += v;
    if (total == 0) return SPH_OK;
    std::vector<float> cuts((size_t)nr + 1);
    cuts[0] = gmin;
    cuts[(size_t)nr] = gmax;
    uint64_t cum = 0;
    int b = 0;
    for (int r = 1; r < nr; r++) {
        const uint64_t target = total * (uint64_t)r / (uint64_t)nr;
        while (b < REBALANCE_BINS && cum + hist[0][b] < target) cum += hist[0][b++];
        // the cut sits at the upper edge of the bin in which the cumulative count reaches the target
        cuts[(size_t)r] = gmin + (float)(b + 1) * binw;
    }
    // every slab must stay wider than two ghost layers (build_ghost_layer refuses narrower ones): keep the old cuts otherwise
    const float min_width = 4.5f * M[0].c->h_max_step;
    for (int r = 0; r < nr; r++)
        if (!(cuts[(size_t)r + 1] - cuts[(size_t)r] >= min_width)) return SPH_OK;
    for (auto& m : M) {
        auto& d = m.c->dist;
        if (d.rank > 0) d.cut_lo = cuts[(size_t)d.rank];
        if (d.rank + 1 < nr) d.cut_hi = cuts[(size_t)d.rank + 1];
        d.rebalances++;
    }
    *applied = true;
    return SPH_OK;
}

// part 2: ghost layer -- owned particles within halo_width of a cut are copied to that neighbour, in array
// order (stable partition again)
int build_ghost_layer(Group& G, std::vector<Member>& M, float base_k, float slack_w, float h_max, int status_in)
{
    const size_t nm = M.size();
    int rc = SPH_OK;
    std::vector<uint32_t> tl(nm), tr(nm), fl(nm), fr(nm);
    std::vector<Xfer> x(nm);
    // ---- H per cut: the largest h of either rank's particles within a global-width layer of the cut (one neighbour round)
    const float region = base_k * h_max + slack_w;
    for (auto& m : M) {
        sph_ctx* c = m.c;
        auto& d = c->dist;
        (void)hipSetDevice(c->device);
        const uint32_t n = (uint32_t)c->n;
        (void)hipMemsetAsync(d.counts.p, 0, 64, c->stream);
        if (n)
            hipLaunchKernelGGL(k_region_hmax, dim3((n + 255) / 256), dim3(256), 0, c->stream, n, c->pm[c->pcur].as<float4>(), d.cut_lo + region, d.cut_hi - region,
                               d.rank > 0 ? 1 : 0, d.rank + 1 < d.nranks ? 1 : 0, d.counts.as<uint32_t>() + RC_HL);
        HIPCHK(c, hipMemcpyAsync(d.counts_host + RC_HL, d.counts.as<uint32_t>() + RC_HL, 8, hipMemcpyDeviceToHost, c->stream));
    }
    if ((rc = agree(G, wait_all(G)))) return rc;
    {
        std::vector<uint32_t> hl(nm), hr(nm), nl(nm), nr(nm);
        for (size_t i = 0; i < nm; i++) {
            hl[i] = M[i].c->dist.counts_host[RC_HL];
            hr[i] = M[i].c->dist.counts_host[RC_HR];
        }
        if ((rc = G.comm->neighbour_counts(G, hl, hr, nl, nr, nullptr))) return rc;   // (float bits of non-negative values: the exchange is word-wise)
        for (size_t i = 0; i < nm; i++) {
            auto& d = M[i].c->dist;
            float a, b, na, nb;
            memcpy(&a, &hl[i], 4); memcpy(&b, &hr[i], 4); memcpy(&na, &nl[i], 4); memcpy(&nb, &nr[i], 4);
            d.hcut[0] = d.rank > 0 ? fmaxf(a, na) : 0.f;
            d.hcut[1] = d.rank + 1 < d.nranks ? fmaxf(b, nb) : 0.f;
            for (int s = 0; s < 2; s++) d.halo_w[s] = d.hcut[s] > 0.f ? base_k * d.hcut[s] + slack_w : 0.f;
        }
    }
    for (auto& m : M) {
        sph_ctx* c = m.c;
        auto& d = c->dist;
        (void)hipSetDevice(c->device);
        const uint32_t n = (uint32_t)c->n;
        (void)hipMemsetAsync(d.counts.p, 0, 64, c->stream);
        if (n) {
            ProfScope ps(&c->prof, "slab_halo_select", c->stream);
            hipLaunchKernelGGL(k_classify_halo, dim3((n + 255) / 256), dim3(256), 0, c->stream, n, c->pm[c->pcur].as<float4>(), d.cut_lo + d.halo_w[0],
                               d.cut_hi - d.halo_w[1], d.rank > 0 ? 1 : 0, d.rank + 1 < d.nranks ? 1 : 0, c->key[0].as<uint32_t>(),
                               c->val[0].as<uint32_t>(), d.counts.as<uint32_t>());
            int res = radix_sort_pairs(c->stream, &c->prof, c->key[0].as<uint32_t>(), c->val[0].as<uint32_t>(), c->key[1].as<uint32_t>(),
                                       c->val[1].as<uint32_t>(), n, 2, c->sort_scratch.as<uint32_t>());
            if (res == 1) {
                std::swap(c->key[0], c->key[1]);
                std::swap(c->val[0], c->val[1]);
            }
        }
    }
    // one round trip: the halo counts to the host and to the x-neighbours, and one agreement over all ranks on the width check
    // (a slab narrower than two ghost layers) and on whatever the caller brings (`status_in`) -- every rank leaves together
    for (auto& m : M) m.c->hint_word = nullptr;
    {
        int agreed = status_in;
        if ((rc = G.comm->counts_round(G, 4, nullptr, &agreed, tl, tr, fl, fr))) return rc;
        for (size_t i = 0; i < nm; i++) {
            auto& d = M[i].c->dist;
            if (d.counts_host[4 + 3]) rc = M[i].c->fail(SPH_ERR_UNSUPPORTED, "slab of rank %d is narrower than two ghost layers", d.rank);
            d.counts_host[4 + 0] = (uint32_t)M[i].c->n - tl[i] - tr[i] - d.counts_host[4 + 3];   // class 0 is not counted on the device
        }
        if (rc) return rc;
        if (status_in) return status_in;
        if (agreed) return M[0].c->fail(agreed, "another rank of the slab decomposition reported status %d", agreed);
    }
    for (size_t i = 0; i < nm; i++) {
        sph_ctx* c = M[i].c;
        auto& d = c->dist;
        (void)hipSetDevice(c->device);
        const uint32_t n = (uint32_t)c->n;
        d.ghosts_ok = true;
        if ((uint64_t)n + fl[i] + fr[i] > c->cap) {   // the ghost records are received and dropped: no ghost slots in this step
            slab_out_of_room(c, (unsigned long long)n + fl[i] + fr[i]);
            d.ghosts_ok = false;
        }
        const uint32_t n_none = d.counts_host[4 + 0];
        d.n_halo[0] = tl[i];
        d.n_halo[1] = tr[i];
        d.n_ghost[0] = fl[i];
        d.n_ghost[1] = fr[i];
        const uint32_t nh = tl[i] + tr[i];
        // halo index list = sorted values behind the `none` class: [left..., right...]
        if (nh) HIPCHK(c, hipMemcpyAsync(d.halo_idx.p, c->val[0].as<uint32_t>() + n_none, (size_t)nh * 4, hipMemcpyDeviceToDevice, c->stream));
        const int k = c->cur;
        for (int side = 0; side < 2; side++) {
            const uint32_t cnt = d.n_halo[side], off = side == 0 ? 0 : d.n_halo[0];
            if (cnt)
                hipLaunchKernelGGL(k_pack_ghosts, dim3((cnt + 255) / 256), dim3(256), 0, c->stream, d.halo_idx.as<uint32_t>() + off, cnt,
                                   c->pm[c->pcur].as<float4>(), c->vel[k].as<float2>(), c->orig[k].as<uint32_t>(), d.send[side].as<float>());
            x[i].send[side] = d.send[side].p;
            x[i].send_bytes[side] = (size_t)cnt * GHOST_WORDS * 4;
            x[i].recv[side] = d.recv[side].p;
            x[i].recv_bytes[side] = (size_t)d.n_ghost[side] * GHOST_WORDS * 4;
        }
    }
    if ((rc = G.comm->exchange(G, x))) return rc;
    for (size_t i = 0; i < nm; i++) {
        sph_ctx* c = M[i].c;
        auto& d = c->dist;
        (void)hipSetDevice(c->device);
        const int k = c->cur;
        const uint32_t n = (uint32_t)c->n;
        const uint32_t base[2] = {n, n + d.n_ghost[0]};
        for (int side = 0; side < 2; side++)
            if (d.n_ghost[side] && d.ghosts_ok)
                hipLaunchKernelGGL(k_unpack_ghosts, dim3((d.n_ghost[side] + 255) / 256), dim3(256), 0, c->stream, base[side], d.n_ghost[side],
                                   (const float*)x[i].recv[side], c->pm[c->pcur].as<float4>(), c->vel[k].as<float2>(), c->orig[k].as<uint32_t>(),
                                   c->lvl[k].as<float>(), c->lvlold[k].as<float>(), d.ring1_src.as<uint8_t>(), side == 0 ? 0u : d.n_ghost[0],
                                   side == 0 ? d.cut_lo - 2.f * d.hcut[0] : d.cut_hi + 2.f * d.hcut[1], side);
        d.n_tot = d.ghosts_ok ? n + d.n_ghost[0] + d.n_ghost[1] : n;
        M[i].n = d.n_tot;
        // pre-sort index -> position in my halo list
        if (d.n_tot) (void)hipMemsetAsync(d.halo_pos.p, 0xff, (size_t)d.n_tot * 4, c->stream);
        const uint32_t nh = d.n_halo[0] + d.n_halo[1];
        if (nh) hipLaunchKernelGGL(k_halo_pos, dim3((nh + 255) / 256), dim3(256), 0, c->stream, d.halo_idx.as<uint32_t>(), nh, d.halo_pos.as<uint32_t>());
    }
    return SPH_OK;
}

// The ordinary step's slab maintenance in ONE round trip (the general path above takes two, plus two partition sorts and a
// reorder of every array): classify every slot of the previous arrays once -- stay / stay in a halo / migrate / old ghost --,
// exchange the counts, hand over the migrants, append them and the neighbours' ghost records BEHIND the previous arrays, and let
// the cell sort drop the slots that left (key = one past the last cell).  A migrant belongs to its receiver's halo towards the
// sender, so each side knows the ghost-record counts from the first round: mine from the left = the left rank's staying halo
// members + my own migrants to it.
// `h_pred`: the ghost width is a multiple of the all-reduced h_max, which only arrives with the round -- the previous step's
// value stands in (masses do not change inside a step) and is compared afterwards.  *fused = false: nothing was applied
// (a rank without a prediction, a different h_max, a deep migrant, a narrow slab) and the caller takes the general path;
// `red` is reduced either way.
int slab_refresh_fused(Group& G, std::vector<Member>& M, std::vector<std::vector<float>>& red, float base_k, float slack_k, bool* fused)
{
    const size_t nm = M.size();
    int rc = SPH_OK;
    *fused = false;
    int fallback = 0, status = SPH_OK;
    const float h_pred = M[0].c->h_max_step;
    std::vector<uint32_t> n_prev_of(nm);
    for (size_t i = 0; i < nm; i++) {
        sph_ctx* c = M[i].c;
        auto& d = c->dist;
        (void)hipSetDevice(c->device);
        if ((rc = ensure_dist_buffers(c, M[i].n))) return rc;
        const uint32_t n_prev = d.have_flags ? d.n_tot : (uint32_t)c->n;
        n_prev_of[i] = n_prev;
        if (!(c->h_max_step > 0.f) || c->h_max_step != h_pred) fallback = 1;
        // the layer widths as the previous step's H per cut predicts them (agreed with the neighbours then; compared with this step's
        // after the round: a larger particle that came near a cut sends the step to the general path, which measures first)
        const float slack_w = slack_k * h_pred, region = base_k * h_pred + slack_w;
        const float w_l = d.hcut[0] > 0.f ? base_k * d.hcut[0] + slack_w : 0.f, w_r = d.hcut[1] > 0.f ? base_k * d.hcut[1] + slack_w : 0.f;
        const bool has_l = d.rank > 0, has_r = d.rank + 1 < d.nranks;
        if (has_l && has_r && !(d.cut_hi - w_r >= d.cut_lo + w_l)) fallback = 1;   // narrower than two ghost layers: the general path reports it
        // two launches: classify + per-block counts; scan, totals and the staging of the round (no memset, no copy, no stage kernel)
        const uint32_t nb = (n_prev + 255u) / 256u;
        RefreshStage rs{};
        for (int k = 0; k < 8; k++) rs.red[k] = red[i][(size_t)k];
        rs.status = 0u;
        rs.hpred[0] = d.hcut[0];
        rs.hpred[1] = d.hcut[1];
        rs.fallback = (uint32_t)fallback;   // what this process knows so far (a later member's verdict reaches the round through the host)
        ProfScope ps(&c->prof, "slab_refresh", c->stream);
        if (n_prev)
            hipLaunchKernelGGL(k_slab_classify, dim3(nb), dim3(256), 0, c->stream, n_prev, c->pm[c->pcur].as<float4>(),
                               d.have_flags ? d.owned.as<uint8_t>() : (const uint8_t*)nullptr, d.cut_lo, d.cut_hi, w_l, w_r, region, has_l ? 1 : 0, has_r ? 1 : 0,
                               d.cls.as<uint8_t>(), d.blk.as<uint32_t>(), d.counts.as<uint32_t>());
        hipLaunchKernelGGL(k_slab_scan, dim3(1), dim3(1024), 0, c->stream, nb, d.blk.as<uint32_t>(), d.blk.as<uint32_t>() + (size_t)nb * 4,
                           d.counts.as<uint32_t>(), d.counts.as<uint32_t>() + 16, rs);
        c->hint_word = nullptr;
    }
    for (auto& m : M) dbg_sync(m.c, "fused: classify + scan", 0);
    std::vector<RefreshCounts> rcs(nm);
    if ((rc = G.comm->refresh_round(G, &red, &status, &fallback, rcs))) return rc;
    if (M[0].c->opt.debug_counts)
        for (size_t i = 0; i < nm; i++)
            fprintf(stderr, "[sph debug] rank %zu: n_prev %u owned %llu mig %u %u halo %u %u in_mig %u %u in_halo %u %u fallback %d cap %llu\n", i, n_prev_of[i],
                    (unsigned long long)M[i].c->n, rcs[i].mig[0], rcs[i].mig[1], rcs[i].halo[0], rcs[i].halo[1], rcs[i].in_mig[0], rcs[i].in_mig[1],
                    rcs[i].in_halo[0], rcs[i].in_halo[1], fallback, (unsigned long long)M[i].c->cap);
    if (status) return M[0].c->fail(status, "another rank of the slab decomposition reported status %d", status);
    if (red[0][3] < 0.f) return SPH_OK;                       // a rank's header wait failed: the caller reports it (same value everywhere)
    if (fallback || !(-red[0][0] == h_pred)) return SPH_OK;   // identical on every rank: all-reduced values only
    for (size_t i = 0; i < nm; i++) {   // the layer is selected with the predicted widths (>= this step's: a region whose H grew raised `fallback`)
        auto& d = M[i].c->dist;
        const float slack_w = slack_k * h_pred;
        for (int s = 0; s < 2; s++) d.halo_w[s] = d.hcut[s] > 0.f ? base_k * d.hcut[s] + slack_w : 0.f;
    }

    std::vector<Xfer> x(nm);
    std::vector<char> over(nm, 0);
    for (size_t i = 0; i < nm; i++) {
        sph_ctx* c = M[i].c;
        auto& d = c->dist;
        const RefreshCounts& q = rcs[i];
        const uint32_t n_prev = n_prev_of[i];
        const uint64_t n_pre = (uint64_t)n_prev + q.in_mig[0] + q.in_mig[1] + q.in_halo[0] + q.mig[0] + q.in_halo[1] + q.mig[1];
        d.ghosts_ok = true;
        if (n_pre > c->cap) {
            // no room: the rank keeps its previous slots, receives what was agreed and drops it (see slab_out_of_room)
            (void)hipSetDevice(c->device);
            slab_out_of_room(c, (unsigned long long)n_pre);
            over[i] = 1;
            d.ghosts_ok = false;
        }
    }
    // ---- migrants
    for (size_t i = 0; i < nm; i++) {
        sph_ctx* c = M[i].c;
        auto& d = c->dist;
        const RefreshCounts& q = rcs[i];
        (void)hipSetDevice(c->device);
        const int k = c->cur;
        const uint32_t n_prev = n_prev_of[i];
        d.n_halo[0] = q.halo[0] + q.in_mig[0];
        d.n_halo[1] = q.halo[1] + q.in_mig[1];
        d.n_ghost[0] = q.in_halo[0] + q.mig[0];
        d.n_ghost[1] = q.in_halo[1] + q.mig[1];
        if (n_prev && (q.mig[0] | q.mig[1] | q.halo[0] | q.halo[1])) {
            ProfScope ps(&c->prof, "slab_refresh", c->stream);
            const uint32_t nb = (n_prev + 255u) / 256u;
            hipLaunchKernelGGL(k_slab_pack, dim3(nb), dim3(256), 0, c->stream, n_prev, d.cls.as<uint8_t>(), d.blk.as<uint32_t>() + (size_t)nb * 4,
                               c->pm[c->pcur].as<float4>(), c->vel[k].as<float2>(), c->orig[k].as<uint32_t>(), c->lvl[k].as<float>(),
                               c->lvlold[k].as<float>(), c->h2n[k].as<float>(), c->lam_sum.as<float>(), c->szc[k].as<uint8_t>(), d.send[0].as<float>(),
                               d.send[1].as<float>(), d.halo_idx.as<uint32_t>(), d.n_halo[0]);
        }
        for (int side = 0; side < 2; side++) {
            x[i].send[side] = d.send[side].p;
            x[i].send_bytes[side] = (size_t)q.mig[side] * MIG_WORDS * 4;
            x[i].recv[side] = d.recv[side].p;
            x[i].recv_bytes[side] = (size_t)q.in_mig[side] * MIG_WORDS * 4;
        }
    }
    for (auto& m : M) dbg_sync(m.c, "fused: pack", 1);
    bool any_mig = false;
    for (size_t i = 0; i < nm; i++) any_mig = any_mig || rcs[i].mig[0] || rcs[i].mig[1] || rcs[i].in_mig[0] || rcs[i].in_mig[1];
    // (one rank per process: no migrant in either direction = nothing to pair up, the neighbours see the same zeros)
    if (any_mig && (rc = G.comm->exchange(G, x))) return rc;
    for (auto& m : M) dbg_sync(m.c, "fused: migrants exchanged", 1);
    // ---- arrivals behind the previous arrays, then the ghost records: [previous slots | from left | from right | ghosts left | ghosts right]
    for (size_t i = 0; i < nm; i++) {
        sph_ctx* c = M[i].c;
        auto& d = c->dist;
        const RefreshCounts& q = rcs[i];
        (void)hipSetDevice(c->device);
        const int k = c->cur;
        const uint32_t n_prev = n_prev_of[i];
        const uint32_t base[2] = {n_prev, n_prev + q.in_mig[0]};
        const uint32_t hoff[2] = {q.halo[0], d.n_halo[0] + q.halo[1]};   // the arrivals close the halo list of the side they came from
        for (int side = 0; side < 2; side++) {
            const uint32_t cnt = q.in_mig[side];
            if (!cnt) continue;
            if (over[i]) {   // dropped: their places in the halo list (the agreed length stays) name slot 0 -- in bounds, content irrelevant
                hipLaunchKernelGGL(k_fill_u32, dim3((cnt + 255) / 256), dim3(256), 0, c->stream, d.halo_idx.as<uint32_t>() + hoff[side], cnt, 0u);
                continue;
            }
            hipLaunchKernelGGL(k_unpack_migrants, dim3((cnt + 255) / 256), dim3(256), 0, c->stream, base[side], cnt, (const float*)x[i].recv[side],
                               c->pm[c->pcur].as<float4>(), c->vel[k].as<float2>(), c->orig[k].as<uint32_t>(), c->lvl[k].as<float>(),
                               c->lvlold[k].as<float>(), c->h2n[k].as<float>(), c->lam_sum.as<float>(), c->szc[k].as<uint8_t>());
            hipLaunchKernelGGL(k_iota_u32, dim3((cnt + 255) / 256), dim3(256), 0, c->stream, d.halo_idx.as<uint32_t>() + hoff[side], cnt, base[side]);
        }
        for (int side = 0; side < 2; side++) {
            const uint32_t cnt = d.n_halo[side], off = side == 0 ? 0 : d.n_halo[0];
            if (cnt)
                hipLaunchKernelGGL(k_pack_ghosts, dim3((cnt + 255) / 256), dim3(256), 0, c->stream, d.halo_idx.as<uint32_t>() + off, cnt,
                                   c->pm[c->pcur].as<float4>(), c->vel[k].as<float2>(), c->orig[k].as<uint32_t>(), d.send[side].as<float>());
            x[i].send[side] = d.send[side].p;
            x[i].send_bytes[side] = (size_t)cnt * GHOST_WORDS * 4;
            x[i].recv[side] = d.recv[side].p;
            x[i].recv_bytes[side] = (size_t)d.n_ghost[side] * GHOST_WORDS * 4;
        }
    }
    for (auto& m : M) dbg_sync(m.c, "fused: arrivals unpacked, ghost records packed", 1);
    if ((rc = G.comm->exchange(G, x))) return rc;
    for (auto& m : M) dbg_sync(m.c, "fused: ghost records exchanged", 1);
    for (size_t i = 0; i < nm; i++) {
        sph_ctx* c = M[i].c;
        auto& d = c->dist;
        const RefreshCounts& q = rcs[i];
        (void)hipSetDevice(c->device);
        const int k = c->cur;
        const uint32_t n_prev = n_prev_of[i];
        const uint32_t n_own_prev = (uint32_t)c->n;                         // owned slots of the previous arrays
        const uint32_t n_stay = n_own_prev - q.mig[0] - q.mig[1];
        const uint32_t n_in = over[i] ? 0u : q.in_mig[0] + q.in_mig[1];
        const uint32_t own_end = n_prev + n_in;
        const uint32_t base[2] = {own_end, own_end + d.n_ghost[0]};
        if (over[i]) {   // every entry of the halo map must name a slot that exists (the dropped arrivals' never get one)
            const uint32_t nh_all = d.n_halo[0] + d.n_halo[1];
            if (nh_all) (void)hipMemsetAsync(d.halo_src.p, 0, (size_t)nh_all * 4, c->stream);
        }
        for (int side = 0; side < 2; side++)
            if (d.n_ghost[side] && d.ghosts_ok)
                hipLaunchKernelGGL(k_unpack_ghosts, dim3((d.n_ghost[side] + 255) / 256), dim3(256), 0, c->stream, base[side], d.n_ghost[side],
                                   (const float*)x[i].recv[side], c->pm[c->pcur].as<float4>(), c->vel[k].as<float2>(), c->orig[k].as<uint32_t>(),
                                   c->lvl[k].as<float>(), c->lvlold[k].as<float>(), d.ring1_src.as<uint8_t>(), side == 0 ? 0u : d.n_ghost[0],
                                   side == 0 ? d.cut_lo - 2.f * d.hcut[0] : d.cut_hi + 2.f * d.hcut[1], side);
        dbg_sync(c, "fused: ghosts unpacked (this rank)", 1);
        d.pre = true;
        d.pre_cls_n = n_prev;
        d.pre_own = own_end;
        const uint32_t ng_slots = d.ghosts_ok ? d.n_ghost[0] + d.n_ghost[1] : 0u;
        d.pre_n = own_end + ng_slots;
        c->n = n_stay + n_in;
        d.n_tot = (uint32_t)c->n + ng_slots;
        d.have_flags = false;
        M[i].n = d.n_tot;
        M[i].n_sort = d.pre_n;
        // (halo_pos is not cleared: k_build_maps consults it for halo members only -- class byte 1 / 2, or an arrival)
        const uint32_t nh = d.n_halo[0] + d.n_halo[1];
        if (nh) hipLaunchKernelGGL(k_halo_pos, dim3((nh + 255) / 256), dim3(256), 0, c->stream, d.halo_idx.as<uint32_t>(), nh, d.halo_pos.as<uint32_t>());
        dbg_sync(c, "fused: halo_pos set", 1);
    }
    for (auto& m : M) dbg_sync(m.c, "fused: ghosts unpacked", 2);
    for (size_t i = 0; i < nm; i++) {   // the next step predicts with THIS step's H per cut (both ranks of a cut hold the same two figures)
        auto& d = M[i].c->dist;
        d.hcut[0] = d.rank > 0 ? fmaxf(rcs[i].hreg[0], rcs[i].in_hreg[0]) : 0.f;
        d.hcut[1] = d.rank + 1 < d.nranks ? fmaxf(rcs[i].hreg[1], rcs[i].in_hreg[1]) : 0.f;
    }
    *fused = true;
    return SPH_OK;
}


int slab_maps_after_sort(sph_ctx* c, uint32_t n, bool pre, hipStream_t s)
{
    auto& d = c->dist;
    if (n)
        hipLaunchKernelGGL(k_build_maps, dim3((n + 255) / 256), dim3(256), 0, s, n, pre ? d.pre_own : (uint32_t)c->n, c->val[0].as<uint32_t>(),
                           d.halo_pos.as<uint32_t>(), d.halo_src.as<uint32_t>(), d.ghost_dst.as<uint32_t>(), d.owned.as<uint8_t>(),
                           d.ring1_src.as<uint8_t>(), d.ring1.as<uint8_t>(), pre ? d.cls.as<uint8_t>() : (const uint8_t*)nullptr, pre ? d.pre_cls_n : 0u);
    // split sweep A: the slots whose pressure acceleration reads a ghost (the halo members: every owned particle with a ghost
    // in reach is one) or is one
    if (n && d.nranks > 1) {
        HIPCHK(c, hipMemsetAsync(d.edge.p, 0, n, s));
        const uint32_t nh = d.n_halo[0] + d.n_halo[1], ng = d.ghosts_ok ? d.n_ghost[0] + d.n_ghost[1] : 0u;
        if (nh + ng)
            hipLaunchKernelGGL(k_edge_mark, dim3((nh + ng + 255) / 256), dim3(256), 0, s, d.halo_src.as<uint32_t>(), nh, d.ghost_dst.as<uint32_t>(), ng,
                               d.edge.as<uint8_t>());
    }
    // the {x, y, a^p} records of the ghosts: sweep A writes a record for every owned particle and every ghost of the first
    // ring, sweep B gathers neighbours' records (all of them written) -- but a particle WITHOUT a recorded list walks the
    // candidates of its 3 x 3 cells through those records (OpJacobiU), and a second-ring ghost among them must carry its
    // position, not whatever the buffer held: seeded here, once per step, on the ghosts' own slots
    const uint32_t ng_seed = d.ghosts_ok ? d.n_ghost[0] + d.n_ghost[1] : 0u;
    if (n && ng_seed)
        hipLaunchKernelGGL(k_seed_ghost_records, dim3((ng_seed + 255) / 256), dim3(256), 0, s, d.ghost_dst.as<uint32_t>(), ng_seed,
                           c->pm[c->pcur].as<float4>(), c->pacc.as<float4>(), c->prec0.as<float4>(), c->prec1.as<float4>());
    d.have_flags = true;
    return SPH_OK;
}

void slab_ghost_mrho(sph_ctx* c, const SweepArgs& a)
{
    const uint32_t ng = c->dist.ghosts_ok ? c->dist.n_ghost[0] + c->dist.n_ghost[1] : 0u;
    if (ng) hipLaunchKernelGGL(k_ghost_mrho, dim3((ng + 255) / 256), dim3(256), 0, c->stream, c->dist.ghost_dst.as<uint32_t>(), ng, a.pm, a.rho, a.mrho);
}

