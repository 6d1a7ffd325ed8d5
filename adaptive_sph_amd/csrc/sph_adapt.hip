// Adaptivity data path on the device: the apply half of single_step_adaptivity
// (/root/reference/src/simulation/simulation.rs:2732-2796).  The partner DECISIONS are the host's -- the reference takes them
// in a sequential loop over the particles (adaptivity/particle_sharing.rs:14-117, particle_merging.rs:16-125) -- and arrive as
// the merge_partner / merge_counter arrays of ParticleVec; what those arrays imply for the particle data is computed here, on
// the device-resident state, in the reference's index space:
//
//   share_particles   particle_sharing.rs:152-240   gather form: a receiver reads its donor, a donor only itself
//   merge_particles   particle_merging.rs:270-370   the same transfer + the swap-with-the-last deletion loop, restated as
//                                                    "the k-th hole from the front receives the k-th survivor from the back"
//                                                    (prefix sums over the delete flags in host order)
//   split_particles   splitting.rs:19-82            children appended in the order of the parents' indices (prefix sums over
//                                                    the child counts), positions from the SplitPatterns table
//
// Particles live in cell-sorted order on the device; orig[] maps a slot to the host index, slot_of[] back.  Merging and
// splitting end in a regather into host order (regather_host_order, sph_api.hip), like a sparse edit.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <vector>

#include "sph_context.hpp"

#define ADAPT_CHECK(c)                                                                                                          \
    if (!(c) || !p || !ap) return SPH_ERR_INVALID_ARGUMENT;                                                                     \
    if ((c)->dist.on) return (c)->fail(SPH_ERR_UNSUPPORTED, "the adaptivity data path on a slab context is not covered yet"); \
    if ((c)->poisoned) return (c)->fail(SPH_ERR_POISONED, "an earlier step failed inside the step: the particle state is undefined until sph_upload")

// ---- exclusive prefix sum (u32), three launches: block sums, scan of the block sums, local scans ------------------------
#define SCAN_BLOCK 256
#define SCAN_ITEMS 8
#define SCAN_TILE (SCAN_BLOCK * SCAN_ITEMS)

__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* total)
{
    __shared__ uint32_t s_w[SCAN_BLOCK / 64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t x = v;
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t y = __shfl_up(x, o, 64);
        if (lane >= o) x += y;
    }
    if (lane == 63) s_w[w] = x;
    __syncthreads();
    uint32_t base = 0, tot = 0;
    for (int k = 0; k < SCAN_BLOCK / 64; k++) {
        if (k < w) base += s_w[k];
        tot += s_w[k];
    }
    __syncthreads();
    *total = tot;
    return base + x - v;
}

__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_block_sums(const uint32_t* __restrict__ in, uint32_t n, uint32_t* __restrict__ sums)
{
    const uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    uint32_t v = 0;
    for (int k = 0; k < SCAN_ITEMS; k++)
        if (base + k < n) v += in[base + k];
    uint32_t tot;
    block_exclusive_scan(v, &tot);
    if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}
__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_sums(uint32_t* __restrict__ sums, uint32_t nb, uint32_t* __restrict__ total)
{
    uint32_t carry = 0;
    for (uint32_t b0 = 0; b0 < nb; b0 += SCAN_BLOCK) {   // one block walks the block sums in chunks (nb <= n / 2048)
        const uint32_t k = b0 + threadIdx.x;
        const uint32_t v = k < nb ? sums[k] : 0u;
        uint32_t tot;
        const uint32_t ex = block_exclusive_scan(v, &tot);
        if (k < nb) sums[k] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) *total = carry;
}
__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_apply(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t n, const uint32_t* __restrict__ sums)
{
    const uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    uint32_t loc[SCAN_ITEMS], v = 0;
    for (int k = 0; k < SCAN_ITEMS; k++) {
        loc[k] = base + k < n ? in[base + k] : 0u;
        v += loc[k];
    }
    uint32_t tot;
    uint32_t run = sums[blockIdx.x] + block_exclusive_scan(v, &tot);
    for (int k = 0; k < SCAN_ITEMS; k++) {
        if (base + k < n) out[base + k] = run;
        run += loc[k];
    }
}
void device_exclusive_scan_u32(hipStream_t s, const uint32_t* in, uint32_t* out, uint32_t n, uint32_t* scratch, uint32_t* total)
{
    const uint32_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
    if (nb) hipLaunchKernelGGL(k_scan_block_sums, dim3(nb), dim3(SCAN_BLOCK), 0, s, in, n, scratch);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(SCAN_BLOCK), 0, s, scratch, nb, total);
    if (nb) hipLaunchKernelGGL(k_scan_apply, dim3(nb), dim3(SCAN_BLOCK), 0, s, in, out, n, scratch);
}

// ---- LevelEstimationState::target_mass (simulation.rs:213-237) -- the same IEEE operations as k_classify -----------------
struct TargetP {
    float max_surface_distance, rest_density, radius_fine, radius_base;
    int sizing_function;
};
__device__ __forceinline__ float target_mass(float lv, const TargetP& t)
{
    const float lvl = fmaxf(lv, -t.max_surface_distance);
    const float interp = lvl / -t.max_surface_distance;
    const float mass_fine = (SPH_PI_F * t.radius_fine * t.radius_fine) * t.rest_density;
    const float mass_base = (SPH_PI_F * t.radius_base * t.radius_base) * t.rest_density;
    if (t.sizing_function == SPH_SIZING_MASS) return mass_fine * (1.f - interp) + mass_base * interp;
    if (t.sizing_function == SPH_SIZING_RADIUS) {
        const float r = t.radius_fine * (1.f - interp) + t.radius_base * interp;
        return (SPH_PI_F * r * r) * t.rest_density;
    }
    const float e = 1.f / 2.f;
    const float r = t.radius_fine * (1.f - powf(interp, e)) + t.radius_base * powf(interp, e);
    return (SPH_PI_F * r * r) * t.rest_density;
}
static TargetP target_params(const sph_params* p)
{
    return TargetP{p->maximum_surface_distance, p->rest_density, p->particle_radius_fine, p->particle_radius_base, p->sizing_function};
}

__global__ __launch_bounds__(256) void k_slot_of(uint32_t n, const uint32_t* __restrict__ orig, uint32_t* __restrict__ slot_of)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) slot_of[orig[i]] = i;
}

// ---- pass 1 of share_particles / merge_particles: the receivers (particle_sharing.rs:165-215, particle_merging.rs:277-325).
// A receiver writes only itself and reads only its donor, which no thread writes in this launch (merge_partner[donor] is
// DELETE, so the donor is nobody's receiver).  merging: the donor drops its whole mass (dropped_mass_merging, :372-385).
__global__ __launch_bounds__(256) void k_transfer_receive(uint32_t n, int merging, const uint32_t* __restrict__ orig, const uint32_t* __restrict__ slot_of,
                                                           const uint32_t* __restrict__ partner, const uint16_t* __restrict__ counter, uint32_t min_partners,
                                                           float dt, float max_transfer, TargetP tp, float4* __restrict__ pm, float2* __restrict__ vel,
                                                           const float* __restrict__ lvl, float* __restrict__ h2n, DeviceStatus* status)
{
    const uint32_t s = blockIdx.x * 256 + threadIdx.x;
    if (s >= n) return;
    const uint32_t i = orig[s];
    const uint32_t j = partner[i];
    if (j == SPH_MERGE_PARTNER_AVAILABLE || j == SPH_MERGE_PARTNER_DELETE) return;
    if (j >= n) {
        if (atomicCAS(&status->error, 0u, (uint32_t)SPH_ERR_INVALID_ARGUMENT) == 0u) status->info = i;
        return;
    }
    const uint32_t cj = counter[j];
    if (cj < min_partners) return;
    const uint32_t sj = slot_of[j];
    float4 Pi = pm[s];
    const float4 Pj = pm[sj];
    float2 vi = vel[s];
    const float2 vj = vel[sj];
    float dropped;
    if (merging) dropped = Pj.z;
    else {
        const float target = target_mass(lvl[sj], tp);
        dropped = fminf(Pj.z - target, target * max_transfer * dt);
    }
    const float mass_i = Pi.z;
    const float mass_n = dropped / (float)cj;
    const float mass = mass_i + mass_n;
    vi.x = (mass_i * vi.x + mass_n * vj.x) / mass;
    vi.y = (mass_i * vi.y + mass_n * vj.y) / mass;
    Pi.x = (mass_i * Pi.x + mass_n * Pj.x) / mass;
    Pi.y = (mass_i * Pi.y + mass_n * Pj.y) / mass;
    Pi.z = mass;
    pm[s] = Pi;
    vel[s] = vi;
    h2n[s] = h_from_mass(mass, tp.rest_density);
}

// ---- pass 2: the donors.  sharing (particle_sharing.rs:217-239): mass -= dropped, h2_next from it.  merging
// (particle_merging.rs:339-355): mass -= dropped (h2_next untouched) and the particle is deleted if what is left is < 1e-6.
__global__ __launch_bounds__(256) void k_transfer_donate(uint32_t n, int merging, const uint32_t* __restrict__ orig, const uint32_t* __restrict__ partner,
                                                          const uint16_t* __restrict__ counter, uint32_t min_partners, float dt, float max_transfer,
                                                          TargetP tp, float4* __restrict__ pm, const float* __restrict__ lvl, float* __restrict__ h2n,
                                                          uint32_t* __restrict__ del_host)
{
    const uint32_t s = blockIdx.x * 256 + threadIdx.x;
    if (s >= n) return;
    const uint32_t i = orig[s];
    uint32_t del = 0;
    if (partner[i] == SPH_MERGE_PARTNER_DELETE && counter[i] >= min_partners) {
        float4 P = pm[s];
        if (merging) {
            P.z -= P.z;   // dropped_mass_merging returns the whole mass
            del = P.z < 0.000001f ? 1u : 0u;
        } else {
            const float target = target_mass(lvl[s], tp);
            const float dropped = fminf(P.z - target, target * max_transfer * dt);
            P.z -= dropped;
            h2n[s] = h_from_mass(P.z, tp.rest_density);
        }
        pm[s] = P;
    }
    if (del_host) del_host[i] = del;
}

// ---- deletion order of merge_particles' swap loop (particle_merging.rs:337-365) in closed form.  n' = n - #deleted.
// Holes = deleted indices < n' (ascending); tail survivors = kept indices >= n' (descending); hole k <- tail survivor k.
__global__ __launch_bounds__(256) void k_merge_holes(uint32_t n, uint32_t n_new, const uint32_t* __restrict__ del, const uint32_t* __restrict__ del_before,
                                                      uint32_t* __restrict__ hole_by_rank, EditSrc* __restrict__ src)
{
    const uint32_t f = blockIdx.x * 256 + threadIdx.x;
    if (f >= n_new) return;
    src[f] = EditSrc{f, 0xffffffffu};           // kept particles below n' stay where they are
    if (del[f]) hole_by_rank[del_before[f]] = f;
}
__global__ __launch_bounds__(256) void k_merge_fill(uint32_t n, uint32_t n_new, uint32_t n_del, const uint32_t* __restrict__ del, const uint32_t* __restrict__ del_before,
                                                     const uint32_t* __restrict__ hole_by_rank, EditSrc* __restrict__ src)
{
    const uint32_t t = n_new + blockIdx.x * 256 + threadIdx.x;
    if (t >= n || del[t]) return;
    // survivors with a larger index: (n - 1 - t) positions behind t, minus the deleted ones among them
    const uint32_t del_after = n_del - del_before[t] - del[t];
    const uint32_t rank = (n - 1u - t) - del_after;
    src[hole_by_rank[rank]] = EditSrc{t, 0xffffffffu};
}

// ---- split_particles (splitting.rs:19-82) ---------------------------------------------------------------------------
// child count of every host index (0 for a particle that does not split): num_children - 1 goes into the prefix sum
__global__ __launch_bounds__(256) void k_split_count(uint32_t n, const uint32_t* __restrict__ orig, const float4* __restrict__ pm, const float* __restrict__ lvl,
                                                      const uint8_t* __restrict__ szc, TargetP tp, uint32_t max_children, int fail_on_missing,
                                                      uint32_t* __restrict__ extra_host, DeviceStatus* status)
{
    const uint32_t s = blockIdx.x * 256 + threadIdx.x;
    if (s >= n) return;
    const uint32_t i = orig[s];
    uint32_t extra = 0;
    if (szc[s] == 4 /* ParticleSizeClass::TooLarge */) {
        const float lv = lvl[s];
        const float target = target_mass(lv, tp);
        const float r = roundf(pm[s].z / target);            // FT::round: half away from zero
        // `as usize`: saturating, NaN -> 0
        uint32_t nc = !(r > 0.f) ? 0u : (r >= 4294967040.f ? 0xffffffffu : (uint32_t)r);
        bool bad = isnan(lv);                                  // LevelEstimationState::level() of FluidInterior: unreachable!()
        if (nc > max_children) {
            if (fail_on_missing) bad = true;                   // panic!("no split pattern for a 1-to-{} split")
            nc = max_children;
        }
        if (!(nc > 1u)) bad = true;                            // assert!(num_children > 1)
        if (bad) {
            if (atomicCAS(&status->error, 0u, (uint32_t)SPH_ERR_NO_SPLIT_PATTERN) == 0u) status->info = i;
        } else extra = nc - 1u;
    }
    extra_host[i] = extra;
}

// final index f -> what it holds: an old particle (the parent slot takes child 0's values) or an appended child
__global__ __launch_bounds__(256) void k_split_plan(uint32_t n, uint32_t n_new, const uint32_t* __restrict__ slot_of, const float4* __restrict__ pm,
                                                     const float2* __restrict__ vel, const float* __restrict__ lvl, const uint32_t* __restrict__ extra,
                                                     const uint32_t* __restrict__ base, const float2* __restrict__ patterns, float rest_density,
                                                     EditSrc* __restrict__ src, EditSet* __restrict__ sets)
{
    const uint32_t f = blockIdx.x * 256 + threadIdx.x;
    if (f >= n_new) return;
    uint32_t parent, child_id;
    if (f < n) {
        parent = f;
        child_id = 0;
        if (extra[f] == 0u) {
            src[f] = EditSrc{f, 0xffffffffu};
            return;
        }
    } else {
        // the parent p with base[p] <= c < base[p] + extra[p]: the last index whose exclusive prefix sum is <= c
        const uint32_t c = f - n;
        uint32_t lo = 0, hi = n - 1u;
        while (lo < hi) {
            const uint32_t mid = (lo + hi + 1u) >> 1;
            if (base[mid] <= c) lo = mid;
            else hi = mid - 1u;
        }
        parent = lo;
        child_id = c - base[parent] + 1u;
    }
    const uint32_t sp = slot_of[parent];
    const float4 P = pm[sp];
    const uint32_t nc = extra[parent] + 1u;
    const float2 off = patterns[(nc - 1u) * nc / 2u - 1u + child_id];         // patterns of 2, 3, ... children, concatenated
    const float scale = sqrtf((P.z / 1.f /* INIT_REST_DENSITY */) * SPH_FRAC_1_PI_F);   // DU::sphere_volume_to_radius (sph_kernels.rs:203-206)
    const float child_mass = P.z / (float)nc;
    EditSet q{};
    q.mass = child_mass;
    q.px = P.x + off.x * scale;
    q.py = P.y + off.y * scale;
    q.h2 = q.h2_next = h_from_mass(child_mass, rest_density);
    if (child_id == 0u) {
        // the parent slot: mass, position, h2, h2_next (velocity, level_estimation, level_old are rewritten with their own values)
        q.fields = SPH_EDIT_F_MASS | SPH_EDIT_F_POSITION | SPH_EDIT_F_H2 | SPH_EDIT_F_H2_NEXT;
        src[f] = EditSrc{parent, f};
    } else {
        // an appended child: ParticleVec defaults + mass, velocity, position, h2_next, level_estimation.  h2 and level_old are
        // written to the PARENT's slot by the reference (splitting.rs:73, 76) and stay 0 here.
        const float2 v = vel[sp];
        q.vx = v.x;
        q.vy = v.y;
        q.lvl = lvl[sp];
        q.fields = SPH_EDIT_F_MASS | SPH_EDIT_F_POSITION | SPH_EDIT_F_VELOCITY | SPH_EDIT_F_H2_NEXT | SPH_EDIT_F_LEVEL_ESTIMATION;
        src[f] = EditSrc{0xfffffffeu /* a default particle */, f};
    }
    sets[f] = q;
}

// ---- host side -----------------------------------------------------------------------------------------------------------
static int check_status(sph_ctx* c, const char* what)
{
    DeviceStatus st;
    HIPCHK(c, hipMemcpyAsync(&st, c->status.p, sizeof st, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (st.error) {
        (void)hipMemsetAsync(c->status.p, 0, sizeof(DeviceStatus), c->stream);
        return c->fail((int)st.error, "%s (particle i=%u)", what, st.info);
    }
    return SPH_OK;
}

static int transfer(sph_ctx* c, const sph_params* p, const sph_adapt_params* ap, const uint32_t* partner, const uint16_t* counter, int merging)
{
    HIPCHK(c, hipSetDevice(c->device));
    const uint32_t n = (uint32_t)c->n;
    if (n == 0) return SPH_OK;
    if (!partner || !counter) return SPH_ERR_INVALID_ARGUMENT;
    // (before anything is launched: a bad index must not leave some receivers and donors already modified)
    for (uint32_t i = 0; i < n; i++)
        if (partner[i] >= n && partner[i] != SPH_MERGE_PARTNER_AVAILABLE && partner[i] != SPH_MERGE_PARTNER_DELETE)
            return c->fail(SPH_ERR_INVALID_ARGUMENT, "merge_partner holds an index outside the particle vector (particle i=%u: %u)", i, partner[i]);
    hipStream_t s = c->stream;
    const int k = c->cur;
    TmpBuf d_partner, d_counter, d_slot, d_del, d_before, d_scratch, d_holes, d_src, d_sets;
    auto release = [&] {
        for (TmpBuf* b : {&d_partner, &d_counter, &d_slot, &d_del, &d_before, &d_scratch, &d_holes, &d_src, &d_sets}) b->release();
    };
    auto guard = [&](hipError_t e) { return e == hipSuccess; };
    if (!guard(d_partner.ensure((size_t)n * 4)) || !guard(d_counter.ensure((size_t)n * 2)) || !guard(d_slot.ensure((size_t)n * 4))) {
        release();
        return c->fail(SPH_ERR_DEVICE, "out of device memory");
    }
    HIPCHK(c, hipMemcpyAsync(d_partner.p, partner, (size_t)n * 4, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(d_counter.p, counter, (size_t)n * 2, hipMemcpyHostToDevice, s));
    const dim3 grid((n + 255) / 256), blk(256);
    const TargetP tp = target_params(p);
    const uint32_t min_partners = merging ? ap->minimum_merge_partners : ap->minimum_share_partners;
    if (merging) {
        if (!guard(d_del.ensure((size_t)n * 4)) || !guard(d_before.ensure((size_t)n * 4 + 4)) || !guard(d_scratch.ensure(((size_t)n / SCAN_TILE + 4) * 4)) ||
            !guard(d_holes.ensure((size_t)n * 4))) {
            release();
            return c->fail(SPH_ERR_DEVICE, "out of device memory");
        }
    }
    hipLaunchKernelGGL(k_slot_of, grid, blk, 0, s, n, c->orig[k].as<uint32_t>(), d_slot.as<uint32_t>());
    hipLaunchKernelGGL(k_transfer_receive, grid, blk, 0, s, n, merging, c->orig[k].as<uint32_t>(), d_slot.as<uint32_t>(), d_partner.as<uint32_t>(),
                       d_counter.as<uint16_t>(), min_partners, ap->dt, ap->max_mass_transfer_sharing, tp, c->pm[c->pcur].as<float4>(), c->vel[k].as<float2>(),
                       c->lvl[k].as<float>(), c->h2n[k].as<float>(), c->status.as<DeviceStatus>());
    hipLaunchKernelGGL(k_transfer_donate, grid, blk, 0, s, n, merging, c->orig[k].as<uint32_t>(), d_partner.as<uint32_t>(), d_counter.as<uint16_t>(), min_partners,
                       ap->dt, ap->max_mass_transfer_sharing, tp, c->pm[c->pcur].as<float4>(), c->lvl[k].as<float>(), c->h2n[k].as<float>(),
                       merging ? d_del.as<uint32_t>() : nullptr);
    int rc = check_status(c, "merge_partner holds an index outside the particle vector");
    // positions and masses changed: what the last step left behind no longer describes the state
    c->hdr_ahead = false;
    c->grid_valid = false;
    c->lists_after = false;
    // (sharing neither reorders nor resizes the vector: stash and the step's flags keep describing the particles in their slots,
    //  as the reference's ParticleVec keeps them through share_particles -- snapshots taken after single_step show them)
    if (rc || !merging) {
        release();
        return rc;
    }
    // ---- delete (particle_merging.rs:337-369)
    uint32_t* d_total = d_before.as<uint32_t>() + n;
    device_exclusive_scan_u32(s, d_del.as<uint32_t>(), d_before.as<uint32_t>(), n, d_scratch.as<uint32_t>(), d_total);
    uint32_t n_del = 0;
    HIPCHK(c, hipMemcpyAsync(&n_del, d_total, 4, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    if (n_del == 0) {
        release();
        return SPH_OK;
    }
    c->have_level = false;    // the deletion reorders the vector: per-step outputs that do not travel (stash, flags) are gone
    c->have_reduced = false;
    // (the reference's loop never removes the particle it ends on when everything is deleted: last_particle_id is a usize that
    //  would underflow -- it panics there; an empty vector is what truncate(0) would leave)
    const uint32_t n_new = n - n_del;
    if (!guard(d_src.ensure(((size_t)n_new + 1) * sizeof(EditSrc))) || !guard(d_sets.ensure(sizeof(EditSet)))) {
        release();
        return c->fail(SPH_ERR_DEVICE, "out of device memory");
    }
    if (n_new) {
        hipLaunchKernelGGL(k_merge_holes, dim3((n_new + 255) / 256), blk, 0, s, n, n_new, d_del.as<uint32_t>(), d_before.as<uint32_t>(), d_holes.as<uint32_t>(),
                           d_src.as<EditSrc>());
        hipLaunchKernelGGL(k_merge_fill, dim3((n_del + 255) / 256), blk, 0, s, n, n_new, n_del, d_del.as<uint32_t>(), d_before.as<uint32_t>(),
                           d_holes.as<uint32_t>(), d_src.as<EditSrc>());
    }
    rc = regather_host_order(c, n_new, d_src.as<EditSrc>(), d_sets.as<EditSet>());
    release();
    return rc;
}

extern "C" int sph_share_particles(sph_ctx* c, const sph_params* p, const sph_adapt_params* ap, const uint32_t* partner, const uint16_t* counter)
{
    ADAPT_CHECK(c);
    return transfer(c, p, ap, partner, counter, 0);
}

extern "C" int sph_merge_particles(sph_ctx* c, const sph_params* p, const sph_adapt_params* ap, const uint32_t* partner, const uint16_t* counter)
{
    ADAPT_CHECK(c);
    return transfer(c, p, ap, partner, counter, 1);
}

extern "C" int sph_set_split_patterns(sph_ctx* c, uint32_t n_patterns, const float* pos_s_xy)
{
    if (!c || (n_patterns && !pos_s_xy)) return SPH_ERR_INVALID_ARGUMENT;
    HIPCHK(c, hipSetDevice(c->device));
    // SplitPatterns::new (splitting.rs:99-105): pattern k has k + 2 children
    size_t n_pos = 0;
    for (uint32_t k = 0; k < n_patterns; k++) n_pos += (size_t)k + 2;
    HIPCHK(c, c->split_patterns.ensure((n_pos ? n_pos : 1) * sizeof(float2)));
    if (n_pos) HIPCHK(c, hipMemcpy(c->split_patterns.p, pos_s_xy, n_pos * sizeof(float2), hipMemcpyHostToDevice));
    c->n_split_patterns = n_patterns;
    return SPH_OK;
}

extern "C" int sph_split_particles(sph_ctx* c, const sph_params* p, const sph_adapt_params* ap)
{
    ADAPT_CHECK(c);
    HIPCHK(c, hipSetDevice(c->device));
    const uint32_t n = (uint32_t)c->n;
    if (n == 0) return SPH_OK;
    hipStream_t s = c->stream;
    const int k = c->cur;
    const uint32_t max_children = c->n_split_patterns + 1u;   // SplitPatterns::get_max_num_children (splitting.rs:115-117)
    TmpBuf d_extra, d_base, d_scratch, d_slot, d_src, d_sets;
    auto release = [&] {
        for (TmpBuf* b : {&d_extra, &d_base, &d_scratch, &d_slot, &d_src, &d_sets}) b->release();
    };
    if (d_extra.ensure((size_t)n * 4) != hipSuccess || d_base.ensure((size_t)n * 4 + 4) != hipSuccess ||
        d_scratch.ensure(((size_t)n / SCAN_TILE + 4) * 4) != hipSuccess || d_slot.ensure((size_t)n * 4) != hipSuccess) {
        release();
        return c->fail(SPH_ERR_DEVICE, "out of device memory");
    }
    const dim3 grid((n + 255) / 256), blk(256);
    const TargetP tp = target_params(p);
    hipLaunchKernelGGL(k_slot_of, grid, blk, 0, s, n, c->orig[k].as<uint32_t>(), d_slot.as<uint32_t>());
    hipLaunchKernelGGL(k_split_count, grid, blk, 0, s, n, c->orig[k].as<uint32_t>(), c->pm[c->pcur].as<float4>(), c->lvl[k].as<float>(), c->szc[k].as<uint8_t>(), tp,
                       max_children, ap->fail_on_missing_split_pattern, d_extra.as<uint32_t>(), c->status.as<DeviceStatus>());
    uint32_t* d_total = d_base.as<uint32_t>() + n;
    device_exclusive_scan_u32(s, d_extra.as<uint32_t>(), d_base.as<uint32_t>(), n, d_scratch.as<uint32_t>(), d_total);
    uint32_t n_extra = 0;
    HIPCHK(c, hipMemcpyAsync(&n_extra, d_total, 4, hipMemcpyDeviceToHost, s));
    int rc = check_status(c, max_children < 2 ? "no split pattern for a 1-to-2 split (sph_set_split_patterns was not called)"
                                              : "no split pattern for this split, or num_children <= 1");
    if (rc || n_extra == 0) {
        release();
        return rc;
    }
    if ((uint64_t)n + n_extra > c->cap) {
        release();
        return c->fail(SPH_ERR_CAPACITY, "splitting needs %llu particles, capacity %llu", (unsigned long long)n + n_extra, (unsigned long long)c->cap);
    }
    const uint32_t n_new = n + n_extra;
    if (d_src.ensure((size_t)n_new * sizeof(EditSrc)) != hipSuccess || d_sets.ensure((size_t)n_new * sizeof(EditSet)) != hipSuccess) {
        release();
        return c->fail(SPH_ERR_DEVICE, "out of device memory");
    }
    hipLaunchKernelGGL(k_split_plan, dim3((n_new + 255) / 256), blk, 0, s, n, n_new, d_slot.as<uint32_t>(), c->pm[c->pcur].as<float4>(), c->vel[k].as<float2>(),
                       c->lvl[k].as<float>(), d_extra.as<uint32_t>(), d_base.as<uint32_t>(), c->split_patterns.as<float2>(), p->rest_density,
                       d_src.as<EditSrc>(), d_sets.as<EditSet>());
    rc = regather_host_order(c, n_new, d_src.as<EditSrc>(), d_sets.as<EditSet>());
    release();
    return rc;
}

// ---- the partner searches as host code (sph_ffi.h: sph_host_find_partners) ------------------------------------------------
static float host_target_mass(float lv, const sph_params* p)
{
    const float lvl = fmaxf(lv, -p->maximum_surface_distance);
    const float interp = lvl / -p->maximum_surface_distance;
    const float mass_fine = (SPH_PI_F * p->particle_radius_fine * p->particle_radius_fine) * p->rest_density;
    const float mass_base = (SPH_PI_F * p->particle_radius_base * p->particle_radius_base) * p->rest_density;
    if (p->sizing_function == SPH_SIZING_MASS) return mass_fine * (1.f - interp) + mass_base * interp;
    if (p->sizing_function == SPH_SIZING_RADIUS) {
        const float r = p->particle_radius_fine * (1.f - interp) + p->particle_radius_base * interp;
        return (SPH_PI_F * r * r) * p->rest_density;
    }
    const float e = 1.f / 2.f;
    const float r = p->particle_radius_fine * (1.f - powf(interp, e)) + p->particle_radius_base * powf(interp, e);
    return (SPH_PI_F * r * r) * p->rest_density;
}

extern "C" int sph_host_find_partners(int kind, uint64_t n, const uint8_t* cls, const float* mass, const float* level, const float* pos, const float* h2,
                                      const uint32_t* off, const uint32_t* idx, const sph_params* p, const sph_adapt_params* ap, uint32_t* partner,
                                      uint16_t* counter, uint64_t* n_transfers)
{
    if ((kind != 0 && kind != 1) || !p || !ap || !n_transfers || (n && (!cls || !mass || !level || !pos || !h2 || !off || !idx || !partner || !counter)))
        return SPH_ERR_INVALID_ARGUMENT;
    const bool share = kind == 0;
    const uint8_t donor_class = share ? 3 /* Large */ : 0 /* TooSmall */;
    const float mass_base = (SPH_PI_F * p->particle_radius_base * p->particle_radius_base) * p->rest_density;   // SimulationParams::mass_base
    const float max_dist_factor = share ? ap->max_share_distance : ap->max_merge_distance;
    for (uint64_t i = 0; i < n; i++) partner[i] = SPH_MERGE_PARTNER_AVAILABLE;
    uint64_t total = 0;
    for (uint64_t i = 0; i < n; i++) {
        counter[i] = 0;
        if (cls[i] != donor_class) continue;
        float dropped;
        if (share) {
            const float target = host_target_mass(level[i], p);     // dropped_mass_sharing (particle_sharing.rs:242-253)
            dropped = fminf(mass[i] - target, target * ap->max_mass_transfer_sharing * ap->dt);
        } else dropped = mass[i];                                   // dropped_mass_merging (particle_merging.rs:372-385)
        for (uint32_t q = off[i]; q < off[i + 1]; q++) {
            const uint64_t j = idx[q];
            if (j == i) continue;
            if (j >= n) return SPH_ERR_INVALID_ARGUMENT;
            bool can;
            if (share) can = cls[j] == 1 || (cls[j] == 0 && ap->allow_share_with_too_small_particle) || (cls[j] == 2 && ap->allow_share_with_optimal_particle);
            else {
                can = cls[j] == 1 || cls[j] == 0 || (cls[j] == 2 && ap->allow_merge_with_optimal_particle);
                if (ap->allow_merge_on_size_difference && mass[j] > 5.f * mass[i]) can = true;
            }
            if (!can) continue;
            // XXX: VERY IMPORTANT long distance shares lead to popping/unstable behavior
            const float dx = pos[2 * i] - pos[2 * j], dy = pos[2 * i + 1] - pos[2 * j + 1];
            const float max_dist = ((h2[i] + h2[j]) * 0.5f) * max_dist_factor;
            if (dx * dx + dy * dy > max_dist * max_dist) continue;
            const float new_mass_j = mass[j] + dropped / (float)(counter[i] + 1);
            const float target_j = host_target_mass(level[j], p);
            if (new_mass_j >= target_j * 1.1f /* PARTICLE_SIZE_FACTOR_LARGE */) continue;
            if (new_mass_j > mass_base) continue;
            if (partner[j] != SPH_MERGE_PARTNER_AVAILABLE) continue;   // the neighbour is somebody's partner already
            if (counter[i] == 0) {
                if (partner[i] != SPH_MERGE_PARTNER_AVAILABLE) continue;   // this particle is itself somebody's partner
                partner[i] = SPH_MERGE_PARTNER_DELETE;
            }
            partner[j] = (uint32_t)i;
            counter[i] += 1;
            total += 1;
            if (!(counter[i] < 1000)) return SPH_ERR_INVALID_ARGUMENT;   // assert!(merge_counter[i] < 1000)
        }
    }
    // validate_share_partners / validate_merge_partners (particle_sharing.rs:119-150, particle_merging.rs:226-268)
    for (uint64_t i = 0; i < n; i++) {
        if (counter[i] > 0) {
            if (cls[i] != donor_class || partner[i] != SPH_MERGE_PARTNER_DELETE) return SPH_ERR_INVALID_ARGUMENT;
            uint32_t c2 = 0;
            for (uint32_t q = off[i]; q < off[i + 1]; q++) c2 += partner[idx[q]] == (uint32_t)i ? 1u : 0u;
            if (c2 != counter[i]) return SPH_ERR_INVALID_ARGUMENT;
        } else {
            if (partner[i] == SPH_MERGE_PARTNER_DELETE) return SPH_ERR_INVALID_ARGUMENT;
            if (partner[i] != SPH_MERGE_PARTNER_AVAILABLE && partner[partner[i]] != SPH_MERGE_PARTNER_DELETE) return SPH_ERR_INVALID_ARGUMENT;
        }
    }
    *n_transfers = total;
    return SPH_OK;
}
