// libsph_hip.so: the C ABI of include/sph_ffi.h over the HIP kernels of this directory.
//
// One context = one FluidSimulation's device-resident state on ONE MI355X (one process per GPU).
// The persistent particle SoA stays on the device in cell-sorted order between steps; `orig`
// maps each sorted slot back to the host's particle index, so uploads/downloads speak the
// reference's indices.
//
// This file: context lifetime, upload / download (host particle order), boundary description, the step header kernels and
// the measurement hooks.  The step itself (single_step_without_adaptivity, simulation.rs:1980-2730) is sequenced in
// sph_step.hip.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "sph_context.hpp"
#include "sph_lambda.hpp"

// ------------------------------------------------------------------------------------------------
// profiler
// ------------------------------------------------------------------------------------------------
int Profiler::find(const char* name)
{
    for (size_t i = 0; i < recs.size(); i++)
        if (recs[i].name == name) return (int)i;
    recs.push_back(Rec{name, 0, 0, {}});
    return (int)recs.size() - 1;
}
bool Profiler::wants(const char* name) const
{
    // mode 2: the density kernel only, on every 8th step (a timing-enabled event record forces a command
    // flush; sampling keeps the perturbation of the timed region below 1 %)
    return mode == 1 || mode == 3 || mode == 4 || (mode == 2 && (step_index & 7u) == 0u && strncmp(name, "density", 7) == 0);
}
hipEvent_t Profiler::get_event()
{
    if (!pool.empty()) {
        hipEvent_t e = pool.back();
        pool.pop_back();
        return e;
    }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}
// Scopes nest in slab mode (the partition / halo selection run the radix sort and the reorder, which open their own): only the
// outermost scope is timed, the inner launches are part of it.
void Profiler::begin(const char* name, hipStream_t s, bool single_launch)
{
    if (depth++ > 0) return;
    cur = find(name);
    ext_open = mode == 3 && single_launch && ts_next < (uint32_t)TS_RING;
    ext_slot = -1;
    if (ext_open) return;   // timed by the kernel itself (take_slot), or not at all
    kev_open = mode == 4 && single_launch;
    kev_taken = false;
    cur_a = get_event();
    if (!kev_open) (void)hipEventRecord(cur_a, s);   // (mode 4, one launch: the dispatch carries the pair, take_events)
}
bool Profiler::take_events(hipEvent_t* a, hipEvent_t* b)
{
    if (!kev_open || kev_taken || depth != 1) return false;
    kev_taken = true;
    cur_b = get_event();
    *a = cur_a;
    *b = cur_b;
    return true;
}
unsigned long long* Profiler::take_slot()
{
    if (!ext_open || ext_slot >= 0 || depth != 1) return nullptr;
    if (!ts_dev) {
        if (hipMalloc((void**)&ts_dev, 2 * (size_t)TS_RING * sizeof(unsigned long long)) != hipSuccess) return nullptr;
        (void)hipMemset(ts_dev, 0xff, (size_t)TS_RING * sizeof(unsigned long long));
        (void)hipMemset(ts_dev + TS_RING, 0, (size_t)TS_RING * sizeof(unsigned long long));
    }
    ext_slot = (int)ts_next++;
    return ts_dev + ext_slot;
}
void Profiler::end(hipStream_t s)
{
    if (--depth > 0) return;
    if (ext_open) {
        if (ext_slot >= 0) pending.push_back(Pending{cur, nullptr, nullptr, ext_slot});   // (else: the scope launched nothing, no sample)
        ext_open = false;
        cur = -1;
        return;
    }
    if (kev_open) {
        kev_open = false;
        if (kev_taken) pending.push_back(Pending{cur, cur_a, cur_b, -1});
        else pool.push_back(cur_a);   // (the scope launched nothing)
        cur = -1;
        return;
    }
    hipEvent_t b = get_event();
    (void)hipEventRecord(b, s);
    pending.push_back(Pending{cur, cur_a, b, -1});
    cur = -1;
}
void Profiler::collect()
{
    // (called behind a wait for the stream: every stamped launch has finished)
    std::vector<unsigned long long> ts;
    if (ts_next) {
        ts.resize(2 * (size_t)ts_next);
        (void)hipMemcpy(ts.data(), ts_dev, (size_t)ts_next * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        (void)hipMemcpy(ts.data() + ts_next, ts_dev + TS_RING, (size_t)ts_next * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        (void)hipMemset(ts_dev, 0xff, (size_t)ts_next * sizeof(unsigned long long));
        (void)hipMemset(ts_dev + TS_RING, 0, (size_t)ts_next * sizeof(unsigned long long));
    }
    static FILE* dump = getenv("SPH_TS_DUMP") ? fopen(getenv("SPH_TS_DUMP"), "w") : nullptr;   // (diagnostic: raw stamps, one line per launch)
    std::vector<Pending> later;
    for (auto& p : pending) {
        if (p.slot >= 0) {
            const unsigned long long t0 = ts[(size_t)p.slot], t1 = ts[(size_t)ts_next + (size_t)p.slot];
            if (dump) fprintf(dump, "%s %llu %llu\n", recs[p.rec].name.c_str(), t0, t1);
            if (t1 >= t0 && t0 != ~0ull) {
                const float ms = (float)((double)(t1 - t0) * 1e-5);   // 100 MHz ticks
                recs[p.rec].launches++;
                recs[p.rec].total_ms += ms;
                recs[p.rec].samples.push_back(ms);
            }
            continue;
        }
        float ms = 0.f;
        const hipError_t e = hipEventElapsedTime(&ms, p.a, p.b);
        if (e == hipErrorNotReady) {   // (a launch queued behind the step's last wait -- the build queued ahead: the next collection takes it)
            later.push_back(p);
            continue;
        }
        if (e == hipSuccess) {
            recs[p.rec].launches++;
            recs[p.rec].total_ms += ms;
            recs[p.rec].samples.push_back(ms);
        }
        pool.push_back(p.a);
        pool.push_back(p.b);
    }
    pending.swap(later);
    ts_next = 0;
    if (dump) fflush(dump);
}
void Profiler::reset()
{
    collect();
    recs.clear();
}
Profiler::~Profiler()
{
    for (auto& p : pending) {
        if (p.a) (void)hipEventDestroy(p.a);
        if (p.b) (void)hipEventDestroy(p.b);
    }
    for (auto e : pool) (void)hipEventDestroy(e);
    if (ts_dev) (void)hipFree(ts_dev);
}

// ------------------------------------------------------------------------------------------------
// small kernels owned by this TU
// ------------------------------------------------------------------------------------------------
#define HDR_BLOCKS 256

// h_next_from_mass (simulation.rs:1865-1871) + the per-step scalars: particle bounding box (CellGrid,
// neighborhood_search.rs:261-275), h_max/h_min, CFL term min_i (2h_i)^2 / (|v_i|^2 + 0.01)
// (simulation.rs:2182-2189)
__global__ __launch_bounds__(256) void k_header(float4* __restrict__ pm, const float2* __restrict__ vel, uint32_t n, float rest_density,
                                                 int from_mass, float* __restrict__ h2_next, HeaderOut* __restrict__ partials,
                                                 const uint8_t* __restrict__ owned)
{
    const float INF = __uint_as_float(0x7f800000u);
    float mnx = INF, mny = INF, mxx = -INF, mxy = -INF, hmx = 0.f, hmn = INF, cfl = INF;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        if (owned && !owned[i]) continue;   // slab decomposition: last step's ghosts are still interleaved
        float4 p = pm[i];
        if (from_mass == 1) {
            p.w = h_from_mass(p.z, rest_density);
            pm[i] = p;
        } else if (from_mass == 2) {
            // FromDistribution*: "only apply the support length that was estimated in the last step": mem::swap(h2, h2_next)
            // (simulation.rs:2004-2014)
            const float t = p.w;
            p.w = h2_next[i];
            h2_next[i] = t;
            pm[i] = p;
        }
        float2 v = vel[i];
        mnx = fminf(mnx, p.x); mxx = fmaxf(mxx, p.x);
        mny = fminf(mny, p.y); mxy = fmaxf(mxy, p.y);
        hmx = fmaxf(hmx, p.w); hmn = fminf(hmn, p.w);
        float sr = p.w * 2.f;
        float c = sr * sr / ((v.x * v.x + v.y * v.y) + 0.01f);
        cfl = c < cfl ? c : cfl;  // partial_cmp-min: NaN never wins, like min_by(partial_cmp) on finite data
    }
    mnx = wave_min(mnx); mny = wave_min(mny); mxx = wave_max(mxx); mxy = wave_max(mxy);
    hmx = wave_max(hmx); hmn = wave_min(hmn); cfl = wave_min(cfl);
    __shared__ HeaderOut s[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) s[w] = HeaderOut{mnx, mny, mxx, mxy, hmx, hmn, cfl, 0};
    __syncthreads();
    if (threadIdx.x == 0) {
        HeaderOut o = s[0];
        for (int k = 1; k < 4; k++) {
            o.min_x = fminf(o.min_x, s[k].min_x); o.min_y = fminf(o.min_y, s[k].min_y);
            o.max_x = fmaxf(o.max_x, s[k].max_x); o.max_y = fmaxf(o.max_y, s[k].max_y);
            o.h_max = fmaxf(o.h_max, s[k].h_max); o.h_min = fminf(o.h_min, s[k].h_min);
            o.min_cfl = fminf(o.min_cfl, s[k].min_cfl);
        }
        partials[blockIdx.x] = o;
    }
}

__global__ __launch_bounds__(256) void k_header_final(const HeaderOut* __restrict__ partials, int nparts, HeaderOut* __restrict__ out, uint32_t seq)
{
    const float INF = __uint_as_float(0x7f800000u);
    HeaderOut o{INF, INF, -INF, -INF, 0.f, INF, INF, 0};
    if ((int)threadIdx.x < nparts) o = partials[threadIdx.x];
    o.min_x = wave_min(o.min_x); o.min_y = wave_min(o.min_y); o.max_x = wave_max(o.max_x); o.max_y = wave_max(o.max_y);
    o.h_max = wave_max(o.h_max); o.h_min = wave_min(o.h_min); o.min_cfl = wave_min(o.min_cfl);
    __shared__ HeaderOut s[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) s[w] = o;
    __syncthreads();
    if (threadIdx.x == 0) {
        HeaderOut r = s[0];
        for (int k = 1; k < 4; k++) {
            r.min_x = fminf(r.min_x, s[k].min_x); r.min_y = fminf(r.min_y, s[k].min_y);
            r.max_x = fmaxf(r.max_x, s[k].max_x); r.max_y = fmaxf(r.max_y, s[k].max_y);
            r.h_max = fmaxf(r.h_max, s[k].h_max); r.h_min = fminf(r.h_min, s[k].h_min);
            r.min_cfl = fminf(r.min_cfl, s[k].min_cfl);
        }
        r.pad = seq - 1u;
        *out = r;
        if (seq) {   // the host may be spinning on the pad word of the mapped copy (wait hint)
            __threadfence_system();
            ((volatile HeaderOut*)out)->pad = seq;
        }
    }
}

// the same reduction over the partials the integrating final sweep wrote (one per 256-particle block); it runs behind that
// sweep and, like it, only once the stop decision has been taken
// `h_ctrl` != nullptr: the launch is the last one before a host wait and does k_publish's job as well (control block + guard
// word to mapped host memory, the sequence number last) -- one launch less on the step's critical path
// (1024 threads, four partials per thread requested together: the launch sits on the step's critical path -- the host waits for its
//  last store -- and 16 dependent load round trips per thread were most of its 8 us)
__global__ __launch_bounds__(1024) void k_header_ahead(const HeaderOut* __restrict__ partials, uint32_t nparts, const SolverCtrl* __restrict__ ctrl,
                                                       HeaderOut* __restrict__ out, const DeviceStatus* __restrict__ status, SolverCtrl* __restrict__ h_ctrl,
                                                       DeviceStatus* __restrict__ h_status, uint32_t seq)
{
    if (ctrl->done == 0u) {
        if (h_ctrl && threadIdx.x == 0) {
            SolverCtrl v = *ctrl;
            v.seq = seq - 1u;
            *h_ctrl = v;
            *h_status = *status;
            __threadfence_system();
            ((volatile SolverCtrl*)h_ctrl)->seq = seq;
        }
        return;
    }
    const float INF = __uint_as_float(0x7f800000u);
    HeaderOut o{INF, INF, -INF, -INF, 0.f, INF, INF, 0};
    for (uint32_t k0 = threadIdx.x; k0 < nparts; k0 += 4096u) {
        HeaderOut q[4];
#pragma unroll
        for (uint32_t u = 0; u < 4u; u++) q[u] = partials[min(k0 + 1024u * u, nparts - 1u)];   // (a repeated partial changes no min / max)
#pragma unroll
        for (uint32_t u = 0; u < 4u; u++) {
            o.min_x = fminf(o.min_x, q[u].min_x); o.min_y = fminf(o.min_y, q[u].min_y);
            o.max_x = fmaxf(o.max_x, q[u].max_x); o.max_y = fmaxf(o.max_y, q[u].max_y);
            o.h_max = fmaxf(o.h_max, q[u].h_max); o.h_min = fminf(o.h_min, q[u].h_min);
            o.min_cfl = fminf(o.min_cfl, q[u].min_cfl);
        }
    }
    o.min_x = wave_min(o.min_x); o.min_y = wave_min(o.min_y); o.max_x = wave_max(o.max_x); o.max_y = wave_max(o.max_y);
    o.h_max = wave_max(o.h_max); o.h_min = wave_min(o.h_min); o.min_cfl = wave_min(o.min_cfl);
    __shared__ HeaderOut s[16];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) s[w] = o;
    __syncthreads();
    if (threadIdx.x == 0) {
        HeaderOut r = s[0];
        for (int k = 1; k < 16; k++) {
            r.min_x = fminf(r.min_x, s[k].min_x); r.min_y = fminf(r.min_y, s[k].min_y);
            r.max_x = fmaxf(r.max_x, s[k].max_x); r.max_y = fmaxf(r.max_y, s[k].max_y);
            r.h_max = fmaxf(r.h_max, s[k].h_max); r.h_min = fminf(r.h_min, s[k].h_min);
            r.min_cfl = fminf(r.min_cfl, s[k].min_cfl);
        }
        *out = r;
        if (h_ctrl) {
            SolverCtrl v = *ctrl;
            v.seq = seq - 1u;
            *h_ctrl = v;
            *h_status = *status;
            __threadfence_system();
            ((volatile SolverCtrl*)h_ctrl)->seq = seq;
        }
    }
}

__global__ __launch_bounds__(256) void k_pack_upload(uint32_t n, const float* __restrict__ mass, const float2* __restrict__ pos,
                                                      const float2* __restrict__ velin, float4* __restrict__ pm, float2* __restrict__ vel,
                                                      uint32_t* __restrict__ orig, float* __restrict__ lvl, float* __restrict__ lvlold,
                                                      float* __restrict__ h2_next)
{
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float2 p = pos[i];
    pm[i] = make_float4(p.x, p.y, mass[i], 0.f);
    h2_next[i] = h_from_mass(mass[i], 1.f);   // FluidSimulation::new: h_init with INIT_REST_DENSITY (simulation.rs:505-520, 344)
    vel[i] = velin[i];
    orig[i] = i;
    lvl[i] = __uint_as_float(0x7fc00000u);  // LevelEstimationState::FluidInterior
    lvlold[i] = 0.f;
}

// copy the Jacobi control block and the error word into mapped pinned host memory (one lane); the host
// busy-polls an event recorded right behind this kernel
// `seq` lands in the host copy's last pad word AFTER everything else: the host spins on that word of mapped memory instead of
// polling an event (sync_ctrl), which shortens the device -> host leg of the two waits of a step
__global__ void k_publish(const SolverCtrl* __restrict__ ctrl, const DeviceStatus* __restrict__ status, SolverCtrl* __restrict__ h_ctrl,
                          DeviceStatus* __restrict__ h_status, uint32_t seq)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        SolverCtrl v = *ctrl;
        v.seq = seq - 1u;
        *h_ctrl = v;
        *h_status = *status;
        __threadfence_system();
        ((volatile SolverCtrl*)h_ctrl)->seq = seq;
    }
}

enum { G_F32 = 0, G_F32X2 = 1, G_PM_X = 2, G_PM_M = 3, G_PM_H = 4, G_U32 = 5, G_H2NEXT = 6, G_U8 = 7, G_F32X4_ZW = 8 };

// dst[orig[i]] = field[i]: back to host particle order
__global__ __launch_bounds__(256) void k_to_host_order(uint32_t n, int kind, const uint32_t* __restrict__ orig, const void* __restrict__ src,
                                                        void* __restrict__ dst)
{
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint32_t o = orig ? orig[i] : i;   // orig == nullptr: keep device order (slab mode)
    switch (kind) {
    case G_F32: ((float*)dst)[o] = ((const float*)src)[i]; break;
    case G_U32: ((uint32_t*)dst)[o] = ((const uint32_t*)src)[i]; break;
    case G_U8: ((uint8_t*)dst)[o] = ((const uint8_t*)src)[i]; break;
    case G_F32X2: ((float2*)dst)[o] = ((const float2*)src)[i]; break;
    case G_PM_X: { float4 p = ((const float4*)src)[i]; ((float2*)dst)[o] = make_float2(p.x, p.y); } break;
    case G_F32X4_ZW: { float4 p = ((const float4*)src)[i]; ((float2*)dst)[o] = make_float2(p.z, p.w); } break;   // a^p out of its {x, y, a^p} record
    case G_PM_M: ((float*)dst)[o] = ((const float4*)src)[i].z; break;
    case G_PM_H: ((float*)dst)[o] = ((const float4*)src)[i].w; break;
    case G_H2NEXT: ((float*)dst)[o] = h_from_mass(((const float4*)src)[i].z, 1.f); break;  // simulation.rs:505-520
    }
}

// field[i] = src[orig[i]]: host-order upload of one field into the sorted SoA
__global__ __launch_bounds__(256) void k_from_host_order(uint32_t n, int kind, const uint32_t* __restrict__ orig, const void* __restrict__ src,
                                                          void* __restrict__ dst)
{
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint32_t o = orig ? orig[i] : i;
    switch (kind) {
    case G_F32: ((float*)dst)[i] = ((const float*)src)[o]; break;
    case G_U32: ((uint32_t*)dst)[i] = ((const uint32_t*)src)[o]; break;
    case G_F32X2: ((float2*)dst)[i] = ((const float2*)src)[o]; break;
    case G_U8: ((uint8_t*)dst)[i] = ((const uint8_t*)src)[o]; break;
    case G_PM_X: { float2 p = ((const float2*)src)[o]; float4 q = ((float4*)dst)[i]; q.x = p.x; q.y = p.y; ((float4*)dst)[i] = q; } break;
    case G_PM_M: { float4 q = ((float4*)dst)[i]; q.z = ((const float*)src)[o]; ((float4*)dst)[i] = q; } break;
    case G_PM_H: { float4 q = ((float4*)dst)[i]; q.w = ((const float*)src)[o]; ((float4*)dst)[i] = q; } break;
    default: break;
    }
}

// dst[slot[r]] = src[r]: the rows of a slab context's owned particles (download order) into their slots
__global__ __launch_bounds__(256) void k_rows_to_slots(uint32_t n, int kind, const uint32_t* __restrict__ slot, const void* __restrict__ src, void* __restrict__ dst)
{
    const uint32_t r = blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    const uint32_t i = slot[r];
    switch (kind) {
    case G_F32: ((float*)dst)[i] = ((const float*)src)[r]; break;
    case G_U32: ((uint32_t*)dst)[i] = ((const uint32_t*)src)[r]; break;
    case G_F32X2: ((float2*)dst)[i] = ((const float2*)src)[r]; break;
    case G_U8: ((uint8_t*)dst)[i] = ((const uint8_t*)src)[r]; break;
    case G_PM_X: { float2 p = ((const float2*)src)[r]; float4 q = ((float4*)dst)[i]; q.x = p.x; q.y = p.y; ((float4*)dst)[i] = q; } break;
    case G_PM_M: { float4 q = ((float4*)dst)[i]; q.z = ((const float*)src)[r]; ((float4*)dst)[i] = q; } break;
    case G_PM_H: { float4 q = ((float4*)dst)[i]; q.w = ((const float*)src)[r]; ((float4*)dst)[i] = q; } break;
    default: break;
    }
}

__global__ __launch_bounds__(256) void k_cell_index_host(uint32_t n, GridP g, const uint32_t* __restrict__ orig, const float4* __restrict__ pm,
                                                          uint32_t* __restrict__ dst)
{
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float4 p = pm[i];
    int cx = (int)floorf(p.x / g.cs) - g.minx, cy = (int)floorf(p.y / g.cs) - g.miny;
    dst[orig[i]] = (uint32_t)cx + (uint32_t)cy * (uint32_t)g.sx;
}

// list lengths of the extended lists, host particle order
__global__ __launch_bounds__(256) void k_ext_counts(uint32_t n, const uint4* __restrict__ nl_ext, const uint32_t* __restrict__ orig, uint32_t* __restrict__ dst)
{
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[orig[i]] = nl_ext[i].w & 0xffffu;
}

// CSR export of the neighbour lists (NeighborhoodCache) in host particle order
__global__ __launch_bounds__(256) void k_fill_neighbors(uint32_t n, GridP g, TileP t, const uint32_t* __restrict__ cell_start,
                                                         const uint32_t* __restrict__ cxy, const uint32_t* __restrict__ orig,
                                                         const float4* __restrict__ pm, const uint32_t* __restrict__ offsets_host,
                                                         uint32_t* __restrict__ indices, float k, const uint32_t* __restrict__ rowmap)
{
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    // rowmap (slab context): row of slot i in the export = its rank among the owned slots, 0xffffffff for a ghost; the
    // neighbours are named by their global ids (ghosts carry theirs)
    const uint32_t row = rowmap ? rowmap[i] : orig[i];
    if (row == 0xffffffffu) return;
    const float4 Ai = pm[i];
    const uint32_t c = cxy[i];
    const int cx = c & 0xffffu, cy = c >> 16;
    uint32_t w = offsets_host[row];
    const int R = stencil_radius(g, t, Ai.w, cx, cy, k);
    for (int dy = -R; dy <= R; dy++) {
        int yy = cy + dy;
        if (yy < 0 || yy >= g.sy) continue;
        uint32_t b = cell_start[(uint32_t)yy * g.sx + max(cx - R, 0)];
        uint32_t e = cell_start[(uint32_t)yy * g.sx + min(cx + R + 1, g.sx)];
        for (uint32_t j = b; j < e; j++) {
            const float4 Aj = pm[j];
            const float dx = Ai.x - Aj.x, dyy = Ai.y - Aj.y;
            const float r2 = dx * dx + dyy * dyy;
            const float s = ((Ai.w + Aj.w) * 0.5f) * k;
            if (r2 < s * s) indices[w++] = orig[j];
        }
    }
}

// check_correct_neighborhood (simulation.rs:1810-1863) against the O(N^2) definition: the sweeps only
// ever visit pairs that satisfy the predicate, so equal COUNTS imply equal sets.
__global__ __launch_bounds__(256) void k_check_neighborhood(uint32_t n, const float4* __restrict__ pm, const uint32_t* __restrict__ ncount,
                                                             const uint32_t* __restrict__ orig, DeviceStatus* status, const uint8_t* __restrict__ owned)
{
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    // slab decomposition: an owned particle's neighbours are all among owned + ghosts, so the O(N^2) count over the local arrays
    // is the global one; ghost lanes have no list
    if (owned && !owned[i]) return;
    const float4 Ai = pm[i];
    uint32_t cnt = 0;
    for (uint32_t j = 0; j < n; j++) {
        const float4 Aj = pm[j];
        const float dx = Ai.x - Aj.x, dy = Ai.y - Aj.y;
        const float s = ((Ai.w + Aj.w) * 0.5f) * 2.f;
        cnt += (dx * dx + dy * dy < s * s) ? 1u : 0u;
    }
    if (cnt != ncount[i] && atomicCAS(&status->error, 0u, (uint32_t)SPH_ERR_CHECK_NEIGHBORHOOD) == 0u) status->info = orig[i];
}

// ------------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------------
static int alloc_particle_buffers(sph_ctx* c)
{
    const size_t n = c->cap ? c->cap : 1;
    for (int k = 0; k < 2; k++) {
        HIPCHK(c, c->pm[k].ensure(n * sizeof(float4)));
        HIPCHK(c, c->vel[k].ensure(n * sizeof(float2)));
        HIPCHK(c, c->orig[k].ensure(n * sizeof(uint32_t)));
        HIPCHK(c, c->lvl[k].ensure(n * sizeof(float)));
        HIPCHK(c, c->lvlold[k].ensure(n * sizeof(float)));
        HIPCHK(c, c->h2n[k].ensure(n * sizeof(float)));
        HIPCHK(c, c->szc[k].ensure(n));
        HIPCHK(c, c->key[k].ensure(n * sizeof(uint32_t)));
        HIPCHK(c, c->val[k].ensure(n * sizeof(uint32_t)));
    }
    HIPCHK(c, c->vel_tmp.ensure(n * sizeof(float2)));
    HIPCHK(c, c->sort_scratch.ensure(radix_sort_scratch_elems((uint32_t)n) * sizeof(uint32_t)));
    HIPCHK(c, c->cxy.ensure(n * sizeof(uint32_t)));
    DevBuf* f1[] = {&c->rho, &c->lam_sum, &c->constf, &c->aii, &c->src, &c->p0, &c->p1, &c->dens_err, &c->stat, &c->ncount,
                    &c->mrho, &c->pt0, &c->pt1};
    for (auto b : f1) HIPCHK(c, b->ensure(n * sizeof(float)));
    HIPCHK(c, c->lam_grad.ensure(n * sizeof(float2)));
    if (c->exact) {   // the boundary handler's per-SDF entries (MathExact, sph_device.h)
        HIPCHK(c, c->wall_pl.ensure(n * sizeof(float2) * SPH_MAX_PLANES));
        HIPCHK(c, c->wall_cnt.ensure(n));
    }
    HIPCHK(c, c->lam_prev.ensure(n * sizeof(float)));
    HIPCHK(c, c->omega.ensure(n * sizeof(float)));
    HIPCHK(c, c->pacc.ensure(n * sizeof(float4)));   // {x, y, a^p}
    HIPCHK(c, c->prec0.ensure(n * sizeof(float4)));  // {x, y, p / rho^2, p} of pressure buffer 0 / 1 (OpPressureAccelU)
    HIPCHK(c, c->prec1.ensure(n * sizeof(float4)));
    HIPCHK(c, c->xv.ensure(n * sizeof(float4)));     // {x, y, v} for the source-term sweeps (OpSourceU)
    HIPCHK(c, c->scratch.ensure(n * sizeof(float4)));
    HIPCHK(c, c->nl.ensure(sweep_list_bytes((uint32_t)n)));
    HIPCHK(c, c->nl_ok.ensure(n));
    HIPCHK(c, c->red_partials.ensure(sizeof(SolverPartial) * (size_t)solver_reduce_blocks((uint32_t)n)));
    HIPCHK(c, c->hdr_ahead_partials.ensure(sizeof(HeaderOut) * (((size_t)n + 255) / 256)));   // one per 256-thread block of k_solver_tail (not a sweep block)
    return SPH_OK;
}

// every switch the library reads from the environment, in one place (sph_context.hpp: Options)
Options options_from_env()
{
    Options o;
    auto num = [](const char* name, int dflt) {
        const char* e = getenv(name);
        return e && e[0] ? atoi(e) : dflt;
    };
    auto flag = [](const char* name) { return getenv(name) != nullptr ? 1 : 0; };
    // the product library reads the math policy, the transport's behaviour and the debug aids -- nothing that selects between kernels
    o.exact = num("SPH_HIP_EXACT", 0) == 1 ? 1 : 0;
    o.overlap = num("SPH_OVERLAP", -1);
    o.loopback_sync = num("SPH_LOOPBACK_SYNC", 0) != 0 ? 1 : 0;
    o.force_slab_mode = flag("SPH_FORCE_SLAB_MODE");
    o.event_wait = flag("SPH_EVENT_WAIT");
    o.debug_sync = num("SPH_DEBUG_SYNC", 0);
    o.debug_counts = flag("SPH_DEBUG_COUNTS");
    o.comm_delay_us = num("SPH_DEBUG_COMM_DELAY_US", 0);
    o.hip_trace = num("SPH_HIP_TRACE", 0) == 1 ? 1 : 0;
#ifdef SPH_LAB
    // the LABORATORY build (-DSPH_LAB: adaptive_sph_amd/build.py build_lab -> libsph_lab.so, same sources): the ablation switches that put
    // an older or alternative form of a kernel / a queueing policy beside the product's -- what the bit-identity tests compare against
    // (tests/conftest.py: lab_lib) and what scripts/variants and scripts/gpu_*.py time.  Defaults = the product's behaviour.
    o.paced = num("SPH_PACED", 1) != 0 ? 1 : 0;
    o.pace_lead = num("SPH_PACE_LEAD", 0);
    o.pace_pred = num("SPH_PACE_PRED", 0xffff);
    o.chain = num("SPH_CHAIN", -1);
    o.accel_generic = flag("SPH_ACCEL_GENERIC");
    o.jacobi_generic = flag("SPH_JACOBI_GENERIC");
    o.source_generic = flag("SPH_SOURCE_GENERIC");
    o.slab_general = flag("SPH_SLAB_GENERAL");
    o.slab_level_plain = flag("SPH_SLAB_LEVEL_PLAIN");
    o.level_serial = flag("SPH_LEVEL_SERIAL");
    o.level_batch8 = flag("SPH_LEVEL_BATCH8");
    o.level_queue = num("SPH_LEVEL_QUEUE", 1) != 0 ? 1 : 0;
    o.offset_lists = num("SPH_OFFSET_LISTS", 1) != 0 ? 1 : 0;
    o.no_fuse = flag("SPH_NO_FUSE");
    o.side_stream_normal = flag("SPH_SIDE_STREAM_NORMAL");
    o.tile = num("SPH_TILE", 0);
    o.ahead_build = num("SPH_AHEAD_BUILD", 1) != 0 ? 1 : 0;
    o.inc_sort = num("SPH_INC_SORT", 1);
    o.slab_paced = num("SPH_SLAB_PACED", 1) != 0 ? 1 : 0;
    o.slab_records = num("SPH_SLAB_RECORDS", 1) != 0 ? 1 : 0;
    o.side_cus = num("SPH_SIDE_CUS", 0);
    o.main_exclude = num("SPH_MAIN_EXCLUDE", 0) != 0 ? 1 : 0;
#else
    // the product ignores the laboratory's switches -- and says so, once per process (advisor r5: a bisect script that sets them against
    // libsph_hip.so would otherwise run the defaults in every row and report nothing)
    static const char* const lab_only[] = {"SPH_PACED", "SPH_PACE_LEAD", "SPH_PACE_PRED", "SPH_CHAIN", "SPH_ACCEL_GENERIC", "SPH_JACOBI_GENERIC", "SPH_SOURCE_GENERIC",
                                           "SPH_SLAB_GENERAL", "SPH_SLAB_LEVEL_PLAIN", "SPH_LEVEL_SERIAL", "SPH_LEVEL_BATCH8", "SPH_LEVEL_QUEUE", "SPH_OFFSET_LISTS",
                                           "SPH_NO_FUSE", "SPH_SIDE_STREAM_NORMAL", "SPH_TILE", "SPH_AHEAD_BUILD", "SPH_INC_SORT", "SPH_SLAB_PACED", "SPH_SLAB_RECORDS",
                                           "SPH_SIDE_CUS", "SPH_MAIN_EXCLUDE"};
    static bool warned = false;
    if (!warned)
        for (const char* name : lab_only)
            if (getenv(name)) {
                fprintf(stderr, "libsph_hip: %s is a laboratory switch and is IGNORED by the product library (use SPH_HIP_LIBRARY=libsph_lab.so, scripts/README.md)\n", name);
                warned = true;
            }
#endif
    return o;
}

extern "C" int sph_create(uint64_t n_capacity, int device_id, const sph_plane* planes, int n_planes, sph_ctx** out)
{
    if (!out || n_planes < 0 || n_planes > SPH_MAX_PLANES || (n_planes > 0 && !planes)) return SPH_ERR_INVALID_ARGUMENT;
    if (n_capacity >= (1ull << 31)) return SPH_ERR_CAPACITY;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device_id < 0 || device_id >= ndev) return SPH_ERR_DEVICE;
    if (hipSetDevice(device_id) != hipSuccess) return SPH_ERR_DEVICE;
    sph_ctx* c = new sph_ctx();
    c->device = device_id;
    c->cap = n_capacity;
    c->n_planes = n_planes;
    for (int k = 0; k < n_planes; k++) c->bnd_h.planes[k] = PlaneP{planes[k].dir_x, planes[k].dir_y, planes[k].delta};
    c->opt = options_from_env();
    c->exact = c->opt.exact;
    auto bail = [&](int code) {
        sph_destroy(c);
        return code;
    };
    // (lab) CU masks: bit i of the mask = CU i / 8 of XCD i % 8 (the dispatcher's round-robin over the 8 XCDs); the side stream gets the
    // first `side_cus` CUs of every XCD, the main stream -- if asked -- everything else
    uint32_t side_mask[8] = {0, 0, 0, 0, 0, 0, 0, 0}, main_mask[8];
    for (int b = 0; b < 8 * c->opt.side_cus && b < 256; b++) side_mask[b >> 5] |= 1u << (b & 31);
    for (int w = 0; w < 8; w++) main_mask[w] = ~side_mask[w];
    if (c->opt.side_cus > 0 && c->opt.main_exclude) {
        if (hipExtStreamCreateWithCUMask(&c->stream, 8, main_mask) != hipSuccess) return bail(SPH_ERR_DEVICE);
    } else if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) return bail(SPH_ERR_DEVICE);
    if (c->opt.side_cus > 0) {
        if (hipExtStreamCreateWithCUMask(&c->stream2, 8, side_mask) != hipSuccess) return bail(SPH_ERR_DEVICE);
    } else
    {
        // the side stream carries the level-set propagation: ~100 tiny dependent launches that must slot in between the waves of
        // the main stream's sweeps -- highest priority, so that the dispatcher serves it first whenever a CU has room
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        if (c->opt.side_stream_normal || hipStreamCreateWithPriority(&c->stream2, hipStreamNonBlocking, hi) != hipSuccess)
            if (hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking) != hipSuccess) return bail(SPH_ERR_DEVICE);
    }
    if (hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess) return bail(SPH_ERR_DEVICE);
    for (auto& e : c->ev)
        if (hipEventCreate(&e) != hipSuccess) return bail(SPH_ERR_DEVICE);
    if (alloc_particle_buffers(c) != SPH_OK) return bail(SPH_ERR_DEVICE);
    bool ok = c->planes_d.ensure(sizeof(BoundaryP)) == hipSuccess && c->lam_lut.ensure(10001 * 4) == hipSuccess &&
              c->dlam_lut.ensure(10001 * 4) == hipSuccess && c->hdr_partials.ensure(sizeof(HeaderOut) * HDR_BLOCKS) == hipSuccess &&
              c->hdr_out.ensure(sizeof(HeaderOut)) == hipSuccess && c->ctrl.ensure(3 * sizeof(SolverCtrl)) == hipSuccess &&
              c->status.ensure(sizeof(DeviceStatus)) == hipSuccess && c->n_tiles.ensure(16) == hipSuccess;
    if (!ok) return bail(SPH_ERR_DEVICE);
    if (hipHostMalloc((void**)&c->hdr_host, sizeof(HeaderOut), hipHostMallocMapped) != hipSuccess) return bail(SPH_ERR_DEVICE);
    if (hipHostMalloc((void**)&c->ctrl_host, 3 * sizeof(SolverCtrl), hipHostMallocMapped) != hipSuccess) return bail(SPH_ERR_DEVICE);
    if (hipHostMalloc((void**)&c->status_host, sizeof(DeviceStatus), hipHostMallocMapped) != hipSuccess) return bail(SPH_ERR_DEVICE);
    if (hipHostMalloc((void**)&c->lvl_changed, 64 * sizeof(uint32_t), hipHostMallocMapped) != hipSuccess) return bail(SPH_ERR_DEVICE);
    if (hipHostGetDevicePointer((void**)&c->lvl_changed_dev, c->lvl_changed, 0) != hipSuccess) return bail(SPH_ERR_DEVICE);
    memset(c->lvl_changed, 0, 64 * sizeof(uint32_t));
    if (hipHostGetDevicePointer((void**)&c->hdr_host_dev, c->hdr_host, 0) != hipSuccess) return bail(SPH_ERR_DEVICE);
    if (hipHostGetDevicePointer((void**)&c->ctrl_host_dev, c->ctrl_host, 0) != hipSuccess) return bail(SPH_ERR_DEVICE);
    if (hipHostGetDevicePointer((void**)&c->status_host_dev, c->status_host, 0) != hipSuccess) return bail(SPH_ERR_DEVICE);
    if (hipEventCreateWithFlags(&c->ev_sync, hipEventDisableTiming) != hipSuccess) return bail(SPH_ERR_DEVICE);
    memset(c->status_host, 0, sizeof(DeviceStatus));
    memset((void*)c->ctrl_host, 0, 3 * sizeof(SolverCtrl));
    c->prog_host = (volatile uint32_t*)(c->ctrl_host + 2);   // (the third block: only its first word is used)
    c->prog_host_dev = (uint32_t*)(c->ctrl_host_dev + 2);
    // BoundaryWinchenbach2020::new (boundary_winchenbach2020.rs:33-36)
    std::vector<float> lam, dlam;
    sph_lambda::build_luts(lam, dlam);
    if (hipMemcpy(c->lam_lut.p, lam.data(), lam.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(c->dlam_lut.p, dlam.data(), dlam.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(c->planes_d.p, &c->bnd_h, sizeof(BoundaryP), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemset(c->status.p, 0, sizeof(DeviceStatus)) != hipSuccess || hipMemset(c->ctrl.p, 0, sizeof(SolverCtrl)) != hipSuccess ||
        hipDeviceSynchronize() != hipSuccess)
        return bail(SPH_ERR_DEVICE);
    *out = c;
    return SPH_OK;
}

// Sdf2D::new_boundary_box / Sdf2DConnectedComponents::from_points (sdf/sdf2d.rs:36-75, 167-179)
extern "C" int sph_set_boundary_polygon(sph_ctx* c, const float* pts, int n)
{
    if (!c || !pts || n < 3 || n > SPH_MAX_POLYGON_POINTS) return SPH_ERR_INVALID_ARGUMENT;
    HIPCHK(c, hipSetDevice(c->device));
    BoundaryP b = c->bnd_h;
    for (int i = 0; i < n; i++) {
        b.px[i] = pts[2 * i];
        b.py[i] = pts[2 * i + 1];
    }
    for (int i = 0; i < n; i++) {
        const int i1 = (i + 1) % n;
        const float dx = b.px[i1] - b.px[i], dy = b.py[i1] - b.py[i];
        const float n2 = dx * dx + dy * dy;
        if (!(n2 > 0.00001f)) return c->fail(SPH_ERR_INVALID_ARGUMENT, "assertion failed: line_dir.norm_squared() > 0.00001");
        const float nn = sqrtf(n2);   // normalize_mut
        b.dx[i] = dx / nn;
        b.dy[i] = dy / nn;
    }
    for (int i = 0; i < n; i++) {
        const int pi = i == 0 ? n - 1 : i - 1;
        const float px = -b.dy[pi] + -b.dy[i], py = b.dx[pi] + b.dx[i];   // rotate_left_90_degrees of both edge directions
        if (!(px * px + py * py > 0.00001f)) return c->fail(SPH_ERR_INVALID_ARGUMENT, "assertion failed: pseudo_normal.norm_squared() > 0.00001");
        b.nx[i] = px;
        b.ny[i] = py;
    }
    b.poly_n = n;
    c->bnd_h = b;
    c->n_planes = 1;   // one Sdf
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(c->planes_d.p, &c->bnd_h, sizeof(BoundaryP), hipMemcpyHostToDevice));
    return SPH_OK;
}

extern "C" void sph_destroy(sph_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->stream2) (void)hipStreamSynchronize(c->stream2);
    dist_release(c);
    DevBuf* all[] = {&c->pm[0], &c->pm[1], &c->vel[0], &c->vel[1], &c->orig[0], &c->orig[1], &c->lvl[0], &c->lvl[1], &c->lvlold[0],
                     &c->lvlold[1], &c->vel_tmp, &c->key[0], &c->key[1], &c->val[0], &c->val[1], &c->sort_scratch, &c->cxy, &c->cell_start,
                     &c->cs_scratch, &c->hdr_ahead_partials, &c->h2n[0], &c->h2n[1], &c->lam_prev, &c->nl, &c->nlx, &c->tile_raw, &c->tile_h, &c->tile_h_ext, &c->lvl_changed_d, &c->lvl_tmp, &c->lvl_nrm, &c->lvl_state, &c->lvl_when, &c->lvl_mark, &c->lvl_queue, &c->nloff, &c->nlh, &c->flag_surface,
                     &c->flag_insufficient, &c->con_thr, &c->con_consumed, &c->con_h, &c->flag_reduced, &c->szc[0], &c->szc[1], &c->omega, &c->stash, &c->nl_ext, &c->nlx_ext, &c->nl_ok, &c->mrho, &c->pt0, &c->pt1, &c->prec0, &c->prec1, &c->xv, &c->rho, &c->lam_sum, &c->lam_grad, &c->wall_pl, &c->wall_cnt, &c->constf, &c->aii, &c->src, &c->p0, &c->p1, &c->pacc, &c->dens_err,
                     &c->stat, &c->ncount, &c->planes_d, &c->lam_lut, &c->dlam_lut, &c->hdr_partials, &c->hdr_out, &c->ctrl, &c->status,
                     &c->n_tiles, &c->red_partials, &c->scratch, &c->split_patterns, &c->akey[0], &c->akey[1], &c->aval[0], &c->aval[1], &c->acxy, &c->acell_start, &c->pm2,
                     &c->atile_raw, &c->atile_h, &c->inc_head, &c->inc_next, &c->inc_bsum, &c->inc_movers, &c->export_d_off, &c->export_d_idx};
    for (auto b : all) b->release();
    if (c->hdr_host) (void)hipHostFree(c->hdr_host);
    if (c->ctrl_host) (void)hipHostFree(c->ctrl_host);
    if (c->status_host) (void)hipHostFree(c->status_host);
    if (c->lvl_changed) (void)hipHostFree(c->lvl_changed);
    for (auto& e : c->ev)
        if (e) (void)hipEventDestroy(e);
    if (c->ev_sync) (void)hipEventDestroy(c->ev_sync);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->stream2) (void)hipStreamDestroy(c->stream2);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    delete c;
}

extern "C" const char* sph_last_error(const sph_ctx* c) { return c ? c->err.c_str() : "null context"; }
extern "C" uint64_t sph_num_particles(const sph_ctx* c) { return c ? c->n : 0; }
extern "C" float sph_time(const sph_ctx* c) { return c ? c->time : 0.f; }
extern "C" int sph_set_time(sph_ctx* c, float t, uint64_t step)
{
    if (!c) return SPH_ERR_INVALID_ARGUMENT;
    c->time = t;
    c->step_number = step;
    return SPH_OK;
}

// include/sph_ffi.h: the arithmetic of the sweeps, chosen by the host through the ABI (SPH_HIP_EXACT only sets the initial value).
// The particle state is policy-free; what a step derives from it (lists, cell table, the build queued ahead, the header computed ahead)
// is dropped, so the next step starts like the first one after an upload -- and the EXACT policy's per-SDF boundary entries get their
// buffers here if the context was created without them.
extern "C" int sph_set_math_policy(sph_ctx* c, int policy)
{
    if (!c || (policy != SPH_MATH_FAST && policy != SPH_MATH_EXACT)) return SPH_ERR_INVALID_ARGUMENT;
    if (c->poisoned) return c->fail(SPH_ERR_POISONED, "sph_set_math_policy: the context is poisoned until sph_upload");
    if (policy == c->exact) return SPH_OK;
    HIPCHK(c, hipSetDevice(c->device));
    if (int rc = wait_stream(c)) return rc;   // a build queued ahead may still be running on the old policy's buffers
    c->exact = policy;
    if (c->exact) {
        size_t n = c->cap ? c->cap : 1;
        HIPCHK(c, c->wall_pl.ensure(n * sizeof(float2) * SPH_MAX_PLANES));
        HIPCHK(c, c->wall_cnt.ensure(n));
    }
    c->grid_valid = false;
    c->ahead.valid = false;
    c->hdr_ahead = false;
    c->lists_after = false;
    if (c->ctrl_host) ((uint32_t*)(c->ctrl_host + 2))[1] = 0u;
    c->inc_count_valid = false;
    return SPH_OK;
}
extern "C" int sph_get_math_policy(const sph_ctx* c) { return c ? c->exact : -1; }

extern "C" int sph_upload(sph_ctx* c, uint64_t n, const float* mass, const float* pos, const float* vel)
{
    if (!c || (n && (!mass || !pos || !vel))) return SPH_ERR_INVALID_ARGUMENT;
    if (n > c->cap) return c->fail(SPH_ERR_CAPACITY, "n=%llu exceeds capacity %llu", (unsigned long long)n, (unsigned long long)c->cap);
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    c->n = n;
    c->cur = 0;
    c->pcur = 0;
    c->poisoned = false;
    c->dist.have_flags = false;
    c->dist.n_tot = (uint32_t)n;
    c->grid_valid = false;   // lists, cell indices and per-step outputs belong to the vector before this call
    if (c->ctrl_host) ((uint32_t*)(c->ctrl_host + 2))[1] = 0u;
    c->inc_count_valid = false;   // ... and so does the incremental sort's last mover count (advisor r4)
    c->export_d_off.release();   // (the CSR export's device buffers: a host that re-uploads is not exporting every step)
    c->export_d_idx.release();
    c->have_level = false;
    c->have_reduced = false;
    c->lists_after = false;
    c->hdr_ahead = false;
    if (n == 0) return SPH_OK;
    // stage host arrays through scratch buffers: mass -> key[1], pos -> scratch, vel -> vel_tmp
    HIPCHK(c, hipMemcpyAsync(c->key[1].p, mass, n * sizeof(float), hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(c->scratch.p, pos, n * sizeof(float2), hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(c->vel_tmp.p, vel, n * sizeof(float2), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_pack_upload, dim3((n + 255) / 256), dim3(256), 0, s, (uint32_t)n, c->key[1].as<float>(), c->scratch.as<float2>(),
                       c->vel_tmp.as<float2>(), c->pm[0].as<float4>(), c->vel[0].as<float2>(), c->orig[0].as<uint32_t>(),
                       c->lvl[0].as<float>(), c->lvlold[0].as<float>(), c->h2n[0].as<float>());
    HIPCHK(c, hipMemsetAsync(c->szc[0].p, 2, n, s));   // ParticleSizeClass::Optimal (simulation.rs:318)
    DevBuf* zero[] = {&c->rho, &c->lam_sum, &c->constf, &c->aii, &c->src, &c->p0, &c->p1, &c->dens_err, &c->ncount};
    for (auto b : zero) HIPCHK(c, hipMemsetAsync(b->p, 0, n * sizeof(float), s));
    HIPCHK(c, hipMemsetAsync(c->lam_grad.p, 0, n * sizeof(float2), s));
    HIPCHK(c, hipMemsetAsync(c->pacc.p, 0, n * sizeof(float4), s));
    c->pressure_cur = 0;
    HIPCHK(c, hipStreamSynchronize(s));
    return SPH_OK;
}

struct FieldRef {
    int kind;         // G_*
    const void* src;  // device array in sorted order
    size_t elem;      // bytes per particle on the host side
    bool uploadable;
};

static bool field_ref(sph_ctx* c, int field, FieldRef* r)
{
    const int k = c->cur;
    switch (field) {
    case SPH_F_MASS: *r = {G_PM_M, c->pm[c->pcur].p, 4, true}; return true;
    case SPH_F_POSITION: *r = {G_PM_X, c->pm[c->pcur].p, 8, true}; return true;
    case SPH_F_VELOCITY: *r = {G_F32X2, c->vel[k].p, 8, true}; return true;
    case SPH_F_PRESSURE_ACCEL: *r = {G_F32X4_ZW, c->pacc.p, 8, false}; return true;
    case SPH_F_DENSITY: *r = {G_F32, c->rho.p, 4, false}; return true;
    case SPH_F_PPE_SOURCE_TERM: *r = {G_F32, c->src.p, 4, false}; return true;
    case SPH_F_PRESSURE: *r = {G_F32, c->pressure_cur ? c->p1.p : c->p0.p, 4, false}; return true;
    case SPH_F_AII: *r = {G_F32, c->aii.p, 4, false}; return true;
    case SPH_F_DENSITY_ERROR: *r = {G_F32, c->dens_err.p, 4, false}; return true;
    case SPH_F_H2: *r = {G_PM_H, c->pm[c->pcur].p, 4, true}; return true;
    case SPH_F_H2_NEXT: *r = {G_F32, c->h2n[k].p, 4, true}; return true;
    case SPH_F_CONSTANT_FIELD: *r = {G_F32, c->constf.p, 4, false}; return true;
    case SPH_F_NEIGHBOR_COUNT: *r = {G_U32, c->ncount.p, 4, false}; return true;
    case SPH_F_LEVEL_ESTIMATION: *r = {G_F32, c->lvl[k].p, 4, true}; return true;
    case SPH_F_LEVEL_OLD: *r = {G_F32, c->lvlold[k].p, 4, true}; return true;
    case SPH_F_PARTICLE_SIZE_CLASS: *r = {G_U8, c->szc[k].p, 1, true}; return true;
    case SPH_F_LAMBDA_SUM: *r = {G_F32, c->lam_sum.p, 4, false}; return true;
    case SPH_F_LAMBDA_GRAD_SUM: *r = {G_F32X2, c->lam_grad.p, 8, false}; return true;
    default: return false;
    }
}

// Slab mode: the arrays hold owned particles and (after a step) interleaved ghosts, and host indices have no
// meaning across migrations; fields come back in DEVICE order of the owned particles, identities via
// SPH_F_PARTICLE_ID (same order for every field until the next step).
static int download_slab(sph_ctx* c, int kind, const void* src, size_t elem, void* dst, uint64_t bytes)
{
    hipStream_t s = c->stream;
    const uint32_t nt = c->dist.have_flags ? c->dist.n_tot : (uint32_t)c->n;
    if (bytes != (uint64_t)c->n * elem) return c->fail(SPH_ERR_INVALID_ARGUMENT, "size mismatch (slab holds %llu particles)", (unsigned long long)c->n);
    if (nt == 0) return SPH_OK;
    std::vector<uint8_t> raw((size_t)nt * elem), flags(nt, 1);
    hipLaunchKernelGGL(k_to_host_order, dim3((nt + 255) / 256), dim3(256), 0, s, nt, kind, (const uint32_t*)nullptr, src, c->scratch.p);
    HIPCHK(c, hipMemcpyAsync(raw.data(), c->scratch.p, raw.size(), hipMemcpyDeviceToHost, s));
    if (c->dist.have_flags) HIPCHK(c, hipMemcpyAsync(flags.data(), c->dist.owned.p, nt, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    size_t w = 0;
    for (uint32_t i = 0; i < nt; i++)
        if (flags[i]) {
            if (w >= c->n) return c->fail(SPH_ERR_DEVICE, "owned-particle count mismatch");
            memcpy((uint8_t*)dst + w * elem, raw.data() + (size_t)i * elem, elem);
            w++;
        }
    return SPH_OK;
}

extern "C" int sph_download(sph_ctx* c, int field, void* dst, uint64_t bytes)
{
    if (!c || !dst) return SPH_ERR_INVALID_ARGUMENT;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    const uint32_t n = (uint32_t)c->n;
    const int k = c->cur;
    if (field == SPH_F_PARTICLE_ID) {
        if (c->dist.on) return download_slab(c, G_U32, c->orig[k].p, 4, dst, bytes);
        if (bytes != (uint64_t)n * 4) return c->fail(SPH_ERR_INVALID_ARGUMENT, "field %d: size mismatch", field);
        for (uint32_t i = 0; i < n; i++) ((uint32_t*)dst)[i] = i;
        return SPH_OK;
    }
    if (field == SPH_F_CELL_INDEX) {
        if (!c->grid_valid) return c->fail(SPH_ERR_INVALID_ARGUMENT, "no grid yet: run a step first");
        if (c->dist.on) return download_slab(c, G_U32, c->key[0].p, 4, dst, bytes);
        if (bytes != (uint64_t)n * 4) return c->fail(SPH_ERR_INVALID_ARGUMENT, "field %d: size mismatch", field);
        if (n == 0) return SPH_OK;
        const bool same_grid = c->fgrid.minx == c->grid.minx && c->fgrid.miny == c->grid.miny && c->fgrid.sx == c->grid.sx && c->fgrid.sy == c->grid.sy;
        if (c->uniform_h && same_grid) {
            // sorted cell keys of the positions the last step started from
            hipLaunchKernelGGL(k_to_host_order, dim3((n + 255) / 256), dim3(256), 0, s, n, (int)G_U32, c->orig[k].as<uint32_t>(),
                               (const void*)c->key[0].p, c->scratch.p);
        } else {
            // multi-resolution scenes sort by a finer grid, a step that adopted the build queued ahead by a wider one (predicted
            // bounding box): recompute the reference-convention index (cell = largest support) from the sorted pre-step snapshot pm[pcur ^ 1]
            hipLaunchKernelGGL(k_cell_index_host, dim3((n + 255) / 256), dim3(256), 0, s, n, c->grid, c->orig[k].as<uint32_t>(),
                               c->pm[c->pcur ^ 1].as<float4>(), c->scratch.as<uint32_t>());
        }
        HIPCHK(c, hipMemcpyAsync(dst, c->scratch.p, bytes, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipStreamSynchronize(s));
        return SPH_OK;
    }
    if (field == SPH_F_STASH || field == SPH_F_FLAG_IS_FLUID_SURFACE || field == SPH_F_FLAG_INSUFFICIENT_NEIGHS) {
        // level-estimation outputs (simulation.rs:539-927).  Before the first step with a level_estimation_method they are
        // the defaults of ParticleVec.
        size_t elem = field == SPH_F_STASH ? 4 : 1;
        if (bytes != (uint64_t)n * elem) return c->fail(SPH_ERR_INVALID_ARGUMENT, "field %d: size mismatch", field);
        if (!c->have_level) {
            memset(dst, 0, bytes);
            return SPH_OK;
        }
        if (n == 0) return SPH_OK;
        const void* src = field == SPH_F_STASH ? c->stash.p : field == SPH_F_FLAG_IS_FLUID_SURFACE ? c->flag_surface.p : c->flag_insufficient.p;
        if (c->dist.on) return download_slab(c, field == SPH_F_STASH ? (int)G_F32 : (int)G_U8, src, elem, dst, bytes);
        hipLaunchKernelGGL(k_to_host_order, dim3((n + 255) / 256), dim3(256), 0, s, n, field == SPH_F_STASH ? (int)G_F32 : (int)G_U8,
                           c->orig[k].as<uint32_t>(), src, c->scratch.p);
        HIPCHK(c, hipMemcpyAsync(dst, c->scratch.p, bytes, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipStreamSynchronize(s));
        return SPH_OK;
    }
    if (field == SPH_F_FLAG_NEIGHBORHOOD_REDUCED) {   // simulation.rs:2164-2168; false before the first constrained step
        if (bytes != (uint64_t)n) return c->fail(SPH_ERR_INVALID_ARGUMENT, "field %d: size mismatch", field);
        if (!c->have_reduced) {
            memset(dst, 0, bytes);
            return SPH_OK;
        }
        if (c->dist.on) return download_slab(c, G_U8, c->flag_reduced.p, 1, dst, bytes);
        if (n == 0) return SPH_OK;
        hipLaunchKernelGGL(k_to_host_order, dim3((n + 255) / 256), dim3(256), 0, s, n, (int)G_U8, c->orig[k].as<uint32_t>(),
                           (const void*)c->flag_reduced.p, c->scratch.p);
        HIPCHK(c, hipMemcpyAsync(dst, c->scratch.p, bytes, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipStreamSynchronize(s));
        return SPH_OK;
    }
    FieldRef r;
    if (!field_ref(c, field, &r)) return c->fail(SPH_ERR_INVALID_ARGUMENT, "unknown field %d", field);
    if (c->dist.on) return download_slab(c, r.kind, r.src, r.elem, dst, bytes);
    if (bytes != (uint64_t)n * r.elem) return c->fail(SPH_ERR_INVALID_ARGUMENT, "field %d: size mismatch", field);
    if (n == 0) return SPH_OK;
    hipLaunchKernelGGL(k_to_host_order, dim3((n + 255) / 256), dim3(256), 0, s, n, r.kind, c->orig[k].as<uint32_t>(), r.src, c->scratch.p);
    HIPCHK(c, hipMemcpyAsync(dst, c->scratch.p, bytes, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    return SPH_OK;
}

// ------------------------------------------------------------------------------------------------
// sparse edits (sph_ffi.h): the script is resolved on the host into "final index f holds object o" + the last value
// written to each field of each touched object; one kernel then gathers the persistent state into final-index order
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_edit_inverse(uint32_t n_old, const uint32_t* __restrict__ orig, uint32_t* __restrict__ slot_of)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n_old) slot_of[orig[i]] = i;
}

__global__ __launch_bounds__(256) void k_edit_apply(uint32_t n_new, uint32_t n_old, const EditSrc* __restrict__ src, const EditSet* __restrict__ sets,
                                                     const uint32_t* __restrict__ slot_of, const float4* __restrict__ pm_in,
                                                     const float2* __restrict__ vel_in, const float* __restrict__ lvl_in,
                                                     const float* __restrict__ lvlold_in, const float* __restrict__ h2n_in,
                                                     const float* __restrict__ lam_in, const uint8_t* __restrict__ szc_in,
                                                     uint8_t* __restrict__ szc_out, float4* __restrict__ pm_out, float2* __restrict__ vel_out,
                                                     uint32_t* __restrict__ orig_out, float* __restrict__ lvl_out, float* __restrict__ lvlold_out,
                                                     float* __restrict__ h2n_out, float* __restrict__ lam_out, const uint32_t* __restrict__ id_in)
{
    const uint32_t f = blockIdx.x * 256 + threadIdx.x;
    if (f >= n_new) return;
    const EditSrc e = src[f];
    uint32_t id = f;   // slab context (id_in): a surviving particle keeps its global id, a new one has none until the host uploads it
    if (id_in) id = e.obj < n_old ? id_in[slot_of[e.obj]] : 0xffffffffu;
    // ParticleVec defaults (simulation.rs:284-334)
    float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
    float2 v = make_float2(0.f, 0.f);
    float lv = __uint_as_float(0x7fc00000u), lo = 0.f, hn = 0.f, lam = 0.f;
    uint8_t cls = 2;   // ParticleSizeClass::Optimal
    if (e.obj < n_old) {
        const uint32_t s = slot_of[e.obj];
        cls = szc_in[s];
        p = pm_in[s];
        v = vel_in[s];
        lv = lvl_in[s];
        lo = lvlold_in[s];
        hn = h2n_in[s];
        lam = lam_in[s];
    }
    if (e.set_idx != 0xffffffffu) {
        const EditSet q = sets[e.set_idx];
        if (q.fields & SPH_EDIT_F_MASS) p.z = q.mass;
        if (q.fields & SPH_EDIT_F_POSITION) { p.x = q.px; p.y = q.py; }
        if (q.fields & SPH_EDIT_F_VELOCITY) { v.x = q.vx; v.y = q.vy; }
        if (q.fields & SPH_EDIT_F_H2) p.w = q.h2;
        if (q.fields & SPH_EDIT_F_H2_NEXT) hn = q.h2_next;
        if (q.fields & SPH_EDIT_F_LEVEL_ESTIMATION) lv = q.lvl;
        if (q.fields & SPH_EDIT_F_LEVEL_OLD) lo = q.lvlold;
    }
    pm_out[f] = p;
    vel_out[f] = v;
    orig_out[f] = id;
    lvl_out[f] = lv;
    lvlold_out[f] = lo;
    h2n_out[f] = hn;
    lam_out[f] = lam;
    szc_out[f] = cls;
}

extern "C" int sph_apply_edits(sph_ctx* c, const sph_edit_op* ops, uint64_t n_ops)
{
    if (!c || (n_ops && !ops)) return SPH_ERR_INVALID_ARGUMENT;
    if (c->poisoned) return c->fail(SPH_ERR_POISONED, "an earlier step failed inside the step: the particle state is undefined until sph_upload");
    HIPCHK(c, hipSetDevice(c->device));
    const uint32_t n_old = (uint32_t)c->n;
    // ---- resolve the script: which object sits at which index, and the last value written to each field of an object
    std::vector<uint32_t> at(n_old);
    for (uint32_t i = 0; i < n_old; i++) at[i] = i;
    uint32_t next_obj = n_old;
    std::vector<EditSet> sets;
    std::vector<uint32_t> set_of;   // object -> record (grown on demand)
    auto record = [&](uint32_t obj) -> EditSet& {
        if (set_of.size() <= obj) set_of.resize((size_t)obj + 1, 0xffffffffu);
        if (set_of[obj] == 0xffffffffu) {
            set_of[obj] = (uint32_t)sets.size();
            sets.push_back(EditSet{0u, 0, 0, 0, 0, 0, 0, 0, 0, 0});
        }
        return sets[set_of[obj]];
    };
    for (uint64_t k = 0; k < n_ops; k++) {
        const sph_edit_op& o = ops[k];
        switch (o.kind) {
        case SPH_EDIT_SET: {
            if (o.a >= at.size()) return c->fail(SPH_ERR_INVALID_ARGUMENT, "edit %llu: index %u out of bounds (len %zu)", (unsigned long long)k, o.a, at.size());
            EditSet& q = record(at[o.a]);
            q.fields |= o.fields;
            if (o.fields & SPH_EDIT_F_MASS) q.mass = o.mass;
            if (o.fields & SPH_EDIT_F_POSITION) { q.px = o.position[0]; q.py = o.position[1]; }
            if (o.fields & SPH_EDIT_F_VELOCITY) { q.vx = o.velocity[0]; q.vy = o.velocity[1]; }
            if (o.fields & SPH_EDIT_F_H2) q.h2 = o.h2;
            if (o.fields & SPH_EDIT_F_H2_NEXT) q.h2_next = o.h2_next;
            if (o.fields & SPH_EDIT_F_LEVEL_ESTIMATION) q.lvl = o.level_estimation;
            if (o.fields & SPH_EDIT_F_LEVEL_OLD) q.lvlold = o.level_old;
        } break;
        case SPH_EDIT_SWAP:
            if (o.a >= at.size() || o.b >= at.size())
                return c->fail(SPH_ERR_INVALID_ARGUMENT, "edit %llu: swap(%u, %u) out of bounds (len %zu)", (unsigned long long)k, o.a, o.b, at.size());
            std::swap(at[o.a], at[o.b]);
            break;
        case SPH_EDIT_TRUNCATE:
            if (o.a < at.size()) at.resize(o.a);   // Vec::truncate: no effect if len is greater
            break;
        case SPH_EDIT_EXTEND:
            if (at.size() + (size_t)o.a > c->cap)
                return c->fail(SPH_ERR_CAPACITY, "edit %llu: %zu particles exceed the capacity %llu", (unsigned long long)k, at.size() + (size_t)o.a,
                               (unsigned long long)c->cap);
            for (uint32_t q = 0; q < o.a; q++) at.push_back(next_obj++);
            break;
        default: return c->fail(SPH_ERR_INVALID_ARGUMENT, "edit %llu: unknown kind %d", (unsigned long long)k, o.kind);
        }
    }
    const uint32_t n_new = (uint32_t)at.size();
    std::vector<EditSrc> src(n_new ? n_new : 1);
    for (uint32_t f = 0; f < n_new; f++) src[f] = EditSrc{at[f], at[f] < set_of.size() ? set_of[at[f]] : 0xffffffffu};
    if (sets.empty()) sets.push_back(EditSet{0u, 0, 0, 0, 0, 0, 0, 0, 0, 0});

    // ---- one gather on the device into final-index order (slot f = host index f, like a fresh upload)
    hipStream_t s = c->stream;
    TmpBuf d_src, d_sets;
    HIPCHK(c, d_src.ensure(src.size() * sizeof(EditSrc)));
    HIPCHK(c, d_sets.ensure(sets.size() * sizeof(EditSet)));
    HIPCHK(c, hipMemcpyAsync(d_src.p, src.data(), src.size() * sizeof(EditSrc), hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(d_sets.p, sets.data(), sets.size() * sizeof(EditSet), hipMemcpyHostToDevice, s));
    const int rc = regather_host_order(c, n_new, d_src.as<EditSrc>(), d_sets.as<EditSet>());
    d_src.release();
    d_sets.release();
    return rc;
}

int regather_host_order(sph_ctx* c, uint32_t n_new, const EditSrc* d_src, const EditSet* d_sets)
{
    hipStream_t s = c->stream;
    const uint32_t n_old = (uint32_t)c->n;
    if (n_new > c->cap) return c->fail(SPH_ERR_CAPACITY, "%u particles exceed the capacity %llu", n_new, (unsigned long long)c->cap);
    TmpBuf d_slot, d_lam;
    HIPCHK(c, d_slot.ensure(((size_t)n_old + 1) * 4));
    HIPCHK(c, d_lam.ensure(((size_t)n_new + 1) * 4));
    const int k = c->cur;
    const bool slab = c->dist.on;
    if (slab) {
        // slab context: "host index" = row of the owned order (what sph_download returns); the arrays also hold the ghosts
        std::vector<uint32_t> slot_of(n_old ? n_old : 1);
        if (c->dist.have_flags) {
            const uint32_t nt = c->dist.n_tot;
            std::vector<uint8_t> flags(nt ? nt : 1);
            if (nt) HIPCHK(c, hipMemcpy(flags.data(), c->dist.owned.p, nt, hipMemcpyDeviceToHost));
            uint32_t w = 0;
            for (uint32_t i = 0; i < nt; i++)
                if (flags[i]) {
                    if (w >= n_old) return c->fail(SPH_ERR_DEVICE, "owned-particle count mismatch");
                    slot_of[w++] = i;
                }
            if (w != n_old) return c->fail(SPH_ERR_DEVICE, "owned-particle count mismatch");
        } else
            for (uint32_t i = 0; i < n_old; i++) slot_of[i] = i;
        if (n_old) HIPCHK(c, hipMemcpyAsync(d_slot.p, slot_of.data(), (size_t)n_old * 4, hipMemcpyHostToDevice, s));
        HIPCHK(c, hipStreamSynchronize(s));   // (slot_of is a local)
    } else if (n_old) hipLaunchKernelGGL(k_edit_inverse, dim3((n_old + 255) / 256), dim3(256), 0, s, n_old, c->orig[k].as<uint32_t>(), d_slot.as<uint32_t>());
    if (n_new)
        hipLaunchKernelGGL(k_edit_apply, dim3((n_new + 255) / 256), dim3(256), 0, s, n_new, n_old, d_src, d_sets,
                           d_slot.as<uint32_t>(), c->pm[c->pcur].as<float4>(), c->vel[k].as<float2>(), c->lvl[k].as<float>(),
                           c->lvlold[k].as<float>(), c->h2n[k].as<float>(), c->lam_sum.as<float>(), c->szc[k].as<uint8_t>(), c->szc[k ^ 1].as<uint8_t>(),
                           c->pm[c->pcur ^ 1].as<float4>(),
                           c->vel[k ^ 1].as<float2>(), c->orig[k ^ 1].as<uint32_t>(), c->lvl[k ^ 1].as<float>(), c->lvlold[k ^ 1].as<float>(),
                           c->h2n[k ^ 1].as<float>(), d_lam.as<float>(), slab ? c->orig[k].as<uint32_t>() : (const uint32_t*)nullptr);
    if (n_new) HIPCHK(c, hipMemcpyAsync(c->lam_sum.p, d_lam.p, (size_t)n_new * 4, hipMemcpyDeviceToDevice, s));
    HIPCHK(c, hipStreamSynchronize(s));
    d_slot.release();
    d_lam.release();
    c->cur = k ^ 1;
    c->pcur ^= 1;
    c->n = n_new;
    c->dist.n_tot = n_new;
    c->dist.have_flags = false;   // (slab context: owned particles only, in row order; the next step selects new ghosts)
    c->dist.n_ghost[0] = c->dist.n_ghost[1] = c->dist.n_halo[0] = c->dist.n_halo[1] = 0;
    c->grid_valid = false;   // lists, cell indices and per-step outputs belong to the vector before this call
    if (c->ctrl_host) ((uint32_t*)(c->ctrl_host + 2))[1] = 0u;
    c->inc_count_valid = false;   // ... and so does the incremental sort's last mover count (advisor r4)
    c->have_level = false;
    c->have_reduced = false;
    c->lists_after = false;
    c->hdr_ahead = false;
    return SPH_OK;
}

extern "C" int sph_upload_field(sph_ctx* c, int field, const void* src, uint64_t bytes)
{
    if (!c || !src) return SPH_ERR_INVALID_ARGUMENT;
    HIPCHK(c, hipSetDevice(c->device));
    FieldRef r;
    if (field == SPH_F_PARTICLE_ID) {
        if (!c->dist.on) return c->fail(SPH_ERR_INVALID_ARGUMENT, "particle ids can only be set on a slab context (they are the host indices otherwise)");
        r = FieldRef{G_U32, c->orig[c->cur].p, 4, true};
    } else if (!field_ref(c, field, &r) || !r.uploadable)
        return c->fail(SPH_ERR_INVALID_ARGUMENT, "field %d cannot be uploaded", field);
    const uint32_t n = (uint32_t)c->n;
    c->hdr_ahead = false;   // the header computed at the end of the last step no longer describes the state
    if (bytes != (uint64_t)n * r.elem) return c->fail(SPH_ERR_INVALID_ARGUMENT, "field %d: size mismatch", field);
    if (n == 0) return SPH_OK;
    hipStream_t s = c->stream;
    if (c->dist.on && c->dist.have_flags) {
        // a slab context behind a step: the rows are the OWNED particles in download order, the arrays also hold the ghosts (whose
        // values stay: they are refreshed from their owners whenever something reads them)
        const uint32_t nt = c->dist.n_tot;
        std::vector<uint8_t> flags(nt ? nt : 1);
        HIPCHK(c, hipMemcpy(flags.data(), c->dist.owned.p, nt, hipMemcpyDeviceToHost));
        std::vector<uint32_t> slots;
        slots.reserve(n);
        for (uint32_t i = 0; i < nt; i++)
            if (flags[i]) slots.push_back(i);
        if (slots.size() != n) return c->fail(SPH_ERR_DEVICE, "owned-particle count mismatch");
        TmpBuf d_slots;
        HIPCHK(c, d_slots.ensure((size_t)n * 4));
        HIPCHK(c, hipMemcpyAsync(d_slots.p, slots.data(), (size_t)n * 4, hipMemcpyHostToDevice, s));
        HIPCHK(c, hipMemcpyAsync(c->scratch.p, src, bytes, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_rows_to_slots, dim3((n + 255) / 256), dim3(256), 0, s, n, r.kind, d_slots.as<uint32_t>(), (const void*)c->scratch.p, (void*)r.src);
        HIPCHK(c, hipStreamSynchronize(s));
        return SPH_OK;
    }
    HIPCHK(c, hipMemcpyAsync(c->scratch.p, src, bytes, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_from_host_order, dim3((n + 255) / 256), dim3(256), 0, s, n, r.kind,
                       c->dist.on ? (const uint32_t*)nullptr : c->orig[c->cur].as<uint32_t>(), (const void*)c->scratch.p, (void*)r.src);
    HIPCHK(c, hipStreamSynchronize(s));
    return SPH_OK;
}

extern "C" int sph_grid(const sph_ctx* c, sph_grid_info* out)
{
    if (!c || !out) return SPH_ERR_INVALID_ARGUMENT;
    out->cell_size = c->grid.cs;
    out->cells_min_x = c->grid.minx;
    out->cells_min_y = c->grid.miny;
    out->size_x = c->grid.sx;
    out->size_y = c->grid.sy;
    return SPH_OK;
}

extern "C" int sph_profile_enable(sph_ctx* c, int enable)
{
    if (!c) return SPH_ERR_INVALID_ARGUMENT;
    c->prof.mode = enable;
    return SPH_OK;
}
extern "C" int sph_profile_reset(sph_ctx* c)
{
    if (!c) return SPH_ERR_INVALID_ARGUMENT;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    c->prof.reset();
    return SPH_OK;
}
extern "C" int sph_profile_get(sph_ctx* c, sph_kernel_time* out, int capacity, int* n_out)
{
    if (!c || !n_out) return SPH_ERR_INVALID_ARGUMENT;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    c->prof.collect();
    int k = 0;
    for (auto& r : c->prof.recs) {
        if (out && k < capacity) {
            memset(&out[k], 0, sizeof(out[k]));
            strncpy(out[k].name, r.name.c_str(), sizeof(out[k].name) - 1);
            out[k].launches = r.launches;
            out[k].total_ms = r.total_ms;
            // a launch "did work" if it took more than a quarter of the kernel's 90th-percentile duration (a launch that returns at
            // once behind a stop decision takes 2-4 us).  Not of the maximum: one hiccup of 100 us among 400 launches of 17 us
            // made every ordinary launch "idle" and the kernel's working average that of its outliers
            float ref = 0.f;
            if (!r.samples.empty()) {
                std::vector<float> sorted(r.samples);
                const size_t k = (sorted.size() - 1) * 9 / 10;
                std::nth_element(sorted.begin(), sorted.begin() + k, sorted.end());
                ref = sorted[k];
            }
            for (float v : r.samples)
                if (v > 0.25f * ref) {
                    out[k].working_launches++;
                    out[k].working_ms += v;
                }
        }
        k++;
    }
    *n_out = k < capacity ? k : capacity;
    return SPH_OK;
}

__global__ void k_empty() {}

extern "C" int sph_profile_event_overhead(sph_ctx* c, double* microseconds)
{
    if (!c || !microseconds) return SPH_ERR_INVALID_ARGUMENT;
    HIPCHK(c, hipSetDevice(c->device));
    // event pair around ONE empty kernel: T1 = o + e ; around TWO: T2 = o + 2 e  =>  o = 2 T1 - T2 is what the pair itself
    // adds (e = an empty kernel's own dispatch + execution, which a real kernel's duration already contains)
    const int reps = 200;
    std::vector<hipEvent_t> ev(2 * reps);
    for (auto& e : ev) HIPCHK(c, hipEventCreate(&e));
    double t[2] = {0, 0};
    for (int nk = 1; nk <= 2; nk++) {
        for (int k = 0; k < reps; k++) {
            HIPCHK(c, hipEventRecord(ev[2 * k], c->stream));
            for (int q = 0; q < nk; q++) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, c->stream);
            HIPCHK(c, hipEventRecord(ev[2 * k + 1], c->stream));
        }
        HIPCHK(c, hipStreamSynchronize(c->stream));
        for (int k = 0; k < reps; k++) {
            float ms = 0.f;
            HIPCHK(c, hipEventElapsedTime(&ms, ev[2 * k], ev[2 * k + 1]));
            t[nk - 1] += ms;
        }
    }
    for (auto& e : ev) (void)hipEventDestroy(e);
    const double o = (2.0 * t[0] - t[1]) / reps * 1e3;
    *microseconds = o > 0 ? o : 0;
    return SPH_OK;
}

// Calibration of the profiler's marker events: a one-wave kernel that spins for `us` microseconds of the device's constant 100 MHz
// clock.  rocprofv3 reports its duration as us + 0.44 (the launch / exit of a one-wave dispatch: profiles/r4_event_calibration.md);
// what a marker-event pair around it reads beyond that is what the pair adds to ANY kernel it brackets in the same place of the
// queue.  Profiler mode 1 launches one such kernel per step right behind the density sweep -- in the middle of the step's busy queue,
// bracketed like every other kernel -- under the name "calibration_spin10"; bench.py subtracts (its bracket - 10.44 us) from the
// sweeps' brackets.  (Alone on an idle queue the same bracket reads 3.9 us more, in the step ~2.4: the calibration must sit where
// the kernels it corrects sit.)
__global__ void k_spin_calib(uint32_t us)
{
    const uint64_t t0 = wall_clock64();   // 100 MHz
    while (wall_clock64() - t0 < (uint64_t)us * 100ull) __builtin_amdgcn_s_sleep(4);
}
void launch_profile_calibration(sph_ctx* c)
{
    if (c->prof.mode != 1 && c->prof.mode != 4) return;
    {
        ProfScope ps(&c->prof, "calibration_spin10", c->stream);
        hipLaunchKernelGGL(k_spin_calib, dim3(1), dim3(64), 0, c->stream, 10u);
    }
    if (c->prof.mode == 4) {   // ... and timed the way mode 4 times the sweeps: by the dispatch's own timestamps (must read rocprofv3's 10.44 us)
        ProfScope ps(&c->prof, "calibration_spin10_dispatch", c->stream, true);
        hipEvent_t e0, e1;
        if (c->prof.take_events(&e0, &e1)) hipExtLaunchKernelGGL(k_spin_calib, dim3(1), dim3(64), 0, c->stream, e0, e1, 0, 10u);
    }
}
extern "C" int sph_profile_dispatch_bracket(sph_ctx* c, uint32_t spin_us, int reps, double* mean_bracket_us)
{
    if (!c || !mean_bracket_us || reps < 1 || reps > 10000) return SPH_ERR_INVALID_ARGUMENT;
    HIPCHK(c, hipSetDevice(c->device));
    std::vector<hipEvent_t> ev(2 * (size_t)reps);
    for (auto& e : ev) HIPCHK(c, hipEventCreate(&e));
    for (int k = 0; k < reps; k++) {
        HIPCHK(c, hipEventRecord(ev[2 * k], c->stream));
        hipLaunchKernelGGL(k_spin_calib, dim3(1), dim3(64), 0, c->stream, spin_us);
        HIPCHK(c, hipEventRecord(ev[2 * k + 1], c->stream));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    double t = 0;
    for (int k = 0; k < reps; k++) {
        float ms = 0.f;
        HIPCHK(c, hipEventElapsedTime(&ms, ev[2 * k], ev[2 * k + 1]));
        t += ms;
    }
    for (auto& e : ev) (void)hipEventDestroy(e);
    *mean_bracket_us = t / reps * 1e3;
    return SPH_OK;
}

// The device's achievable streaming rate, measured in the same run as the sweeps it is compared with.  The guide's figure (6.29 TB/s,
// MI355X_MICROARCH.md) needs several independent 16-byte accesses in flight per lane and a grid that fills every CU several times over:
// each thread moves U float4 a block-stride apart (loads first, then stores), one tile of 256 U records per workgroup, nontemporal or
// plain.  (The grid-stride loop of rounds 1-4 -- ONE access in flight per lane, 4096 workgroups -- reached 4.6-4.8 TB/s.)
typedef float copy_v4f __attribute__((ext_vector_type(4)));
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_copy4u(const float4* __restrict__ src4, float4* __restrict__ dst4, size_t n)
{
    const copy_v4f* __restrict__ src = reinterpret_cast<const copy_v4f*>(src4);
    copy_v4f* __restrict__ dst = reinterpret_cast<copy_v4f*>(dst4);
    const size_t base = (size_t)blockIdx.x * (256u * U) + threadIdx.x;
    copy_v4f v[U];
#pragma unroll
    for (int k = 0; k < U; k++) {
        const size_t i = base + (size_t)k * 256u;
        if (i < n) v[k] = NT ? __builtin_nontemporal_load(src + i) : src[i];
    }
#pragma unroll
    for (int k = 0; k < U; k++) {
        const size_t i = base + (size_t)k * 256u;
        if (i < n) {
            if (NT) __builtin_nontemporal_store(v[k], dst + i);
            else dst[i] = v[k];
        }
    }
}
__global__ __launch_bounds__(256) void k_copy4(const float4* __restrict__ src, float4* __restrict__ dst, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

// best of the forms (first repetition of each warms up); *form (optional) names the winner: 0 grid-stride, 1 U=4, 2 U=8, 3 U=4 nt, 4 U=8 nt
static int copy_bandwidth(sph_ctx* c, uint64_t bytes, double* gb_per_s, int* form)
{
    HIPCHK(c, hipSetDevice(c->device));
    TmpBuf a, b;
    HIPCHK(c, a.ensure(bytes));
    HIPCHK(c, b.ensure(bytes));
    HIPCHK(c, hipMemsetAsync(a.p, 0, bytes, c->stream));
    hipEvent_t e0, e1;
    HIPCHK(c, hipEventCreate(&e0));
    HIPCHK(c, hipEventCreate(&e1));
    const size_t n4 = bytes / 16;
    double best = 0;
    int best_form = 0;
    for (int f = 0; f < 5; f++) {
        for (int rep = 0; rep < 5; rep++) {
            HIPCHK(c, hipEventRecord(e0, c->stream));
            const float4 *src = a.as<float4>();
            float4* dst = b.as<float4>();
            switch (f) {
            case 0: hipLaunchKernelGGL(k_copy4, dim3(256 * 16), dim3(256), 0, c->stream, src, dst, n4); break;
            case 1: hipLaunchKernelGGL((k_copy4u<4, false>), dim3((unsigned)((n4 + 1023) / 1024)), dim3(256), 0, c->stream, src, dst, n4); break;
            case 2: hipLaunchKernelGGL((k_copy4u<8, false>), dim3((unsigned)((n4 + 2047) / 2048)), dim3(256), 0, c->stream, src, dst, n4); break;
            case 3: hipLaunchKernelGGL((k_copy4u<4, true>), dim3((unsigned)((n4 + 1023) / 1024)), dim3(256), 0, c->stream, src, dst, n4); break;
            default: hipLaunchKernelGGL((k_copy4u<8, true>), dim3((unsigned)((n4 + 2047) / 2048)), dim3(256), 0, c->stream, src, dst, n4); break;
            }
            HIPCHK(c, hipEventRecord(e1, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            float ms = 0.f;
            HIPCHK(c, hipEventElapsedTime(&ms, e0, e1));
            const double g = 2.0 * (double)(n4 * 16) / (ms * 1e-3) / 1e9;
            if (rep > 0 && g > best) {
                best = g;
                best_form = f;
            }
        }
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    a.release();
    b.release();
    *gb_per_s = best;
    if (form) *form = best_form;
    return SPH_OK;
}
extern "C" int sph_profile_copy_bandwidth(sph_ctx* c, uint64_t bytes, double* gb_per_s)
{
    if (!c || !gb_per_s || bytes < 1024) return SPH_ERR_INVALID_ARGUMENT;
    return copy_bandwidth(c, bytes, gb_per_s, nullptr);
}

// list words of the last step by form (measurement hook): NL_OK 0x80000000, NL_WALL 0x40000000, NL_IDX 0x20000000 (sph_sweeps.hip)
__global__ __launch_bounds__(256) void k_list_forms(uint32_t n, const uint4* __restrict__ nl, const uint8_t* __restrict__ owned, const uint8_t* __restrict__ ring1,
                                                     unsigned long long* __restrict__ out)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    const bool on = i < n && (!owned || owned[i] || (ring1 && ring1[i]));
    const uint32_t w = on ? nl[i].w : 0u;
    const unsigned long long b_on = __ballot(on), b_ok = __ballot(on && (w & 0x80000000u)), b_ix = __ballot(on && !(w & 0x80000000u) && (w & 0x20000000u)),
                             b_wall = __ballot(on && (w & 0x40000000u));
    if ((threadIdx.x & 63) == 0 && b_on) {
        const unsigned long long c_on = __popcll(b_on), c_ok = __popcll(b_ok), c_ix = __popcll(b_ix);
        atomicAdd(&out[0], c_on);
        atomicAdd(&out[1], c_ok);
        atomicAdd(&out[2], c_ix);
        atomicAdd(&out[3], c_on - c_ok - c_ix);
        atomicAdd(&out[4], (unsigned long long)__popcll(b_wall));
    }
}

extern "C" int sph_profile_list_forms(sph_ctx* c, sph_list_forms* out)
{
    if (!c || !out) return SPH_ERR_INVALID_ARGUMENT;
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->grid_valid) return c->fail(SPH_ERR_INVALID_ARGUMENT, "no neighbour lists yet: run a step first");
    const uint32_t n = c->dist.on && c->dist.have_flags ? c->dist.n_tot : (uint32_t)c->n;
    TmpBuf acc;
    HIPCHK(c, acc.ensure(5 * sizeof(unsigned long long)));
    HIPCHK(c, hipMemsetAsync(acc.p, 0, 5 * sizeof(unsigned long long), c->stream));
    const bool flags = c->dist.on && c->dist.have_flags;
    if (n)
        hipLaunchKernelGGL(k_list_forms, dim3((n + 255) / 256), dim3(256), 0, c->stream, n, c->nl.as<uint4>(), flags ? c->dist.owned.as<uint8_t>() : nullptr,
                           flags ? c->dist.ring1.as<uint8_t>() : nullptr, acc.as<unsigned long long>());
    unsigned long long h[5] = {0, 0, 0, 0, 0};
    HIPCHK(c, hipMemcpyAsync(h, acc.p, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    acc.release();
    out->n_lists = h[0];
    out->n_mask = h[1];
    out->n_index = h[2];
    out->n_walk = h[3];
    out->n_wall = h[4];
    return SPH_OK;
}

extern "C" int sph_download_neighbors(sph_ctx* c, uint32_t* offsets, uint32_t* indices, uint64_t cap, uint64_t* n_indices)
{
    if (!c) return SPH_ERR_INVALID_ARGUMENT;
    HIPCHK(c, hipSetDevice(c->device));
    const uint32_t n = (uint32_t)c->n;
    if (!c->grid_valid) return c->fail(SPH_ERR_INVALID_ARGUMENT, "no neighbour lists yet: run a step first");
    // (persistent host staging: two fresh 4 n-byte vectors per call were ~3 ms of page faults and zero-filling at 4 M particles)
    std::vector<uint32_t>&cnt = c->export_cnt, &off = c->export_off;
    if (cnt.size() < n) cnt.resize(n);
    if (off.size() < (size_t)n + 1) off.resize((size_t)n + 1);
    int rc = SPH_OK;
    hipStream_t s = c->stream;
    if (c->dist.on) {
        // Slab context: one row per OWNED particle, in the order of sph_download(SPH_F_PARTICLE_ID); the indices are global particle
        // ids (an owned particle's neighbours are all among owned + ghosts, and the ghost records carry their ids).
        if (!c->dist.have_flags) return c->fail(SPH_ERR_INVALID_ARGUMENT, "no neighbour lists yet: run a step first");
        const uint32_t nt = c->dist.n_tot;
        const bool ext = c->lists_after;   // (as on a plain context: the extended lists of the ADVECTED positions)
        std::vector<uint8_t> flags(nt);
        if (nt) HIPCHK(c, hipMemcpy(flags.data(), c->dist.owned.p, nt, hipMemcpyDeviceToHost));
        std::vector<uint32_t> rowmap(nt);
        uint32_t w = 0;
        for (uint32_t i = 0; i < nt; i++) rowmap[i] = flags[i] ? w++ : 0xffffffffu;
        if (w != n) return c->fail(SPH_ERR_DEVICE, "owned-particle count mismatch");
        if (!ext) {
            rc = sph_download(c, SPH_F_NEIGHBOR_COUNT, cnt.data(), (uint64_t)n * 4);
            if (rc) return rc;
        } else if (nt) {   // counts of the extended lists: low 16 bits of the list words, owned slots in slot order
            std::vector<uint4> words(nt);
            HIPCHK(c, hipMemcpy(words.data(), c->nl_ext.p, (size_t)nt * sizeof(uint4), hipMemcpyDeviceToHost));
            for (uint32_t i = 0; i < nt; i++)
                if (flags[i]) cnt[rowmap[i]] = words[i].w & 0xffffu;
        }
        uint64_t tot = 0;
        for (uint32_t i = 0; i < n; i++) {
            off[i] = (uint32_t)tot;
            tot += cnt[i];
        }
        off[n] = (uint32_t)tot;
        if (n_indices) *n_indices = tot;
        if (offsets) memcpy(offsets, off.data(), ((size_t)n + 1) * 4);
        if (!indices) return SPH_OK;
        if (cap < tot) return c->fail(SPH_ERR_INVALID_ARGUMENT, "indices buffer too small");
        if (tot == 0) return SPH_OK;
        TmpBuf d_off, d_idx, d_row;
        HIPCHK(c, d_off.ensure(((size_t)n + 1) * 4));
        HIPCHK(c, d_idx.ensure((size_t)tot * 4));
        HIPCHK(c, d_row.ensure((size_t)nt * 4));
        HIPCHK(c, hipMemcpyAsync(d_off.p, off.data(), ((size_t)n + 1) * 4, hipMemcpyHostToDevice, s));
        HIPCHK(c, hipMemcpyAsync(d_row.p, rowmap.data(), (size_t)nt * 4, hipMemcpyHostToDevice, s));
        if (!ext)
            hipLaunchKernelGGL(k_fill_neighbors, dim3((nt + 255) / 256), dim3(256), 0, s, nt, c->fgrid,
                               TileP{c->tile_ts, c->tile_tsx, c->tile_tsy, c->tile_h.as<uint32_t>(), 0.f}, c->cell_start.as<uint32_t>(), c->cxy.as<uint32_t>(),
                               c->orig[c->cur].as<uint32_t>(), c->pm[c->pcur ^ 1].as<float4>(), d_off.as<uint32_t>(), d_idx.as<uint32_t>(), 2.f,
                               d_row.as<uint32_t>());
        else   // cells (cxy) of the pre-step positions, geometry of the advected ones (pm[pcur]: the ghosts' were refreshed), ranges widened by the slack
            hipLaunchKernelGGL(k_fill_neighbors, dim3((nt + 255) / 256), dim3(256), 0, s, nt, c->fgrid,
                               TileP{c->tile_ts, c->tile_tsx, c->tile_tsy, c->tile_h_ext.as<uint32_t>(), c->lists_after_slack}, c->cell_start.as<uint32_t>(),
                               c->cxy.as<uint32_t>(), c->orig[c->cur].as<uint32_t>(), c->pm[c->pcur].as<float4>(), d_off.as<uint32_t>(), d_idx.as<uint32_t>(),
                               c->lists_after_k, d_row.as<uint32_t>());
        HIPCHK(c, hipMemcpyAsync(indices, d_idx.p, (size_t)tot * 4, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipStreamSynchronize(s));
        d_off.release();
        d_idx.release();
        d_row.release();
        return SPH_OK;
    }
    // level_estimation_after_advection: `self.neighs` was rebuilt at the end of the step (simulation.rs:2678-2689) -- the cache
    // then holds the EXTENDED lists of the ADVECTED positions, and that is what the host's partner searches iterate
    const bool ext = c->lists_after;
    if (!ext) {
        rc = sph_download(c, SPH_F_NEIGHBOR_COUNT, cnt.data(), (uint64_t)n * 4);
        if (rc) return rc;
    } else if (n) {
        HIPCHK(c, c->scratch.ensure((size_t)n * 4));
        hipLaunchKernelGGL(k_ext_counts, dim3((n + 255) / 256), dim3(256), 0, s, n, c->nl_ext.as<uint4>(), c->orig[c->cur].as<uint32_t>(), c->scratch.as<uint32_t>());
        HIPCHK(c, hipMemcpyAsync(cnt.data(), c->scratch.p, (size_t)n * 4, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipStreamSynchronize(s));
    }
    uint64_t tot = 0;
    for (uint32_t i = 0; i < n; i++) {
        off[i] = (uint32_t)tot;
        tot += cnt[i];
    }
    off[n] = (uint32_t)tot;
    if (n_indices) *n_indices = tot;
    if (offsets) memcpy(offsets, off.data(), ((size_t)n + 1) * 4);
    if (!indices) return SPH_OK;
    if (cap < tot) return c->fail(SPH_ERR_INVALID_ARGUMENT, "indices buffer too small");
    if (tot == 0) return SPH_OK;
    // The lists are those of the positions the last step STARTED from (NeighborhoodCache after a
    // step): pm[pcur ^ 1] still holds that sorted pre-step snapshot.
    DevBuf &d_off = c->export_d_off, &d_idx = c->export_d_idx;   // (kept across calls: an adaptive host exports every step)
    HIPCHK(c, d_off.ensure(((size_t)n + 1) * 4));
    HIPCHK(c, d_idx.ensure((size_t)tot * 4));
    HIPCHK(c, hipMemcpyAsync(d_off.p, off.data(), ((size_t)n + 1) * 4, hipMemcpyHostToDevice, s));
    if (!ext)
        hipLaunchKernelGGL(k_fill_neighbors, dim3((n + 255) / 256), dim3(256), 0, s, n, c->fgrid,
                           TileP{c->tile_ts, c->tile_tsx, c->tile_tsy, c->tile_h.as<uint32_t>(), 0.f}, c->cell_start.as<uint32_t>(), c->cxy.as<uint32_t>(),
                           c->orig[c->cur].as<uint32_t>(), c->pm[c->pcur ^ 1].as<float4>(), d_off.as<uint32_t>(), d_idx.as<uint32_t>(), 2.f, (const uint32_t*)nullptr);
    else   // cells (cxy) of the pre-step positions, geometry of the advected ones (pm[pcur]), ranges widened by the slack
        hipLaunchKernelGGL(k_fill_neighbors, dim3((n + 255) / 256), dim3(256), 0, s, n, c->fgrid,
                           TileP{c->tile_ts, c->tile_tsx, c->tile_tsy, c->tile_h_ext.as<uint32_t>(), c->lists_after_slack}, c->cell_start.as<uint32_t>(),
                           c->cxy.as<uint32_t>(), c->orig[c->cur].as<uint32_t>(), c->pm[c->pcur].as<float4>(), d_off.as<uint32_t>(), d_idx.as<uint32_t>(),
                           c->lists_after_k, (const uint32_t*)nullptr);
    HIPCHK(c, hipMemcpyAsync(indices, d_idx.p, (size_t)tot * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    return SPH_OK;
}


// ------------------------------------------------------------------------------------------------
// launch wrappers used by the step driver (sph_step.hip)
// ------------------------------------------------------------------------------------------------
void launch_header(sph_ctx* c, uint32_t n, float rest_density, int from_mass /* 0 keep h, 1 from mass, 2 swap with h2_next */, HeaderOut* out_dev,
                   const uint8_t* owned)
{
    hipStream_t s = c->stream;
    ProfScope ps(&c->prof, "header", s);
    int nb = (int)((n + 255) / 256);
    if (nb > HDR_BLOCKS) nb = HDR_BLOCKS;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(k_header, dim3(nb), dim3(256), 0, s, c->pm[c->pcur].as<float4>(), c->vel[c->cur].as<float2>(), n, rest_density, from_mass,
                       c->h2n[c->cur].as<float>(), c->hdr_partials.as<HeaderOut>(), owned);
    c->publish_seq++;
    if (c->publish_seq == 0u) c->publish_seq = 1u;
    const bool mapped = out_dev == c->hdr_host_dev;
    hipLaunchKernelGGL(k_header_final, dim3(1), dim3(256), 0, s, c->hdr_partials.as<HeaderOut>(), nb, out_dev, mapped ? c->publish_seq : 0u);
    if (mapped) {   // nothing may be queued between this launch and the wait that uses the hint
        c->hint_word = &((volatile HeaderOut*)c->hdr_host)->pad;
        c->hint_seq = c->publish_seq;
    }
}

void launch_header_ahead(sph_ctx* c, uint32_t nblocks, HeaderOut* out_dev, bool publish)
{
    ProfScope ps(&c->prof, "header_ahead", c->stream);
    uint32_t seq = 0;
    if (publish) {   // the launch_publish that follows has nothing left to do
        c->publish_seq++;
        if (c->publish_seq == 0u) c->publish_seq = 1u;
        seq = c->publish_seq;
        c->publish_folded = true;
    }
    hipLaunchKernelGGL(k_header_ahead, dim3(1), dim3(1024), 0, c->stream, c->hdr_ahead_partials.as<HeaderOut>(), nblocks, c->ctrl.as<SolverCtrl>(), out_dev,
                       c->status.as<DeviceStatus>(), publish ? c->ctrl_host_dev : (SolverCtrl*)nullptr, publish ? c->status_host_dev : (DeviceStatus*)nullptr, seq);
}

void launch_publish(sph_ctx* c)
{
    if (c->publish_folded) {   // k_header_ahead, queued just before, publishes
        c->publish_folded = false;
        return;
    }
    c->publish_seq++;
    if (c->publish_seq == 0u) c->publish_seq = 1u;
    hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, c->stream, c->ctrl.as<SolverCtrl>(), c->status.as<DeviceStatus>(), c->ctrl_host_dev,
                       c->status_host_dev, c->publish_seq);
}

void launch_check_neighborhood(sph_ctx* c, const SweepArgs& a)
{
    ProfScope ps(&c->prof, "check_neighborhood", c->stream);
    hipLaunchKernelGGL(k_check_neighborhood, dim3((a.n + 255) / 256), dim3(256), 0, c->stream, a.n, a.pm, a.ncount, a.orig, a.status, a.owned);
}
