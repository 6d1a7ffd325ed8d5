// libsph_hip.so: the C ABI of include/sph_ffi.h over the HIP kernels of this directory.
//
// One context = one FluidSimulation's device-resident state on ONE MI355X (one process per GPU).
// The persistent particle SoA stays on the device in cell-sorted order between steps; `orig`
// maps each sorted slot back to the host's particle index, so uploads/downloads speak the
// reference's indices.
//
// sph_step sequences the sweeps exactly like single_step_without_adaptivity
// (/root/reference/src/simulation/simulation.rs:1980-2730); the citations sit next to each call.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "sph_internal.hpp"
#include "sph_lambda.hpp"

// ------------------------------------------------------------------------------------------------
// profiler
// ------------------------------------------------------------------------------------------------
int Profiler::find(const char* name)
{
    for (size_t i = 0; i < recs.size(); i++)
        if (recs[i].name == name) return (int)i;
    recs.push_back(Rec{name, 0, 0});
    return (int)recs.size() - 1;
}
bool Profiler::wants(const char* name) const
{
    // mode 2: the density kernel only, on every 8th step (a timing-enabled event record forces a command
    // flush; sampling keeps the perturbation of the timed region below 1 %)
    return mode == 1 || (mode == 2 && (step_index & 7u) == 0u && strncmp(name, "density", 7) == 0);
}
hipEvent_t Profiler::get_event()
{
    if (!pool.empty()) {
        hipEvent_t e = pool.back();
        pool.pop_back();
        return e;
    }
    hipEvent_t e;
    hipEventCreate(&e);
    return e;
}
void Profiler::begin(const char* name, hipStream_t s)
{
    cur = find(name);
    cur_a = get_event();
    hipEventRecord(cur_a, s);
}
void Profiler::end(hipStream_t s)
{
    hipEvent_t b = get_event();
    hipEventRecord(b, s);
    pending.push_back(Pending{cur, cur_a, b});
    cur = -1;
}
void Profiler::collect()
{
    for (auto& p : pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            recs[p.rec].launches++;
            recs[p.rec].total_ms += ms;
        }
        pool.push_back(p.a);
        pool.push_back(p.b);
    }
    pending.clear();
}
void Profiler::reset()
{
    collect();
    recs.clear();
}
Profiler::~Profiler()
{
    for (auto& p : pending) {
        hipEventDestroy(p.a);
        hipEventDestroy(p.b);
    }
    for (auto e : pool) hipEventDestroy(e);
}

// ------------------------------------------------------------------------------------------------
// small kernels owned by this TU
// ------------------------------------------------------------------------------------------------
#define HDR_BLOCKS 256

// h_next_from_mass (simulation.rs:1865-1871) + the per-step scalars: particle bounding box (CellGrid,
// neighborhood_search.rs:261-275), h_max/h_min, CFL term min_i (2h_i)^2 / (|v_i|^2 + 0.01)
// (simulation.rs:2182-2189)
__global__ __launch_bounds__(256) void k_header(float4* __restrict__ pm, const float2* __restrict__ vel, uint32_t n, float rest_density,
                                                 int from_mass, HeaderOut* __restrict__ partials)
{
    const float INF = __uint_as_float(0x7f800000u);
    float mnx = INF, mny = INF, mxx = -INF, mxy = -INF, hmx = 0.f, hmn = INF, cfl = INF;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        float4 p = pm[i];
        if (from_mass) {
            p.w = h_from_mass(p.z, rest_density);
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

__global__ __launch_bounds__(256) void k_header_final(const HeaderOut* __restrict__ partials, int nparts, HeaderOut* __restrict__ out)
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
        *out = r;
    }
}

__global__ __launch_bounds__(256) void k_pack_upload(uint32_t n, const float* __restrict__ mass, const float2* __restrict__ pos,
                                                      const float2* __restrict__ velin, float4* __restrict__ pm, float2* __restrict__ vel,
                                                      uint32_t* __restrict__ orig, float* __restrict__ lvl, float* __restrict__ lvlold)
{
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float2 p = pos[i];
    pm[i] = make_float4(p.x, p.y, mass[i], 0.f);
    vel[i] = velin[i];
    orig[i] = i;
    lvl[i] = __uint_as_float(0x7fc00000u);  // LevelEstimationState::FluidInterior
    lvlold[i] = 0.f;
}

// copy the Jacobi control block and the error word into mapped pinned host memory (one lane); the host
// busy-polls an event recorded right behind this kernel
__global__ void k_publish(const SolverCtrl* __restrict__ ctrl, const DeviceStatus* __restrict__ status, SolverCtrl* __restrict__ h_ctrl,
                          DeviceStatus* __restrict__ h_status)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        *h_ctrl = *ctrl;
        *h_status = *status;
    }
}

enum { G_F32 = 0, G_F32X2 = 1, G_PM_X = 2, G_PM_M = 3, G_PM_H = 4, G_U32 = 5, G_H2NEXT = 6 };

// dst[orig[i]] = field[i]: back to host particle order
__global__ __launch_bounds__(256) void k_to_host_order(uint32_t n, int kind, const uint32_t* __restrict__ orig, const void* __restrict__ src,
                                                        void* __restrict__ dst)
{
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint32_t o = orig[i];
    switch (kind) {
    case G_F32: ((float*)dst)[o] = ((const float*)src)[i]; break;
    case G_U32: ((uint32_t*)dst)[o] = ((const uint32_t*)src)[i]; break;
    case G_F32X2: ((float2*)dst)[o] = ((const float2*)src)[i]; break;
    case G_PM_X: { float4 p = ((const float4*)src)[i]; ((float2*)dst)[o] = make_float2(p.x, p.y); } break;
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
    uint32_t o = orig[i];
    switch (kind) {
    case G_F32: ((float*)dst)[i] = ((const float*)src)[o]; break;
    case G_F32X2: ((float2*)dst)[i] = ((const float2*)src)[o]; break;
    case G_PM_X: { float2 p = ((const float2*)src)[o]; float4 q = ((float4*)dst)[i]; q.x = p.x; q.y = p.y; ((float4*)dst)[i] = q; } break;
    case G_PM_M: { float4 q = ((float4*)dst)[i]; q.z = ((const float*)src)[o]; ((float4*)dst)[i] = q; } break;
    case G_PM_H: { float4 q = ((float4*)dst)[i]; q.w = ((const float*)src)[o]; ((float4*)dst)[i] = q; } break;
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

// CSR export of the neighbour lists (NeighborhoodCache) in host particle order
__global__ __launch_bounds__(256) void k_fill_neighbors(uint32_t n, GridP g, const uint32_t* __restrict__ cell_start,
                                                         const uint32_t* __restrict__ cxy, const uint32_t* __restrict__ orig,
                                                         const float4* __restrict__ pm, const uint32_t* __restrict__ offsets_host,
                                                         uint32_t* __restrict__ indices)
{
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float4 Ai = pm[i];
    const uint32_t c = cxy[i];
    const int cx = c & 0xffffu, cy = c >> 16;
    uint32_t w = offsets_host[orig[i]];
    for (int dy = -1; dy <= 1; dy++) {
        int yy = cy + dy;
        if (yy < 0 || yy >= g.sy) continue;
        uint32_t b = cell_start[(uint32_t)yy * g.sx + max(cx - 1, 0)];
        uint32_t e = cell_start[(uint32_t)yy * g.sx + min(cx + 2, g.sx)];
        for (uint32_t j = b; j < e; j++) {
            const float4 Aj = pm[j];
            const float dx = Ai.x - Aj.x, dyy = Ai.y - Aj.y;
            const float r2 = dx * dx + dyy * dyy;
            const float s = ((Ai.w + Aj.w) * 0.5f) * 2.f;
            if (r2 < s * s) indices[w++] = orig[j];
        }
    }
}

// check_correct_neighborhood (simulation.rs:1810-1863) against the O(N^2) definition: the sweeps only
// ever visit pairs that satisfy the predicate, so equal COUNTS imply equal sets.
__global__ __launch_bounds__(256) void k_check_neighborhood(uint32_t n, const float4* __restrict__ pm, const uint32_t* __restrict__ ncount,
                                                             const uint32_t* __restrict__ orig, DeviceStatus* status)
{
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
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
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    hipError_t ensure(size_t need)
    {
        if (need <= bytes) return hipSuccess;
        if (p) hipFree(p);
        p = nullptr;
        bytes = 0;
        size_t grow = need + need / 4 + 256;
        hipError_t e = hipMalloc(&p, grow);
        if (e == hipSuccess) bytes = grow;
        return e;
    }
    void release()
    {
        if (p) hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    template <class T>
    T* as() const { return (T*)p; }
};

struct sph_ctx {
    int device = 0;
    uint64_t cap = 0, n = 0;
    hipStream_t stream = nullptr;
    int n_planes = 0;
    PlaneP planes_h[SPH_MAX_PLANES];
    float time = 0.f;
    uint64_t step_number = 0;
    std::string err;
    Profiler prof;
    int exact = 0;

    // persistent SoA (ping-pong across the per-step reorder)
    DevBuf pm[2], vel[2], orig[2], lvl[2], lvlold[2];
    int cur = 0;   // which of the (vel, orig, lvl, lvlold) ping-pong set is live
    int pcur = 0;  // which pm buffer is live; the other one holds the sorted PRE-step positions after a step
    DevBuf vel_tmp;
    // per-step
    DevBuf key[2], val[2], sort_scratch, cxy, cell_start, nl, nl_ok, mrho, pt0, pt1;
    bool uniform_h = false;
    float h_uniform = 0.f;
    DevBuf rho, lam_sum, lam_grad, constf, aii, src, p0, p1, pacc, dens_err, stat, ncount;
    DevBuf planes_d, lam_lut, dlam_lut, hdr_partials, hdr_out, ctrl, status, n_tiles, red_partials, scratch;
    // mapped pinned host memory: written by kernels directly (no D2H copy launches)
    HeaderOut* hdr_host = nullptr;
    SolverCtrl* ctrl_host = nullptr;
    DeviceStatus* status_host = nullptr;
    HeaderOut* hdr_host_dev = nullptr;
    SolverCtrl* ctrl_host_dev = nullptr;
    DeviceStatus* status_host_dev = nullptr;
    hipEvent_t ev_sync = nullptr;

    GridP grid{};
    bool grid_valid = false;
    uint32_t pressure_cur = 0;
    uint32_t last_div_iters = 2, last_dens_iters = 2;
    hipEvent_t ev[8];

    int fail(int code, const char* fmt, ...)
    {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        err = buf;
        return code;
    }
};

#define HIPCHK(ctx, call)                                                                                  \
    do {                                                                                                   \
        hipError_t e_ = (call);                                                                            \
        if (e_ != hipSuccess) return (ctx)->fail(SPH_ERR_DEVICE, "%s failed: %s", #call, hipGetErrorString(e_)); \
    } while (0)

static int alloc_particle_buffers(sph_ctx* c)
{
    const size_t n = c->cap ? c->cap : 1;
    for (int k = 0; k < 2; k++) {
        HIPCHK(c, c->pm[k].ensure(n * sizeof(float4)));
        HIPCHK(c, c->vel[k].ensure(n * sizeof(float2)));
        HIPCHK(c, c->orig[k].ensure(n * sizeof(uint32_t)));
        HIPCHK(c, c->lvl[k].ensure(n * sizeof(float)));
        HIPCHK(c, c->lvlold[k].ensure(n * sizeof(float)));
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
    HIPCHK(c, c->pacc.ensure(n * sizeof(float2)));
    HIPCHK(c, c->scratch.ensure(n * sizeof(float4)));
    HIPCHK(c, c->nl.ensure(sweep_list_bytes((uint32_t)n)));
    HIPCHK(c, c->nl_ok.ensure(n));
    HIPCHK(c, c->red_partials.ensure(sizeof(SolverPartial) * (size_t)solver_reduce_blocks((uint32_t)n)));
    return SPH_OK;
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
    for (int k = 0; k < n_planes; k++) c->planes_h[k] = PlaneP{planes[k].dir_x, planes[k].dir_y, planes[k].delta};
    const char* ex = getenv("SPH_HIP_EXACT");
    c->exact = (ex && ex[0] == '1') ? 1 : 0;
    auto bail = [&](int code) {
        sph_destroy(c);
        return code;
    };
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) return bail(SPH_ERR_DEVICE);
    for (auto& e : c->ev)
        if (hipEventCreate(&e) != hipSuccess) return bail(SPH_ERR_DEVICE);
    if (alloc_particle_buffers(c) != SPH_OK) return bail(SPH_ERR_DEVICE);
    bool ok = c->planes_d.ensure(sizeof(PlaneP) * SPH_MAX_PLANES) == hipSuccess && c->lam_lut.ensure(10001 * 4) == hipSuccess &&
              c->dlam_lut.ensure(10001 * 4) == hipSuccess && c->hdr_partials.ensure(sizeof(HeaderOut) * HDR_BLOCKS) == hipSuccess &&
              c->hdr_out.ensure(sizeof(HeaderOut)) == hipSuccess && c->ctrl.ensure(sizeof(SolverCtrl)) == hipSuccess &&
              c->status.ensure(sizeof(DeviceStatus)) == hipSuccess && c->n_tiles.ensure(16) == hipSuccess;
    if (!ok) return bail(SPH_ERR_DEVICE);
    if (hipHostMalloc((void**)&c->hdr_host, sizeof(HeaderOut), hipHostMallocMapped) != hipSuccess) return bail(SPH_ERR_DEVICE);
    if (hipHostMalloc((void**)&c->ctrl_host, sizeof(SolverCtrl), hipHostMallocMapped) != hipSuccess) return bail(SPH_ERR_DEVICE);
    if (hipHostMalloc((void**)&c->status_host, sizeof(DeviceStatus), hipHostMallocMapped) != hipSuccess) return bail(SPH_ERR_DEVICE);
    if (hipHostGetDevicePointer((void**)&c->hdr_host_dev, c->hdr_host, 0) != hipSuccess) return bail(SPH_ERR_DEVICE);
    if (hipHostGetDevicePointer((void**)&c->ctrl_host_dev, c->ctrl_host, 0) != hipSuccess) return bail(SPH_ERR_DEVICE);
    if (hipHostGetDevicePointer((void**)&c->status_host_dev, c->status_host, 0) != hipSuccess) return bail(SPH_ERR_DEVICE);
    if (hipEventCreateWithFlags(&c->ev_sync, hipEventDisableTiming) != hipSuccess) return bail(SPH_ERR_DEVICE);
    memset(c->status_host, 0, sizeof(DeviceStatus));
    // BoundaryWinchenbach2020::new (boundary_winchenbach2020.rs:33-36)
    std::vector<float> lam, dlam;
    sph_lambda::build_luts(lam, dlam);
    hipMemcpy(c->lam_lut.p, lam.data(), lam.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(c->dlam_lut.p, dlam.data(), dlam.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(c->planes_d.p, c->planes_h, sizeof(PlaneP) * SPH_MAX_PLANES, hipMemcpyHostToDevice);
    hipMemset(c->status.p, 0, sizeof(DeviceStatus));
    hipMemset(c->ctrl.p, 0, sizeof(SolverCtrl));
    hipDeviceSynchronize();
    *out = c;
    return SPH_OK;
}

extern "C" void sph_destroy(sph_ctx* c)
{
    if (!c) return;
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    DevBuf* all[] = {&c->pm[0], &c->pm[1], &c->vel[0], &c->vel[1], &c->orig[0], &c->orig[1], &c->lvl[0], &c->lvl[1], &c->lvlold[0],
                     &c->lvlold[1], &c->vel_tmp, &c->key[0], &c->key[1], &c->val[0], &c->val[1], &c->sort_scratch, &c->cxy, &c->cell_start,
                     &c->nl, &c->nl_ok, &c->mrho, &c->pt0, &c->pt1, &c->rho, &c->lam_sum, &c->lam_grad, &c->constf, &c->aii, &c->src, &c->p0, &c->p1, &c->pacc, &c->dens_err,
                     &c->stat, &c->ncount, &c->planes_d, &c->lam_lut, &c->dlam_lut, &c->hdr_partials, &c->hdr_out, &c->ctrl, &c->status,
                     &c->n_tiles, &c->red_partials, &c->scratch};
    for (auto b : all) b->release();
    if (c->hdr_host) hipHostFree(c->hdr_host);
    if (c->ctrl_host) hipHostFree(c->ctrl_host);
    if (c->status_host) hipHostFree(c->status_host);
    for (auto& e : c->ev)
        if (e) hipEventDestroy(e);
    if (c->ev_sync) hipEventDestroy(c->ev_sync);
    if (c->stream) hipStreamDestroy(c->stream);
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

extern "C" int sph_upload(sph_ctx* c, uint64_t n, const float* mass, const float* pos, const float* vel)
{
    if (!c || (n && (!mass || !pos || !vel))) return SPH_ERR_INVALID_ARGUMENT;
    if (n > c->cap) return c->fail(SPH_ERR_CAPACITY, "n=%llu exceeds capacity %llu", (unsigned long long)n, (unsigned long long)c->cap);
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    c->n = n;
    c->cur = 0;
    c->pcur = 0;
    c->grid_valid = false;
    if (n == 0) return SPH_OK;
    // stage host arrays through scratch buffers: mass -> key[1], pos -> scratch, vel -> vel_tmp
    HIPCHK(c, hipMemcpyAsync(c->key[1].p, mass, n * sizeof(float), hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(c->scratch.p, pos, n * sizeof(float2), hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(c->vel_tmp.p, vel, n * sizeof(float2), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_pack_upload, dim3((n + 255) / 256), dim3(256), 0, s, (uint32_t)n, c->key[1].as<float>(), c->scratch.as<float2>(),
                       c->vel_tmp.as<float2>(), c->pm[0].as<float4>(), c->vel[0].as<float2>(), c->orig[0].as<uint32_t>(),
                       c->lvl[0].as<float>(), c->lvlold[0].as<float>());
    DevBuf* zero[] = {&c->rho, &c->lam_sum, &c->constf, &c->aii, &c->src, &c->p0, &c->p1, &c->dens_err, &c->ncount};
    for (auto b : zero) HIPCHK(c, hipMemsetAsync(b->p, 0, n * sizeof(float), s));
    HIPCHK(c, hipMemsetAsync(c->lam_grad.p, 0, n * sizeof(float2), s));
    HIPCHK(c, hipMemsetAsync(c->pacc.p, 0, n * sizeof(float2), s));
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
    case SPH_F_PRESSURE_ACCEL: *r = {G_F32X2, c->pacc.p, 8, false}; return true;
    case SPH_F_DENSITY: *r = {G_F32, c->rho.p, 4, false}; return true;
    case SPH_F_PPE_SOURCE_TERM: *r = {G_F32, c->src.p, 4, false}; return true;
    case SPH_F_PRESSURE: *r = {G_F32, c->pressure_cur ? c->p1.p : c->p0.p, 4, false}; return true;
    case SPH_F_AII: *r = {G_F32, c->aii.p, 4, false}; return true;
    case SPH_F_DENSITY_ERROR: *r = {G_F32, c->dens_err.p, 4, false}; return true;
    case SPH_F_H2: *r = {G_PM_H, c->pm[c->pcur].p, 4, true}; return true;
    case SPH_F_H2_NEXT: *r = {G_H2NEXT, c->pm[c->pcur].p, 4, false}; return true;
    case SPH_F_CONSTANT_FIELD: *r = {G_F32, c->constf.p, 4, false}; return true;
    case SPH_F_NEIGHBOR_COUNT: *r = {G_U32, c->ncount.p, 4, false}; return true;
    case SPH_F_LEVEL_ESTIMATION: *r = {G_F32, c->lvl[k].p, 4, true}; return true;
    case SPH_F_LEVEL_OLD: *r = {G_F32, c->lvlold[k].p, 4, true}; return true;
    case SPH_F_LAMBDA_SUM: *r = {G_F32, c->lam_sum.p, 4, false}; return true;
    case SPH_F_LAMBDA_GRAD_SUM: *r = {G_F32X2, c->lam_grad.p, 8, false}; return true;
    default: return false;
    }
}

extern "C" int sph_download(sph_ctx* c, int field, void* dst, uint64_t bytes)
{
    if (!c || !dst) return SPH_ERR_INVALID_ARGUMENT;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    const uint32_t n = (uint32_t)c->n;
    const int k = c->cur;
    if (field == SPH_F_CELL_INDEX) {
        if (bytes != (uint64_t)n * 4) return c->fail(SPH_ERR_INVALID_ARGUMENT, "field %d: size mismatch", field);
        if (!c->grid_valid) return c->fail(SPH_ERR_INVALID_ARGUMENT, "no grid yet: run a step first");
        if (n == 0) return SPH_OK;
        // sorted cell keys of the positions the last step started from
        hipLaunchKernelGGL(k_to_host_order, dim3((n + 255) / 256), dim3(256), 0, s, n, (int)G_U32, c->orig[k].as<uint32_t>(),
                           (const void*)c->key[0].p, c->scratch.p);
        HIPCHK(c, hipMemcpyAsync(dst, c->scratch.p, bytes, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipStreamSynchronize(s));
        return SPH_OK;
    }
    if (field == SPH_F_STASH || field == SPH_F_FLAG_IS_FLUID_SURFACE || field == SPH_F_FLAG_INSUFFICIENT_NEIGHS ||
        field == SPH_F_PARTICLE_SIZE_CLASS) {
        // level-estimation outputs: not produced on the device yet -> defaults of ParticleVec
        size_t elem = field == SPH_F_STASH ? 4 : 1;
        if (bytes != (uint64_t)n * elem) return c->fail(SPH_ERR_INVALID_ARGUMENT, "field %d: size mismatch", field);
        memset(dst, field == SPH_F_PARTICLE_SIZE_CLASS ? 2 : 0, bytes);
        return SPH_OK;
    }
    FieldRef r;
    if (!field_ref(c, field, &r)) return c->fail(SPH_ERR_INVALID_ARGUMENT, "unknown field %d", field);
    if (bytes != (uint64_t)n * r.elem) return c->fail(SPH_ERR_INVALID_ARGUMENT, "field %d: size mismatch", field);
    if (n == 0) return SPH_OK;
    hipLaunchKernelGGL(k_to_host_order, dim3((n + 255) / 256), dim3(256), 0, s, n, r.kind, c->orig[k].as<uint32_t>(), r.src, c->scratch.p);
    HIPCHK(c, hipMemcpyAsync(dst, c->scratch.p, bytes, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    return SPH_OK;
}

extern "C" int sph_upload_field(sph_ctx* c, int field, const void* src, uint64_t bytes)
{
    if (!c || !src) return SPH_ERR_INVALID_ARGUMENT;
    HIPCHK(c, hipSetDevice(c->device));
    FieldRef r;
    if (!field_ref(c, field, &r) || !r.uploadable) return c->fail(SPH_ERR_INVALID_ARGUMENT, "field %d cannot be uploaded", field);
    const uint32_t n = (uint32_t)c->n;
    if (bytes != (uint64_t)n * r.elem) return c->fail(SPH_ERR_INVALID_ARGUMENT, "field %d: size mismatch", field);
    if (n == 0) return SPH_OK;
    hipStream_t s = c->stream;
    HIPCHK(c, hipMemcpyAsync(c->scratch.p, src, bytes, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_from_host_order, dim3((n + 255) / 256), dim3(256), 0, s, n, r.kind, c->orig[c->cur].as<uint32_t>(),
                       (const void*)c->scratch.p, (void*)r.src);
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
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    c->prof.reset();
    return SPH_OK;
}
extern "C" int sph_profile_get(sph_ctx* c, sph_kernel_time* out, int capacity, int* n_out)
{
    if (!c || !n_out) return SPH_ERR_INVALID_ARGUMENT;
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    c->prof.collect();
    int k = 0;
    for (auto& r : c->prof.recs) {
        if (out && k < capacity) {
            memset(&out[k], 0, sizeof(out[k]));
            strncpy(out[k].name, r.name.c_str(), sizeof(out[k].name) - 1);
            out[k].launches = r.launches;
            out[k].total_ms = r.total_ms;
        }
        k++;
    }
    *n_out = k < capacity ? k : capacity;
    return SPH_OK;
}

// ------------------------------------------------------------------------------------------------
// the step
// ------------------------------------------------------------------------------------------------
static SweepArgs make_args(sph_ctx* c, const StepP& sp)
{
    SweepArgs a{};
    const int k = c->cur;
    a.g = c->grid;
    a.sp = sp;
    a.n = (uint32_t)c->n;
    a.exact = c->exact;
    a.cell_start = c->cell_start.as<uint32_t>();
    a.orig = c->orig[k].as<uint32_t>();
    a.pm = c->pm[c->pcur].as<float4>();
    a.vel = c->vel[k].as<float2>();
    a.vel_tmp = c->vel_tmp.as<float2>();
    a.rho = c->rho.as<float>();
    a.lam_sum = c->lam_sum.as<float>();
    a.lam_grad = c->lam_grad.as<float2>();
    a.constf = c->constf.as<float>();
    a.aii = c->aii.as<float>();
    a.src = c->src.as<float>();
    a.p0 = c->p0.as<float>();
    a.p1 = c->p1.as<float>();
    a.pacc = c->pacc.as<float2>();
    a.dens_err = c->dens_err.as<float>();
    a.stat = c->stat.as<float>();
    a.ncount = c->ncount.as<uint32_t>();
    a.nl = c->nl.as<uint4>();
    a.partials = c->red_partials.as<float>();
    a.mrho = c->mrho.as<float>();
    a.pt0 = c->pt0.as<float>();
    a.pt1 = c->pt1.as<float>();
    a.uniform_h = c->uniform_h ? 1 : 0;
    a.h_uniform = c->h_uniform;
    a.planes = c->planes_d.as<PlaneP>();
    a.lam_lut = c->lam_lut.as<float>();
    a.dlam_lut = c->dlam_lut.as<float>();
    a.ctrl = c->ctrl.as<SolverCtrl>();
    a.status = c->status.as<DeviceStatus>();
    return a;
}

static const char* status_message(uint32_t code)
{
    switch (code) {
    case SPH_ERR_DENSITY_NOT_FINITE: return "assertion failed: p_density.is_finite()";
    case SPH_ERR_DENSITY_TOO_SMALL: return "assertion failed: *p_density > 0.0001";
    case SPH_ERR_AII_NOT_FINITE: return "assertion failed: (*p_aii).is_finite()";
    case SPH_ERR_AII_NEGATIVE: return "AII should not be negative!";
    case SPH_ERR_AP_NOT_FINITE: return "'!a_p.is_finite()' failed. Pressure values probably have exploded!";
    case SPH_ERR_PRESSURE_NOT_FINITE: return "'!p_pressure_next_iter.is_finite()' failed.";
    case SPH_ERR_TOO_MANY_NEIGHBORS: return "exceeded maximum allowed number of 20000 neighbors";
    case SPH_ERR_VELOCITY_NOT_FINITE: return "Assertion 'p_velocity[d].is_finite()' failed!";
    case SPH_ERR_POSITION_NOT_FINITE: return "Assertion 'p_position[d].is_finite()' failed!";
    case SPH_ERR_VISCOSITY_NOT_FINITE: return "Assertion 'viscosity_accel[d].is_finite()' failed!";
    case SPH_ERR_CHECK_NEIGHBORHOOD: return "neighbour list differs from the brute-force definition";
    default: return "device-side guard failed";
    }
}

// Wait for everything queued on the context's stream.  Busy-polls an event instead of
// hipStreamSynchronize: the blocking wait's wake-up latency (tens of microseconds, more once another
// runtime user in the process has switched the device to blocking-sync scheduling) would otherwise be
// paid three times per step.
static int wait_stream(sph_ctx* c)
{
    HIPCHK(c, hipEventRecord(c->ev_sync, c->stream));
    auto t0 = std::chrono::steady_clock::now();
    for (uint64_t spins = 0;; spins++) {
        hipError_t e = hipEventQuery(c->ev_sync);
        if (e == hipSuccess) return SPH_OK;
        if (e != hipErrorNotReady) return c->fail(SPH_ERR_DEVICE, "hipEventQuery failed: %s", hipGetErrorString(e));
        if ((spins & 0xfffu) == 0xfffu &&
            std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 60.0)
            return c->fail(SPH_ERR_DEVICE, "device did not finish the queued work within 60 s");
    }
}

// publish ctrl + status to the host and wait; returns the device error code (0 if none)
static int sync_ctrl(sph_ctx* c)
{
    hipStream_t s = c->stream;
    hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, s, c->ctrl.as<SolverCtrl>(), c->status.as<DeviceStatus>(), c->ctrl_host_dev,
                       c->status_host_dev);
    int rc = wait_stream(c);
    if (rc) return rc;
    if (c->status_host->error) {
        uint32_t code = c->status_host->error, info = c->status_host->info;
        (void)hipMemsetAsync(c->status.p, 0, sizeof(DeviceStatus), s);
        c->status_host->error = 0;
        return c->fail((int)code, "%s (particle i=%u)", status_message(code), info);
    }
    return SPH_OK;
}

// iisph_pressure_iterations (simulation.rs:1377-1516).  Iteration 0 was folded into the source-term
// sweep (closed form, see OpSource); its statistics are reduced here.  Iterations are enqueued
// speculatively up to the predicted count, followed by the FINAL pressure-acceleration sweep with its
// fused tail (v += dt a^p / integrate); every kernel checks the device-side `done` flag first, so
// iterations queued past the stop decision cost a launch and nothing else, and the final sweep only
// runs once the decision is taken.  One host sync per chunk.
static int pressure_iterations(sph_ctx* c, SweepArgs& a, float max_avg_error, int residual_density, uint32_t max_iters,
                               uint32_t predicted_iters, int tail, float4* pm_out, sph_solver_stats* st)
{
    hipStream_t s = c->stream;
    Profiler* prof = &c->prof;
    // ctrl was zeroed before the source sweep; iteration 0's stop decision:
    launch_solver_reduce(s, prof, a, 0, residual_density, max_avg_error, max_iters, c->red_partials.as<float>());
    uint32_t k = 1;
    uint32_t upto = predicted_iters > 2 ? predicted_iters : 2;  // iterations 0..upto (iters is the index of the last one)
    for (;;) {
        for (; k <= upto && k <= max_iters; k++) {
            launch_pressure_accel(s, prof, a, (int)k, 0, nullptr);
            launch_jacobi_update(s, prof, a, (int)k, residual_density);
            launch_solver_reduce(s, prof, a, (int)k, residual_density, max_avg_error, max_iters, c->red_partials.as<float>());
        }
        launch_pressure_accel(s, prof, a, -1, tail, pm_out);
        int rc = sync_ctrl(c);
        if (rc) return rc;
        if (c->ctrl_host->done) break;
        if (k > max_iters) break;  // cannot happen: iteration max_iters always sets done
        upto = k + 1;
    }
    const SolverCtrl& h = *c->ctrl_host;
    c->pressure_cur = h.cur;
    st->iters = h.iters;
    st->converged = 1;
    st->normal_count = h.normal;
    st->singular_count = h.singular;
    st->negative_count = h.negative;
    st->avg_error = h.normal > 0 ? h.sum_err / (float)h.normal : NAN;
    st->max_error = h.max_err;
    return SPH_OK;
}

static int ilog2_ceil(uint32_t v)
{
    int b = 0;
    while ((1ull << b) < (uint64_t)v) b++;
    return b;
}

// host-side timeline of one step (SPH_HIP_TRACE=1): where the CPU thread spends its time
struct HostTrace {
    bool on;
    std::chrono::steady_clock::time_point t0;
    double acc[8] = {0};
    int steps = 0;
    HostTrace() { const char* e = getenv("SPH_HIP_TRACE"); on = e && e[0] == '1'; }
    void start() { if (on) t0 = std::chrono::steady_clock::now(); }
    void mark(int k)
    {
        if (!on) return;
        auto t1 = std::chrono::steady_clock::now();
        acc[k] += std::chrono::duration<double, std::micro>(t1 - t0).count();
        t0 = t1;
    }
    void end_step()
    {
        if (!on) return;
        if (++steps % 20 == 0) {
            fprintf(stderr, "[sph trace] per step us: header+sync %.1f | sort+grid launches %.1f | sweeps launches %.1f | div solve %.1f | mid %.1f | dens solve %.1f | tail+sync %.1f\n",
                    acc[0] / steps, acc[1] / steps, acc[2] / steps, acc[3] / steps, acc[4] / steps, acc[5] / steps, acc[6] / steps);
        }
    }
};
static HostTrace g_trace;

extern "C" int sph_step(sph_ctx* c, const sph_params* p, sph_step_stats* out)
{
    if (!c || !p) return SPH_ERR_INVALID_ARGUMENT;
    g_trace.start();
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    Profiler* prof = &c->prof;
    const uint32_t n = (uint32_t)c->n;

    if (c->n_planes == 0) return c->fail(SPH_ERR_NO_BOUNDARY, "not implemented: NoBoundaryHandler::iisph_aii");
    if (p->support_length_estimation != SPH_H_FROM_MASS)
        return c->fail(SPH_ERR_UNSUPPORTED, "support_length_estimation other than FromMass is not covered yet");
    if (p->constrain_neighborhood_count) return c->fail(SPH_ERR_UNSUPPORTED, "constrain_neighborhood_count is not covered yet");
    if (p->pressure_solver_method == SPH_SOLVER_IISPH2) return c->fail(SPH_ERR_UNSUPPORTED, "IISPH2 is not covered yet");
    if (p->level_estimation_method != SPH_LEVEL_NONE)
        return c->fail(SPH_ERR_UNSUPPORTED, "level estimation on the device is not covered yet (SURVEY.md 8f rank 1)");
    if (p->check_aii) return c->fail(SPH_ERR_UNSUPPORTED, "check_aii is not covered yet");
    if (!p->level_estimation_after_advection && !p->use_extended_range_for_level_estimation)
        return c->fail(SPH_ERR_INVALID_ARGUMENT, "assertion failed: simulation_params.use_extended_range_for_level_estimation");
    if (n == 0) return c->fail(SPH_ERR_INVALID_ARGUMENT, "called `Option::unwrap()` on a `None` value (no particles)");

    // phase timing with HIP events only while profiling: a timing-enabled hipEventRecord forces a command
    // flush, and seven of them cost ~0.3 ms per step; otherwise the host wall clock fills ms_simulation_step
    const bool tev = c->prof.mode == 1;
    const auto wall0 = std::chrono::steady_clock::now();
    if (tev) hipEventRecord(c->ev[0], s);
    int k = c->cur;

    // ---- step header: h from mass (simulation.rs:1998-2003), bounding box, CFL term -------------
    {
        ProfScope ps(prof, "header", s);
        int nb = (int)((n + 255) / 256);
        if (nb > HDR_BLOCKS) nb = HDR_BLOCKS;
        hipLaunchKernelGGL(k_header, dim3(nb), dim3(256), 0, s, c->pm[c->pcur].as<float4>(), c->vel[k].as<float2>(), n, p->rest_density, 1,
                           c->hdr_partials.as<HeaderOut>());
        hipLaunchKernelGGL(k_header_final, dim3(1), dim3(256), 0, s, c->hdr_partials.as<HeaderOut>(), nb, c->hdr_host_dev);
    }
    {
        int rcw = wait_stream(c);
        if (rcw) return rcw;
    }
    const HeaderOut hdr = *c->hdr_host;
    g_trace.mark(0);
    if (!std::isfinite(hdr.min_x) || !std::isfinite(hdr.max_x) || !std::isfinite(hdr.min_y) || !std::isfinite(hdr.max_y) ||
        !(hdr.h_max > 0.f))
        return c->fail(SPH_ERR_POSITION_NOT_FINITE, "particle positions or smoothing lengths are not finite");

    // CellGrid (neighborhood_search.rs:261-275) with cell = support radius of the largest particle
    GridP g{};
    g.cs = hdr.h_max * 2.f;
    g.minx = (int)floorf(hdr.min_x / g.cs) - 1;
    g.miny = (int)floorf(hdr.min_y / g.cs) - 1;
    const long long sx = (long long)((int)floorf(hdr.max_x / g.cs) + 2) - g.minx;
    const long long sy = (long long)((int)floorf(hdr.max_y / g.cs) + 2) - g.miny;
    if (sx <= 0 || sy <= 0 || sx >= 65536 || sy >= 65536 || sx * sy >= (1ll << 31))
        return c->fail(SPH_ERR_UNSUPPORTED, "cell grid %lld x %lld is too large for this build", sx, sy);
    g.sx = (int)sx;
    g.sy = (int)sy;
    g.ncells = (uint32_t)(sx * sy);
    g.ntx = g.nty = 0;
    c->grid = g;
    c->grid_valid = true;
    HIPCHK(c, c->cell_start.ensure(((size_t)g.ncells + 1) * sizeof(uint32_t)));
    c->uniform_h = (hdr.h_min == hdr.h_max);
    c->h_uniform = hdr.h_max;

    // CFL (simulation.rs:2190-2191)
    const float cfl_dt = p->cfl_factor * sqrtf(hdr.min_cfl);
    const float dt = fminf(p->max_dt, cfl_dt);

    StepP sp{};
    sp.rest_density = p->rest_density;
    sp.viscosity = p->viscosity;
    sp.gravity = p->gravity;
    sp.jacobi_omega = p->jacobi_omega;
    sp.dt = dt;
    sp.sdf_eps = p->sdf_gradient_eps;
    sp.pull_x = p->pull_fluid_to[0];
    sp.pull_y = p->pull_fluid_to[1];
    sp.hyb_vfactor = fminf(dt * p->hybrid_dfsph_factor, 1.f);
    sp.viscosity_type = p->viscosity_type;
    sp.penalty = p->boundary_penalty_term;
    sp.opdisc = p->operator_discretization;
    sp.has_pull = p->has_pull_fluid_to;
    sp.n_planes = c->n_planes;

    // ---- neighbourhood: cell index -> radix sort -> reorder -> cell ranges -> tiles -------------
    // (replaces build_neighborhood_list + filter_down, simulation.rs:2018-2070; same neighbour set)
    launch_cell_keys(s, prof, c->pm[c->pcur].as<float4>(), n, g, c->key[0].as<uint32_t>(), c->val[0].as<uint32_t>());
    int bits = ilog2_ceil(g.ncells);
    int res = radix_sort_pairs(s, prof, c->key[0].as<uint32_t>(), c->val[0].as<uint32_t>(), c->key[1].as<uint32_t>(),
                               c->val[1].as<uint32_t>(), n, bits, c->sort_scratch.as<uint32_t>());
    if (res == 1) {  // keep the sorted keys in key[0] / val[0]
        std::swap(c->key[0], c->key[1]);
        std::swap(c->val[0], c->val[1]);
    }
    launch_reorder(s, prof, n, g, c->key[0].as<uint32_t>(), c->val[0].as<uint32_t>(), c->pm[c->pcur].as<float4>(), c->vel[k].as<float2>(),
                   c->orig[k].as<uint32_t>(), c->lvl[k].as<float>(), c->lvlold[k].as<float>(), c->pm[c->pcur ^ 1].as<float4>(),
                   c->vel[k ^ 1].as<float2>(), c->orig[k ^ 1].as<uint32_t>(), c->lvl[k ^ 1].as<float>(), c->lvlold[k ^ 1].as<float>(),
                   c->cxy.as<uint32_t>());
    c->cur = k ^ 1;
    c->pcur ^= 1;
    k = c->cur;
    launch_cell_start(s, prof, c->key[0].as<uint32_t>(), n, g.ncells, c->cell_start.as<uint32_t>());
    if (tev) hipEventRecord(c->ev[1], s);
    g_trace.mark(1);

    SweepArgs a = make_args(c, sp);
    sph_step_stats st;
    memset(&st, 0, sizeof st);
    st.n_particles = n;
    st.dt = dt;

    // ---- density + boundary lambda + neighbour count (simulation.rs:2072-2074, 2179-2180, 2204) --
    launch_density(s, prof, a);
    if (p->check_neighborhood) {
        ProfScope ps(prof, "check_neighborhood", s);
        hipLaunchKernelGGL(k_check_neighborhood, dim3((n + 255) / 256), dim3(256), 0, s, n, a.pm, a.ncount, a.orig, a.status);
    }
    // ---- constant_field + a_ii (simulation.rs:2235-2259) -----------------------------------------
    launch_aii_const(s, prof, a);

    int rc = SPH_OK;
    auto non_pressure = [&]() {  // update_velocity_with_non_pressure_accel: velocity_temp, then mem::swap
        launch_non_pressure(s, prof, a);
        std::swap(c->vel[k], c->vel_tmp);
        a.vel = c->vel[k].as<float2>();
        a.vel_tmp = c->vel_tmp.as<float2>();
    };

    auto begin_solve = [&](int kind, int residual_density) {
        hipMemsetAsync(c->ctrl.p, 0, sizeof(SolverCtrl), s);
        launch_source_term(s, prof, a, kind, residual_density);  // + Jacobi iteration 0
    };
    float4* pm_next = c->pm[c->pcur ^ 1].as<float4>();
    enum { T_NONE = 0, T_VEL = 1, T_VX = 2, T_HYBRID = 3 };  // TAIL_* of sph_sweeps.hip

    switch (p->pressure_solver_method) {
    case SPH_SOLVER_IISPH:  // simulation.rs:2389-2446
        non_pressure();
        if (tev) hipEventRecord(c->ev[4], s);
        begin_solve(1, 1);
        rc = pressure_iterations(c, a, p->iisph_max_avg_density_error, 1, p->max_iters, c->last_dens_iters, T_VX, pm_next, &st.density_solver);
        if (rc) return rc;
        if (tev) hipEventRecord(c->ev[5], s);
        break;
    case SPH_SOLVER_ONLY_DIVERGENCE:  // simulation.rs:2448-2500
        non_pressure();
        if (tev) hipEventRecord(c->ev[2], s);
        begin_solve(0, 0);
        rc = pressure_iterations(c, a, p->hybrid_dfsph_max_avg_divergence_error, 0, p->max_iters, c->last_div_iters, T_VX, pm_next, &st.div_solver);
        if (rc) return rc;
        if (tev) hipEventRecord(c->ev[3], s);
        break;
    default:  // HybridDFSPH, simulation.rs:2502-2670
        if (p->hybrid_dfsph_non_pressure_accel_before_divergence_free) non_pressure();
        if (tev) hipEventRecord(c->ev[2], s);
        begin_solve(0, 0);
        g_trace.mark(2);
        rc = pressure_iterations(c, a, p->hybrid_dfsph_max_avg_divergence_error, 0, p->max_iters, c->last_div_iters, T_VEL, nullptr, &st.div_solver);
        if (rc) return rc;
        g_trace.mark(3);
        if (tev) hipEventRecord(c->ev[3], s);
        if (!p->hybrid_dfsph_non_pressure_accel_before_divergence_free) non_pressure();
        if (tev) hipEventRecord(c->ev[4], s);
        begin_solve(p->hybrid_dfsph_density_source_term == SPH_ONLY_DENSITY ? 2 : 1, 1);
        g_trace.mark(4);
        rc = pressure_iterations(c, a, p->hybrid_dfsph_max_avg_density_error, 1, p->max_iters, c->last_dens_iters, T_HYBRID, pm_next, &st.density_solver);
        if (rc) return rc;
        g_trace.mark(5);
        if (tev) hipEventRecord(c->ev[5], s);
        break;
    }
    if (tev) hipEventRecord(c->ev[6], s);
    c->pcur ^= 1;  // integrated positions live in the other pm buffer; the old one keeps the pre-step snapshot
    if (p->viscosity_type == SPH_VISC_XSPH)  // simulation.rs:2673-2676
        return c->fail(SPH_ERR_XSPH_TODO, "not yet implemented (XSPH velocity smoothing)");

    c->last_div_iters = st.div_solver.iters;
    c->last_dens_iters = st.density_solver.iters;
    c->time += dt;  // simulation.rs:2724-2725
    c->step_number += 1;
    st.time = c->time;
    st.step_number = c->step_number;
    st.ms_simulation_step = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
    if (tev) {
        float ms = 0.f;
        (void)hipEventSynchronize(c->ev[6]);
        if (hipEventElapsedTime(&ms, c->ev[0], c->ev[6]) == hipSuccess) st.ms_simulation_step = ms;
        if (hipEventElapsedTime(&ms, c->ev[0], c->ev[1]) == hipSuccess) st.ms_neighborhood = ms;
        const bool has_div = p->pressure_solver_method == SPH_SOLVER_ONLY_DIVERGENCE || p->pressure_solver_method == SPH_SOLVER_HYBRID_DFSPH;
        const bool has_dens = p->pressure_solver_method != SPH_SOLVER_ONLY_DIVERGENCE;
        if (has_div && hipEventElapsedTime(&ms, c->ev[2], c->ev[3]) == hipSuccess) st.ms_div_solver = ms;
        if (has_dens && hipEventElapsedTime(&ms, c->ev[4], c->ev[5]) == hipSuccess) st.ms_density_solver = ms;
    }
    if (prof->mode) prof->collect();
    prof->step_index++;
    if (out) *out = st;
    g_trace.mark(6);
    g_trace.end_step();
    return SPH_OK;
}

extern "C" int sph_download_neighbors(sph_ctx* c, uint32_t* offsets, uint32_t* indices, uint64_t cap, uint64_t* n_indices)
{
    if (!c) return SPH_ERR_INVALID_ARGUMENT;
    HIPCHK(c, hipSetDevice(c->device));
    const uint32_t n = (uint32_t)c->n;
    if (!c->grid_valid) return c->fail(SPH_ERR_INVALID_ARGUMENT, "no neighbour lists yet: run a step first");
    std::vector<uint32_t> cnt(n), off((size_t)n + 1);
    int rc = sph_download(c, SPH_F_NEIGHBOR_COUNT, cnt.data(), (uint64_t)n * 4);
    if (rc) return rc;
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
    DevBuf d_off, d_idx;
    HIPCHK(c, d_off.ensure(((size_t)n + 1) * 4));
    HIPCHK(c, d_idx.ensure((size_t)tot * 4));
    hipStream_t s = c->stream;
    HIPCHK(c, hipMemcpyAsync(d_off.p, off.data(), ((size_t)n + 1) * 4, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_fill_neighbors, dim3((n + 255) / 256), dim3(256), 0, s, n, c->grid, c->cell_start.as<uint32_t>(), c->cxy.as<uint32_t>(),
                       c->orig[c->cur].as<uint32_t>(), c->pm[c->pcur ^ 1].as<float4>(), d_off.as<uint32_t>(), d_idx.as<uint32_t>());
    HIPCHK(c, hipMemcpyAsync(indices, d_idx.p, (size_t)tot * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    d_off.release();
    d_idx.release();
    return SPH_OK;
}

// ------------------------------------------------------------------------------------------------
// multi-GPU (slab decomposition over RCCL) -- see sph_comm.hip
// ------------------------------------------------------------------------------------------------
extern "C" int sph_comm_unique_id(uint8_t id_out[128])
{
    (void)id_out;
    return SPH_ERR_UNSUPPORTED;
}
extern "C" int sph_comm_init(sph_ctx* c, const uint8_t id[128], int rank, int n_ranks)
{
    (void)id;
    (void)rank;
    if (!c) return SPH_ERR_INVALID_ARGUMENT;
    if (n_ranks == 1) return SPH_OK;
    return c->fail(SPH_ERR_UNSUPPORTED, "multi-GPU slab decomposition is not built yet");
}
