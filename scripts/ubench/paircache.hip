// Micro-benchmark: would a per-step pair cache (m_j grad W_ij stored once, streamed by every iteration sweep) beat recomputing
// grad W from gathered positions?  (development aid, not part of the library)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/paircache scripts/ubench/paircache.hip && /tmp/paircache
// n = 2^20 lanes, CNT neighbour slots per lane in the layout [slot][lane] (coalesced across lanes), neighbour-like indices.
//   stream_idx : per slot   8 B cached gradient + 4 B cached index (coalesced)  + one 8 B payload gather + ~8 VALU
//   stream_mask: per slot   8 B cached gradient, index decoded from three row masks + one 8 B payload gather
//   recompute  : per slot   16 B position gather + 8 B payload gather + rsq / spline / gradient (what the sweeps do now)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_stream_idx(uint32_t n, int cnt, const float2* __restrict__ pc, const uint32_t* __restrict__ idx,
                                                     const float2* __restrict__ pay, const float* __restrict__ rho, float* __restrict__ out)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float2 ai = pay[i];
    const float inv = 1.f / rho[i];
    float sum = 0.f;
    for (int k = 0; k < cnt; k += 4) {
        float2 g[4]; uint32_t j[4]; float2 aj[4];
#pragma unroll
        for (int q = 0; q < 4; q++) { const int kk = k + q < cnt ? k + q : k; g[q] = pc[(size_t)kk * n + i]; j[q] = idx[(size_t)kk * n + i]; }
#pragma unroll
        for (int q = 0; q < 4; q++) aj[q] = pay[j[q]];
#pragma unroll
        for (int q = 0; q < 4; q++)
            if (k + q < cnt) sum += inv * ((aj[q].x - ai.x) * g[q].x + (aj[q].y - ai.y) * g[q].y);
    }
    out[i] = sum;
}

__global__ __launch_bounds__(256) void k_stream_mask(uint32_t n, const float2* __restrict__ pc, const uint4* __restrict__ nl, const uint32_t* __restrict__ rowbase,
                                                      const float2* __restrict__ pay, const float* __restrict__ rho, float* __restrict__ out)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float2 ai = pay[i];
    const float inv = 1.f / rho[i];
    const uint4 lw = nl[i];
    const uint32_t masks[3] = {lw.x, lw.y, lw.z};
    float sum = 0.f;
    uint32_t slot = 0;
#pragma unroll
    for (int dr = 0; dr < 3; dr++) {
        uint32_t mk = masks[dr];
        const uint32_t base = rowbase[dr * n + i];
        while (mk) {
            uint32_t b[4]; bool v[4];
#pragma unroll
            for (int q = 0; q < 4; q++) { v[q] = mk != 0; b[q] = v[q] ? __ffs(mk) - 1 : 0; mk &= mk - 1; }
            float2 g[4], aj[4];
#pragma unroll
            for (int q = 0; q < 4; q++) { g[q] = pc[(size_t)(slot + (v[q] ? q : 0)) * n + i]; aj[q] = pay[base + b[q]]; }
#pragma unroll
            for (int q = 0; q < 4; q++)
                if (v[q]) { sum += inv * ((aj[q].x - ai.x) * g[q].x + (aj[q].y - ai.y) * g[q].y); slot++; }
        }
    }
    out[i] = sum;
}

__global__ __launch_bounds__(256) void k_recompute(uint32_t n, const float4* __restrict__ pm, const uint4* __restrict__ nl, const uint32_t* __restrict__ rowbase,
                                                    const float2* __restrict__ pay, const float* __restrict__ rho, float nf, float inv2h, float* __restrict__ out)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float4 Ai = pm[i];
    const float2 ai = pay[i];
    const float inv = 1.f / rho[i];
    const uint4 lw = nl[i];
    const uint32_t masks[3] = {lw.x, lw.y, lw.z};
    float sum = 0.f;
#pragma unroll
    for (int dr = 0; dr < 3; dr++) {
        uint32_t mk = masks[dr];
        const uint32_t base = rowbase[dr * n + i];
        while (mk) {
            uint32_t b[4]; bool v[4];
#pragma unroll
            for (int q = 0; q < 4; q++) { v[q] = mk != 0; b[q] = v[q] ? __ffs(mk) - 1 : 0; mk &= mk - 1; }
            float4 A[4]; float2 aj[4];
#pragma unroll
            for (int q = 0; q < 4; q++) { A[q] = pm[base + b[q]]; aj[q] = pay[base + b[q]]; }
#pragma unroll
            for (int q = 0; q < 4; q++)
                if (v[q]) {
                    const float dx = Ai.x - A[q].x, dy = Ai.y - A[q].y;
                    const float r2 = dx * dx + dy * dy;
                    const float rinv = __builtin_amdgcn_rsqf(r2);
                    const float qq = (r2 * rinv) * inv2h;
                    const float a = 18.f * qq * qq - 12.f * qq, vv = 1.f - qq, bb = -6.f * vv * vv;
                    float s = nf * (qq < 0.5f ? a : (qq < 1.f ? bb : 0.f)) * inv2h * rinv;
                    s = qq > 1e-5f ? s : 0.f;
                    sum += A[q].z * inv * ((aj[q].x - ai.x) * (s * dx) + (aj[q].y - ai.y) * (s * dy));
                }
        }
    }
    out[i] = sum;
}

template <class F>
float timeit(F f)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int r = 0; r < 8; r++) {
        hipEventRecord(e0);
        f();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (r > 1 && ms < best) best = ms;
    }
    return best * 1e3f;
}

int main(int argc, char** argv)
{
    const int side = argc > 1 ? atoi(argv[1]) : 1024;
    const uint32_t n = (uint32_t)side * side;
    const int SLOTS = 16;
    // a lattice sorted by rows of cells of 2 x 2 particles... keep it simple: row-major lattice, neighbours = the 13 lattice sites
    // within 2 spacings (x in [-2, 2], y in [-1, 1] -> 3 rows of up to 5, minus self = 14; use 12-14)
    std::vector<uint32_t> idx((size_t)SLOTS * n), rowbase((size_t)3 * n);
    std::vector<uint4> nl(n);
    std::vector<float4> pm(n);
    std::vector<float2> pay(n), pc((size_t)SLOTS * n);
    std::vector<float> rho(n, 1000.f);
    int cnt_max = 0;
    for (uint32_t i = 0; i < n; i++) {
        const int x = i % side, y = i / side;
        pm[i] = make_float4(x * 1.f / side, y * 1.f / side, 1.f, 1.2f / side);
        pay[i] = make_float2(0.001f * x, 0.002f * y);
        int k = 0;
        uint32_t m[3] = {0, 0, 0};
        for (int dy = -1; dy <= 1; dy++) {
            const int yy = y + dy;
            const int x0 = x - 2 < 0 ? 0 : x - 2;
            rowbase[(size_t)(dy + 1) * n + i] = (yy >= 0 && yy < side) ? (uint32_t)yy * side + x0 : i;
            if (yy < 0 || yy >= side) continue;
            for (int xx = x0; xx <= x + 2 && xx < side; xx++) {
                if (dy == 0 && xx == x) continue;
                if (dy != 0 && (xx == x - 2 || xx == x + 2) && ((x + y) & 1)) continue;   // 12..14 neighbours
                m[dy + 1] |= 1u << (xx - x0);
                idx[(size_t)k * n + i] = (uint32_t)yy * side + xx;
                pc[(size_t)k * n + i] = make_float2(0.1f * (xx - x), 0.1f * dy);
                k++;
            }
        }
        for (int q = k; q < SLOTS; q++) idx[(size_t)q * n + i] = i;
        nl[i] = make_uint4(m[0], m[1], m[2], (uint32_t)k);
        cnt_max = k > cnt_max ? k : cnt_max;
    }
    float4* d_pm; float2 *d_pay, *d_pc; uint32_t *d_idx, *d_rb; uint4* d_nl; float *d_rho, *d_out;
    CHECK(hipMalloc(&d_pm, n * 16)); CHECK(hipMalloc(&d_pay, n * 8)); CHECK(hipMalloc(&d_pc, (size_t)SLOTS * n * 8));
    CHECK(hipMalloc(&d_idx, (size_t)SLOTS * n * 4)); CHECK(hipMalloc(&d_rb, (size_t)3 * n * 4)); CHECK(hipMalloc(&d_nl, n * 16));
    CHECK(hipMalloc(&d_rho, n * 4)); CHECK(hipMalloc(&d_out, n * 4));
    CHECK(hipMemcpy(d_pm, pm.data(), n * 16, hipMemcpyHostToDevice)); CHECK(hipMemcpy(d_pay, pay.data(), n * 8, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_pc, pc.data(), (size_t)SLOTS * n * 8, hipMemcpyHostToDevice)); CHECK(hipMemcpy(d_idx, idx.data(), (size_t)SLOTS * n * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_rb, rowbase.data(), (size_t)3 * n * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(d_nl, nl.data(), n * 16, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_rho, rho.data(), n * 4, hipMemcpyHostToDevice));
    const dim3 grid((n + 255) / 256), blk(256);
    printf("n = %u, up to %d neighbours per lane (12-14)\n", n, cnt_max);
    const float t1 = timeit([&] { hipLaunchKernelGGL(k_stream_idx, grid, blk, 0, 0, n, cnt_max, d_pc, d_idx, d_pay, d_rho, d_out); });
    printf("stream_idx  (8 B gradient + 4 B index streamed, 8 B payload gather): %7.1f us   [%.0f MB streamed]\n", t1, cnt_max * 12.0 * n / 1e6);
    const float t2 = timeit([&] { hipLaunchKernelGGL(k_stream_mask, grid, blk, 0, 0, n, d_pc, d_nl, d_rb, d_pay, d_rho, d_out); });
    printf("stream_mask (8 B gradient streamed, index from row masks, 8 B payload gather): %7.1f us   [%.0f MB streamed]\n", t2, 13 * 8.0 * n / 1e6);
    const float t3 = timeit([&] { hipLaunchKernelGGL(k_recompute, grid, blk, 0, 0, n, d_pm, d_nl, d_rb, d_pay, d_rho, 1.f, 0.5f * side / 1.2f, d_out); });
    printf("recompute   (16 B position gather + 8 B payload gather + gradient): %7.1f us\n", t3);
    return 0;
}
