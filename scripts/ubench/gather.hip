// Micro-benchmark: what does a neighbour gather cost on gfx950?  (development aid, not part of the library)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/gather scripts/ubench/gather.hip && /tmp/gather
// 2^20 lanes (16384 waves, 256-thread blocks) each do K "trips" of G independent gathers of W bytes at j = i + d(k,g) with a
// small pseudo-random jitter (neighbour-like locality), then a dependent FMA chain of V instructions per trip.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int W, int G, int V, int ACT>
__global__ __launch_bounds__(256) void k_gather(const float4* __restrict__ a4, const float2* __restrict__ a2, const float* __restrict__ a1,
                                                 const int* __restrict__ off, int K, uint32_t n, float* __restrict__ out)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float acc = 0.f;
    for (int k = 0; k < K; k++) {
        float v[G];
#pragma unroll
        for (int g = 0; g < G; g++) {
            uint32_t j = i + (uint32_t)off[(k * G + g) * 64 + (threadIdx.x & 63)];
            j = j < n ? j : i;
            v[g] = 0.f;
            // ACT of every 8 lanes (pseudo-randomly per gather) really issue the load: what exec-masked padding slots cost
            if ((((threadIdx.x * 2654435761u) >> 13) + g * 3 + k * 5) % 8u < (unsigned)ACT) {
                if (W == 16) { float4 q = a4[j]; v[g] = q.x + q.w; }
                else if (W == 8) { float2 q = a2[j]; v[g] = q.x + q.y; }
                else v[g] = a1[j];
            }
        }
#pragma unroll
        for (int g = 0; g < G; g++) {
            float t = v[g];
#pragma unroll
            for (int q = 0; q < V; q++) t = fmaf(t, 1.0001f, 0.5f);
            acc += t;
        }
    }
    out[i] = acc;
}

template <int W, int G, int V, int ACT = 8>
float run(const float4* a4, const float2* a2, const float* a1, const int* off, int K, uint32_t n, float* out)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int r = 0; r < 5; r++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_gather<W, G, V, ACT>), dim3((n + 255) / 256), dim3(256), 0, 0, a4, a2, a1, off, K, n, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (r > 0 && ms < best) best = ms;
    }
    return best * 1e3f;
}

int main()
{
    const uint32_t n = 1u << 20;
    float4* a4; float2* a2; float* a1; float* out; int* off;
    CHECK(hipMalloc(&a4, n * 16)); CHECK(hipMalloc(&a2, n * 8)); CHECK(hipMalloc(&a1, n * 4)); CHECK(hipMalloc(&out, n * 4));
    CHECK(hipMemset(a4, 0, n * 16)); CHECK(hipMemset(a2, 0, n * 8)); CHECK(hipMemset(a1, 0, n * 4));
    // offsets: trip k, gather g, lane l -> row (k % 3 - 1) * 2200 + (g + 4 * (k / 3)) - 6 + jitter(l) in {-1, 0, 1, 2}
    const int KMAX = 8, GMAX = 8;
    std::vector<int> h(KMAX * GMAX * 64);
    unsigned s = 12345u;
    for (int k = 0; k < KMAX; k++)
        for (int g = 0; g < GMAX; g++)
            for (int l = 0; l < 64; l++) {
                s = s * 1664525u + 1013904223u;
                h[(k * GMAX + g) * 64 + l] = (k % 3 - 1) * 2200 + (g + 4 * (k / 3)) - 6 + (int)((s >> 24) & 3u) - 1;
            }
    CHECK(hipMalloc(&off, h.size() * 4));
    // note: kernels index off with stride G, so fill per G below
    auto fill = [&](int G) {
        std::vector<int> t(KMAX * G * 64);
        for (int k = 0; k < KMAX; k++)
            for (int g = 0; g < G; g++)
                for (int l = 0; l < 64; l++) t[(k * G + g) * 64 + l] = h[(k * GMAX + g) * 64 + l];
        hipMemcpy(off, t.data(), t.size() * 4, hipMemcpyHostToDevice);
    };
    printf("%-28s %8s\n", "variant (K trips x G gathers)", "us");
#define RUN(W, G, V, K)                                                                              \
    {                                                                                                \
        fill(G);                                                                                     \
        float us = run<W, G, V>(a4, a2, a1, off, K, n, out);                                         \
        printf("W=%2d G=%d V=%3d K=%d  gathers/lane=%2d : %7.1f us  (%.2f us per gather-instruction per lane-set)\n", W, G, V, K, G * K, us, us / (G * K)); \
    }
    RUN(16, 4, 8, 1) RUN(16, 4, 8, 2) RUN(16, 4, 8, 4) RUN(16, 4, 8, 6)
    RUN(8, 4, 8, 4) RUN(4, 4, 8, 4)
    RUN(16, 8, 8, 2) RUN(16, 8, 8, 3) RUN(8, 8, 8, 4) RUN(4, 8, 8, 4)
    RUN(16, 4, 32, 4) RUN(16, 4, 64, 4) RUN(16, 4, 128, 4)
    RUN(16, 2, 8, 8) RUN(16, 1, 8, 8)
    printf("fraction of lanes active per gather (W=16, 4x4 gathers):\n");
    { fill(4); printf("  8/8 %.1f us  6/8 %.1f us  4/8 %.1f us  2/8 %.1f us\n", run<16, 4, 8, 8>(a4, a2, a1, off, 4, n, out), run<16, 4, 8, 6>(a4, a2, a1, off, 4, n, out), run<16, 4, 8, 4>(a4, a2, a1, off, 4, n, out), run<16, 4, 8, 2>(a4, a2, a1, off, 4, n, out)); }
    return 0;
}
