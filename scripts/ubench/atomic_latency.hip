// Round-trip latency of dependent memory operations from ONE wave (MI355X): a returning atomic add, loads at wavefront / workgroup (sc0) /
// agent (sc1) scope (pointer chase through a 1 MB ring), with the device idle -- what a device-side barrier or a queue tail costs per step.
// build: hipcc --offload-arch=gfx950 -O3 -o atomic_latency atomic_latency.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_atomic(unsigned* p, int n, unsigned long long* out)
{
    unsigned v = 0;
    const unsigned long long t0 = wall_clock64();
    for (int k = 0; k < n; k++) v = atomicAdd(p + (v & 1u), 1u);   // (the next address depends on the returned value)
    const unsigned long long t1 = wall_clock64();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = v; }
}
template <int SCOPE>
__global__ void k_chase(const unsigned* ring, int n, unsigned long long* out)
{
    unsigned j = threadIdx.x;
    const unsigned long long t0 = wall_clock64();
    for (int k = 0; k < n; k++) {
        if (SCOPE == 0) j = ring[j];
        else if (SCOPE == 1) j = __hip_atomic_load(ring + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else j = __hip_atomic_load(ring + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const unsigned long long t1 = wall_clock64();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = j; }
}
int main()
{
    const int n = 2000, ring_n = 1 << 18;
    unsigned *p, *ring;
    unsigned long long* out;
    CHK(hipMalloc(&p, 256));
    CHK(hipMemset(p, 0, 256));
    CHK(hipMalloc(&ring, ring_n * 4));
    CHK(hipMalloc(&out, 16));
    std::vector<unsigned> h(ring_n);
    for (int i = 0; i < ring_n; i++) h[i] = (unsigned)((i * 40503ull + 12345ull) % ring_n);   // a scattered permutation-ish walk
    CHK(hipMemcpy(ring, h.data(), ring_n * 4, hipMemcpyHostToDevice));
    unsigned long long r[2];
    auto report = [&](const char* name) {
        CHK(hipDeviceSynchronize());
        CHK(hipMemcpy(r, out, 16, hipMemcpyDeviceToHost));
        printf("| %s | %.0f ns per dependent operation |\n", name, r[0] * 10.0 / n);   // wall_clock64: 100 MHz
        return 0;
    };
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(k_atomic, dim3(1), dim3(64), 0, 0, p, n, out);
        if (report("returning atomicAdd, one wave, same two words (64 lanes)")) return 1;
        hipLaunchKernelGGL(k_atomic, dim3(1), dim3(1), 0, 0, p, n, out);
        if (report("returning atomicAdd, one lane")) return 1;
        hipLaunchKernelGGL(k_chase<0>, dim3(1), dim3(64), 0, 0, ring, n, out);
        if (report("plain load chase (L1 / L2 as they come)")) return 1;
        hipLaunchKernelGGL(k_chase<1>, dim3(1), dim3(64), 0, 0, ring, n, out);
        if (report("workgroup-scope load chase (sc0)")) return 1;
        hipLaunchKernelGGL(k_chase<2>, dim3(1), dim3(64), 0, 0, ring, n, out);
        if (report("agent-scope load chase (sc1)")) return 1;
    }
    return 0;
}
