// Cross-stream dependency latency on one GPU: two streams ping-pong through events (record on one, hipStreamWaitEvent on the other),
// a tiny kernel per hop; against the same number of kernels on ONE stream.  What a split sweep pays per extra hop.
// build: hipcc --offload-arch=gfx950 -O2 -o /tmp/xstream_hop scripts/ubench/xstream_hop.hip ; run: /tmp/xstream_hop [hops] [kernel_us]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
__global__ void k_spin(uint32_t us, uint32_t* sink)
{
    const uint64_t t0 = wall_clock64();
    while (wall_clock64() - t0 < (uint64_t)us * 100ull) __builtin_amdgcn_s_sleep(4);
    if (sink && threadIdx.x == 1024) *sink = 1;
}
int main(int argc, char** argv)
{
    const int hops = argc > 1 ? atoi(argv[1]) : 2000;
    const uint32_t us = argc > 2 ? (uint32_t)atoi(argv[2]) : 2u;
    hipStream_t s[2];
    hipEvent_t e[2];
    for (int i = 0; i < 2; i++) {
        hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking);
        hipEventCreateWithFlags(&e[i], hipEventDisableTiming);
    }
    auto run = [&](bool two) {
        hipDeviceSynchronize();
        auto t0 = std::chrono::steady_clock::now();
        for (int h = 0; h < hops; h++) {
            const int a = two ? (h & 1) : 0;
            if (two && h) hipStreamWaitEvent(s[a], e[a ^ 1], 0);
            hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s[a], us, (uint32_t*)nullptr);
            if (two) hipEventRecord(e[a], s[a]);
        }
        hipDeviceSynchronize();
        return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / hops;
    };
    run(false);
    run(true);
    const double one = run(false), two = run(true);
    printf("%d kernels of %u us: one stream %.2f us per kernel, two streams alternating through events %.2f us per kernel -> %.2f us per cross-stream hop\n", hops, us,
           one, two, two - one);
    // a long kernel on stream 0 while stream 1 runs short ones that each wait for an event recorded on stream 0 BEFORE the long kernel: does the wait resolve early?
    return 0;
}
