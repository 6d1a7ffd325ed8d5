// Do two HIP streams of one process run their kernels side by side on this device?  Stream A: a chain of N_SMALL dependent small kernels
// (a few workgroups, ~5 us each: the level propagation's sweeps); stream B: N_BIG kernels that fill the device (~25 us each: the step's
// sweeps).  Times: A alone, B alone, both queued together, for stream A at default / highest priority, and for A queued first / B first.
// build: hipcc --offload-arch=gfx950 -O3 -o two_queues two_queues.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_small(float* p, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float v = p[i % n];
    for (int k = 0; k < 6; k++) v = p[(int)(fabsf(v) * 1e-9f) + (i + 64 * k) % n] + 1.f;   // a chain of dependent loads
    p[i % n] = v * 0.f;
}
__global__ void k_big(const float4* a, float4* b, size_t n)
{
    // (grid-stride: with a grid of 512 workgroups the kernel leaves most wave slots of every CU free)
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float4 v = a[i];
        v.x += 1.f;
        b[i] = v;
    }
}
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    const int N_SMALL = 100, N_BIG = 24;
    const size_t nb = 1 << 22;   // 64 MB in, 64 MB out per big kernel
    float* ps;
    float4 *a, *b;
    CHK(hipMalloc(&ps, 1 << 20));
    CHK(hipMemset(ps, 0, 1 << 20));
    CHK(hipMalloc(&a, nb * 16));
    CHK(hipMalloc(&b, nb * 16));
    CHK(hipMemset(a, 0, nb * 16));
    int lo, hi;
    CHK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    for (int variant = 0; variant < 8; variant++) {
        const bool high = variant & 1, b_first = variant & 2;
        const unsigned big_grid = (variant & 4) ? 512u : (unsigned)(nb / 256);
        hipStream_t sa, sb;
        CHK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
        if (high) CHK(hipStreamCreateWithPriority(&sa, hipStreamNonBlocking, hi));
        else CHK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
        auto run_a = [&]() { for (int k = 0; k < N_SMALL; k++) hipLaunchKernelGGL(k_small, dim3(128), dim3(256), 0, sa, ps, 1 << 18); };
        auto run_b = [&]() { for (int k = 0; k < N_BIG; k++) hipLaunchKernelGGL(k_big, dim3(big_grid), dim3(256), 0, sb, a, b, nb); };
        double ta = 0, tb = 0, tab = 0;
        for (int rep = 0; rep < 5; rep++) {
            CHK(hipDeviceSynchronize());
            double t0 = now();
            run_a();
            CHK(hipStreamSynchronize(sa));
            double t1 = now();
            run_b();
            CHK(hipStreamSynchronize(sb));
            double t2 = now();
            if (b_first) { run_b(); run_a(); } else { run_a(); run_b(); }
            CHK(hipStreamSynchronize(sa));
            CHK(hipStreamSynchronize(sb));
            double t3 = now();
            ta = t1 - t0; tb = t2 - t1; tab = t3 - t2;
        }
        printf("| B grid %u | stream A %s priority, %s queued first | A alone %.0f us (%d small kernels) | B alone %.0f us (%d big kernels) | both %.0f us | sum %.0f us |\n",
               big_grid, high ? "highest" : "default", b_first ? "B" : "A", ta, N_SMALL, tb, N_BIG, tab, ta + tb);
        CHK(hipStreamDestroy(sa));
        CHK(hipStreamDestroy(sb));
    }
    return 0;
}
