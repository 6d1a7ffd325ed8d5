// FETCH_SIZE / WRITE_SIZE calibration on the access patterns of the SPH sweeps (round 6).
// MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a WIDE COALESCED streaming read (16 B / lane); other widths
// and WRITE_SIZE are uncalibrated.  scripts/summarize_profile.py doubled FETCH_SIZE for every kernel.  Each kernel below moves a KNOWN
// number of unique bytes in one pattern the sweeps use, on arrays far larger than the 256 MB Infinity Cache (so that no launch is served
// from a cache filled by the launch before it), one launch per pattern:
//   build: hipcc --offload-arch=gfx950 -O3 -o fetch_calib scripts/ubench/fetch_calib.hip
//   run:   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out/fetch -o p -- ./fetch_calib     (and the same with WRITE_SIZE)
// scripts/fetch_calib_table.py joins the two passes with the byte counts this program prints.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// n lanes, one element each; the result is folded into one store per WAVE (so that the launch's writes stay negligible: n / 64 x 4 B)
__device__ __forceinline__ void fold(float v, float* out, uint32_t i)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0) out[i >> 6] = v;
}
__global__ __launch_bounds__(256) void read_stream_16(const float4* __restrict__ a, float* __restrict__ out, uint32_t n)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    const float4 q = a[i];
    fold(q.x + q.y + q.z + q.w, out, i);
}
__global__ __launch_bounds__(256) void read_stream_8(const float2* __restrict__ a, float* __restrict__ out, uint32_t n)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    const float2 q = a[i];
    fold(q.x + q.y, out, i);
}
__global__ __launch_bounds__(256) void read_stream_4(const float* __restrict__ a, float* __restrict__ out, uint32_t n)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    fold(a[i], out, i);
}
__global__ __launch_bounds__(256) void read_stream_1(const uint8_t* __restrict__ a, float* __restrict__ out, uint32_t n)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    fold((float)a[i], out, i);
}
// the replay sweeps' pattern: every lane gathers twelve 16-byte records at the rest lattice's offsets in a cell-sorted array (row pitch
// PITCH slots: +-1, +-2 in its own row of cells, the rest one or two rows away), blocks in the XCD-band order of the product.  Unique bytes =
// 16 n (+ 2 PITCH records at the ends): every record is some lane's neighbour twelve times, and should leave memory once.
__global__ __launch_bounds__(256) void read_gather_16x12(const float4* __restrict__ a, float* __restrict__ out, uint32_t n, uint32_t nblocks, int pitch)
{
    const uint32_t per_xcd = (nblocks + 7) >> 3;
    const uint32_t blk = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    if (blk >= nblocks) return;
    const uint32_t i = blk * 256 + threadIdx.x;
    const int off[12] = {-2 * pitch, -pitch - 1, -pitch, -pitch + 1, -2, -1, 1, 2, pitch - 1, pitch, pitch + 1, 2 * pitch};
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 12; k++) {
        long long j = (long long)i + off[k];
        j = j < 0 ? 0 : (j >= (long long)n ? n - 1 : j);
        const float4 q = a[j];
        s += q.x + q.w;
    }
    fold(s, out, i);
}
// one 16-byte record per lane from a line of its own (stride 256 B): what a request for a barely used line is tallied as
__global__ __launch_bounds__(256) void read_sparse_16(const float4* __restrict__ a, float* __restrict__ out, uint32_t n)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    const float4 q = a[(size_t)i * 16];
    fold(q.x + q.w, out, i);
}
__global__ __launch_bounds__(256) void write_stream_16(float4* __restrict__ a, uint32_t n)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    a[i] = make_float4((float)i, 1.f, 2.f, 3.f);
}
__global__ __launch_bounds__(256) void write_stream_8(float2* __restrict__ a, uint32_t n)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    a[i] = make_float2((float)i, 1.f);
}
__global__ __launch_bounds__(256) void write_stream_4(float* __restrict__ a, uint32_t n)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    a[i] = (float)i;
}
__global__ __launch_bounds__(256) void write_stream_1(uint8_t* __restrict__ a, uint32_t n)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    a[i] = (uint8_t)i;
}
// the BUILD sweep's offset-list store: group g of particle i at [g n + i], 8 bytes each, three groups
__global__ __launch_bounds__(256) void write_groups_8x3(uint2* __restrict__ a, uint32_t n)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
#pragma unroll
    for (int g = 0; g < 3; g++) a[(size_t)g * n + i] = make_uint2(i, (uint32_t)g);
}
__global__ __launch_bounds__(256) void flush_cache(float4* __restrict__ a, uint32_t n)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    float4 q = a[i];
    q.x += 1.f;
    a[i] = q;
}

int main()
{
    const uint32_t n = 1u << 26;                // 64 Mi lanes: 1 GiB of 16-byte records
    const size_t big = (size_t)n * 16 * 16;     // (read_sparse_16 strides 256 B: 16 GiB)
    void *a = nullptr, *out = nullptr, *junk = nullptr;
    CHECK(hipMalloc(&a, big));
    CHECK(hipMalloc(&out, (size_t)n / 64 * 4 + 4096));
    CHECK(hipMalloc(&junk, (size_t)1 << 30));
    CHECK(hipMemset(a, 0, big));
    CHECK(hipMemset(junk, 0, (size_t)1 << 30));
    const dim3 g(n / 256), b(256);
    auto flush = [&]() { hipLaunchKernelGGL(flush_cache, dim3((1u << 26) / 256), b, 0, 0, (float4*)junk, 1u << 26); };   // 1 GiB through the caches between two patterns
    printf("kernel,unique_read_bytes,unique_write_bytes\n");
    flush(); hipLaunchKernelGGL(read_stream_16, g, b, 0, 0, (const float4*)a, (float*)out, n);
    printf("read_stream_16,%zu,%zu\n", (size_t)n * 16, (size_t)n / 64 * 4);
    flush(); hipLaunchKernelGGL(read_stream_8, g, b, 0, 0, (const float2*)a, (float*)out, n);
    printf("read_stream_8,%zu,%zu\n", (size_t)n * 8, (size_t)n / 64 * 4);
    flush(); hipLaunchKernelGGL(read_stream_4, g, b, 0, 0, (const float*)a, (float*)out, n);
    printf("read_stream_4,%zu,%zu\n", (size_t)n * 4, (size_t)n / 64 * 4);
    flush(); hipLaunchKernelGGL(read_stream_1, g, b, 0, 0, (const uint8_t*)a, (float*)out, n);
    printf("read_stream_1,%zu,%zu\n", (size_t)n, (size_t)n / 64 * 4);
    flush(); hipLaunchKernelGGL(read_gather_16x12, dim3(((n / 256 + 7) / 8) * 8), b, 0, 0, (const float4*)a, (float*)out, n, n / 256, 2126);
    printf("read_gather_16x12,%zu,%zu\n", (size_t)n * 16, (size_t)n / 64 * 4);
    flush(); hipLaunchKernelGGL(read_sparse_16, g, b, 0, 0, (const float4*)a, (float*)out, n);
    printf("read_sparse_16,%zu,%zu\n", (size_t)n * 16, (size_t)n / 64 * 4);
    flush(); hipLaunchKernelGGL(write_stream_16, g, b, 0, 0, (float4*)a, n);
    printf("write_stream_16,0,%zu\n", (size_t)n * 16);
    flush(); hipLaunchKernelGGL(write_stream_8, g, b, 0, 0, (float2*)a, n);
    printf("write_stream_8,0,%zu\n", (size_t)n * 8);
    flush(); hipLaunchKernelGGL(write_stream_4, g, b, 0, 0, (float*)a, n);
    printf("write_stream_4,0,%zu\n", (size_t)n * 4);
    flush(); hipLaunchKernelGGL(write_stream_1, g, b, 0, 0, (uint8_t*)a, n);
    printf("write_stream_1,0,%zu\n", (size_t)n);
    flush(); hipLaunchKernelGGL(write_groups_8x3, g, b, 0, 0, (uint2*)a, n);
    printf("write_groups_8x3,0,%zu\n", (size_t)n * 24);
    CHECK(hipDeviceSynchronize());
    return 0;
}
