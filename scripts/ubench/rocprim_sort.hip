// rocPRIM's radix_sort_pairs against this repo's two-pass LSD sort (59.6 us at N = 1M, 18 key bits): 1M (cell key, index) pairs.
// build: hipcc --offload-arch=gfx950 -O2 -o /tmp/rocprim_sort scripts/ubench/rocprim_sort.hip ; run: /tmp/rocprim_sort [n] [bits]
#include <hip/hip_runtime.h>
#include <cstring>
#include <string.h>
#include <rocprim/rocprim.hpp>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
int main(int argc, char** argv)
{
    const size_t n = argc > 1 ? (size_t)atol(argv[1]) : (1u << 20);
    const unsigned bits = argc > 2 ? (unsigned)atoi(argv[2]) : 18u;
    std::vector<uint32_t> hk(n), hv(n);
    // nearly sorted keys, as a step leaves them: key = i / 4 with a few per cent displaced by one row of cells
    uint32_t s = 12345u;
    for (size_t i = 0; i < n; i++) {
        s = s * 1664525u + 1013904223u;
        uint32_t k = (uint32_t)(i / 4);
        if ((s >> 27) == 0) k = (k + 512u) & ((1u << bits) - 1u);
        hk[i] = k & ((1u << bits) - 1u);
        hv[i] = (uint32_t)i;
    }
    uint32_t *k0, *k1, *v0, *v1;
    hipMalloc(&k0, n * 4); hipMalloc(&k1, n * 4); hipMalloc(&v0, n * 4); hipMalloc(&v1, n * 4);
    hipMemcpy(k0, hk.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(v0, hv.data(), n * 4, hipMemcpyHostToDevice);
    size_t tmp_bytes = 0;
    rocprim::radix_sort_pairs(nullptr, tmp_bytes, k0, k1, v0, v1, n, 0, bits, 0);
    void* tmp;
    hipMalloc(&tmp, tmp_bytes);
    hipStream_t st;
    hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 5; w++) rocprim::radix_sort_pairs(tmp, tmp_bytes, k0, k1, v0, v1, n, 0, bits, st);
    hipStreamSynchronize(st);
    const int reps = 50;
    hipEventRecord(e0, st);
    for (int r = 0; r < reps; r++) rocprim::radix_sort_pairs(tmp, tmp_bytes, k0, k1, v0, v1, n, 0, bits, st);
    hipEventRecord(e1, st);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("rocprim::radix_sort_pairs n=%zu bits=%u: %.2f us per sort (temporary storage %zu bytes)\n", n, bits, ms * 1e3 / reps, tmp_bytes);
    return 0;
}
