// VALU issue rate on gfx950, measured: how many shader clocks does ONE SIMD need per wave64 instruction, by instruction kind and
// by the number of resident waves per SIMD?  (MI355X_MICROARCH.md: SIMD-32, a wave64 VALU op issues over 2 cycles; DESIGN.md round 2
// priced the sweeps at 4.)  Every "floor" of DESIGN.md section 3 follows from this number.
//
//   build: hipcc --offload-arch=gfx950 -O2 -o /tmp/valu_issue scripts/ubench/valu_issue.hip ; run: /tmp/valu_issue
//
// Method: a kernel whose body is UNROLL x CHAINS independent dependency chains of one instruction (inline asm, so the compiler
// neither folds nor reorders them), REPS trips; grid = 256 CUs x 1 block, block = 64 x 4 x W threads, i.e. W waves on each of the
// 4 SIMDs of every CU.  Lane 0 of every wave reads s_memtime (shader clock) before and after; the table shows
//   clk/inst = (longest wave's cycles) / (W x instructions per wave)  -- the SIMD's issue cost per wave-instruction when W waves share it.
// With CHAINS = 8 independent accumulators per wave the chains never wait on their own latency once W >= 1.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CHAINS 8
#define UNROLL 16

#define OP8(STR)                                                                                                                           \
    asm volatile(STR : "+v"(a0) : "v"(b), "v"(c) : "s20", "s22", "s23", "vcc", "scc");                                                                                         \
    asm volatile(STR : "+v"(a1) : "v"(b), "v"(c) : "s20", "s22", "s23", "vcc", "scc");                                                                                         \
    asm volatile(STR : "+v"(a2) : "v"(b), "v"(c) : "s20", "s22", "s23", "vcc", "scc");                                                                                         \
    asm volatile(STR : "+v"(a3) : "v"(b), "v"(c) : "s20", "s22", "s23", "vcc", "scc");                                                                                         \
    asm volatile(STR : "+v"(a4) : "v"(b), "v"(c) : "s20", "s22", "s23", "vcc", "scc");                                                                                         \
    asm volatile(STR : "+v"(a5) : "v"(b), "v"(c) : "s20", "s22", "s23", "vcc", "scc");                                                                                         \
    asm volatile(STR : "+v"(a6) : "v"(b), "v"(c) : "s20", "s22", "s23", "vcc", "scc");                                                                                         \
    asm volatile(STR : "+v"(a7) : "v"(b), "v"(c) : "s20", "s22", "s23", "vcc", "scc");

#define KERNEL(NAME, STR)                                                                                                                  \
    __global__ void NAME(int reps, float* sink, unsigned long long* cyc)                                                                   \
    {                                                                                                                                      \
        float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;                   \
        float b = 1.0000001f, c = 1e-9f;                                                                                                   \
        const unsigned long long w0 = wall_clock64();                                                                                      \
        const unsigned long long t0 = __builtin_readcyclecounter();                                                                        \
        for (int r = 0; r < reps; r++) {                                                                                                   \
            _Pragma("unroll") for (int u = 0; u < UNROLL; u++) { OP8(STR) }                                                                \
        }                                                                                                                                  \
        const unsigned long long t1 = __builtin_readcyclecounter();                                                                        \
        const unsigned long long w1 = wall_clock64();                                                                                      \
        if ((threadIdx.x & 63) == 0) {                                                                                                     \
            cyc[2 * (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6))] = t1 - t0;                                                      \
            cyc[2 * (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) + 1] = w1 - w0;                                                  \
        }                                                                                                                                  \
        if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == -1.f) *sink = a0;                                                                      \
    }

// two 32-bit registers per chain for the packed / 64-bit forms
#define OP8P(STR)                                                                                                                          \
    asm volatile(STR : "+v"(p0) : "v"(pb), "v"(pc) : "s20", "s22", "s23", "vcc", "scc");                                                                                       \
    asm volatile(STR : "+v"(p1) : "v"(pb), "v"(pc) : "s20", "s22", "s23", "vcc", "scc");                                                                                       \
    asm volatile(STR : "+v"(p2) : "v"(pb), "v"(pc) : "s20", "s22", "s23", "vcc", "scc");                                                                                       \
    asm volatile(STR : "+v"(p3) : "v"(pb), "v"(pc) : "s20", "s22", "s23", "vcc", "scc");                                                                                       \
    asm volatile(STR : "+v"(p4) : "v"(pb), "v"(pc) : "s20", "s22", "s23", "vcc", "scc");                                                                                       \
    asm volatile(STR : "+v"(p5) : "v"(pb), "v"(pc) : "s20", "s22", "s23", "vcc", "scc");                                                                                       \
    asm volatile(STR : "+v"(p6) : "v"(pb), "v"(pc) : "s20", "s22", "s23", "vcc", "scc");                                                                                       \
    asm volatile(STR : "+v"(p7) : "v"(pb), "v"(pc) : "s20", "s22", "s23", "vcc", "scc");
typedef float float2v __attribute__((ext_vector_type(2)));
#define KERNELP(NAME, STR)                                                                                                                 \
    __global__ void NAME(int reps, float* sink, unsigned long long* cyc)                                                                   \
    {                                                                                                                                      \
        float2v p0 = {(float)threadIdx.x, 1.f}, p1 = p0 + 1.f, p2 = p0 + 2.f, p3 = p0 + 3.f, p4 = p0 + 4.f, p5 = p0 + 5.f, p6 = p0 + 6.f,   \
                p7 = p0 + 7.f;                                                                                                             \
        float2v pb = {1.0000001f, 0.9999999f}, pc = {1e-9f, 2e-9f};                                                                        \
        const unsigned long long w0 = wall_clock64();                                                                                      \
        const unsigned long long t0 = __builtin_readcyclecounter();                                                                        \
        for (int r = 0; r < reps; r++) {                                                                                                   \
            _Pragma("unroll") for (int u = 0; u < UNROLL; u++) { OP8P(STR) }                                                               \
        }                                                                                                                                  \
        const unsigned long long t1 = __builtin_readcyclecounter();                                                                        \
        const unsigned long long w1 = wall_clock64();                                                                                      \
        if ((threadIdx.x & 63) == 0) {                                                                                                     \
            cyc[2 * (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6))] = t1 - t0;                                                      \
            cyc[2 * (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) + 1] = w1 - w0;                                                  \
        }                                                                                                                                  \
        float2v t = p0 + p1 + p2 + p3 + p4 + p5 + p6 + p7;                                                                                 \
        if (t.x + t.y == -1.f) *sink = t.x;                                                                                                \
    }

KERNEL(k_fma, "v_fma_f32 %0, %1, %0, %2")
KERNEL(k_fmac, "v_fmac_f32 %0, %1, %2")
KERNEL(k_sub, "v_sub_f32 %0, %1, %0")
KERNEL(k_min, "v_min_f32 %0, %1, %0")
KERNEL(k_mov, "v_mov_b32 %0, %1")
KERNEL(k_addu, "v_add_u32 %0, %1, %0")
KERNEL(k_lshl, "v_lshlrev_b32 %0, 1, %0")
KERNEL(k_bfe, "v_bfe_u32 %0, %0, 1, 5")
KERNEL(k_mullo, "v_mul_lo_u32 %0, %1, %0")
KERNEL(k_cvt, "v_cvt_f32_u32 %0, %0")
KERNEL(k_floor, "v_floor_f32 %0, %0")
KERNEL(k_cndmask_s, "v_cndmask_b32 %0, %1, %0, s[22:23]")
KERNEL(k_cmp_s, "v_cmp_lt_f32 s[22:23], %1, %0")
KERNEL(k_cmp_cnd, "v_cmp_lt_f32 vcc, %1, %0\n v_cndmask_b32 %0, %2, %0, vcc")
KERNEL(k_med3, "v_med3_f32 %0, %1, %0, %2")
KERNEL(k_sffs, "s_ff1_i32_b64 s20, s[22:23]")
KERNEL(k_sand, "s_and_b64 s[22:23], s[22:23], exec")
KERNEL(k_snop, "s_nop 0")
KERNEL(k_add, "v_add_f32 %0, %1, %0")
KERNEL(k_mul, "v_mul_f32 %0, %1, %0")
KERNEL(k_max, "v_max_f32 %0, %1, %0")
KERNEL(k_rsq, "v_rsq_f32 %0, %0")
KERNEL(k_rcp, "v_rcp_f32 %0, %0")
KERNEL(k_sqrt, "v_sqrt_f32 %0, %0")
KERNEL(k_cndmask, "v_cndmask_b32 %0, %1, %0, vcc")
KERNEL(k_cmp, "v_cmp_lt_f32 vcc, %1, %0")
KERNEL(k_and, "v_and_b32 %0, %1, %0")
KERNEL(k_lshl_add, "v_lshl_add_u32 %0, %0, 1, %1")
KERNEL(k_ffbl, "v_ffbl_b32 %0, %0")
KERNEL(k_mov_dpp, "v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf")
KERNEL(k_add_dpp, "v_add_f32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf")
KERNEL(k_readlane, "v_readlane_b32 s20, %0, 3")
KERNEL(k_bpermute, "ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)")
KERNELP(k_pk_fma, "v_pk_fma_f32 %0, %1, %0, %2")
KERNELP(k_pk_add, "v_pk_add_f32 %0, %1, %0")
KERNELP(k_pk_mul, "v_pk_mul_f32 %0, %1, %0")
KERNELP(k_lshl_add_u64, "v_lshl_add_u64 %0, %0, 0, %1")

typedef void (*kern_t)(int, float*, unsigned long long*);

int main()
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    printf("device: %s, %d CUs, clockRate %d kHz\n", prop.gcnArchName, cus, prop.clockRate);
    float* sink;
    unsigned long long* cyc;
    hipMalloc(&sink, 4);
    hipMalloc(&cyc, sizeof(unsigned long long) * cus * 64 * 2);
    struct { const char* name; kern_t k; } ks[] = {
        {"v_fma_f32", k_fma}, {"v_fmac_f32", k_fmac}, {"v_sub_f32", k_sub}, {"v_min_f32", k_min}, {"v_mov_b32", k_mov}, {"v_add_u32", k_addu}, {"v_lshlrev_b32", k_lshl},
        {"v_bfe_u32", k_bfe}, {"v_mul_lo_u32", k_mullo}, {"v_cvt_f32_u32", k_cvt}, {"v_floor_f32", k_floor}, {"v_med3_f32", k_med3},
        {"v_cndmask_b32 (sgpr mask)", k_cndmask_s}, {"v_cmp_lt_f32 -> sgpr", k_cmp_s}, {"v_cmp + v_cndmask (2 inst)", k_cmp_cnd},
        {"s_ff1_i32_b64", k_sffs}, {"s_and_b64", k_sand}, {"s_nop 0", k_snop}, {"v_add_f32", k_add}, {"v_mul_f32", k_mul}, {"v_max_f32", k_max}, {"v_cndmask_b32", k_cndmask}, {"v_cmp_lt_f32", k_cmp},
        {"v_and_b32", k_and}, {"v_lshl_add_u32", k_lshl_add}, {"v_ffbl_b32", k_ffbl}, {"v_lshl_add_u64", k_lshl_add_u64},
        {"v_pk_fma_f32", k_pk_fma}, {"v_pk_add_f32", k_pk_add}, {"v_pk_mul_f32", k_pk_mul},
        {"v_rsq_f32", k_rsq}, {"v_rcp_f32", k_rcp}, {"v_sqrt_f32", k_sqrt},
        {"v_mov_b32_dpp row_shr", k_mov_dpp}, {"v_add_f32_dpp row_shr", k_add_dpp}, {"v_readlane_b32", k_readlane}, {"ds_bpermute_b32 (+wait)", k_bpermute},
    };
    const int reps = 200;
    const double inst = (double)reps * UNROLL * CHAINS;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    printf("| instruction | W=1 clk/inst | W=2 | W=4 | W=8 | W=8 wall: wave-inst/ns/SIMD | s_memtime ticks per us (wall_clock64 = 100 MHz) |\n|---|---|---|---|---|---|---|\n");
    for (auto& k : ks) {
        printf("| %s |", k.name);
        fflush(stdout);
        double wall_rate = 0, tick_per_us = 0;
        for (int W : {1, 2, 4, 8}) {
            const int threads = 64 * 4 * W;   // W waves on each SIMD of the CU (one block per CU; 1024 threads max => two blocks at W = 8)
            const int blocks_per_cu = threads > 1024 ? 2 : 1;
            const int bt = threads / blocks_per_cu;
            const int nb = cus * blocks_per_cu;
            hipLaunchKernelGGL(k.k, dim3(nb), dim3(bt), 0, 0, reps, sink, cyc);   // warm-up
            hipDeviceSynchronize();
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k.k, dim3(nb), dim3(bt), 0, 0, reps, sink, cyc);
            hipEventRecord(e1, 0);
            hipDeviceSynchronize();
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned long long> h2((size_t)nb * (bt / 64) * 2), h, hw;
            hipMemcpy(h2.data(), cyc, h2.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
            for (size_t q = 0; q < h2.size(); q += 2) { h.push_back(h2[q]); hw.push_back(h2[q + 1]); }
            std::sort(h.begin(), h.end());
            std::sort(hw.begin(), hw.end());
            const double med = (double)h[h.size() / 2];
            printf(" %.2f |", med / (W * inst));
            if (W == 8) { wall_rate = (W * inst) / (ms * 1e6); tick_per_us = med / ((double)hw[hw.size() / 2] / 100.0); }
        }
        printf(" %.3f | %.0f |\n", wall_rate, tick_per_us);
        fflush(stdout);
    }
    printf("\n(clk = s_memtime ticks between the wave's first and last instruction, median over all waves; W waves per SIMD share the SIMD, so\n"
           " clk/inst = wave cycles / (W x instructions per wave).  2.0 = SIMD-32 issuing a wave64 op in two passes; 4.0 = quarter rate.)\n");
    return 0;
}
