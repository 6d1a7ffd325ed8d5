// BUILD-sweep laboratory: the density sweep that also records the neighbour list (OpDensity, BUILD = true, of csrc/sph_sweeps.hip)
// of a uniform-h scene as a stand-alone program, on TWO grids:
//   R = 1: cells of one SUPPORT radius (2 h), 3 x 3 stencil, three contiguous candidate rows, list word = 3 row masks + count (16 B)
//          -- the product's form;
//   R = 2: cells of HALF a support radius (h), 5 x 5 stencil, five contiguous candidate rows, list word = 5 row masks + count (24 B)
//          -- 25 h^2 of candidate area instead of 36 h^2.
// Both walk their rows in trips of 4 with the reference's predicate (r2 < (2 h)^2, strict), sum the cubic-spline density over the
// accepted candidates and store the masks, the count and rho.  Each scene is sorted by ITS grid (x-fastest cells, stable).
//
//   build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o build_lab scripts/ubench/build_lab.hip ; run: ./build_lab [side=1024] [jitter=0.15] [reps=50]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#define CHECK(x)                                                                                                   \
    do {                                                                                                           \
        hipError_t e_ = (x);                                                                                       \
        if (e_ != hipSuccess) {                                                                                    \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));                              \
            exit(1);                                                                                               \
        }                                                                                                          \
    } while (0)

struct Grid {
    float cs;
    int minx, miny, sx, sy;
};
struct BArgs {
    uint32_t n, nblocks;
    Grid g;
    float h, nf, inv2h, s2, mass;
    const uint32_t* __restrict__ cell_start;
    const float4* __restrict__ pm;   // x, y, m, h
    const float2* __restrict__ xy;   // x, y only (uniform scenes need nothing else of a candidate): two candidates per 16-byte load
    uint4* __restrict__ nl_a;        // masks 0..2 (R = 1) or 0..3 (R = 2) [+ count for R = 1]
    uint2* __restrict__ nl_b;        // R = 2: mask 4, count
    float* __restrict__ rho;
    uint32_t* __restrict__ over;     // rows with more than 32 candidates (not representable as a mask)
};

// cubic spline in truncated-power form: W(q) = nf [2 (1 - q)+^3 - 8 (1/2 - q)+^3], q = r / (2 h)
__device__ __forceinline__ float w_spline(const BArgs& A, float r2)
{
    const float q = __builtin_amdgcn_sqrtf(r2) * A.inv2h;   // (v_sqrt_f32, as MathUniform::w of the library)
    const float u = fmaxf(1.f - q, 0.f), t = fmaxf(0.5f - q, 0.f);
    return 2.f * A.nf * fmaf(-4.f * t, t * t, u * (u * u));
}

template <int R, int TRIP>
__global__ __launch_bounds__(256) void k_build(BArgs A)
{
    const uint32_t per_xcd = (A.nblocks + 7) >> 3;
    const uint32_t blk = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    if (blk >= A.nblocks) return;
    const uint32_t i = blk * 256 + threadIdx.x;
    if (i >= A.n) return;
    const float4 Ai = A.pm[i];
    const int cx = (int)floorf(Ai.x / A.g.cs) - A.g.minx, cy = (int)floorf(Ai.y / A.g.cs) - A.g.miny;
    constexpr int ROWS = 2 * R + 1;
    uint32_t rb[ROWS], re[ROWS], mk[ROWS];
#pragma unroll
    for (int dr = 0; dr < ROWS; dr++) {
        const int yy = cy + dr - R;
        const bool ok = yy >= 0 && yy < A.g.sy;
        const uint32_t base = (uint32_t)(ok ? yy : 0) * (uint32_t)A.g.sx;
        rb[dr] = ok ? A.cell_start[base + (uint32_t)max(cx - R, 0)] : 0u;
        re[dr] = ok ? A.cell_start[base + (uint32_t)min(cx + R + 1, A.g.sx)] : 0u;
    }
    float rho = 0.f;
    uint32_t cnt = 0;
#pragma unroll
    for (int dr = 0; dr < ROWS; dr++) {
        const uint32_t b = rb[dr], e = re[dr];
        uint32_t m = 0;
        if (e - b > 32u) atomicAdd(A.over, 1u);
        for (uint32_t j = b; j < e; j += TRIP) {
            float4 Aj[TRIP];
#pragma unroll
            for (int k = 0; k < TRIP; k++) Aj[k] = A.pm[j + k < e ? j + k : j];
#pragma unroll
            for (int k = 0; k < TRIP; k++) {
                const float dx = Ai.x - Aj[k].x, dy = Ai.y - Aj[k].y;
                const float r2 = dx * dx + dy * dy;
                if (j + k < e && r2 < A.s2) {
                    rho += A.mass * w_spline(A, r2);
                    const uint32_t bit = j + k - b;
                    if (bit < 32u) m |= 1u << bit;
                    cnt++;
                }
            }
        }
        mk[dr] = m;
    }
    if (R == 1) A.nl_a[i] = make_uint4(mk[0], mk[1], mk[2], cnt);
    else {
        A.nl_a[i] = make_uint4(mk[0], mk[1], mk[2], mk[ROWS > 3 ? 3 : 0]);
        A.nl_b[i] = make_uint2(mk[ROWS > 4 ? 4 : 0], cnt);
    }
    A.rho[i] = rho;
}

// ---- branch-free forms (R = 1 only).  Every candidate slot of a trip is evaluated by every lane: the truncated-power spline is
// zero beyond the support by itself, the predicate (the reference's operations, strict <) only selects the mask bit and zeroes W for
// a rejected candidate (so that the sum sees exactly the accepted terms); slots behind the row's end are masked by a validity mask
// computed once per row.  PAIRED: the candidates come from the float2 array, two per 16-byte load (rows are contiguous index ranges).
template <int PAIRED>
__global__ __launch_bounds__(256) void k_build_flat(BArgs A)
{
    const uint32_t per_xcd = (A.nblocks + 7) >> 3;
    const uint32_t blk = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    if (blk >= A.nblocks) return;
    const uint32_t i = blk * 256 + threadIdx.x;
    if (i >= A.n) return;
    const float4 Ai = A.pm[i];
    const int cx = (int)floorf(Ai.x / A.g.cs) - A.g.minx, cy = (int)floorf(Ai.y / A.g.cs) - A.g.miny;
    uint32_t rb[3], re[3], mk[3];
#pragma unroll
    for (int dr = 0; dr < 3; dr++) {
        const int yy = cy + dr - 1;
        const bool ok = yy >= 0 && yy < A.g.sy;
        const uint32_t base = (uint32_t)(ok ? yy : 0) * (uint32_t)A.g.sx;
        rb[dr] = ok ? A.cell_start[base + (uint32_t)max(cx - 1, 0)] : 0u;
        re[dr] = ok ? A.cell_start[base + (uint32_t)min(cx + 2, A.g.sx)] : 0u;
    }
    float rho = 0.f;
#pragma unroll
    for (int dr = 0; dr < 3; dr++) {
        const uint32_t b = rb[dr], e = re[dr];
        const uint32_t len = e - b;
        if (len > 32u) atomicAdd(A.over, 1u);
        uint32_t m = 0;
        constexpr uint32_t PER = PAIRED ? 8u : 4u;
        for (uint32_t k0 = 0; __any(k0 < len); k0 += PER) {   // (wave-uniform trip count)
            float cxs[PER], cys[PER];
            if (PAIRED) {
#pragma unroll
                for (uint32_t g = 0; g < 4; g++) {
                    float4 v;
                    __builtin_memcpy(&v, A.xy + b + k0 + 2 * g, 16);
                    cxs[2 * g] = v.x; cys[2 * g] = v.y; cxs[2 * g + 1] = v.z; cys[2 * g + 1] = v.w;
                }
            } else {
#pragma unroll
                for (uint32_t g = 0; g < 4; g++) {
                    const float4 v = A.pm[b + k0 + g];
                    cxs[g] = v.x; cys[g] = v.y;
                }
            }
#pragma unroll
            for (uint32_t g = 0; g < PER; g++) {
                const float dx = Ai.x - cxs[g], dy = Ai.y - cys[g];
                const float r2 = dx * dx + dy * dy;
                const bool in = r2 < A.s2;
                const float w = w_spline(A, r2);
                const bool take = in && (k0 + g < len);
                rho += A.mass * (take ? w : 0.f);
                m |= (take ? 1u : 0u) << ((k0 + g) & 31u);
            }
        }
        mk[dr] = m;
    }
    A.nl_a[i] = make_uint4(mk[0], mk[1], mk[2], (uint32_t)(__popc(mk[0]) + __popc(mk[1]) + __popc(mk[2])));
    A.rho[i] = rho;
}

// ---- two phases: (1) the predicate alone over the candidate rows, branch-free, two candidates per 16-byte load of the float2
// array, into the three row masks; (2) the density sum over the accepted bits only, by the mask replay every later sweep uses
// (trips of 4 set bits per row, float2 gathers).  Same visiting order as the walk: the sums are bit-identical. ----
template <int PAIRED>
__global__ __launch_bounds__(256) void k_build_2phase(BArgs A)
{
    const uint32_t per_xcd = (A.nblocks + 7) >> 3;
    const uint32_t blk = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    if (blk >= A.nblocks) return;
    const uint32_t i = blk * 256 + threadIdx.x;
    if (i >= A.n) return;
    const float2 Ai = A.xy[i];
    const int cx = (int)floorf(Ai.x / A.g.cs) - A.g.minx, cy = (int)floorf(Ai.y / A.g.cs) - A.g.miny;
    uint32_t rb[3], re[3], mk[3];
#pragma unroll
    for (int dr = 0; dr < 3; dr++) {
        const int yy = cy + dr - 1;
        const bool ok = yy >= 0 && yy < A.g.sy;
        const uint32_t base = (uint32_t)(ok ? yy : 0) * (uint32_t)A.g.sx;
        rb[dr] = ok ? A.cell_start[base + (uint32_t)max(cx - 1, 0)] : 0u;
        re[dr] = ok ? A.cell_start[base + (uint32_t)min(cx + 2, A.g.sx)] : 0u;
    }
#pragma unroll
    for (int dr = 0; dr < 3; dr++) {
        const uint32_t b = rb[dr];
        const uint32_t len = re[dr] - b;
        if (len > 32u) atomicAdd(A.over, 1u);
        uint32_t m = 0;
        constexpr uint32_t PER = PAIRED ? 8u : 4u;
        for (uint32_t k0 = 0; __any(k0 < len); k0 += PER) {
            float cxs[PER], cys[PER];
            if (PAIRED) {
#pragma unroll
                for (uint32_t g = 0; g < 4; g++) {
                    float4 v;
                    __builtin_memcpy(&v, A.xy + b + k0 + 2 * g, 16);
                    cxs[2 * g] = v.x; cys[2 * g] = v.y; cxs[2 * g + 1] = v.z; cys[2 * g + 1] = v.w;
                }
            } else {
#pragma unroll
                for (uint32_t g = 0; g < 4; g++) {
                    const float2 v = A.xy[b + k0 + g];
                    cxs[g] = v.x; cys[g] = v.y;
                }
            }
            uint32_t bits = 0;
#pragma unroll
            for (uint32_t g = 0; g < PER; g++) {
                const float dx = Ai.x - cxs[g], dy = Ai.y - cys[g];
                const float r2 = dx * dx + dy * dy;
                bits |= (r2 < A.s2 ? 1u : 0u) << g;
            }
            m |= bits << (k0 & 31u);
        }
        mk[dr] = m & (len >= 32u ? 0xffffffffu : ((1u << len) - 1u));
    }
    float rho = 0.f;
#pragma unroll
    for (int dr = 0; dr < 3; dr++) {
        uint32_t m = mk[dr];
        const uint32_t b = rb[dr];
        while (m) {
            uint32_t bb[4];
            bool v[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                v[k] = m != 0u;
                bb[k] = v[k] ? (uint32_t)__ffs(m) - 1u : bb[0];
                m &= m - 1u;
            }
            float2 R[4];
#pragma unroll
            for (int k = 0; k < 4; k++) R[k] = A.xy[b + bb[k]];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const float dx = Ai.x - R[k].x, dy = Ai.y - R[k].y;
                const float r2 = dx * dx + dy * dy;
                if (v[k]) rho += A.mass * w_spline(A, r2);
            }
        }
    }
    A.nl_a[i] = make_uint4(mk[0], mk[1], mk[2], (uint32_t)(__popc(mk[0]) + __popc(mk[1]) + __popc(mk[2])));
    A.rho[i] = rho;
}

struct Scene {
    Grid g;
    std::vector<float4> pm;
    std::vector<uint32_t> cell_start, orig;
    double candidates;
};

static Scene make_scene(const std::vector<float>& x, const std::vector<float>& y, float cs, int R, float mass, float h)
{
    const uint32_t n = (uint32_t)x.size();
    float mnx = 1e9f, mny = 1e9f, mxx = -1e9f, mxy = -1e9f;
    for (uint32_t i = 0; i < n; i++) {
        mnx = std::min(mnx, x[i]); mny = std::min(mny, y[i]); mxx = std::max(mxx, x[i]); mxy = std::max(mxy, y[i]);
    }
    Scene S;
    S.g.cs = cs;
    S.g.minx = (int)floorf(mnx / cs) - 1;
    S.g.miny = (int)floorf(mny / cs) - 1;
    S.g.sx = (int)floorf(mxx / cs) + 2 - S.g.minx;
    S.g.sy = (int)floorf(mxy / cs) + 2 - S.g.miny;
    const uint32_t ncells = (uint32_t)S.g.sx * (uint32_t)S.g.sy;
    std::vector<uint32_t> key(n);
    S.orig.resize(n);
    for (uint32_t i = 0; i < n; i++) {
        const int cx = (int)floorf(x[i] / cs) - S.g.minx, cy = (int)floorf(y[i] / cs) - S.g.miny;
        key[i] = (uint32_t)cy * (uint32_t)S.g.sx + (uint32_t)cx;
        S.orig[i] = i;
    }
    std::stable_sort(S.orig.begin(), S.orig.end(), [&](uint32_t a, uint32_t b) { return key[a] < key[b]; });
    S.pm.resize(n);
    S.cell_start.assign(ncells + 1, 0);
    for (uint32_t s = 0; s < n; s++) {
        const uint32_t i = S.orig[s];
        S.pm[s] = make_float4(x[i], y[i], mass, h);
        S.cell_start[key[i] + 1]++;
    }
    for (uint32_t c = 0; c < ncells; c++) S.cell_start[c + 1] += S.cell_start[c];
    S.candidates = 0;
    for (uint32_t s = 0; s < n; s++) {
        const uint32_t k = key[S.orig[s]];
        const int cx = (int)(k % (uint32_t)S.g.sx), cy = (int)(k / (uint32_t)S.g.sx);
        for (int yy = std::max(cy - R, 0); yy <= std::min(cy + R, S.g.sy - 1); yy++)
            S.candidates += S.cell_start[(uint32_t)yy * S.g.sx + std::min(cx + R + 1, S.g.sx)] - S.cell_start[(uint32_t)yy * S.g.sx + std::max(cx - R, 0)];
    }
    return S;
}

template <class T>
static T* dev_alloc(size_t count)
{
    void* p;
    CHECK(hipMalloc(&p, count * sizeof(T)));
    CHECK(hipMemset(p, 0, count * sizeof(T)));
    return (T*)p;
}

int main(int argc, char** argv)
{
    const int side = argc > 1 ? atoi(argv[1]) : 1024;
    const float jitter = argc > 2 ? (float)atof(argv[2]) : 0.15f;
    const int reps = argc > 3 ? atoi(argv[3]) : 50;
    const uint32_t n = (uint32_t)side * (uint32_t)side;
    const float d = 1.f / 1024.f, rho0 = 1.f, mass = rho0 * d * d;
    const float h = 1.9f * sqrtf((mass / rho0) * 0.318309873342514038086f);
    std::mt19937 rng(1234);
    std::uniform_real_distribution<float> U(-jitter * d, jitter * d);
    std::vector<float> x(n), y(n);
    for (int r = 0; r < side; r++)
        for (int c = 0; c < side; c++) {
            x[(size_t)r * side + c] = -1.9995f + c * d + U(rng);
            y[(size_t)r * side + c] = -0.9995f + r * d + U(rng);
        }
    setvbuf(stdout, nullptr, _IOLBF, 0);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    std::vector<float> rho_ref(n, 0.f);
    std::vector<uint32_t> cnt_ref(n, 0u);
    printf("n = %u, jitter %.2f d, h = %.4f d\n", n, jitter, h / d);
    printf("| grid | cells | candidates / particle | accepted / particle | rows > 32 | list word | trips of | us per launch (HIP events, %d back to back) | max rel diff of rho vs the first row | counts equal |\n|---|---|---|---|---|---|---|---|---|---|\n", reps);
    for (int R = 1; R <= 2; R++) {
        const float cs = 2.f * h / (float)R;
        Scene S = make_scene(x, y, cs, R, mass, h);
        BArgs A{};
        A.n = n;
        A.nblocks = (n + 255) / 256;
        A.g = S.g;
        A.h = h;
        A.nf = 10.f / (7.f * 3.14159274101257324219f * (h * h));
        A.inv2h = 1.f / (2.f * h);
        A.s2 = (2.f * h) * (2.f * h);
        A.mass = mass;
        uint32_t* cs_d = dev_alloc<uint32_t>(S.cell_start.size());
        CHECK(hipMemcpy(cs_d, S.cell_start.data(), S.cell_start.size() * 4, hipMemcpyHostToDevice));
        float4* pm_d = dev_alloc<float4>(n + 16);
        CHECK(hipMemcpy(pm_d, S.pm.data(), (size_t)n * 16, hipMemcpyHostToDevice));
        std::vector<float2> xyh(n + 16, make_float2(1e9f, 1e9f));
        for (uint32_t q = 0; q < n; q++) xyh[q] = make_float2(S.pm[q].x, S.pm[q].y);
        float2* xy_d = dev_alloc<float2>(n + 16);
        CHECK(hipMemcpy(xy_d, xyh.data(), (size_t)(n + 16) * 8, hipMemcpyHostToDevice));
        A.xy = xy_d;
        A.cell_start = cs_d;
        A.pm = pm_d;
        A.nl_a = dev_alloc<uint4>(n);
        A.nl_b = dev_alloc<uint2>(n);
        A.rho = dev_alloc<float>(n);
        A.over = dev_alloc<uint32_t>(1);
        const uint32_t grid = ((A.nblocks + 7) / 8) * 8;
        for (int trip : {4, 2, 104, 108, 204, 208}) {
            if (R != 1 && trip > 100) continue;
            auto launch = [&]() {
                if (trip == 104) hipLaunchKernelGGL((k_build_flat<0>), dim3(grid), dim3(256), 0, 0, A);
                else if (trip == 108) hipLaunchKernelGGL((k_build_flat<1>), dim3(grid), dim3(256), 0, 0, A);
                else if (trip == 204) hipLaunchKernelGGL((k_build_2phase<0>), dim3(grid), dim3(256), 0, 0, A);
                else if (trip == 208) hipLaunchKernelGGL((k_build_2phase<1>), dim3(grid), dim3(256), 0, 0, A);
                else if (R == 1 && trip == 4) hipLaunchKernelGGL((k_build<1, 4>), dim3(grid), dim3(256), 0, 0, A);
                else if (R == 1) hipLaunchKernelGGL((k_build<1, 2>), dim3(grid), dim3(256), 0, 0, A);
                else if (trip == 4) hipLaunchKernelGGL((k_build<2, 4>), dim3(grid), dim3(256), 0, 0, A);
                else hipLaunchKernelGGL((k_build<2, 2>), dim3(grid), dim3(256), 0, 0, A);
            };
            CHECK(hipMemset(A.over, 0, 4));
            launch();
            CHECK(hipDeviceSynchronize());
            std::vector<float> rho(n);
            std::vector<uint4> na(n);
            std::vector<uint2> nb(n);
            uint32_t over = 0;
            CHECK(hipMemcpy(rho.data(), A.rho, (size_t)n * 4, hipMemcpyDeviceToHost));
            CHECK(hipMemcpy(na.data(), A.nl_a, (size_t)n * 16, hipMemcpyDeviceToHost));
            CHECK(hipMemcpy(nb.data(), A.nl_b, (size_t)n * 8, hipMemcpyDeviceToHost));
            CHECK(hipMemcpy(&over, A.over, 4, hipMemcpyDeviceToHost));
            double err = 0, acc = 0;
            bool same = true;
            for (uint32_t s = 0; s < n; s++) {
                const uint32_t i = S.orig[s];
                const uint32_t c = R == 1 ? na[s].w : nb[s].y;
                acc += c;
                if (R == 1 && trip == 4) {
                    rho_ref[i] = rho[s];
                    cnt_ref[i] = c;
                } else {
                    err = std::max(err, (double)fabsf(rho[s] - rho_ref[i]) / (double)rho_ref[i]);
                    same = same && c == cnt_ref[i];
                }
            }
            CHECK(hipEventRecord(e0, 0));
            for (int r = 0; r < reps; r++) launch();
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipDeviceSynchronize());
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            printf("| %s | %d x %d | %.1f | %.2f | %u | %d B | %d | %.2f | %.2e | %s |\n", R == 1 ? "cell = 2 h, 3 x 3" : "cell = h, 5 x 5", S.g.sx, S.g.sy,
                   S.candidates / n, acc / n, over, R == 1 ? 16 : 24, trip, ms * 1e3 / reps, err, same ? "yes" : "NO");
        }
        CHECK(hipFree(cs_d));
        CHECK(hipFree(pm_d));
        CHECK(hipFree(A.nl_a));
        CHECK(hipFree(A.nl_b));
        CHECK(hipFree(A.rho));
        CHECK(hipFree(A.over));
    }
    return 0;
}
