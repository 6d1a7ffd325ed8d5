// Neighbour sweeps of the SPH step as hand-written HIP kernels for gfx950 (MI355X).
//
// Execution model (MI355X-first, not a translation of the reference's rayon loops over
// Vec<Vec<u32>> neighbour lists):
//   * particles live in a cell-sorted SoA (x,y,m,h packed as one float4 => one 16-B access per
//     particle); thread i of a launch owns sorted particle i, so every per-particle read and
//     write is coalesced and every wave is 64 consecutive, spatially adjacent particles;
//   * the DENSITY sweep (first of the step) walks the 3 rows x 3 cells around each particle --
//     three contiguous index ranges, because cells are ordered x-fastest -- and applies the
//     reference's neighbour predicate
//         |x_ij|^2 < ((h_i + h_j) * 0.5 * 2)^2        (neighborhood_search.rs:143-146)
//     with exactly the reference's operations, so the accepted set IS the reference's list.  It
//     records the accepted candidates as three 32-bit row masks (bit b of row r = candidate b of
//     that row's contiguous range) plus count and flags: one uint4 per particle, one coalesced
//     1-KB load per wave in every later sweep;
//   * all later sweeps of the step (positions do not change until the final integrate) replay
//     that word: per row, up to 4 set bits per trip -> four independent gathers of the neighbour
//     record and its per-neighbour payload -> four pair evaluations -- 13 instead of ~38
//     candidate evaluations per particle, 4-way ILP, no barriers, no LDS, 8 waves per SIMD.
//     Since round 4 the gradient sweeps (both sweeps of a Jacobi iteration, the source terms) replay
//     the same list in a second form the density sweep writes beside the word: 16-bit offsets j - i,
//     flat, four per 8-byte group (k_sweep_off below): no decoding per slot, no cell index and no
//     cell-table loads at the head of the sweep;
//     Neighbours of neighbouring lanes are adjacent in memory (cell-sorted order), so the gathers
//     hit the same few cache lines per wave; per-neighbour derived quantities (p_j/rho_j^2,
//     m_j/rho_j) are produced once per particle by the sweep that owns them, not once per pair;
//   * every particle writes only its own outputs (gather-only, no atomics); the visiting order
//     (rows bottom-to-top, sorted index ascending) is identical on both paths, so results do not
//     depend on which path ran;
//   * multi-resolution scenes sort by the SMALL particles' grid; a particle's stencil is as wide
//     as the largest h that can be around it (TileP, sph_device.h), particles with a wide stencil
//     or crowded rows record explicit index lists (nlx) instead of masks, and only those with
//     more than NLX_CAP neighbours walk their candidates in every sweep;
//   * the level estimation (EmptyAngle surface detection, level-set propagation, smoothing) runs
//     in the same skeleton on its own extended-range lists.
//
// Each Op below cites the reference sweep it implements (src/simulation/simulation.rs and
// src/simulation/boundary_handler/sdf_boundary_handler/boundary_winchenbach2020.rs).
#include <type_traits>
#include "sph_internal.hpp"

#ifndef SWEEP_THREADS
#define SWEEP_THREADS 256   // measured (profiles/r2_variants.md, last table): 128 and 1024 are slower (0.662 / 0.637 vs 0.619 ms/step), 512 is the same
#endif
#ifndef SPH_TILE_DEFAULT
#define SPH_TILE_DEFAULT 0
#endif
#ifndef SPH_FORCE_IDX
#define SPH_FORCE_IDX 0
#endif
// SPH_OFF32 (round 5): the record gathers of the uniform-h Jacobi sweeps address their 16-byte records through a 32-bit BYTE offset from
// the (uniform) base pointer -- `global_load_dwordx4 v, v_off, s[base]` -- instead of a 64-bit index: the zero-extension (one v_mov per
// gather) and the 64-bit shift-add go, and sweep A's register count drops from 80 to 68.  Valid up to 2^28 particles (4 GB per record
// array): larger contexts do not build offset lists (sweeps_want_offset_lists) and replay their mask words through the 64-bit form.
#ifndef SPH_OFF32
#define SPH_OFF32 1
#endif
__device__ __forceinline__ float4 load_record(const float4* __restrict__ base, uint32_t j)
{
#if SPH_OFF32
    return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base) + (uint32_t)(j << 4));
#else
    return base[j];
#endif
}

// Neighbour list word (one uint4 = 16 B per particle, coalesced 1 KB per wave):
//   x, y, z : accepted-candidate bit masks of the three cell rows cy-1, cy, cy+1.  Bit b of row r
//             is sorted particle  cell_start[(cy+r-1)*sx + cx-1] + b  (the row's candidate range
//             is contiguous because cells are ordered x-fastest); the bases are re-read from the
//             L2-resident cell table, which costs no HBM traffic;
//   w       : neighbour count (bits 0..15) | NL_OK   (mask list complete: 3x3 stencil, every row <= 32 candidates)
//                                          | NL_IDX  (explicit index list in nlx: particles whose stencil is
//                                                     wider than 3x3 cells or whose rows hold > 32 candidates --
//                                                     the interface particles of a multi-resolution scene)
//                                          | NL_WALL (particle has boundary terms: lambda != 0)
// Index list: group g (4 neighbour indices, one uint4) of particle i lives at nlx[g * n + i], so a wave reads
// one coalesced 1 KB line per trip.  Particles with more than NLX_CAP neighbours have neither flag and walk
// their candidates in every sweep.
#define NL_OK 0x80000000u
#define NL_WALL 0x40000000u
#define NL_IDX 0x20000000u
#define NLX_GROUPS 32
#define NLX_CAP (4 * NLX_GROUPS)

struct SweepCommon {
    GridP g;       // the grid the particles are sorted by (cell = support of the SMALLEST particle)
    TileP t;       // per-tile bound on the neighbours' h => stencil radius of a particle (uniform scenes: 3x3)
    uint32_t n;
    uint32_t nblocks;
    const uint32_t* __restrict__ cell_start;
    uint4* __restrict__ nl;    // neighbour list words
    uint4* __restrict__ nlx;   // explicit index lists (nullptr in uniform scenes)
    const uint8_t* __restrict__ owned;  // slab decomposition: ghost lanes idle (their values come from their owner)
    const uint8_t* __restrict__ ring1;  // ... except, in RING1 ops, the ghosts within one support radius of the cut
    // slab decomposition, a sweep in two launches (the ghost exchange runs under the first): part 1 = every lane whose `edge`
    // byte is 0 (owned, no ghost within reach); part 2 = the listed slots (the halo members, then the ghosts); 0 = one launch
    int part;
    const uint8_t* __restrict__ edge;
    const uint32_t* __restrict__ elist_a;
    const uint32_t* __restrict__ elist_b;
    uint32_t n_ea, n_eb;
    // relative-offset lists (k_sweep_off; nullptr: not built this step)
    const uint2* __restrict__ nloff;
    const uint8_t* __restrict__ nlh;
    uint2* __restrict__ nloff_out;      // BUILD sweep of a uniform scene: write them (emit_offset_list), else nullptr
    uint8_t* __restrict__ nlh_out;
    // Profiler mode 3 (else nullptr): this launch's timestamp slot -- ts[0] receives the earliest block start, ts[TS_RING] the latest
    // block end, on the device's constant 100 MHz clock (sph_internal.hpp)
    unsigned long long* ts;
};
#define SPH_TS_RING 65536   // == Profiler::TS_RING

// first block start / last block end of a launch on the device clock (launch-uniform branch: one scalar compare when off)
__device__ __forceinline__ void sweep_stamp(unsigned long long* ts, bool end)
{
    // start: the first block of every XCD (blocks are dispatched in index order, block b to XCD b % 8); end: every block, once
    // its last wave is through -- stamped by thread 0 behind a block barrier.  (A stamp per wave -- 32 k same-address atomics per
    // launch -- made the sweeps 2.5 us longer.)
    if (!ts) return;
    if (!end) {
        if (blockIdx.x < 8u && threadIdx.x == 0) atomicMin(ts, (unsigned long long)wall_clock64());
        return;
    }
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(ts + SPH_TS_RING, (unsigned long long)wall_clock64());
}

__device__ __forceinline__ void raise_error(DeviceStatus* st, uint32_t code, uint32_t info)
{
    if (atomicCAS(&st->error, 0u, code) == 0u) st->info = info;
}

// a solve's new pressure of particle i: p, and p / rho^2 for the neighbours' sweep A -- as a field of its own, or (uniform-h scenes
// on one context, OpPressureAccelU) inside the 16-byte record {x, y, p / rho^2, p} that sweep gathers whole
__device__ __forceinline__ void store_pressure(float* __restrict__ p_out, float* __restrict__ pterm_out, float4* __restrict__ rec_out, uint32_t i,
                                               const float4& Ai, float p, float pt)
{
    p_out[i] = p;
    if (rec_out) rec_out[i] = make_float4(Ai.x, Ai.y, pt, p);
    else pterm_out[i] = pt;
}

// ------------------------------------------------------------------------------------------------
// generic sweep.  BUILD = true: candidate walk + list recording (density sweep).
//
// Op interface:
//   typedef Math;  Math m;                      math policy (sph_device.h)
//   bool   skip()                               launch-uniform early exit (speculative Jacobi iterations)
//   bool   lane_skip(i)                         particle i has nothing to do in this sweep (level-set propagation)
//   float4 loadA(j)                             (x, y, m, h) of particle j
//   NB     nb(acc, j, Aj)                       per-neighbour payload of particle j
//   void   init(acc)                            lane-independent state (e.g. which pressure buffer is current)
//   void   begin(acc, i, Ai)                    load own data, zero accumulators
//   void   pair(acc, Aj, NBj, dx, dy, r2, hij)  one accepted pair
//   void   finish(acc, i, Ai, wall)             boundary terms (only if `wall`), outputs, guards
//   void   epilogue(acc, active, blk)           optional block-level tail (HAS_EPILOGUE)
//   float  krange()                             neighbour predicate |x_ij| < h_ij * krange(): 2 (SPH support), or the
//                                               extended range of the level estimation (EXTENDED ops, own list arrays)
//
// Measured negatives (MI355X, N = 1M; DESIGN.md "What bounds the sweeps"): staging the wave's three row ranges
// in LDS (random ds_read_b128 costs what the L1 gathers cost, and the LDS cuts occupancy: Jacobi 38.9 vs 28.5 us);
// flattening the three rows into one batch loop; FMA / packed-f32 pair math (fewer VALU instructions, same time).
// ------------------------------------------------------------------------------------------------
// SPH_DBG_NOGATHER (measurement only, results are garbage): every lane "gathers" one of the first 64 records, i.e. the loads
// cost what a broadcast costs -- what is left of a sweep's time is its instruction issue
#ifdef SPH_DBG_NOGATHER
#define SPH_GIDX(J) ((J) & 63u)
#else
#define SPH_GIDX(J) (J)
#endif
// SPH_DBG_NOMATH (measurement only): the pair bodies are never executed (r2 is never negative) -- what is left is the loop
// skeleton, the mask decoding and the position gathers
#ifdef SPH_DBG_NOMATH
#define SPH_DBG_PAIR_GATE &&r2 < -1.f
#else
#define SPH_DBG_PAIR_GATE
#endif
#define SPH_FETCH(J, AOUT, NOUT)                                                          \
    const float4 AOUT = op.loadA(SPH_GIDX(J));                                            \
    const typename Op::NB NOUT = op.nb(acc, SPH_GIDX(J), AOUT);
#define SPH_PAIR(AJ, NJ, ON)                                                              \
    {                                                                                     \
        const float dx = Ai.x - AJ.x, dy = Ai.y - AJ.y;                                   \
        const float r2 = dx * dx + dy * dy;                                               \
        const float hij = Math::UNIFORM ? op.m.h : (Ai.w + AJ.w) * 0.5f;                  \
        if ((ON)SPH_DBG_PAIR_GATE) op.pair(acc, AJ, NJ, dx, dy, r2, hij);                 \
    }

// ---- list replay, row masks: per row up to SPH_TRIP set bits per trip (that many independent fetches in flight) ----------
// A wave runs a trip as long as ANY of its lanes has bits left in the row, and every trip issues SPH_TRIP pair slots for all
// lanes: with 3-6 neighbours per row and lane, trips of 4 execute 8 slots per row where trips of 2 execute 6.
#ifndef SPH_TRIP
#define SPH_TRIP 4
#endif
template <class Op>
__device__ __forceinline__ void replay_masks(const Op& op, typename Op::Acc& acc, const float4 Ai, const uint32_t (&rb)[3], const uint4 lw)
{
    typedef typename Op::Math Math;
    const uint32_t masks[3] = {lw.x, lw.y, lw.z};
#pragma unroll
    for (int dr = 0; dr < 3; dr++) {
        uint32_t mk = masks[dr];
        const uint32_t base = rb[dr];
        while (mk) {
            const uint32_t b0 = __ffs(mk) - 1;
            mk &= mk - 1;
            const bool v1 = mk != 0;
            const uint32_t b1 = v1 ? __ffs(mk) - 1 : b0;
            mk &= mk - 1;
#if SPH_TRIP == 4
            const bool v2 = mk != 0;
            const uint32_t b2 = v2 ? __ffs(mk) - 1 : b0;
            mk &= mk - 1;
            const bool v3 = mk != 0;
            const uint32_t b3 = v3 ? __ffs(mk) - 1 : b0;
            mk &= mk - 1;
            const uint32_t j0 = base + b0, j1 = base + b1, j2 = base + b2, j3 = base + b3;
            SPH_FETCH(j0, A0, N0)
            SPH_FETCH(j1, A1, N1)
            SPH_FETCH(j2, A2, N2)
            SPH_FETCH(j3, A3, N3)
            SPH_PAIR(A0, N0, true)
            SPH_PAIR(A1, N1, v1)
            SPH_PAIR(A2, N2, v2)
            SPH_PAIR(A3, N3, v3)
#else
            const uint32_t j0 = base + b0, j1 = base + b1;
            SPH_FETCH(j0, A0, N0)
            SPH_FETCH(j1, A1, N1)
            SPH_PAIR(A0, N0, true)
            SPH_PAIR(A1, N1, v1)
#endif
        }
    }
}

// ---- list replay, row masks, FLAT form (SPH_FLAT16): the three masks are decoded first, into up to 16 neighbour indices held in
// registers (row-major, ascending: the order of replay_masks, so the sums are bit-identical), then trips of 4 run over that flat
// sequence: 16 pair slots for up to 16 neighbours where the per-row trips spend 20-24 (every row pads its last trip).  What is
// left in the masks after 16 (compressed regions) is replayed one neighbour at a time.
// MEASURED NEGATIVE for the headline, kept as a switch (profiles/r3_variants.md): in the laboratory the rest lattice is 20.3 vs 20.1 us and
// a jittered scene 23.4 vs 26.1 us; in the product the driver window of configs[1] (a compressed near-lattice column) LOSES 5 %
// (OpJacobiU 25.2 vs 23.2 us, 1.265 vs 1.204 ms/step: the 16-slot select chain costs what the padding slots cost), the settled
// scene 100 steps later gains 3 % of GPU time (0.522 vs 0.533 ms), configs[3] loses 2 %.
#ifndef SPH_FLAT16
#define SPH_FLAT16 0
#endif
#if SPH_FLAT16   // (laboratory form: compiled only when a variant build asks for it, scripts/variants)
template <class Op>
__device__ __forceinline__ void replay_masks_flat(const Op& op, typename Op::Acc& acc, const float4 Ai, const uint32_t rb0, const uint32_t rb1,
                                                  const uint32_t rb2, const uint4 lw, const uint32_t i)
{
    typedef typename Op::Math Math;
    uint32_t m0 = lw.x, m1 = lw.y, m2 = lw.z;
    const uint32_t cnt = (uint32_t)(__popc(m0) + __popc(m1) + __popc(m2));
    uint32_t j[16];
#pragma unroll
    for (int s = 0; s < 16; s++) {
        const bool u0 = m0 != 0u, u1 = m1 != 0u;
        const uint32_t m = u0 ? m0 : (u1 ? m1 : m2);
        const uint32_t bs = u0 ? rb0 : (u1 ? rb1 : rb2);
        j[s] = m ? bs + (uint32_t)__ffs(m) - 1u : i;   // (an unused slot fetches the particle's own record: a line the wave holds)
        const uint32_t mm = m & (m - 1u);
        m0 = u0 ? mm : m0;
        m1 = (!u0 && u1) ? mm : m1;
        m2 = (!u0 && !u1) ? mm : m2;
    }
#pragma unroll
    for (uint32_t s0 = 0; s0 < 16u; s0 += 4u) {
        if (!__any(s0 < cnt)) break;   // (wave-uniform: the longest list of the wave decides the number of trips)
        SPH_FETCH(j[s0], A0, N0)
        SPH_FETCH(j[s0 + 1], A1, N1)
        SPH_FETCH(j[s0 + 2], A2, N2)
        SPH_FETCH(j[s0 + 3], A3, N3)
        SPH_PAIR(A0, N0, s0 < cnt)
        SPH_PAIR(A1, N1, s0 + 1u < cnt)
        SPH_PAIR(A2, N2, s0 + 2u < cnt)
        SPH_PAIR(A3, N3, s0 + 3u < cnt)
    }
    // more than 16 neighbours (compressed regions): what is left in the masks goes through the per-row trips
    if (__any((m0 | m1 | m2) != 0u)) {
        const uint32_t rbs[3] = {rb0, rb1, rb2};
        replay_masks(op, acc, Ai, rbs, make_uint4(m0, m1, m2, 0u));
    }
}

#endif   // SPH_FLAT16

// ---- list replay, explicit indices: one coalesced uint4 (4 neighbours) per trip ---------------------------
template <class Op>
__device__ __forceinline__ void replay_indices(const Op& op, typename Op::Acc& acc, const float4 Ai, const uint32_t i, const uint32_t cnt,
                                               const uint4* __restrict__ nlx, const uint32_t n)
{
    typedef typename Op::Math Math;
    // (the index quad of trip k + 1 is requested before trip k's gathers: two dependent round trips per trip -- indices, then
    //  records -- were the time of a latency-bound sweep: configs[1] + EmptyAngle 2.07 -> 2.00 ms/step.  Requesting the RECORDS of
    //  trip k + 1 ahead as well measured 2.03-2.05: recorded in profiles/r3_variants.md, not kept)
    uint4 qn = cnt ? nlx[i] : make_uint4(0, 0, 0, 0);
    for (uint32_t k = 0; k < cnt; k += 4) {
        const uint4 q = qn;
        if (k + 4 < cnt) qn = nlx[(size_t)((k >> 2) + 1) * n + i];
        const bool v1 = k + 1 < cnt, v2 = k + 2 < cnt, v3 = k + 3 < cnt;
        const uint32_t j0 = q.x, j1 = v1 ? q.y : q.x, j2 = v2 ? q.z : q.x, j3 = v3 ? q.w : q.x;
        SPH_FETCH(j0, A0, N0)
        SPH_FETCH(j1, A1, N1)
        SPH_FETCH(j2, A2, N2)
        SPH_FETCH(j3, A3, N3)
        SPH_PAIR(A0, N0, true)
        SPH_PAIR(A1, N1, v1)
        SPH_PAIR(A2, N2, v2)
        SPH_PAIR(A3, N3, v3)
    }
}

// recorder of an explicit index list (BUILD): 4 accepted indices are collected in registers, then stored as one uint4
struct IdxRecorder {
    uint4 cur;
    __device__ __forceinline__ void push(uint32_t j, uint32_t nacc, uint4* __restrict__ nlx, uint32_t n, uint32_t i)
    {
        const uint32_t s = nacc & 3u;
        cur.x = s == 0u ? j : cur.x;
        cur.y = s == 1u ? j : cur.y;
        cur.z = s == 2u ? j : cur.z;
        cur.w = s == 3u ? j : cur.w;
        if (s == 3u && nacc < NLX_CAP) nlx[(size_t)(nacc >> 2) * n + i] = cur;
    }
    __device__ __forceinline__ void flush(uint32_t nacc, uint4* __restrict__ nlx, uint32_t n, uint32_t i)
    {
        if ((nacc & 3u) && nacc < NLX_CAP) nlx[(size_t)(nacc >> 2) * n + i] = cur;
    }
};

// one row of the candidate walk: exact reference predicate, 4 candidates per trip.  MASKS: accepted candidates set
// bits of `mk` (bit = j - b); REC: accepted candidates are appended to the explicit index list.
#define SPH_FETCH_W(J, AOUT, NOUT)                                                        \
    const float4 AOUT = op.loadA(J);                                                      \
    const typename Op::NB NOUT = op.nb(acc, J, AOUT);
template <class Op, bool MASKS, bool REC>
__device__ __forceinline__ void walk_row(const Op& op, typename Op::Acc& acc, const float4 Ai, const uint32_t b, const uint32_t e, uint32_t& mk,
                                         uint32_t& nacc, IdxRecorder& rec, uint4* __restrict__ nlx, const uint32_t n, const uint32_t i)
{
    typedef typename Op::Math Math;
    for (uint32_t j = b; j < e; j += 4) {
        const bool v1 = j + 1 < e, v2 = j + 2 < e, v3 = j + 3 < e;
        const uint32_t j1 = v1 ? j + 1 : j, j2 = v2 ? j + 2 : j, j3 = v3 ? j + 3 : j;
        SPH_FETCH_W(j, A0, N0)
        SPH_FETCH_W(j1, A1, N1)
        SPH_FETCH_W(j2, A2, N2)
        SPH_FETCH_W(j3, A3, N3)
#define SPH_CAND(JJ, AJ, NJ, VALID)                                                       \
    {                                                                                     \
        /* neighbour predicate, exactly the reference's operations (no FMA, strict <) */  \
        const float dx = Ai.x - AJ.x, dy = Ai.y - AJ.y;                                   \
        const float r2 = dx * dx + dy * dy;                                               \
        const float hij = Math::UNIFORM ? op.m.h : (Ai.w + AJ.w) * 0.5f;                  \
        const float s = hij * op.krange();                                                \
        if ((VALID) && r2 < s * s) {                                                      \
            op.pair(acc, AJ, NJ, dx, dy, r2, hij);                                        \
            if (MASKS) {                                                                  \
                const uint32_t bit = (JJ) - b;                                            \
                if (bit < 32u) mk |= 1u << bit;                                           \
            }                                                                             \
            if (REC) rec.push(JJ, nacc, nlx, n, i);                                       \
            nacc++;                                                                       \
        }                                                                                 \
    }
        SPH_CAND(j, A0, N0, true)
        SPH_CAND(j1, A1, N1, v1)
        SPH_CAND(j2, A2, N2, v2)
        SPH_CAND(j3, A3, N3, v3)
#undef SPH_CAND
    }
}
#undef SPH_FETCH_W
#undef SPH_FETCH
#undef SPH_PAIR

// optional Op hook `float2 cell_pos(i, Ai)`: the position the particle was SORTED by, when loadA() returns something else
// (level estimation after advection: geometry from the advected positions, cells and list words from the pre-step ones)
template <class Op, class = void>
struct OpCellPos {
    static __device__ __forceinline__ float2 get(const Op&, uint32_t, const float4& Ai) { return make_float2(Ai.x, Ai.y); }
};
template <class Op>
struct OpCellPos<Op, std::void_t<decltype(&Op::cell_pos)>> {
    static __device__ __forceinline__ float2 get(const Op& op, uint32_t i, const float4& Ai) { return op.cell_pos(i, Ai); }
};

// Ops with `static constexpr bool RING1 = true` (density BUILD, pressure acceleration) also run on the FIRST ghost ring of a
// slab: those ghosts then carry a neighbour list and a locally computed a^p, and the Jacobi iteration needs ONE neighbour
// exchange (p / rho^2) instead of two (+ a^p).
template <class Op, class = void>
struct OpRing1 : std::false_type {};
template <class Op>
struct OpRing1<Op, std::void_t<decltype(Op::RING1)>> : std::bool_constant<Op::RING1> {};

// Ops with `static constexpr bool TILE = true` may run through the LDS-staged form (k_sweep_tile): they read the particle's own
// cell from loadA(i) and touch neighbours only through loadA / nb / pair.
template <class Op, class = void>
struct OpTile : std::false_type {};
template <class Op>
struct OpTile<Op, std::void_t<decltype(Op::TILE)>> : std::bool_constant<Op::TILE> {};

// Ops with `static constexpr bool SECOND_PASS = true`: a lane for which second_wanted(acc) holds after its sweep runs the list again
// with the op `second()` returns (type Op::Second) -- see OpLevelPropagate.
template <class Op, class = void>
struct OpSecondPass : std::false_type {};
template <class Op>
struct OpSecondPass<Op, std::void_t<decltype(Op::SECOND_PASS)>> : std::bool_constant<Op::SECOND_PASS> {};

// optional Op hook `bool prologue(raw_block)`: block-uniform work at the start of the launch (the Jacobi stop decision, taken by
// block 0 of the NEXT pressure-acceleration sweep while the other blocks already sweep); returns true to leave.  Default: skip().
template <class Op, class = void>
struct OpPrologue {
    static __device__ __forceinline__ bool run(const Op& op, uint32_t) { return op.skip(); }
};
template <class Op>
struct OpPrologue<Op, std::void_t<decltype(&Op::prologue)>> {
    static __device__ __forceinline__ bool run(const Op& op, uint32_t raw_block) { return op.prologue(raw_block); }
};

// SPH_BUILD_2PHASE (variant, scripts/variants): the BUILD sweep of a uniform scene in two phases, in registers -- the neighbour
// predicate over a row's candidates, branch-free, into the row mask (4 unclamped loads per trip: the records behind a row's end
// are other particles' or the allocation's slack, their bits are masked by the validity test); then the op's pair() over the
// accepted bits only, through the same replay the later sweeps use.  Same visiting order and arithmetic as the walk.
#ifndef SPH_BUILD_2PHASE
#define SPH_BUILD_2PHASE 0
#endif
#if SPH_BUILD_2PHASE   // (laboratory form: compiled only when a variant build asks for it, scripts/variants)
template <class Op>
__device__ __forceinline__ uint32_t predicate_row(const Op& op, const float4 Ai, const uint32_t b, const uint32_t e, const float s2)
{
    uint32_t m = 0u;
    for (uint32_t j = b; j < e; j += 4) {
        const float4 A0 = op.loadA(j), A1 = op.loadA(j + 1), A2 = op.loadA(j + 2), A3 = op.loadA(j + 3);
#define SPH_PRED(AJ, K)                                                                        \
    {                                                                                          \
        const float dx = Ai.x - AJ.x, dy = Ai.y - AJ.y;                                        \
        const float r2 = dx * dx + dy * dy;                                                    \
        m |= ((j + K < e && r2 < s2) ? 1u : 0u) << (j + K - b);                                \
    }
        SPH_PRED(A0, 0u)
        SPH_PRED(A1, 1u)
        SPH_PRED(A2, 2u)
        SPH_PRED(A3, 3u)
#undef SPH_PRED
    }
    return m;
}

#endif   // SPH_BUILD_2PHASE

// BUILD sweep of a uniform scene whose solves run on records: the particle's mask word as 16-bit relative offsets j - i for k_sweep_off
// (described there), written at the end of the density sweep -- it holds the row bases and the masks in registers; a kernel of its
// own re-read 32 B and took 18 us for what costs the BUILD sweep ~2.  The halfwords go through a column of LDS per lane ([slot][lane]:
// conflict-free; a slot number that is only known at run time is an address there, a select chain over twelve registers otherwise).
#define NLOFF_GROUPS 6                  // groups of four offsets = one trip of k_sweep_off = one 8-byte load
#define NLOFF_SLOTS (4 * NLOFF_GROUPS)
#define NLH_OK 0x20u                    // header byte: bits 0..4 count, NLH_OK, NLH_WALL
#define NLH_WALL 0x40u
__device__ __forceinline__ void emit_offset_list(uint2* __restrict__ nloff, uint8_t* __restrict__ nlh, const uint32_t n, const uint32_t i, const uint4 lw,
                                                 const uint32_t (&rb)[3])
{
    __shared__ uint16_t s_half[NLOFF_SLOTS][SWEEP_THREADS];
    uint32_t head = (lw.w & NL_WALL) ? NLH_WALL : 0u;
    if (!(lw.w & NL_OK)) {
        nlh[i] = (uint8_t)head;
        return;
    }
    uint32_t cnt = 0;
    bool fits = true;
    const uint32_t masks[3] = {lw.x, lw.y, lw.z};
#pragma unroll
    for (int dr = 0; dr < 3; dr++) {
        uint32_t mk = masks[dr];
        if (dr == 1) {   // the particle itself is on the reference's list; its pair term in the gradient sweeps is zero
            const uint32_t sb = i - rb[1];
            if (sb < 32u) mk &= ~(1u << sb);
        }
        while (mk) {
            const uint32_t b = (uint32_t)__ffs(mk) - 1u;
            mk &= mk - 1u;
            const int d = (int)(rb[dr] + b) - (int)i;
            fits = fits && d >= -32768 && d <= 32767 && cnt < (uint32_t)NLOFF_SLOTS;
            if (cnt < (uint32_t)NLOFF_SLOTS) s_half[cnt][threadIdx.x] = (uint16_t)d;
            cnt++;
        }
    }
    if (fits) head |= NLH_OK | cnt;
    nlh[i] = (uint8_t)head;
    if (!fits) return;
    // (the sweep reads groups 0..2 of every list before it knows the count: they are always written -- zeros = the particle itself
    //  behind the count; a later group only when the list reaches it, and the sweep does not look at it otherwise)
#pragma unroll
    for (int g = 0; g < NLOFF_GROUPS; g++) {
        if (!(g < 3 || cnt > (uint32_t)(4 * g))) continue;
        uint32_t w[2];
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const uint32_t s0 = (uint32_t)(4 * g + 2 * k);
            const uint32_t lo = s0 < cnt ? (uint32_t)s_half[s0][threadIdx.x] : 0u, hi = s0 + 1u < cnt ? (uint32_t)s_half[s0 + 1u][threadIdx.x] : 0u;
            w[k] = lo | (hi << 16);
        }
        nloff[(size_t)g * n + i] = make_uint2(w[0], w[1]);
    }
}

// one particle of a sweep, every list form: mask word, explicit index list, candidate walk (3 x 3 cells or a wide stencil)
// (Ai, lw: the particle's record and list word, loaded by the caller BEFORE it looks at the slab flags -- one memory round trip
//  at the head of every wave instead of two)
template <class Op, bool BUILD>
__device__ __forceinline__ void sweep_particle(const Op& op, const SweepCommon& c, typename Op::Acc& acc, const uint32_t i, const float4 Ai, uint4 lw)
{
    typedef typename Op::Math Math;
    const GridP g = c.g;
    uint32_t rb_build[3] = {0u, 0u, 0u};   // (BUILD: the row bases of a 3 x 3 stencil, for emit_offset_list)
    op.begin(acc, i, Ai);
    // explicit index lists exist in multi-resolution scenes and for the extended-range lists of the level estimation
    // SPH_FORCE_IDX (variant): explicit index lists in uniform scenes too -- no mask decoding per neighbour slot, one more coalesced
    // 16-byte load per 4 neighbours
    constexpr bool IDX = !Math::UNIFORM || Op::EXTENDED || SPH_FORCE_IDX;
    if (!BUILD && IDX && (lw.w & NL_IDX)) {
        replay_indices(op, acc, Ai, i, lw.w & 0xffffu, c.nlx, c.n);
    } else {
        // own cell (the same IEEE expression the sort key was computed from)
        const float2 cp = OpCellPos<Op>::get(op, i, Ai);
        const int cx = (int)floorf(cp.x / g.cs) - g.minx;
        const int cy = (int)floorf(cp.y / g.cs) - g.miny;
        const bool walk = BUILD || !(lw.w & NL_OK);
        const int R = (!IDX || !walk) ? 1 : stencil_radius(g, c.t, Ai.w, cx, cy, op.krange());
        if (R == 1) {
            // 3 x 3 cells: three contiguous candidate ranges
            uint32_t rb[3], re[3];
            bool ok_list = true;
#pragma unroll
            for (int dr = 0; dr < 3; dr++) {
                const int yy = cy + dr - 1;
                const bool ok = yy >= 0 && yy < g.sy;
                const uint32_t base = (uint32_t)(ok ? yy : 0) * (uint32_t)g.sx;
                rb[dr] = ok ? c.cell_start[base + (uint32_t)max(cx - 1, 0)] : 0u;
                re[dr] = rb[dr];
                if (walk) {
                    re[dr] = ok ? c.cell_start[base + (uint32_t)min(cx + 2, g.sx)] : rb[dr];
                    ok_list = ok_list && !Op::EXTENDED && (re[dr] - rb[dr]) <= 32u;
                }
            }
            if (!walk) {
                // the particle itself is on its own list (the reference keeps it there), but in the gradient sweeps
                // its pair term is exactly zero (grad W(0) = 0, Q_i - Q_i = 0): drop its bit, which brings the middle
                // row of the rest lattice from 5 to 4 set bits = one trip instead of two
                if (Op::SKIP_SELF) {
                    const uint32_t sb = i - rb[1];
                    if (sb < 32u) lw.y &= ~(1u << sb);
                }
#if SPH_FLAT16
                if (Math::UNIFORM && !Op::EXTENDED) replay_masks_flat(op, acc, Ai, rb[0], rb[1], rb[2], lw, i);
                else
#endif
                    replay_masks(op, acc, Ai, rb, lw);
            } else {
                uint32_t mk[3] = {0u, 0u, 0u}, nacc = 0;
                IdxRecorder rec;
                rec.cur = make_uint4(0, 0, 0, 0);
                const bool rec_idx = BUILD && IDX && (!ok_list || (SPH_FORCE_IDX && !Op::EXTENDED));
#if SPH_BUILD_2PHASE
                if (BUILD && Math::UNIFORM && !Op::EXTENDED && !SPH_FORCE_IDX && ok_list) {
                    const float s = op.m.h * op.krange();
#pragma unroll
                    for (int dr = 0; dr < 3; dr++) mk[dr] = predicate_row(op, Ai, rb[dr], re[dr], s * s);
                    replay_masks(op, acc, Ai, rb, make_uint4(mk[0], mk[1], mk[2], 0u));
                    nacc = (uint32_t)(__popc(mk[0]) + __popc(mk[1]) + __popc(mk[2]));
                } else
#endif
                if (rec_idx) {
#pragma unroll
                    for (int dr = 0; dr < 3; dr++) walk_row<Op, false, true>(op, acc, Ai, rb[dr], re[dr], mk[dr], nacc, rec, c.nlx, c.n, i);
                    rec.flush(nacc, c.nlx, c.n, i);
                } else {
#pragma unroll
                    for (int dr = 0; dr < 3; dr++) walk_row<Op, BUILD, false>(op, acc, Ai, rb[dr], re[dr], mk[dr], nacc, rec, c.nlx, c.n, i);
                }
                if (BUILD) {
                    lw = make_uint4(mk[0], mk[1], mk[2],
                                    (nacc & 0xffffu) | (ok_list ? NL_OK : 0u) | (rec_idx && nacc <= NLX_CAP ? NL_IDX : 0u));
                    rb_build[0] = rb[0];
                    rb_build[1] = rb[1];
                    rb_build[2] = rb[2];
                }
            }
        } else {
            // wide stencil (a large neighbour may be around): (2R+1) rows, explicit index list
            uint32_t dummy = 0, nacc = 0;
            IdxRecorder rec;
            rec.cur = make_uint4(0, 0, 0, 0);
            const int x0 = max(cx - R, 0), x1 = min(cx + R + 1, g.sx);
            for (int yy = max(cy - R, 0); yy <= min(cy + R, g.sy - 1); yy++) {
                const uint32_t base = (uint32_t)yy * (uint32_t)g.sx;
                const uint32_t b = c.cell_start[base + (uint32_t)x0], e = c.cell_start[base + (uint32_t)x1];
                if (BUILD) walk_row<Op, false, true>(op, acc, Ai, b, e, dummy, nacc, rec, c.nlx, c.n, i);
                else walk_row<Op, false, false>(op, acc, Ai, b, e, dummy, nacc, rec, c.nlx, c.n, i);
            }
            if (BUILD) {
                rec.flush(nacc, c.nlx, c.n, i);
                lw = make_uint4(0, 0, 0, (nacc & 0xffffu) | (nacc <= NLX_CAP ? NL_IDX : 0u));
            }
        }
    }
    const bool wall = op.finish(acc, i, Ai, BUILD ? true : (lw.w & NL_WALL) != 0u);
    if (BUILD) {
        if (wall) lw.w |= NL_WALL;
        c.nl[i] = lw;
        if constexpr (!Math::EXACT && !Op::EXTENDED) {
            if (c.nloff_out) emit_offset_list(c.nloff_out, c.nlh_out, c.n, i, lw, rb_build);   // (launch-uniform)
        }
    }
}

template <class Op, bool BUILD>
__device__ __forceinline__ void sweep_block(const Op& op, const SweepCommon& c);

// (A form with FEWER workgroups than tiles, each looping over tiles raw, raw + gridDim.x, ..., was measured in round 4 for the steps
//  that run a level propagation on the side stream -- a dispatch with more workgroups than the device holds keeps the workgroup
//  dispatcher to itself, scripts/ubench/two_queues.hip -- and removed: the loop around the tile costs the headline 4.7 % (1.075 ->
//  1.126 ms/step) and the side stream's launches still waited for the main queue to drain: profiles/r4_level_frontier.md.)
template <class Op, bool BUILD>
__global__ __launch_bounds__(SWEEP_THREADS) void k_sweep(Op op, SweepCommon c)
{
    sweep_stamp(c.ts, false);
    sweep_block<Op, BUILD>(op, c);
    sweep_stamp(c.ts, true);
}

template <class Op, bool BUILD>
__device__ __forceinline__ void sweep_block(const Op& op, const SweepCommon& c)
{
    if (OpPrologue<Op>::run(op, blockIdx.x)) return;
    // XCD-aware block order: the dispatcher places block b on XCD b % 8; give every XCD a contiguous
    // band of the cell-sorted array so that vertically adjacent waves (which share neighbour rows)
    // hit the same L2.
    const uint32_t per_xcd = (c.nblocks + 7) >> 3;
    const uint32_t blk = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    if (blk >= c.nblocks) return;
    uint32_t i = blk * SWEEP_THREADS + threadIdx.x;
    if (c.part == 2) i = i < c.n_ea ? c.elist_a[i] : (i < c.n_ea + c.n_eb ? c.elist_b[i - c.n_ea] : c.n);   // (launch-uniform test)
    const uint32_t ic = i < c.n ? i : 0;
    // slab decomposition: the record and the list word are requested together with the flags (one memory round trip at the head
    // of every wave instead of two); otherwise only by the lanes that have work (level-set propagation skips most)
    const bool slab = c.owned != nullptr;   // launch-uniform
    float4 Ai = make_float4(0.f, 0.f, 0.f, 0.f);
    uint4 lw = make_uint4(0, 0, 0, 0);
    if (slab) {
        Ai = op.loadA(ic);
        if (!BUILD) lw = c.nl[ic];
    }
    const bool mine = !slab || c.owned[ic] || (OpRing1<Op>::value && c.ring1 && c.ring1[ic]);
    const bool active = i < c.n && mine && !op.lane_skip(i) && !(c.part == 1 && c.edge[ic]);
    typename Op::Acc acc;
    op.init(acc);  // lane-independent state
    if (active) {
        if (!slab) {
            Ai = op.loadA(i);
            if (!BUILD) lw = c.nl[i];
        }
        sweep_particle<Op, BUILD>(op, c, acc, i, Ai, lw);
        // optional second pass of a lane over the same list with another op (level-set propagation on slabs: OpLevelPropagate)
        if constexpr (OpSecondPass<Op>::value) {
            if (op.second_wanted(acc)) {
                const typename Op::Second op2 = op.second();
                typename Op::Second::Acc acc2;
                op2.init(acc2);
                sweep_particle<typename Op::Second, false>(op2, c, acc2, i, Ai, lw);
            }
        }
    }
    if (Op::HAS_EPILOGUE) op.epilogue(acc, active, blk);
}

// ------------------------------------------------------------------------------------------------
// Replay on RELATIVE-OFFSET LISTS (round 4; VERDICT round 3 item 7, profiles/r4_jacobi_lab.md: sweep B 19.7 -> 16.5 us and sweep A
// 19.9 -> 16.0 us stand-alone on the rest lattice, 25.8 -> 20.1 and 27.4 -> 20.0 jittered).  The neighbours of a particle of a
// cell-sorted array sit within two cell rows of it -- a few thousand slots -- so j - i fits 16 bits.  The density BUILD sweep (emit_offset_list) turns the
// mask word of every particle into up to NLOFF_SLOTS such offsets once per step, in the masks' visiting order (rows bottom to top,
// index ascending: the sums are those of the mask replay bit for bit), the particle itself left out and the slots behind the count
// holding 0 = the particle itself, whose pair term in a gradient sweep is exactly zero (dx = dy = 0, W'(q) / r finite: MathUniform::
// gscale).  A slot then costs one v_bfe_i32 / v_ashrrev_i32 and one add where the mask replay spends ffs / and / compare / select / add,
// no slot carries a predicate, and the head of the sweep needs neither the two IEEE divisions of the cell index nor the three
// dependent cell_start loads -- the neighbours' gathers go out one round trip after the wave starts instead of two.
// Layout: group g (four offsets, 8 bytes: one trip) of particle i at nloff[g n + i]; header BYTE nlh[i]: bits 0..4 count, NLH_OK list
// valid (mask list, <= NLOFF_SLOTS neighbours, every offset within 16 bits), NLH_WALL = NL_WALL.  A lane without NLH_OK takes the mask
// path (sweep_particle) inside the same launch.  (First form: 16-byte quads of eight offsets + a 4-byte header = 36 B read per
// particle where the rest lattice needs 24 + 1: the sweeps had become bandwidth-bound on exactly those bytes, profiles/r4_sq_counters.txt.)
// Ops: `static constexpr bool OFF16 = true` -- gradient sweeps (FAST / UNIFORM math) whose pair term of the particle with itself is exactly
// zero (SKIP_SELF): the two sweeps of a Jacobi iteration (on their records in uniform scenes, record + payload otherwise), the source-term
// sweep, the non-pressure forces when they run alone.  In a multi-resolution scene the lanes with a mask list (3 x 3 stencil: the bulk) take
// the offsets, the interface particles their index lists, inside the same launch.
// ------------------------------------------------------------------------------------------------
template <class Op, class = void>
struct OpOff16 : std::false_type {};
template <class Op>
struct OpOff16<Op, std::void_t<decltype(Op::OFF16)>> : std::bool_constant<Op::OFF16> {};

// ops with `static constexpr int WIDE_TRIPS = k`: the gathers of the first k trips of k_sweep_off leave together (default 1)
template <class Op, class = void>
struct OpWideTrips : std::integral_constant<int, 1> {};
template <class Op>
struct OpWideTrips<Op, std::void_t<decltype(Op::WIDE_TRIPS)>> : std::integral_constant<int, Op::WIDE_TRIPS> {};
// ops with `static constexpr bool OFF16_SELF = true` need the particle itself (W(0) != 0 in one of their sums): they get every slot's
// offset -- pair_off(.., off) instead of pair(..) -- and add their own term where the reference's list has it, between the last
// neighbour in front of the particle and the first behind it (off > 0 for the first time: the list is in ascending order of j); a
// padding slot has off == 0 (no neighbour has); begin_off() behind begin()
template <class Op, class = void>
struct OpOffSelf : std::false_type {};
template <class Op>
struct OpOffSelf<Op, std::void_t<decltype(Op::OFF16_SELF)>> : std::bool_constant<Op::OFF16_SELF> {};
// ops with `static constexpr int OFF_WAVES = 8` (the two record sweeps of a Jacobi iteration): the offset-list replay is compiled for that
// many waves per SIMD.  Sweep A's twelve records in flight made it 80 VGPRs = 6 waves; with the 32-bit record addressing (load_record)
// it fits 64 with one spilled register and runs 8 (round 5, A/B on one box: sweep B 17.15 -> 16.5 us, sweep A 14.5 -> 14.25; the same
// bound on EVERY offset-list op spills the fused a_ii / force sweep to twice its time: profiles/r5_variants.md section 3)
template <class Op, class = void>
struct OpOffWaves : std::integral_constant<int, 1> {};
template <class Op>
struct OpOffWaves<Op, std::void_t<decltype(Op::OFF_WAVES)>> : std::integral_constant<int, Op::OFF_WAVES> {};
template <class Op>
__global__ __launch_bounds__(SWEEP_THREADS, OpOffWaves<Op>::value) void k_sweep_off(Op op, SweepCommon c)
{
    typedef typename Op::Math Math;
    static_assert(!Math::EXACT && (Op::SKIP_SELF || OpOffSelf<Op>::value) && !Op::EXTENDED, "k_sweep_off: gradient sweeps of the FAST / UNIFORM math policies");
    sweep_stamp(c.ts, false);
    if (!OpPrologue<Op>::run(op, blockIdx.x)) {
        const uint32_t per_xcd = (c.nblocks + 7) >> 3;
        const uint32_t blk = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
        if (blk < c.nblocks) {
            uint32_t i = blk * SWEEP_THREADS + threadIdx.x;
            if (c.part == 2) i = i < c.n_ea ? c.elist_a[i] : (i < c.n_ea + c.n_eb ? c.elist_b[i - c.n_ea] : c.n);   // (launch-uniform test)
            const uint32_t ic = i < c.n ? i : 0;
            // the record, the header byte and the first three offset groups are requested together (one round trip at the head of the wave)
            const float4 Ai = op.loadA(ic);
            const uint32_t head = c.nlh[ic];
            const uint2 g0 = c.nloff[ic], g1 = c.nloff[(size_t)c.n + ic], g2 = c.nloff[2 * (size_t)c.n + ic];
            const bool slab = c.owned != nullptr;   // launch-uniform
            const bool mine = !slab || c.owned[ic] || (OpRing1<Op>::value && c.ring1 && c.ring1[ic]);
            const bool active = i < c.n && mine && !op.lane_skip(i) && !(c.part == 1 && c.edge[ic]);
            typename Op::Acc acc;
            op.init(acc);
            if (active && (head & NLH_OK)) {
                op.begin(acc, i, Ai);
                if constexpr (OpOffSelf<Op>::value) op.begin_off(acc, i, Ai);
                const uint32_t cnt = head & 0x1fu;
#define SPH_OFF_TRIP(WA, WB)                                                                                   \
    {                                                                                                          \
        const int o0 = (int)((WA) << 16) >> 16, o1 = (int)(WA) >> 16, o2 = (int)((WB) << 16) >> 16, o3 = (int)(WB) >> 16; \
        const uint32_t j0 = i + (uint32_t)o0, j1 = i + (uint32_t)o1, j2 = i + (uint32_t)o2, j3 = i + (uint32_t)o3; \
        const float4 A0 = op.loadA(j0), A1 = op.loadA(j1), A2 = op.loadA(j2), A3 = op.loadA(j3);               \
        const typename Op::NB N0 = op.nb(acc, j0, A0), N1 = op.nb(acc, j1, A1), N2 = op.nb(acc, j2, A2), N3 = op.nb(acc, j3, A3); \
        SPH_OFF_PAIR(A0, N0, o0) SPH_OFF_PAIR(A1, N1, o1) SPH_OFF_PAIR(A2, N2, o2) SPH_OFF_PAIR(A3, N3, o3)    \
    }
#define SPH_OFF_PAIR(AJ, NJ, OFF)                                                                              \
    {                                                                                                          \
        const float dx = Ai.x - AJ.x, dy = Ai.y - AJ.y;                                                        \
        const float hij_ = Math::UNIFORM ? op.m.h : (Ai.w + AJ.w) * 0.5f;                                      \
        if constexpr (OpOffSelf<Op>::value) op.pair_off(acc, AJ, NJ, dx, dy, dx * dx + dy * dy, hij_, OFF);    \
        else op.pair(acc, AJ, NJ, dx, dy, dx * dx + dy * dy, hij_);                                            \
    }
                // (the trips are wave-uniform: the longest list of the wave decides; a shorter one evaluates its own record, for nothing)
                constexpr int WT = OpWideTrips<Op>::value;   // trips whose gathers leave together (3: twelve records in flight, one round trip instead of three)
                if (WT >= 2 && std::is_empty<typename Op::NB>::value && __any(cnt > 4u * (uint32_t)(WT - 1))) {
                    // the usual wave (12 neighbours on the rest lattice)
                    const uint32_t w[6] = {g0.x, g0.y, g1.x, g1.y, g2.x, g2.y};
                    float4 R[4 * (WT >= 2 ? WT : 2)];
#pragma unroll
                    for (int k = 0; k < 2 * WT; k++) {
                        R[2 * k] = op.loadA(i + (uint32_t)((int)(w[k] << 16) >> 16));
                        R[2 * k + 1] = op.loadA(i + (uint32_t)((int)w[k] >> 16));
                    }
#pragma unroll
                    for (int k = 0; k < 4 * WT; k++) SPH_OFF_PAIR(R[k], typename Op::NB{}, ((k & 1) ? (int)w[k >> 1] >> 16 : (int)(w[k >> 1] << 16) >> 16))
                    if (WT == 2 && __any(cnt > 8u)) SPH_OFF_TRIP(g2.x, g2.y)
                } else {
                    SPH_OFF_TRIP(g0.x, g0.y)
                    if (__any(cnt > 4u)) SPH_OFF_TRIP(g1.x, g1.y)
                    if (__any(cnt > 8u)) SPH_OFF_TRIP(g2.x, g2.y)
                }
#pragma unroll
                for (uint32_t g = 3; g < (uint32_t)NLOFF_GROUPS; g++) {
                    if (!__any(cnt > 4u * g)) break;
                    // (a group behind a list's end was not written: that lane evaluates its own record)
                    const uint2 gg = cnt > 4u * g ? c.nloff[(size_t)g * c.n + i] : make_uint2(0u, 0u);
                    SPH_OFF_TRIP(gg.x, gg.y)
                }
#undef SPH_OFF_TRIP
#undef SPH_OFF_PAIR
                op.finish(acc, i, Ai, (head & NLH_WALL) != 0u);
            } else if (active) {
                sweep_particle<Op, false>(op, c, acc, i, Ai, c.nl[i]);   // no offset list: the mask word (or the candidate walk)
            }
            if (Op::HAS_EPILOGUE) op.epilogue(acc, active, blk);
        }
    }
    sweep_stamp(c.ts, true);
}

#ifdef SPH_LAB   // the LDS-staged form of the sweeps: a laboratory form (libsph_lab.so), measured slower every round (profiles/r2..r6_variants.md); not in the product TU
// ------------------------------------------------------------------------------------------------
// The same sweep with the wave's candidate rows staged in LDS (uniform-h scenes, mask lists).
// The 64 particles of a wave are consecutive in the cell-sorted order; when they sit in ONE row of cells, their three
// candidate rows are three contiguous index ranges shared by all lanes: [start(cy+r-1, cx_first-1), start(cy+r-1, cx_last+2)),
// ~80 records each on the rest lattice.  Those records (and the per-neighbour payload of the op) are loaded ONCE per wave,
// coalesced, into the wave's own LDS region; every lane then reads its candidates / replays its mask bits from LDS instead
// of issuing 16 per-lane global gathers per neighbour slot.  BUILD runs in two phases: the reference predicate over the
// candidates, branch-free, into the three row masks; then the density sum over the accepted bits only (13 of ~38).
// Same visiting order and the same arithmetic as the gather form, so the results are bit-identical; a wave that straddles
// two cell rows, has a row longer than TILE_CAP records, a lane with more than 32 candidates in a row, or a lane without a
// mask list falls back to the gather form (sweep_particle).
// ------------------------------------------------------------------------------------------------
#ifndef TILE_CAP
#define TILE_CAP 96
#endif

template <class NB, bool EMPTY = std::is_empty<NB>::value>
struct TileNB {
    NB v[SWEEP_THREADS / 64][3 * TILE_CAP];
    __device__ __forceinline__ void put(uint32_t w, uint32_t k, const NB& x) { v[w][k] = x; }
    __device__ __forceinline__ NB get(uint32_t w, uint32_t k) const { return v[w][k]; }
};
template <class NB>
struct TileNB<NB, true> {
    __device__ __forceinline__ void put(uint32_t, uint32_t, const NB&) {}
    __device__ __forceinline__ NB get(uint32_t, uint32_t) const { return NB{}; }
};

template <class Op, bool BUILD>
__device__ __forceinline__ void sweep_tile_block(const Op& op, const SweepCommon& c);

template <class Op, bool BUILD>
__global__ __launch_bounds__(SWEEP_THREADS) void k_sweep_tile(Op op, SweepCommon c)
{
    sweep_stamp(c.ts, false);
    sweep_tile_block<Op, BUILD>(op, c);
    sweep_stamp(c.ts, true);
}

template <class Op, bool BUILD>
__device__ __forceinline__ void sweep_tile_block(const Op& op, const SweepCommon& c)
{
    typedef typename Op::Math Math;
    typedef typename Op::NB NB;
    static_assert(Math::UNIFORM && !Op::EXTENDED, "tile sweep: uniform-h scenes, SPH-support lists");
    if (OpPrologue<Op>::run(op, blockIdx.x)) return;
    const uint32_t per_xcd = (c.nblocks + 7) >> 3;
    const uint32_t blk = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    if (blk >= c.nblocks) return;
    __shared__ float4 s_A[SWEEP_THREADS / 64][3 * TILE_CAP];
    __shared__ TileNB<NB> s_nb;
    const uint32_t w = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t wbase = blk * SWEEP_THREADS + w * 64u;
    const uint32_t i = wbase + lane;
    const uint32_t nvalid = __builtin_amdgcn_readfirstlane(wbase < c.n ? min(64u, c.n - wbase) : 0u);
    const uint32_t ic = i < c.n ? i : 0;
    const bool mine = !c.owned || c.owned[ic] || (OpRing1<Op>::value && c.ring1 && c.ring1[ic]);
    const bool active = i < c.n && mine && !op.lane_skip(i);
    const GridP g = c.g;
    typename Op::Acc acc;
    op.init(acc);
    if (nvalid) {   // wave-uniform
        const float4 Ai = op.loadA(lane < nvalid ? i : wbase);
        const int cx = (int)floorf(Ai.x / g.cs) - g.minx;
        const int cy = (int)floorf(Ai.y / g.cs) - g.miny;
        const int cxf = __builtin_amdgcn_readfirstlane(cx), cyf = __builtin_amdgcn_readfirstlane(cy);
        const int cxl = __builtin_amdgcn_readlane(cx, (int)nvalid - 1), cyl = __builtin_amdgcn_readlane(cy, (int)nvalid - 1);
        bool tile_ok = cyf == cyl;   // one row of cells: cx is non-decreasing over the lanes
        uint32_t sb[3], sl[3], rb[3], re[3];
        uint4 lw = make_uint4(0, 0, 0, 0);
        if (!BUILD) lw = c.nl[lane < nvalid ? i : wbase];
        bool lane_ok = true;
#pragma unroll
        for (int dr = 0; dr < 3; dr++) {
            const int yy = cyf + dr - 1;
            const bool ok = yy >= 0 && yy < g.sy;
            const uint32_t base = (uint32_t)(ok ? yy : 0) * (uint32_t)g.sx;
            sb[dr] = ok ? c.cell_start[base + (uint32_t)max(cxf - 1, 0)] : 0u;
            const uint32_t se = ok ? c.cell_start[base + (uint32_t)min(cxl + 2, g.sx)] : sb[dr];
            sl[dr] = se - sb[dr];
            tile_ok = tile_ok && sl[dr] <= (uint32_t)TILE_CAP;
            rb[dr] = ok ? c.cell_start[base + (uint32_t)max(cx - 1, 0)] : 0u;
            re[dr] = rb[dr];
            if (BUILD) {
                re[dr] = ok ? c.cell_start[base + (uint32_t)min(cx + 2, g.sx)] : rb[dr];
                lane_ok = lane_ok && (re[dr] - rb[dr]) <= 32u;
            }
        }
        if (!BUILD) lane_ok = (lw.w & NL_OK) != 0u;
        tile_ok = tile_ok && __ballot(lane < nvalid && !lane_ok) == 0ull;
        if (!tile_ok) {
            if (active) sweep_particle<Op, BUILD>(op, c, acc, i, Ai, lw);
        } else {
            // ---- stage the three rows: coalesced loads, one record (+ payload) per lane and trip
#pragma unroll
            for (int dr = 0; dr < 3; dr++) {
                for (uint32_t k = lane; k < sl[dr]; k += 64u) {
                    const uint32_t j = sb[dr] + k;
                    const float4 Aj = op.loadA(j);
                    s_A[w][dr * TILE_CAP + k] = Aj;
                    s_nb.put(w, dr * TILE_CAP + k, op.nb(acc, j, Aj));
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (active) {
                op.begin(acc, i, Ai);
                uint32_t off[3];
#pragma unroll
                for (int dr = 0; dr < 3; dr++) off[dr] = dr * TILE_CAP + (rb[dr] - sb[dr]);
                if (BUILD) {
                    // phase 1: the neighbour predicate, exactly the reference's operations (no FMA, strict <), branch-free
                    const float s = op.m.h * op.krange();
                    const float s2 = s * s;
                    uint32_t mk[3];
#pragma unroll
                    for (int dr = 0; dr < 3; dr++) {
                        uint32_t m = 0u;
                        const uint32_t len = re[dr] - rb[dr];
                        for (uint32_t t = 0; t < len; t += 2u) {
                            const float4 A0 = s_A[w][off[dr] + t];
                            const float4 A1 = s_A[w][off[dr] + min(t + 1u, len - 1u)];
                            const float dx0 = Ai.x - A0.x, dy0 = Ai.y - A0.y, dx1 = Ai.x - A1.x, dy1 = Ai.y - A1.y;
                            const float r0 = dx0 * dx0 + dy0 * dy0, r1 = dx1 * dx1 + dy1 * dy1;
                            m |= (r0 < s2 ? 1u : 0u) << t;
                            m |= ((t + 1u < len && r1 < s2) ? 2u : 0u) << t;
                        }
                        mk[dr] = m;
                    }
                    lw = make_uint4(mk[0], mk[1], mk[2], 0u);
                }
                uint32_t masks[3] = {lw.x, lw.y, lw.z};
                if (!BUILD && Op::SKIP_SELF) {   // see k_sweep: the own pair term is exactly zero in the gradient sweeps
                    const uint32_t sbit = i - rb[1];
                    if (sbit < 32u) masks[1] &= ~(1u << sbit);
                }
                // phase 2 / replay: the accepted candidates in the order of the gather form (rows bottom to top, index ascending)
#pragma unroll
                for (int dr = 0; dr < 3; dr++) {
                    uint32_t mk = masks[dr];
                    const uint32_t o = off[dr];
                    while (mk) {
                        const uint32_t b0 = __ffs(mk) - 1;
                        mk &= mk - 1;
                        const bool v1 = mk != 0;
                        const uint32_t b1 = v1 ? __ffs(mk) - 1 : b0;
                        mk &= mk - 1;
                        const float4 A0 = s_A[w][o + b0], A1 = s_A[w][o + b1];
                        const NB N0 = s_nb.get(w, o + b0), N1 = s_nb.get(w, o + b1);
                        {
                            const float dx = Ai.x - A0.x, dy = Ai.y - A0.y;
                            op.pair(acc, A0, N0, dx, dy, dx * dx + dy * dy, op.m.h);
                        }
                        if (v1) {
                            const float dx = Ai.x - A1.x, dy = Ai.y - A1.y;
                            op.pair(acc, A1, N1, dx, dy, dx * dx + dy * dy, op.m.h);
                        }
                    }
                }
                if (BUILD) lw.w = (uint32_t)(__popc(lw.x) + __popc(lw.y) + __popc(lw.z)) | NL_OK;
                const bool wall = op.finish(acc, i, Ai, BUILD ? true : (lw.w & NL_WALL) != 0u);
                if (BUILD) {
                    if (wall) lw.w |= NL_WALL;
                    c.nl[i] = lw;
                }
            }
        }
    }
    if (Op::HAS_EPILOGUE) op.epilogue(acc, active, blk);
}

// ------------------------------------------------------------------------------------------------
// Op: density  (+ boundary lambda terms, + neighbour count, + m/rho)
//   calculate_particle_density               simulation.rs:1007-1028, asserts :1046-1047
//   BoundaryWinchenbach2020::update_after_advect   boundary_winchenbach2020.rs:58-152
//   neighbor_count                           simulation.rs:2072-2074
// ------------------------------------------------------------------------------------------------
#endif   // SPH_LAB (k_sweep_tile)

struct NBNone {};

// HDIST: support_length_estimation FromDistribution* -- compiled apart, the density sweep is VALU-bound and the two
// extra sums cost 3.5 us at N = 1M even behind a launch-uniform branch
template <class MathT, bool HDIST>
struct OpDensity {
    static constexpr bool TILE = true;
    typedef MathT Math;
    static constexpr bool HAS_EPILOGUE = false, SKIP_SELF = false;
    static constexpr bool EXTENDED = false, RING1 = true;
    __device__ constexpr float krange() const { return 2.f; }
    typedef NBNone NB;
    MathT m;
    const float4* __restrict__ pm;
    const uint32_t* __restrict__ orig;
    float* __restrict__ rho;
    float* __restrict__ mrho;
    float* __restrict__ lam_sum;
    float2* __restrict__ lam_grad;
    uint32_t* __restrict__ ncount;
    const BoundaryP* __restrict__ planes;
    const float* __restrict__ lam_lut;
    const float* __restrict__ dlam_lut;
    DeviceStatus* status;
    StepP sp;
    // support_length_estimation FromDistribution* (estimate_h_next_from_distribution / _distribution2, simulation.rs:1873-1971):
    // the W sums ride along with the density sum; h2_next holds the h of the previous step on entry
    int h_mode;
    float* __restrict__ h2_next;
    const float* __restrict__ lam_prev;   // lambda_sum(i) of the PREVIOUS step (update_after_advect runs later in the step)
    const uint8_t* __restrict__ owned_flag;   // slab decomposition (else nullptr): ring-1 ghost lanes record their list, nothing more
    struct Acc {
        float sum, lam, wsum, vwsum;
        uint32_t cnt;
        bool wall;
    };
    __device__ bool skip() const { return false; }
    __device__ bool lane_skip(uint32_t) const { return false; }
    __device__ void init(Acc&) const {}
    __device__ void epilogue(Acc&, bool, uint32_t) const {}
    __device__ float4 loadA(uint32_t j) const { return pm[j]; }
    __device__ NB nb(const Acc&, uint32_t, float4) const { return NB{}; }


    __device__ void begin(Acc& a, uint32_t i, float4 Ai) const
    {
        a.sum = 0.f;
        a.wsum = a.vwsum = 0.f;
        a.cnt = 0;
        // semi-analytic boundary: every IEEE op as in the reference (same values as the oracle)
        const float x = Ai.x, y = Ai.y;
        const float sr_i = Ai.w * 2.f;
        float ls = 0.f, gxs = 0.f, gys = 0.f;
        uint32_t n_entries = 0;
        for (int k = 0; k < sp.n_planes; k++) {
            float d = sdf_probe(planes, k, x, y) / sr_i;
            if (!(d < 1.f)) continue;
            const float eps = sp.sdf_eps;
            const float inv_2eps = 1.f / (2.f * eps);
            float gx = (sdf_probe(planes, k, x + eps, y) - sdf_probe(planes, k, x - eps, y)) * inv_2eps;
            float gy = (sdf_probe(planes, k, x, y + eps) - sdf_probe(planes, k, x, y - eps)) * inv_2eps;
            float gn = sqrtf(gx * gx + gy * gy);
            if (!(gn >= 0.00001f)) continue;
            gx /= gn;
            gy /= gn;
            float penalty, dpenalty;
            if (sp.penalty == SPH_PENALTY_NONE) {
                penalty = 1.f;
                dpenalty = 0.f;
            } else if (sp.penalty == SPH_PENALTY_LINEAR) {
                penalty = 1.f - d;
                dpenalty = -1.f;
            } else if (sp.penalty == SPH_PENALTY_QUADRATIC1) {
                if (d > 0.f) { penalty = 1.f; dpenalty = 0.f; }
                else if (d > -1.f) { penalty = 0.5f * d * d + 1.f; dpenalty = d; }
                else { penalty = 0.5f - d; dpenalty = -1.f; }
            } else {
                if (d > 0.f) { penalty = 1.f; dpenalty = 0.f; }
                else if (d > -0.5f) { penalty = d * d + 1.f; dpenalty = 2.f * d; }
                else { penalty = 0.75f - d; dpenalty = -1.f; }
            }
            float lambda, dlambda;
            if (d <= -1.f) { lambda = 1.f; dlambda = 0.f; }
            else { lambda = lut_get(lam_lut, d); dlambda = lut_get(dlam_lut, d); }
            float s = dpenalty * lambda + penalty * dlambda;
            ls += lambda * penalty;
            gxs += gx / sr_i * s;
            gys += gy / sr_i * s;
            if constexpr (MathT::EXACT) {   // the entry itself (boundary_winchenbach2020.rs:139-149)
                m.wall_pl[(size_t)n_entries * m.wall_n + i] = make_float2(gx / sr_i * s, gy / sr_i * s);
                n_entries++;
            }
        }
        if constexpr (MathT::EXACT) m.wall_cnt[i] = (uint8_t)n_entries;
        a.lam = ls;
        a.wall = ls != 0.f || gxs != 0.f || gys != 0.f;
        lam_sum[i] = ls;
        lam_grad[i] = make_float2(gxs, gys);
    }
    __device__ void pair(Acc& a, float4 Aj, NB, float, float, float r2, float hij) const
    {
        const float w = m.w(r2, hij);
        a.sum += Aj.z * w;
        if (HDIST) {
            a.wsum += w;
            a.vwsum += (Aj.z / sp.rest_density) * w;
        }
        a.cnt++;
    }
    __device__ bool finish(Acc& a, uint32_t i, float4 Ai, bool) const
    {
        float d = a.sum + a.lam;
        rho[i] = d;
        mrho[i] = Ai.z / d;
        ncount[i] = a.cnt;
        if (!isfinite(d)) raise_error(status, SPH_ERR_DENSITY_NOT_FINITE, orig[i]);
        else if (!(d > 0.0001f)) raise_error(status, SPH_ERR_DENSITY_TOO_SMALL, orig[i]);
        if (a.cnt > 20000u) raise_error(status, SPH_ERR_TOO_MANY_NEIGHBORS, orig[i]);
        if (HDIST && (!owned_flag || owned_flag[i])) {
            const float bv = lam_prev[i];
            float vol;
            if (h_mode == SPH_H_FROM_DISTRIBUTION2) vol = (Ai.z / sp.rest_density) / (a.vwsum + bv);
            else vol = (1.f - fminf(bv, 0.5f)) / a.wsum;
            if (!(vol >= 0.f)) raise_error(status, SPH_ERR_VOLUME_ESTIMATE, orig[i]);
            const float h_new = SPH_ETA * sqrtf(vol * SPH_FRAC_1_PI_F);
            float hn = 0.5f * h_new + (1.f - 0.5f) * Ai.w;
            if (h_mode == SPH_H_FROM_DISTRIBUTION_CLAMPED1) hn = fminf(hn, 1.f * h_from_mass(Ai.z, sp.rest_density));
            if (h_mode == SPH_H_FROM_DISTRIBUTION_CLAMPED2) hn = fminf(hn, 2.f * h_from_mass(Ai.z, sp.rest_density));
            h2_next[i] = hn;
        }
        return a.wall;
    }
};

// ------------------------------------------------------------------------------------------------
// Op: a_ii + constant_field
//   compute_aii -> BoundaryWinchenbach2020::iisph_aii   simulation.rs:1080-1125, boundary_winchenbach2020.rs:225-306
//   constant_field                                       simulation.rs:2235-2248
// ------------------------------------------------------------------------------------------------
template <class MathT>
struct OpAiiConst {
    static constexpr bool TILE = true;
    typedef MathT Math;
    static constexpr bool HAS_EPILOGUE = false, SKIP_SELF = false;   // W(0) != 0 in the constant field
    static constexpr bool EXTENDED = false;
    __device__ constexpr float krange() const { return 2.f; }
    typedef float NB;  // m_j / rho_j
    MathT m;
    const float4* __restrict__ pm;
    const uint32_t* __restrict__ orig;
    const float* __restrict__ rho;
    const float* __restrict__ mrho;
    const float* __restrict__ lam_sum;
    const float2* __restrict__ lam_grad;
    float* __restrict__ aii;
    float* __restrict__ constf;
    DeviceStatus* status;
    StepP sp;
    float2* __restrict__ ap_unit;   // check_aii only: a^p_i for the pressure field e_i (input of OpCheckAii), else nullptr
    struct Acc {
        float cf, ax, ay, a2, bx, by;
        float self_cf;    // offset lists: the particle's own term of the constant field, (m_i / rho_i) W(0) ...
        bool self_done;   // ... and whether it has been added (a list that names the particle itself: always)
    };
    // (k_sweep_off: the offset list leaves the particle out; of its pair with itself only the constant field's W(0) term is not zero --
    //  the gradient is -- and that term goes where the mask replay has it, in front of the first neighbour behind the particle)
    static constexpr bool OFF16 = !MathT::EXACT, OFF16_SELF = true;
    __device__ void begin_off(Acc& a, uint32_t i, float4 Ai) const
    {
        float wv, sc;
        m.wg(0.f, Ai.w, wv, sc);
        a.self_cf = mrho[i] * wv;
        a.self_done = false;
    }
    __device__ void pair_off(Acc& a, float4 Aj, NB mr, float dx, float dy, float r2, float hij, int off) const
    {
        a.cf += (off > 0 && !a.self_done) ? a.self_cf : 0.f;
        a.self_done = a.self_done || off > 0;
        pair(a, Aj, off != 0 ? mr : 0.f, dx, dy, r2, hij);   // (a padding slot is the particle again: its gradient is zero, its m / rho must not count)
    }
    __device__ bool skip() const { return false; }
    __device__ bool lane_skip(uint32_t) const { return false; }
    __device__ void init(Acc&) const {}
    __device__ void epilogue(Acc&, bool, uint32_t) const {}
    __device__ float4 loadA(uint32_t j) const { return pm[j]; }
    __device__ NB nb(const Acc&, uint32_t j, float4) const { return mrho[j]; }
    __device__ void begin(Acc& a, uint32_t, float4) const
    {
        a.cf = a.ax = a.ay = a.a2 = a.bx = a.by = 0.f;
        a.self_cf = 0.f;
        a.self_done = true;
    }
    __device__ void pair(Acc& a, float4 Aj, NB mr, float dx, float dy, float r2, float hij) const
    {
        float gx, gy;
        if constexpr (MathT::EXACT) {
            a.cf += mr * m.w(r2, hij);
            m.grad(dx, dy, r2, hij, gx, gy);
        } else {   // W and grad W from one rsq (MathFast::wg)
            float wv, sc;
            m.wg(r2, hij, wv, sc);
            a.cf += mr * wv;
            gx = sc * dx;
            gy = sc * dy;
        }
        a.ax += Aj.z * gx;
        a.ay += Aj.z * gy;
        if (sp.opdisc == SPH_OP_WINCHENBACH2020) {
            a.bx += mr * gx;
            a.by += mr * gy;
            a.a2 += mr * (gx * gx + gy * gy);
        } else {
            a.a2 += Aj.z * (gx * gx + gy * gy);
        }
    }
    __device__ bool finish(Acc& a, uint32_t i, float4 Ai, bool wall) const
    {
        float ls = 0.f;
        float2 gl = make_float2(0.f, 0.f);
        if (wall) {
            ls = lam_sum[i];
            gl = lam_grad[i];
        }
        if (!a.self_done) a.cf += a.self_cf;   // (offset list without a neighbour behind the particle)
        constf[i] = a.cf + ls / sp.rest_density;
        const float mi = Ai.z, rho_i = rho[i], rho_b = sp.rest_density;
        const float rho_i_sq = rho_i * rho_i;
        float v;
        if constexpr (MathT::EXACT) {   // the boundary sums entry by entry, as iisph_aii adds them (boundary_winchenbach2020.rs:236-304)
            const float coeff = sp.opdisc == SPH_OP_SYMMETRIC_GRADIENT ? 1.f : 0.f;
            const float f = rho_b * (1.f / (rho_i * rho_i) + coeff / (rho_b * rho_b));
            float sgx = 0.f, sgy = 0.f, rgx = 0.f, rgy = 0.f, sbx = 0.f, sby = 0.f;
            const uint32_t cnt = wall ? m.wall_count(i) : 0u;
            for (uint32_t k = 0; k < cnt; k++) {
                const float2 g = m.wall_entry(i, k);
                sgx += g.x;
                sgy += g.y;
                rgx += rho_b * g.x;
                rgy += rho_b * g.y;
                sbx += f * g.x;
                sby += f * g.y;
            }
            const float lx = a.ax / rho_i_sq + sbx, ly = a.ay / rho_i_sq + sby;
            if (sp.opdisc == SPH_OP_WINCHENBACH2020) {
                const float rx = a.bx + sgx, ry = a.by + sgy;
                v = (lx * rx + ly * ry) + (mi * a.a2 / rho_i_sq);
            } else {
                const float rx = a.ax / rho_i + rgx / rho_i, ry = a.ay / rho_i + rgy / rho_i;
                v = (lx * rx + ly * ry) + (mi * a.a2) / (rho_i * rho_i * rho_i);
            }
        } else if (sp.opdisc == SPH_OP_WINCHENBACH2020) {
            float f = rho_b * (1.f / (rho_i * rho_i) + 0.f / (rho_b * rho_b));
            float lx = a.ax / rho_i_sq + f * gl.x, ly = a.ay / rho_i_sq + f * gl.y;
            float rx = a.bx + gl.x, ry = a.by + gl.y;
            v = (lx * rx + ly * ry) + (mi * a.a2 / rho_i_sq);
        } else {
            float coeff = sp.opdisc == SPH_OP_SIMPLE_GRADIENT ? 0.f : 1.f;
            float f = rho_b * (1.f / (rho_i * rho_i) + coeff / (rho_b * rho_b));
            float rgx = rho_b * gl.x, rgy = rho_b * gl.y;
            float lx = a.ax / rho_i_sq + f * gl.x, ly = a.ay / rho_i_sq + f * gl.y;
            float rx = a.ax / rho_i + rgx / rho_i, ry = a.ay / rho_i + rgy / rho_i;
            v = (lx * rx + ly * ry) + (mi * a.a2) / (rho_i * rho_i * rho_i);
        }
        aii[i] = v;
        if (!isfinite(v)) raise_error(status, SPH_ERR_AII_NOT_FINITE, orig[i]);
        if (ap_unit) {
            // calculate_particle_pressure_accel with p = e_i at particle i itself (simulation.rs:1750-1808, boundary
            // boundary_winchenbach2020.rs:164-194): -sum_k m_k (1/rho_i^2 + 0) grad W_ik - rho_b (1/rho_i^2 + p_ib/rho_b^2) sum grad lambda
            const float pti = 1.f / (rho_i * rho_i);
            const float p_ib = sp.opdisc == SPH_OP_SYMMETRIC_GRADIENT ? 1.f : 0.f;
            const float fb = -rho_b * (1.f / (rho_i * rho_i) + p_ib / (rho_b * rho_b));
            float wx = fb * gl.x, wy = fb * gl.y;
            if constexpr (MathT::EXACT) {
                wx = wy = 0.f;
                const uint32_t cnt = wall ? m.wall_count(i) : 0u;
                for (uint32_t k = 0; k < cnt; k++) {
                    const float2 g = m.wall_entry(i, k);
                    wx += fb * g.x;
                    wy += fb * g.y;
                }
            }
            ap_unit[i] = make_float2(-(pti * a.ax) + wx, -(pti * a.ay) + wy);
        }
        return wall;
    }
};

// ------------------------------------------------------------------------------------------------
// Op: non-pressure acceleration -> velocity_temp
//   update_velocity_with_non_pressure_accel / calculate_particle_non_pressure_accel
//   simulation.rs:1051-1077, 931-1005
// ------------------------------------------------------------------------------------------------
struct NBRhoVel {
    float rho, vx, vy;
};

template <class MathT>
struct OpNonPressure {
    static constexpr bool TILE = true;
    typedef MathT Math;
    static constexpr bool HAS_EPILOGUE = false, SKIP_SELF = true;
    static constexpr bool EXTENDED = false;
    static constexpr bool OFF16 = !MathT::EXACT;   // (k_sweep_off: relative-offset lists, when the step built them)
    __device__ constexpr float krange() const { return 2.f; }
    typedef NBRhoVel NB;
    MathT m;
    const float4* __restrict__ pm;
    const uint32_t* __restrict__ orig;
    const float* __restrict__ rho;
    const float2* __restrict__ vel;
    float2* __restrict__ vel_out;
    DeviceStatus* status;
    StepP sp;
    float4* __restrict__ xv_out;   // {x, y, v'}: the record the source-term sweep gathers (OpSourceU; else nullptr)
    struct Acc {
        float vx, vy, rho_i, vix, viy;
    };
    __device__ bool skip() const { return false; }
    __device__ bool lane_skip(uint32_t) const { return false; }
    __device__ void init(Acc&) const {}
    __device__ void epilogue(Acc&, bool, uint32_t) const {}
    __device__ float4 loadA(uint32_t j) const { return pm[j]; }
    __device__ NB nb(const Acc&, uint32_t j, float4) const
    {
        float2 v = vel[j];
        return NB{rho[j], v.x, v.y};
    }
    __device__ void begin(Acc& a, uint32_t i, float4) const
    {
        a.vx = a.vy = 0.f;
        a.rho_i = rho[i];
        float2 v = vel[i];
        a.vix = v.x;
        a.viy = v.y;
    }
    __device__ void pair(Acc& a, float4 Aj, NB Bj, float dx, float dy, float r2, float hij) const
    {
        const float ux = a.vix - Bj.vx, uy = a.viy - Bj.vy;
        if (sp.viscosity_type == SPH_VISC_APPROX_LAPLACE) {
            const float xv = dx * ux + dy * uy;
            if (xv >= 0.f) return;
            float gx, gy;
            m.grad(dx, dy, r2, hij, gx, gy);
            const float rho_ij = (a.rho_i + Bj.rho) * 0.5f;
            const float den = r2 + 0.01f * hij * hij;
            float coeff;
            if (MathT::EXACT) coeff = 2.f * 4.f * (Aj.z / rho_ij) * xv / den;
            else coeff = 8.f * (Aj.z * xv) * fast_rcp(rho_ij * den);   // (one reciprocal for both divisors)
            const float f = sp.viscosity * coeff;
            a.vx += f * gx;
            a.vy += f * gy;
        } else if (sp.viscosity_type == SPH_VISC_WCSPH) {
            const float est = ux * dx + uy * dy;
            if (est < 0.f) {
                float gx, gy;
                m.grad(dx, dy, r2, hij, gx, gy);
                const float den = r2 + 0.001f * hij * hij;
                float viscous_term, pi_ab;
                if (MathT::EXACT) {
                    viscous_term = 2.f * sp.viscosity * hij * 88.f / (a.rho_i + Bj.rho);
                    pi_ab = -viscous_term * est / den;
                } else {
                    viscous_term = 2.f * sp.viscosity * hij * 88.f * fast_rcp(a.rho_i + Bj.rho);
                    pi_ab = -viscous_term * est * fast_rcp(den);
                }
                const float f = -Aj.z * pi_ab;
                a.vx += f * gx;
                a.vy += f * gy;
            }
        }
    }
    __device__ bool finish(Acc& a, uint32_t i, float4 Ai, bool wall) const
    {
        float px = 0.f, py = 0.f;
        if (sp.has_pull) {
            float tx = sp.pull_x - Ai.x, ty = sp.pull_y - Ai.y;
            float nn = sqrtf(tx * tx + ty * ty);
            px = tx / nn * 13.f;
            py = ty / nn * 13.f;
        }
        const float ax = (a.vx + 0.f) + px;
        const float ay = (a.vy + sp.gravity) + py;
        if (!isfinite(a.vx) || !isfinite(a.vy)) raise_error(status, SPH_ERR_VISCOSITY_NOT_FINITE, orig[i]);
        const float2 vn = make_float2(a.vix + sp.dt * ax, a.viy + sp.dt * ay);
        vel_out[i] = vn;
        if (xv_out) xv_out[i] = make_float4(Ai.x, Ai.y, vn.x, vn.y);   // (launch-uniform)
        return wall;
    }
};

// ------------------------------------------------------------------------------------------------
// Op: PPE source term (+ pressure := 0, p/rho^2 := 0)
//   prepare_ppe_divergence / prepare_full_ppe / prepare_only_density_part_ppe   simulation.rs:1127-1204
//   calculate_source_term_divergence/_full/_only_density_part                   simulation.rs:1633-1676, 1712-1748
//   calculate_divergence_iisph (+ boundary part)    simulation.rs:1552-1592, boundary_winchenbach2020.rs:196-223
// ------------------------------------------------------------------------------------------------
struct NBVecMr {
    float qx, qy, mr;  // vector quantity Q_j, m_j / rho_j
};

// block partial of PressureSolverStatistics in fixed lane order (deterministic); k_solver_final adds the
// block partials in block (= particle) order.  cls: 0 normal, 1 singular, 2 negative, 3 none.
__device__ __forceinline__ void solver_block_partial(SolverPartial* __restrict__ partials, uint32_t cls, float err, uint32_t blk)
{
    __shared__ SolverPartial s_w[SWEEP_THREADS / 64];
    // (the three counts are population counts of ballots: scalar instructions instead of three shuffle reductions)
    const uint32_t normal = (uint32_t)__popcll(__ballot(cls == 0u));
    const uint32_t singular = (uint32_t)__popcll(__ballot(cls == 1u));
    const uint32_t negative = (uint32_t)__popcll(__ballot(cls == 2u));
    // (a wave of the sweep may hold idle lanes -- the array's tail, ghosts: the caller passes cls = 3 for them, and every lane of
    //  the block reaches this point, so the DPP network sees all 64)
    float sum = wave_sum_dpp(cls == 0u ? err : 0.f);
    float mx = wave_max_nonneg_dpp(cls == 0u ? fabsf(err) : 0.f);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) s_w[w] = SolverPartial{normal, singular, negative, sum, mx};
    __syncthreads();
    if (threadIdx.x == 0) {
        SolverPartial t = s_w[0];
        for (int k = 1; k < SWEEP_THREADS / 64; k++) {
            t.normal += s_w[k].normal;
            t.singular += s_w[k].singular;
            t.negative += s_w[k].negative;
            t.sum_err += s_w[k].sum_err;
            t.max_err = fmaxf(t.max_err, s_w[k].max_err);
        }
        partials[blk] = t;
    }
}

// OMEGA: IISPH2's source term (calculate_source_term_full_with_omega, simulation.rs:1678-1710) with the per-particle omega of
// simulation.rs:2263-2311 summed in the same pass (its self term W(0)-like is not zero, so the own bit stays on the list)
template <class MathT, bool OMEGA>
struct OpSource {
    static constexpr bool TILE = true;
    typedef MathT Math;
    static constexpr bool HAS_EPILOGUE = true, SKIP_SELF = !OMEGA;
    static constexpr bool EXTENDED = false;
    static constexpr bool OFF16 = !MathT::EXACT && !OMEGA;   // (k_sweep_off: relative-offset lists, when the step built them)
    __device__ constexpr float krange() const { return 2.f; }
    typedef NBVecMr NB;
    MathT m;
    const float4* __restrict__ pm;
    const uint32_t* __restrict__ orig;
    const float* __restrict__ rho;
    const float* __restrict__ mrho;
    const float2* __restrict__ vel;
    const float2* __restrict__ lam_grad;
    const float* __restrict__ aii;
    float* __restrict__ src;
    float* __restrict__ p_out;       // pressure after iteration 0 (buffer 1)
    float* __restrict__ pterm_out;
    float4* __restrict__ rec_out;    // see store_pressure (else nullptr)
    float* __restrict__ dens_err;
    SolverPartial* __restrict__ partials;
    SolverCtrl* __restrict__ ctrl_reset;   // the solve's control block starts from zero (first kernel of a solve)
    DeviceStatus* status;
    StepP sp;
    int kind;  // 0 divergence, 1 full, 2 only density, 3 full with omega (OMEGA)
    int residual_density;
    float* __restrict__ omega;              // OMEGA only
    const uint8_t* __restrict__ size_class;
    const uint32_t* __restrict__ gate;      // chained solves (else nullptr): 0 = the solve before this one has not ended, leave
    struct Acc {
        float sum, rho_i, inv_rho_i, qx, qy;
        float err;
        uint32_t cls;
        float om, om_c;   // OMEGA: running omega and H_i / (3 rho_i)
        bool large;
    };
    __device__ bool prologue(uint32_t) const { return gate && *gate == 0u; }
    __device__ bool skip() const { return false; }
    __device__ bool lane_skip(uint32_t) const { return false; }
    __device__ void init(Acc&) const {}
    __device__ float4 loadA(uint32_t j) const { return pm[j]; }
    __device__ NB nb(const Acc&, uint32_t j, float4) const
    {
        float2 v = vel[j];
        float mr = 0.f;
        if (sp.opdisc == SPH_OP_WINCHENBACH2020) mr = mrho[j];
        return NB{v.x, v.y, mr};
    }
    __device__ void begin(Acc& a, uint32_t i, float4) const
    {
        a.sum = 0.f;
        a.rho_i = rho[i];
        a.inv_rho_i = fast_rcp(a.rho_i);
        float2 v = vel[i];
        a.qx = v.x;
        a.qy = v.y;
        a.err = 0.f;
        a.cls = 3u;
        a.om = 1.f;
        a.om_c = 0.f;
        a.large = false;
        if (OMEGA) {
            const float4 Ai = pm[i];
            a.om_c = (Ai.w * 2.f) / (3.f * a.rho_i);
            a.large = size_class[i] == 3;   // ParticleSizeClass::Large (adaptivity/mod.rs:12-23)
            if (a.large) a.om += a.om_c * Ai.z * dwdh(0.f, Ai.w * 2.f);   // simulation.rs:2277-2288
        }
    }
    // dwdh of simulation.rs:2267-2276 (2-D): derivative of the normalised kernel with respect to the support radius H
    __device__ static float dwdh(float d, float H)
    {
        const float q = d / H;
        const float cd = 40.f / (7.f * SPH_PI_F);
        const float w = cubic_unnorm(q), wd = cubic_unnorm_deriv(q);
        return cd * -(2.f) / (H * H * H) * w + cd / (H * H) * wd * (-d / (H * H));
    }
    __device__ void pair(Acc& a, float4 Aj, NB Bj, float dx, float dy, float r2, float hij) const
    {
        if (OMEGA && !a.large) a.om += a.om_c * Aj.z * dwdh(sqrtf(r2), hij * 2.f);   // simulation.rs:2289-2305
        if (kind == 2) return;
        if constexpr (MathT::EXACT) {
            float gx, gy;
            m.grad(dx, dy, r2, hij, gx, gy);
            const float dot = (Bj.qx - a.qx) * gx + (Bj.qy - a.qy) * gy;
            if (sp.opdisc == SPH_OP_WINCHENBACH2020) a.sum += Bj.mr * dot;
            else a.sum += Aj.z / a.rho_i * dot;
        } else {   // (Q_j - Q_i) . grad W_ij = s ((Q_j - Q_i) . x_ij): scalars first (MathFast / MathUniform, sph_device.h)
            const float e = fmaf(Bj.qx - a.qx, dx, (Bj.qy - a.qy) * dy);
            const float c = sp.opdisc == SPH_OP_WINCHENBACH2020 ? Bj.mr : Aj.z * a.inv_rho_i;
            a.sum = fmaf(c * m.gscale(r2, hij), e, a.sum);
        }
    }
    __device__ bool finish(Acc& a, uint32_t i, float4 Ai, bool wall) const
    {
        const float rho_i = a.rho_i, rho_b = sp.rest_density, dt = sp.dt;
        float s;
        const float nde = sp.opdisc == SPH_OP_WINCHENBACH2020 ? sp.rest_density : rho_i;
        if (kind == 2) {
            s = -(sp.rest_density - rho_i) / (nde * dt * dt);
        } else {
            float2 gl = make_float2(0.f, 0.f);
            if (wall) gl = lam_grad[i];
            const float bdot = (0.f - a.qx) * gl.x + (0.f - a.qy) * gl.y;
            float bdiv = sp.opdisc == SPH_OP_WINCHENBACH2020 ? bdot : rho_b / rho_i * bdot;
            if constexpr (MathT::EXACT) bdiv = wall ? m.wall_divergence(i, a.qx, a.qy, rho_i, rho_b, sp.opdisc == SPH_OP_WINCHENBACH2020) : 0.f;
            const float vdiv = a.sum + (sp.n_planes ? bdiv : 0.f);
            if (kind == 0) s = -vdiv / dt;
            else if (OMEGA) {
                const float om = fminf(2.5f, fmaxf(a.om, 0.125f));
                omega[i] = om;
                s = -(sp.rest_density - rho_i) / (sp.rest_density * dt * dt) - vdiv / (dt * om);
            } else s = -(sp.rest_density - rho_i) / (nde * dt * dt) - vdiv / dt;
        }
        src[i] = s;
        // ---- Jacobi iteration 0 (iisph_single_pressure_iteration, simulation.rs:1207-1322) in closed form:
        // the iteration starts from p = 0, so a^p = 0 and (Ap)_i = 0 for every particle, and
        //   p' = 0 + w * (s - 0) / a_ii,  err = rho_i dt^2 (s - 0)  or  dt (s - 0)
        // -- the values the generic sweeps A and B would produce, without their two neighbour passes.
        const float aii_i = aii[i];
        if (aii_i < 0.f) raise_error(status, SPH_ERR_AII_NEGATIVE, orig[i]);  // simulation.rs:1390-1403
        if (fabsf(aii_i) < 10e-4f) {
            store_pressure(p_out, pterm_out, rec_out, i, Ai, 0.f, 0.f);
            a.cls = 1u;
            return wall;
        }
        const float a_p = 0.f;
        float pn = 0.f + sp.jacobi_omega * (s - a_p) / aii_i;
        if (!isfinite(pn)) raise_error(status, SPH_ERR_PRESSURE_NOT_FINITE, orig[i]);
        float err;
        if (residual_density) {
            err = rho_i * dt * dt * (s - a_p);
            dens_err[i] = err;
        } else {
            err = dt * (s - a_p);
        }
        if (pn <= 0.f) {
            store_pressure(p_out, pterm_out, rec_out, i, Ai, 0.f, 0.f);
            a.cls = 2u;
        } else {
            store_pressure(p_out, pterm_out, rec_out, i, Ai, pn, pn / (rho_i * rho_i));
            a.cls = 0u;
            a.err = err;
        }
        return wall;
    }
    __device__ void epilogue(Acc& a, bool active, uint32_t blk) const
    {
        solver_block_partial(partials, active ? a.cls : 3u, a.err, blk);
        if (blk == 0u && threadIdx.x == 0) *ctrl_reset = SolverCtrl{};
    }
};

// The same source term for uniform-h scenes with mass-derived smoothing lengths on one context -- the headline -- on ONE gathered record
// per neighbour: {x_j, y_j, v_j}, as the sweep of the non-pressure forces (OpNonPressure::finish) or the divergence solve's tail
// (k_solver_tail, TAIL_VEL) left it, instead of the particle record (16 bytes) plus the velocity (8).  Round 5; the
// reasoning of OpJacobiU / OpPressureAccelU: a replay sweep costs its gather instructions.  The record holds no mass: m_j becomes the
// mass of particle 0 (see OpPressureAccelU) -- with equal masses the pair coefficient m_j / rho_i is the generic sweep's bit for bit,
// and so is everything behind it (same finish, same closed-form iteration 0, same residual statistics).  Divergence and full source
// terms (kind 0 / 1); the step driver says when the record is current (SweepArgs::xv_ok), else the sweep is OpSource.
template <class MathT>
struct OpSourceU : OpSource<MathT, false> {
    static_assert(MathT::UNIFORM, "OpSourceU: uniform-h scenes");
    typedef OpSource<MathT, false> B;
    typedef typename B::Acc Acc;
    static constexpr bool TILE = false;
    static constexpr bool OFF16 = true;    // (k_sweep_off: relative-offset lists)
    static constexpr int WIDE_TRIPS = 3;
    static constexpr int OFF_WAVES = 8;    // (see OpOffWaves)
    typedef NBNone NB;
    const float4* __restrict__ xv;
    __device__ float4 loadA(uint32_t j) const { return load_record(xv, j); }
    __device__ NB nb(const Acc&, uint32_t, float4) const { return NB{}; }
    __device__ void begin(Acc& a, uint32_t i, float4 Ai) const
    {
        a.sum = 0.f;
        a.rho_i = this->rho[i];
        a.inv_rho_i = fast_rcp(a.rho_i);
        a.qx = Ai.z;
        a.qy = Ai.w;
        a.err = 0.f;
        a.cls = 3u;
        a.om = 1.f;
        a.om_c = this->pm[0].z * a.inv_rho_i;   // the pair coefficient m / rho_i (om_c is IISPH2's otherwise)
        a.large = false;
    }
    __device__ void pair(Acc& a, float4 Aj, NB, float dx, float dy, float r2, float hij) const
    {
        const float e = fmaf(Aj.z - a.qx, dx, (Aj.w - a.qy) * dy);
        a.sum = fmaf(a.om_c * this->m.gscale(r2, hij), e, a.sum);
    }
};

// ------------------------------------------------------------------------------------------------
// Op: pressure acceleration (Jacobi sweep A)
//   calculate_particle_pressure_accel(s) / calculate_fluid_fluid_pressure_accel   simulation.rs:1518-1543, 1750-1808
//   iisph_boundary_pressure_accel                                  boundary_winchenbach2020.rs:164-194
// The neighbour payload p_j / rho_j^2 is written once per particle by the sweep that produced p.
// ------------------------------------------------------------------------------------------------
// What follows the LAST pressure-acceleration sweep of a solve (k_solver_tail, a per-particle map):
//   TAIL_NONE                 nothing
//   TAIL_VEL      HybridDFSPH after the divergence solve: v += dt a^p                    (simulation.rs:2547-2560)
//   TAIL_VX       IISPH / OnlyDivergence: v += dt a^p ; x += dt v                         (simulation.rs:2433-2445, 2486-2499)
//   TAIL_HYBRID   HybridDFSPH: x += dt v + dt^2 a^p ; v += dt a^p * min(dt*factor, 1)   (simulation.rs:2644-2646)
enum { TAIL_NONE = 0, TAIL_VEL = 1, TAIL_VX = 2, TAIL_HYBRID = 3 };

// parameters of the stop rule (iisph_pressure_iterations, simulation.rs:1453-1479)
struct SolveP {
    int residual_density;
    float max_avg_error;
    uint32_t max_iters;
    int multi;   // slab decomposition (0: one context -- block 0 of sweep A reduces the partials and decides):
                 // 1: block 0 only adds up this rank's totals (k_solver_totals, block 0 of k_pack_totals);
                 // 2: the launch runs BEHIND the all-reduce of the totals that travelled with the ghost exchange: its block 0 takes the
                 //    decision from them (solver_decide_multi) -- the place and the protocol of the one-context form, so sweep B and the
                 //    tail read one flag whatever the transport;
                 // 3: neither (the two launches of a split sweep A: k_solver_progress decides between them)
    uint32_t* prog = nullptr;   // paced solve: mapped host word that receives every decision (SweepArgs::prog_host), else nullptr
    uint32_t epoch = 0;
    // slab decomposition over RCCL: the ranks' totals arrive as an all-gather (one row per rank, tot_table[8 j ..]) in the iteration's
    // ONE grouped send / receive; whoever reads the "all-reduced" totals adds the rows up in rank order -- the same sum on every rank,
    // without a launch of its own for it.  nullptr: the transport reduced in place (tot[] holds the sums).
    const double* tot_table = nullptr;
    int tot_nr = 0, tot_self = 0;
};
// element k of the ranks' summed totals (solver_reduce_decide's layout)
__device__ __forceinline__ double solver_total(const double* __restrict__ tot, const SolveP& q, int k)
{
    if (!q.tot_table) return tot[k];
    double s = 0.0;
    for (int j = 0; j < q.tot_nr; j++) s += j == q.tot_self ? tot[k] : q.tot_table[8 * j + k];
    return s;
}
static SolveP solve_params(const SweepArgs& a, int residual_density, float max_avg_error, uint32_t max_iters, int multi, bool progress)
{
    SolveP q{residual_density, max_avg_error, max_iters, multi, progress ? a.prog_host : nullptr, a.prog_epoch};
    q.tot_table = a.tot_table;
    q.tot_nr = a.tot_nr;
    q.tot_self = a.tot_self;
    return q;
}

// stopping rule of iisph_pressure_iterations (simulation.rs:1453-1479) for iteration `iter`
__device__ __forceinline__ bool solver_stop_rule(uint32_t normal, float sum_err, int iter, const SolveP& q, float rest_density, float dt)
{
    const float avg = normal > 0 ? sum_err / (float)normal : __uint_as_float(0x7fc00000u);
    bool stop;
    if (q.residual_density) stop = normal == 0 || (fabsf(avg / rest_density) < q.max_avg_error && iter > 1);
    else stop = normal == 0 || (fabsf(avg) < q.max_avg_error / dt && iter > 1);
    if (!stop && (uint32_t)iter == q.max_iters) stop = true;
    return stop;
}

// Slab decomposition: the ranks' totals of iteration `iter` (tot[0..5], all-reduced behind sweep A(iter + 1)) -> the decision,
// evaluated by whoever needs it (every block of sweep B(iter + 1), k_solver_decide before a solve's tail).  tot[5] counts the
// ranks whose device-side guards fired: the solve then ends on every rank together (ctrl->peer_error) instead of leaving the
// others in a collective.  `publish`: this thread also writes the control block.
__device__ __forceinline__ bool solver_decide_multi(const double* __restrict__ tot, SolverCtrl* ctrl, int iter, const SolveP& q, float rest_density, float dt,
                                                    bool publish)
{
    const uint32_t slot = (uint32_t)(iter + 1) & 1u;
    if (ctrl->slot_done[iter & 1] != 0u) {   // decided earlier: hand the flag on (the totals are stale)
        if (publish) {
            ctrl->slot_done[slot] = 1u;
            if (q.prog) *(volatile uint32_t*)q.prog = (q.epoch << 16) | 0x8000u | ((uint32_t)iter & 0x7fffu);
        }
        return true;
    }
    const uint32_t normal = (uint32_t)solver_total(tot, q, 0);
    const float sum_err = (float)solver_total(tot, q, 3);
    const bool failed = solver_total(tot, q, 5) > 0.0;
    const bool stop = failed || solver_stop_rule(normal, sum_err, iter, q, rest_density, dt);
    if (publish) {
        ctrl->normal = normal;
        ctrl->singular = (uint32_t)solver_total(tot, q, 1);
        ctrl->negative = (uint32_t)solver_total(tot, q, 2);
        ctrl->sum_err = sum_err;
        ctrl->max_err = (float)solver_total(tot, q, 4);
        ctrl->iters = (uint32_t)iter;
        ctrl->cur = (uint32_t)((iter + 1) & 1);
        ctrl->slot_done[slot] = stop ? 1u : 0u;
        if (failed) ctrl->peer_error = 1u;
        if (stop) ctrl->done = 1u;
        // paced solve: the host learns of the decision while the launch that took it still sweeps
        if (q.prog) *(volatile uint32_t*)q.prog = (q.epoch << 16) | (stop ? 0x8000u : 0u) | ((uint32_t)iter & 0x7fffu);
    }
    return stop;
}

// PressureSolverStatistics of iteration `iter` from the per-block partials (fixed order: deterministic; the reference's rayon
// tree order is not) and the stop rule of simulation.rs:1453-1479.  Called by ALL threads of one block.
__device__ __forceinline__ void solver_reduce_decide(const SolverPartial* __restrict__ partials, uint32_t nparts, SolverCtrl* ctrl, double* __restrict__ tot,
                                                     int iter, const SolveP& q, float rest_density, float dt, const DeviceStatus* status)
{
    __shared__ SolverPartial s_r[SWEEP_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    SolverPartial t{0, 0, 0, 0.f, 0.f};
    // (four partials requested together, added in the order a one-at-a-time loop adds them: the decision is on the host's critical
    //  path -- a paced solve queues the next iteration when it arrives -- and sixteen dependent round trips were most of its 8 us)
    for (uint32_t k0 = tid; k0 < nparts; k0 += 4u * SWEEP_THREADS) {
        SolverPartial v[4];
#pragma unroll
        for (uint32_t u = 0; u < 4u; u++) {
            const uint32_t k = k0 + u * SWEEP_THREADS;
            v[u] = k < nparts ? partials[k] : SolverPartial{0, 0, 0, 0.f, 0.f};
        }
#pragma unroll
        for (uint32_t u = 0; u < 4u; u++) {
            t.normal += v[u].normal;
            t.singular += v[u].singular;
            t.negative += v[u].negative;
            t.sum_err += v[u].sum_err;
            t.max_err = fmaxf(t.max_err, v[u].max_err);
        }
    }
    t.normal = wave_sum_u32(t.normal);
    t.singular = wave_sum_u32(t.singular);
    t.negative = wave_sum_u32(t.negative);
    t.sum_err = wave_sum(t.sum_err);
    t.max_err = wave_max(t.max_err);
    if (lane == 0) s_r[w] = t;
    __syncthreads();
    if (tid != 0) return;
    t = s_r[0];
    for (int k = 1; k < SWEEP_THREADS / 64; k++) {
        t.normal += s_r[k].normal;
        t.singular += s_r[k].singular;
        t.negative += s_r[k].negative;
        t.sum_err += s_r[k].sum_err;
        t.max_err = fmaxf(t.max_err, s_r[k].max_err);
    }
    if (q.multi) {   // local totals (doubles, exact for the counts) -> all-reduce(sum) over the ranks -> k_solver_decide
        tot[0] = (double)t.normal;
        tot[1] = (double)t.singular;
        tot[2] = (double)t.negative;
        tot[3] = (double)t.sum_err;
        tot[4] = (double)t.max_err;   // summed over the ranks: informational
        tot[5] = status->error != 0u ? 1.0 : 0.0;   // a guard fired on this rank: every rank ends the solve (solver_decide_multi)
        return;
    }
    const bool stop = solver_stop_rule(t.normal, t.sum_err, iter, q, rest_density, dt);
    ctrl->normal = t.normal;
    ctrl->singular = t.singular;
    ctrl->negative = t.negative;
    ctrl->sum_err = t.sum_err;
    ctrl->max_err = t.max_err;
    ctrl->iters = (uint32_t)iter;
    ctrl->cur = (uint32_t)((iter + 1) & 1);  // mem::swap(pressure, pressure_next_iter)
    ctrl->slot_done[(iter + 1) & 1] = stop ? 1u : 0u;
    if (stop) ctrl->done = 1u;
    // paced solve: the host learns of the decision while this launch still sweeps (it queues the next iteration, or the tail)
    if (q.prog) *(volatile uint32_t*)q.prog = (q.epoch << 16) | (stop ? 0x8000u : 0u) | ((uint32_t)iter & 0x7fffu);
}

// Sweep A of iteration `iter` >= 1 reads pressure buffer iter & 1 -- and its block 0 first takes the stop decision of iteration
// iter - 1 from the partials sweep B left behind, while the other blocks already sweep (no reduction kernel, no launch gap
// between B and the next A).  If that decision is "stop", this launch WAS the solve's last pressure-acceleration sweep
// (simulation.rs:1499-1509: a^p from the final pressures) and every later launch of the solve returns at once.  The flag a
// launch writes is never one its own blocks read: slot_done[iter & 1] is written here and read by B(iter) and A(iter + 1);
// slot_done[(iter - 1) & 1] is what this launch tests.
template <class MathT>
struct OpPressureAccel {
    static constexpr bool TILE = true;
    typedef MathT Math;
    static constexpr bool HAS_EPILOGUE = false, SKIP_SELF = true;
    static constexpr bool EXTENDED = false, RING1 = true;
    static constexpr bool OFF16 = !MathT::EXACT;   // (k_sweep_off: relative-offset lists, when the step built them)
    __device__ constexpr float krange() const { return 2.f; }
    typedef float NB;  // p_j / (rho_j * rho_j)
    MathT m;
    const float4* __restrict__ pm;
    const uint32_t* __restrict__ orig;
    const float* __restrict__ rho;
    const float* __restrict__ p0;
    const float* __restrict__ p1;
    const float* __restrict__ pt0;
    const float* __restrict__ pt1;
    const float2* __restrict__ lam_grad;
    float4* __restrict__ pacc;   // {x, y, a^p}: the pressure acceleration travels with the position it belongs to, so that the Jacobi
                                 // update of a uniform-h scene gathers ONE 16-byte record per neighbour (OpJacobiU)
    SolverCtrl* ctrl;
    const SolverPartial* __restrict__ partials;
    uint32_t nparts;
    double* __restrict__ tot;
    DeviceStatus* status;
    StepP sp;
    SolveP solve;
    int iter;  // >= 1: Jacobi iteration `iter`; < 0: explicit sweep with the solve's final pressures (IISPH2 after its rescaling)
    const uint8_t* __restrict__ owned_flag;   // slab decomposition (else nullptr): ring-1 ghost lanes compute a^p as well
    const uint32_t* __restrict__ gate;        // chained solves (else nullptr), see OpSource
    struct Acc {
        float ax, ay, p1t;
        const float* pt;
        const float* p;
    };
    __device__ bool skip() const { return false; }
    __device__ bool prologue(uint32_t raw_block) const
    {
        if (gate && *gate == 0u) return true;
        if (iter < 0) return ctrl->done == 0u;
        if (ctrl->slot_done[(iter - 1) & 1] != 0u) {   // the solve ended before this launch: hand the flag on, leave
            if (raw_block == 0u && threadIdx.x == 0 && (solve.multi == 0 || solve.multi == 2)) {
                ctrl->slot_done[iter & 1] = 1u;
                if (solve.multi == 2 && solve.prog)   // (paced slabs: an unpaced head launch behind the end of the solve)
                    *(volatile uint32_t*)solve.prog = (solve.epoch << 16) | 0x8000u | ((uint32_t)(iter - 1) & 0x7fffu);
            }
            return true;
        }
        if (raw_block == 0u && solve.multi <= 1) solver_reduce_decide(partials, nparts, ctrl, tot, iter - 1, solve, sp.rest_density, sp.dt, status);
        if (raw_block == 0u && solve.multi == 2 && threadIdx.x == 0) solver_decide_multi(tot, ctrl, iter - 1, solve, sp.rest_density, sp.dt, true);
        return false;
    }
    __device__ bool lane_skip(uint32_t) const { return false; }
    __device__ void epilogue(Acc&, bool, uint32_t) const {}
    __device__ float4 loadA(uint32_t j) const { return pm[j]; }
    __device__ NB nb(const Acc& a, uint32_t j, float4) const { return a.pt[j]; }
    __device__ void init(Acc& a) const
    {
        const uint32_t cur = iter >= 0 ? (uint32_t)(iter & 1) : ctrl->cur;
        a.pt = cur ? pt1 : pt0;
        a.p = cur ? p1 : p0;
    }
    __device__ void begin(Acc& a, uint32_t i, float4) const
    {
        a.ax = a.ay = 0.f;
        a.p1t = a.pt[i];
    }
    __device__ void pair(Acc& a, float4 Aj, NB ptj, float dx, float dy, float r2, float hij) const
    {
        if constexpr (MathT::EXACT) {
            float gx, gy;
            m.grad(dx, dy, r2, hij, gx, gy);
            const float f = -Aj.z * (a.p1t + ptj);
            a.ax += f * gx;
            a.ay += f * gy;
        } else {
            const float fs = (-Aj.z * (a.p1t + ptj)) * m.gscale(r2, hij);
            a.ax = fmaf(fs, dx, a.ax);
            a.ay = fmaf(fs, dy, a.ay);
        }
    }
    __device__ bool finish(Acc& a, uint32_t i, float4 Ai, bool wall) const
    {
        float bx = 0.f, by = 0.f;
        const bool own = !owned_flag || owned_flag[i];
        if (sp.n_planes && wall) {
            const float rho_i = rho[i], rho_b = sp.rest_density;
            // a ghost has p / rho^2 (refreshed from its owner every iteration), not p
            const float pr2 = own ? a.p[i] / (rho_i * rho_i) : a.p1t;
            const float p_i = own ? a.p[i] : a.p1t * (rho_i * rho_i);
            const float p_ib = sp.opdisc == SPH_OP_SYMMETRIC_GRADIENT ? p_i : 0.f;
            const float f = -rho_b * (pr2 + p_ib / (rho_b * rho_b));
            const float2 gl = lam_grad[i];
            bx = f * gl.x;
            by = f * gl.y;
            if constexpr (MathT::EXACT) {   // iisph_boundary_pressure_accel entry by entry (boundary_winchenbach2020.rs:164-194)
                bx = by = 0.f;
                const uint32_t cnt = m.wall_count(i);
                for (uint32_t k = 0; k < cnt; k++) {
                    const float2 g = m.wall_entry(i, k);
                    bx += f * g.x;
                    by += f * g.y;
                }
            }
        }
        pacc[i] = make_float4(Ai.x, Ai.y, a.ax + bx, a.ay + by);
        return wall;
    }
};

// Sweep A for uniform-h scenes with mass-derived smoothing lengths on one context -- the headline again -- on ONE gathered record per
// neighbour: {x_j, y_j, p_j / rho_j^2, p_j}, as OpSource / OpJacobiU leave it (store_pressure), instead of the particle record (16
// bytes) plus the 4-byte p / rho^2.  Same reasoning and the same measurement as OpJacobiU (profiles/r3_jacobi_lab.md: 23.6 -> 20.3
// us stand-alone on the rest lattice, 31.3 -> 26.6 jittered; sweep B pays 1-2 us for storing 12 more bytes).  The record holds no
// mass: m_j becomes the mass of particle 0 (one scalar load) -- equal, or a neighbouring float, on every particle of such a scene
// (see OpJacobiU).  The particle's own record brings its own p / rho^2 and p along.  Decision, prologue and output are the base's.
template <class MathT>
struct OpPressureAccelU : OpPressureAccel<MathT> {
    static_assert(MathT::UNIFORM, "OpPressureAccelU: uniform-h scenes");
    typedef OpPressureAccel<MathT> B;
    static constexpr bool TILE = false, RING1 = true;   // (slab decomposition: the first ghost ring computes its own a^p, as in the base)
    static constexpr bool OFF16 = true;                 // (k_sweep_off: relative-offset lists)
    static constexpr int WIDE_TRIPS = 3;                // (measured: 3 and 2 alike, 1 is 1.3 % slower on the driver window)
    static constexpr int OFF_WAVES = 8;                 // (see OpOffWaves)
    typedef NBNone NB;
    const float4* __restrict__ rec;   // of the pressure buffer this iteration reads (slabs: the ghosts' records carry their owners' p / rho^2, refreshed every iteration)
    struct Acc {
        float ax, ay, p1t, mass;
    };
    __device__ float4 loadA(uint32_t j) const { return load_record(rec, j); }
    __device__ NB nb(const Acc&, uint32_t, float4) const { return NB{}; }
    __device__ void init(Acc& a) const { a.mass = this->pm[0].z; }
    __device__ void begin(Acc& a, uint32_t, float4 Ai) const
    {
        a.ax = a.ay = 0.f;
        a.p1t = Ai.z;
    }
    __device__ void pair(Acc& a, float4 Aj, NB, float dx, float dy, float r2, float hij) const
    {
        const float fs = (-a.mass * (a.p1t + Aj.z)) * this->m.gscale(r2, hij);
        a.ax = fmaf(fs, dx, a.ax);
        a.ay = fmaf(fs, dy, a.ay);
    }
    __device__ bool finish(Acc& a, uint32_t i, float4 Ai, bool wall) const
    {
        float bx = 0.f, by = 0.f;
        if (this->sp.n_planes && wall) {
            const float rho_b = this->sp.rest_density;
            // (a ghost's record holds p / rho^2 only: p from it and the ghost's refreshed density, like the base does)
            const bool own = !this->owned_flag || this->owned_flag[i];
            float p_i = Ai.w;
            if (!own) {
                const float rho_i = this->rho[i];
                p_i = Ai.z * (rho_i * rho_i);
            }
            const float p_ib = this->sp.opdisc == SPH_OP_SYMMETRIC_GRADIENT ? p_i : 0.f;
            const float f = -rho_b * (Ai.z + p_ib / (rho_b * rho_b));
            const float2 gl = this->lam_grad[i];
            bx = f * gl.x;
            by = f * gl.y;
        }
        this->pacc[i] = make_float4(Ai.x, Ai.y, a.ax + bx, a.ay + by);
        return wall;
    }
    __device__ void epilogue(Acc&, bool, uint32_t) const {}
};

// The solve's tail: the integrate map of the solver mode (TAIL_*) on the owned particles, run once the stop decision is taken.
// The integrating tails know the positions and velocities the NEXT step starts from: they also reduce that step's header
// (bounding box, h range, CFL term -- what k_header computes) per block, so the next step starts without the header kernels
// and without the host wait behind them (hdr_partials == nullptr: not wanted).
__global__ __launch_bounds__(256) void k_solver_tail(uint32_t n, int tail, float dt, float vfactor, const float4* __restrict__ pm, float4* __restrict__ pm_out,
                                                      float2* __restrict__ vel, const float4* __restrict__ pacc, const uint32_t* __restrict__ orig,
                                                      const uint8_t* __restrict__ owned, const SolverCtrl* __restrict__ ctrl, HeaderOut* __restrict__ hdr_partials,
                                                      DeviceStatus* status, const uint32_t* __restrict__ gate, const double* __restrict__ tot,
                                                      int decide_iter, SolveP solve, float rest_density, SolverCtrl* __restrict__ handoff_host,
                                                      uint32_t* __restrict__ gate_out, IncClassifyP inc, float4* __restrict__ xv)
{
    if (gate && *gate == 0u) return;
    const bool b0t0 = blockIdx.x == 0 && threadIdx.x == 0;
    (void)decide_iter;
    (void)tot;
    const bool done = ctrl->done != 0u;   // (the decision was taken by the last sweep A's block 0, or the launch that stood in for it)
    // chained solves: this is the FIRST solve's tail -- hand its control block to the host and open (or keep shut) the gate of
    // the second solve's launches (k_solver_handoff's job; the next kernel starts after every block of this one has finished)
    if (gate_out && b0t0) {
        const SolverCtrl v = *ctrl;
        *handoff_host = v;
        __threadfence_system();
        *gate_out = done ? 1u : 0u;
    }
    if (!done) return;
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    const bool active = i < n && (!owned || owned[i]);
    float nx = 0.f, ny = 0.f, nh = 0.f, ncfl = 0.f;
    if (active) {
        // (TAIL_VEL moves no particle and reduces no header: it does not need the record)
        const float4 Ai = tail == TAIL_VEL ? make_float4(0.f, 0.f, 0.f, 0.f) : pm[i];
        const float4 rec = pacc[i];
        const float2 ap = make_float2(rec.z, rec.w);
        float2 v = vel[i];
        float4 p = Ai;
        if (tail == TAIL_VEL) {
            v.x += dt * ap.x;
            v.y += dt * ap.y;
            if (!isfinite(v.x) || !isfinite(v.y)) raise_error(status, SPH_ERR_VELOCITY_NOT_FINITE, orig[i]);
            if (xv) xv[i] = make_float4(rec.x, rec.y, v.x, v.y);   // (the a^p record carries the position: sweep A's finish; OpSourceU gathers this)
        } else if (tail == TAIL_VX) {
            v.x += dt * ap.x;
            v.y += dt * ap.y;
            p.x += dt * v.x;
            p.y += dt * v.y;
            if (!isfinite(v.x) || !isfinite(v.y)) raise_error(status, SPH_ERR_VELOCITY_NOT_FINITE, orig[i]);
            pm_out[i] = p;
        } else {
            p.x += dt * v.x + dt * dt * ap.x;
            p.y += dt * v.y + dt * dt * ap.y;
            v.x += dt * ap.x * vfactor;
            v.y += dt * ap.y * vfactor;
            if (!isfinite(p.x) || !isfinite(p.y)) raise_error(status, SPH_ERR_POSITION_NOT_FINITE, orig[i]);
            pm_out[i] = p;
        }
        vel[i] = v;
        // the incremental cell sort queued behind this launch (sph_sort.hip) wants every particle's new cell: the position is at hand
        if (inc.head && tail >= TAIL_VX) inc_classify_particle(inc, i, p.x, p.y);   // (launch-uniform test)
        nx = p.x;
        ny = p.y;
        nh = Ai.w;
        const float sr = Ai.w * 2.f;
        ncfl = sr * sr / ((v.x * v.x + v.y * v.y) + 0.01f);   // simulation.rs:2182-2189
    }
    if (!hdr_partials || tail < TAIL_VX) return;   // launch-uniform
    const float INF = __uint_as_float(0x7f800000u);
    float mnx = active ? nx : INF, mxx = active ? nx : -INF, mny = active ? ny : INF, mxy = active ? ny : -INF;
    float hmx = active ? nh : 0.f, hmn = active ? nh : INF;
    float cfl = active ? ncfl : INF;
    // (every thread of the block is here -- the exits above are launch- or block-uniform: the DPP network sees full waves)
    mnx = wave_min_dpp(mnx); mny = wave_min_dpp(mny); mxx = wave_max_dpp(mxx); mxy = wave_max_dpp(mxy);
    hmx = wave_max_dpp(hmx); hmn = wave_min_dpp(hmn); cfl = wave_min_dpp(cfl);
    __shared__ HeaderOut s_h[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) s_h[w] = HeaderOut{mnx, mny, mxx, mxy, hmx, hmn, cfl, 0};
    __syncthreads();
    if (threadIdx.x == 0) {
        HeaderOut o = s_h[0];
        for (int k = 1; k < 4; k++) {
            o.min_x = fminf(o.min_x, s_h[k].min_x); o.min_y = fminf(o.min_y, s_h[k].min_y);
            o.max_x = fmaxf(o.max_x, s_h[k].max_x); o.max_y = fmaxf(o.max_y, s_h[k].max_y);
            o.h_max = fmaxf(o.h_max, s_h[k].h_max); o.h_min = fminf(o.h_min, s_h[k].h_min);
            o.min_cfl = fminf(o.min_cfl, s_h[k].min_cfl);
        }
        hdr_partials[blockIdx.x] = o;
    }
}

// ------------------------------------------------------------------------------------------------
// Op: relaxed-Jacobi pressure update (sweep B)
//   iisph_single_pressure_iteration    simulation.rs:1241-1319
//   a_ii >= 0 pre-check                simulation.rs:1390-1403 (done here at iteration 0)
// Per-particle residual class goes to stat[] (normal: the error; singular / negative: tagged NaNs),
// reduced in fixed particle order by k_solver_reduce.
// ------------------------------------------------------------------------------------------------
template <class MathT>
struct OpJacobi {
    static constexpr bool TILE = true;
    typedef MathT Math;
    static constexpr bool HAS_EPILOGUE = true, SKIP_SELF = true;
    static constexpr bool EXTENDED = false;
    static constexpr bool OFF16 = !MathT::EXACT;   // (k_sweep_off: relative-offset lists, when the step built them)
    __device__ constexpr float krange() const { return 2.f; }
    typedef NBVecMr NB;
    MathT m;
    const float4* __restrict__ pm;
    const uint32_t* __restrict__ orig;
    const float* __restrict__ rho;
    const float* __restrict__ mrho;
    const float4* __restrict__ pacc;   // {x, y, a^p} (OpPressureAccel)
    const float2* __restrict__ lam_grad;
    const float* __restrict__ aii;
    const float* __restrict__ src;
    const float* __restrict__ p_in;
    float* __restrict__ p_out;
    float* __restrict__ pterm_out;
    float4* __restrict__ rec_out;   // see store_pressure (OpJacobiU on one context, else nullptr)
    float* __restrict__ dens_err;
    SolverPartial* __restrict__ partials;
    SolverCtrl* ctrl;
    DeviceStatus* status;
    StepP sp;
    int iter;
    int residual_density;
    struct Acc {
        float sum, rho_i, inv_rho_i, qx, qy;
        float err;       // residual of a "normal" particle
        uint32_t cls;    // 0 normal, 1 singular, 2 negative (PressureSolverStatistics, simulation.rs:397-445)
        float aii_i, src_i, pin_i;   // what finish() needs of the particle itself, requested with the rest at the head of the wave
    };
    SolveP solve;
    const double* __restrict__ tot;
    const uint32_t* __restrict__ gate;   // chained solves (else nullptr), see OpSource
    __device__ bool skip() const { return false; }
    // the decision on iteration iter - 1 was taken by block 0 of A(iter) -- from the partials (one context), or from the ranks'
    // all-reduced totals (slab decomposition; a split sweep A: by k_solver_progress between its two launches)
    __device__ bool prologue(uint32_t raw_block) const
    {
        if (gate && *gate == 0u) return true;
        return ctrl->slot_done[iter & 1] != 0u;
    }
    __device__ bool lane_skip(uint32_t) const { return false; }
    __device__ void init(Acc&) const {}
    __device__ float4 loadA(uint32_t j) const { return pm[j]; }
    __device__ NB nb(const Acc&, uint32_t j, float4) const
    {
        const float2 a = *(reinterpret_cast<const float2*>(pacc + j) + 1);   // (8 of the record's 16 bytes: this form also gathers pm[j])
        float mr = 0.f;
        if (sp.opdisc == SPH_OP_WINCHENBACH2020) mr = mrho[j];
        return NB{a.x, a.y, mr};
    }
    __device__ void begin(Acc& a, uint32_t i, float4) const
    {
        a.sum = 0.f;
        a.rho_i = rho[i];
        a.inv_rho_i = fast_rcp(a.rho_i);
        const float4 q = pacc[i];
        a.qx = q.z;
        a.qy = q.w;
        a.err = 0.f;
        a.cls = 3u;
        a.aii_i = aii[i];
        a.src_i = src[i];
        a.pin_i = p_in[i];
    }
    __device__ void pair(Acc& a, float4 Aj, NB Bj, float dx, float dy, float r2, float hij) const
    {
        if constexpr (MathT::EXACT) {
            float gx, gy;
            m.grad(dx, dy, r2, hij, gx, gy);
            const float dot = (Bj.qx - a.qx) * gx + (Bj.qy - a.qy) * gy;
            if (sp.opdisc == SPH_OP_WINCHENBACH2020) a.sum += Bj.mr * dot;
            else a.sum += Aj.z / a.rho_i * dot;
        } else {
            const float e = fmaf(Bj.qx - a.qx, dx, (Bj.qy - a.qy) * dy);
            const float c = sp.opdisc == SPH_OP_WINCHENBACH2020 ? Bj.mr : Aj.z * a.inv_rho_i;
            a.sum = fmaf(c * m.gscale(r2, hij), e, a.sum);
        }
    }
    __device__ bool finish(Acc& a, uint32_t i, float4 Ai, bool wall) const
    {
        const float aii_i = a.aii_i;
        if (fabsf(aii_i) < 10e-4f) {
            store_pressure(p_out, pterm_out, rec_out, i, Ai, 0.f, 0.f);
            a.cls = 1u;
            return wall;
        }
        const float rho_i = a.rho_i, rho_b = sp.rest_density, dt = sp.dt;
        float bdiv = 0.f;
        if (sp.n_planes && wall) {
            const float2 gl = lam_grad[i];
            const float bdot = (0.f - a.qx) * gl.x + (0.f - a.qy) * gl.y;
            bdiv = sp.opdisc == SPH_OP_WINCHENBACH2020 ? bdot : rho_b / rho_i * bdot;
            if constexpr (MathT::EXACT) bdiv = m.wall_divergence(i, a.qx, a.qy, rho_i, rho_b, sp.opdisc == SPH_OP_WINCHENBACH2020);
        }
        const float a_p = a.sum + bdiv;
        const float s = a.src_i;
        if (!isfinite(a_p)) raise_error(status, SPH_ERR_AP_NOT_FINITE, orig[i]);
        float pn = a.pin_i + sp.jacobi_omega * (s - a_p) / aii_i;
        if (!isfinite(pn)) raise_error(status, SPH_ERR_PRESSURE_NOT_FINITE, orig[i]);
        float err;
        if (residual_density) {
            err = rho_i * dt * dt * (s - a_p);
            dens_err[i] = err;
        } else {
            err = dt * (s - a_p);
        }
        if (pn <= 0.f) {  // clamp_negative_pressures is true at every call site
            store_pressure(p_out, pterm_out, rec_out, i, Ai, 0.f, 0.f);
            a.cls = 2u;
        } else {
            store_pressure(p_out, pterm_out, rec_out, i, Ai, pn, pn / (rho_i * rho_i));   // p_j / (rho_j * rho_j) of calculate_fluid_fluid_pressure_accel
            a.cls = 0u;
            a.err = err;
        }
        return wall;
    }
    __device__ void epilogue(Acc& a, bool active, uint32_t blk) const { solver_block_partial(partials, active ? a.cls : 3u, a.err, blk); }
};


// The same update for uniform-h scenes with mass-derived smoothing lengths -- BASELINE configs[1] and [3], the headline -- on ONE
// gathered record per neighbour: {x_j, y_j, a^p_j}, as sweep A leaves it (16 bytes), instead of the particle record (16) plus
// the a^p payload (8).  What a replay sweep costs is its gather INSTRUCTIONS more than their bytes (profiles/r3_jacobi_lab.md:
// 27.0 -> 21.9 us for this sweep stand-alone), and the particle's own record brings its own a^p along.  The record holds no mass:
// the pair coefficient m_j / rho_i becomes m_i / rho_i (the density sweep's m / rho of the particle itself).  With h = 1.9 sqrt(m /
// (rho_0 pi)) bit-identical on every particle the masses are equal or neighbouring floats (the map m -> h loses one bit), so
// the substitution is exact or 1.2e-7 relative per pair -- the size of v_rsq_f32's own error.  Everything else is OpJacobi: same
// finish, same residual statistics, same stop decision.  Not for the consistent-gradient-by-volume discretisation (per-neighbour
// m_j / rho_j), EXACT mode, non-uniform h or FromDistribution* support lengths: those take OpJacobi.
template <class MathT>
struct OpJacobiU : OpJacobi<MathT> {
    static_assert(MathT::UNIFORM, "OpJacobiU: uniform-h scenes");
    typedef OpJacobi<MathT> B;
    typedef typename B::Acc Acc;
    static constexpr bool TILE = false;
    static constexpr bool OFF16 = true;   // (k_sweep_off: relative-offset lists)
    static constexpr int WIDE_TRIPS = 3;   // (OpJacobiU 18.2 -> 17.1 us)
    static constexpr int OFF_WAVES = 8;    // (see OpOffWaves)
    typedef NBNone NB;
    __device__ float4 loadA(uint32_t j) const { return load_record(this->pacc, j); }
    __device__ NB nb(const Acc&, uint32_t, float4) const { return NB{}; }
    __device__ void begin(Acc& a, uint32_t i, float4 Ai) const
    {
        a.sum = 0.f;
        a.rho_i = this->rho[i];
        a.inv_rho_i = this->mrho[i];   // m_i / rho_i: the pair coefficient (see above)
        a.qx = Ai.z;
        a.qy = Ai.w;
        a.err = 0.f;
        a.cls = 3u;
        a.aii_i = this->aii[i];
        a.src_i = this->src[i];
        a.pin_i = this->p_in[i];
    }
    __device__ void pair(Acc& a, float4 Aj, NB, float dx, float dy, float r2, float hij) const
    {
        const float e = fmaf(Aj.z - a.qx, dx, (Aj.w - a.qy) * dy);
        a.sum = fmaf(this->m.gscale(r2, hij), e, a.sum);
    }
    __device__ bool finish(Acc& a, uint32_t i, float4 Ai, bool wall) const
    {
        a.sum *= a.inv_rho_i;
        return B::finish(a, i, Ai, wall);
    }
};

// ------------------------------------------------------------------------------------------------
// Op: check_aii (simulation.rs:1347-1375): a_ii against the operator applied to the unit pressure field e_i
//   calculate_aii_inefficiently = div( a^p[e_i] )_i  (simulation.rs:1324-1345, 1594-1631), tolerance 0.01 in f32.
// With p = e_i only two kinds of pressure acceleration are non-zero: a^p_i (OpAiiConst stored it) and, for a neighbour j,
// a^p_j = -m_i (0 + 1/rho_i^2) grad W_ji = (m_i / rho_i^2) grad W_ij -- so the composition is one more sweep.
// ------------------------------------------------------------------------------------------------
template <class MathT>
struct OpCheckAii {
    typedef MathT Math;
    static constexpr bool HAS_EPILOGUE = false, SKIP_SELF = true;
    static constexpr bool EXTENDED = false;
    __device__ constexpr float krange() const { return 2.f; }
    typedef float NB;   // m_j / rho_j
    MathT m;
    const float4* __restrict__ pm;
    const uint32_t* __restrict__ orig;
    const float* __restrict__ rho;
    const float* __restrict__ mrho;
    const float2* __restrict__ lam_grad;
    const float* __restrict__ aii;
    const float2* __restrict__ ap_unit;
    DeviceStatus* status;
    StepP sp;
    struct Acc {
        float sum, rho_i, c_i, qx, qy;
    };
    __device__ bool skip() const { return false; }
    __device__ bool lane_skip(uint32_t) const { return false; }
    __device__ void init(Acc&) const {}
    __device__ void epilogue(Acc&, bool, uint32_t) const {}
    __device__ float4 loadA(uint32_t j) const { return pm[j]; }
    __device__ NB nb(const Acc&, uint32_t j, float4) const { return mrho[j]; }
    __device__ void begin(Acc& a, uint32_t i, float4 Ai) const
    {
        a.sum = 0.f;
        a.rho_i = rho[i];
        a.c_i = Ai.z / (a.rho_i * a.rho_i);   // m_i / rho_i^2
        const float2 q = ap_unit[i];
        a.qx = q.x;
        a.qy = q.y;
    }
    __device__ void pair(Acc& a, float4 Aj, NB mrj, float dx, float dy, float r2, float hij) const
    {
        float gx, gy;
        m.grad(dx, dy, r2, hij, gx, gy);
        // the self pair has grad W = 0: its (Q_i - Q_i) term vanishes either way
        const float dot = (a.c_i * gx - a.qx) * gx + (a.c_i * gy - a.qy) * gy;   // (a^p_j - a^p_i) . grad W_ij
        if (sp.opdisc == SPH_OP_WINCHENBACH2020) a.sum += mrj * dot;
        else a.sum += Aj.z / a.rho_i * dot;
    }
    __device__ bool finish(Acc& a, uint32_t i, float4, bool wall) const
    {
        float bdiv = 0.f;
        if (sp.n_planes && wall) {
            const float2 gl = lam_grad[i];
            const float bdot = (0.f - a.qx) * gl.x + (0.f - a.qy) * gl.y;
            bdiv = sp.opdisc == SPH_OP_WINCHENBACH2020 ? bdot : sp.rest_density / a.rho_i * bdot;
            if constexpr (MathT::EXACT) bdiv = m.wall_divergence(i, a.qx, a.qy, a.rho_i, sp.rest_density, sp.opdisc == SPH_OP_WINCHENBACH2020);
        }
        const float real = a.sum + bdiv;
        const float v = aii[i];
        if (!(real <= v + 0.01f && real >= v - 0.01f)) raise_error(status, SPH_ERR_CHECK_AII, orig[i]);   // assert_ft_approx_eq
        return wall;
    }
};

// ================================================================================================
// Level estimation (distance-to-surface field of the adaptivity; simulation.rs:539-927).  It works on its own
// neighbour lists of range k = level_estimation_range / ETA (use_extended_range_for_level_estimation; build_
// neighborhood_list with that k, simulation.rs:2018-2046), built by OpLevelNormal and replayed by the others.
// ================================================================================================
#define LVL_UNASSIGNED 0xffffffffu   // OpLevelPropagate::when of a particle without a value

// (particle_radius * maximum_range)^2 of is_neighbor_in_level_estimation_range (simulation.rs:698-723), +inf when the rule is off
// (maximum_range < 0 here); particle_radius = sphere_volume_to_radius(m / rho0) = sqrt((m / rho0) * (1 / pi)) (sph_kernels.rs:203-206)
__device__ __forceinline__ float level_range_sq(float mass, float maximum_range, float rest_density)
{
    if (maximum_range < 0.f) return __uint_as_float(0x7f800000u);
    const float pr = sqrtf((mass / rest_density) * SPH_FRAC_1_PI_F);
    const float r = pr * maximum_range;
    return r * r;
}

// Op: surface detection, part 1 (surface_detection_by_empty_angle, simulation.rs:539-583): the "normal"
//   n_i = - sum_j (m_i / rho_0) grad W_ij  and the cheap classifications.  state: 0 interior, 1 surface,
//   2 decided by the cone test of part 2.
template <class MathT>
struct OpLevelNormal {
    typedef MathT Math;
    static constexpr bool HAS_EPILOGUE = false, SKIP_SELF = false, EXTENDED = true;
    typedef NBNone NB;
    MathT m;
    const float4* __restrict__ pm;
    const float4* __restrict__ pm_cell;   // pre-step positions when pm holds the advected ones, else nullptr
    float2* __restrict__ nrm;
    uint8_t* __restrict__ state;
    uint8_t* __restrict__ flag_insufficient;
    const BoundaryP* __restrict__ planes;
    StepP sp;
    float k;
    int boundary_is_fluid_surface;
    struct Acc {
        float nx, ny, f;
        uint32_t cnt;
    };
    __device__ float krange() const { return k; }
    __device__ bool skip() const { return false; }
    __device__ bool lane_skip(uint32_t) const { return false; }
    __device__ void init(Acc&) const {}
    __device__ void epilogue(Acc&, bool, uint32_t) const {}
    __device__ float2 cell_pos(uint32_t i, const float4& Ai) const
    {
        if (!pm_cell) return make_float2(Ai.x, Ai.y);
        const float4 q = pm_cell[i];
        return make_float2(q.x, q.y);
    }
    __device__ float4 loadA(uint32_t j) const { return pm[j]; }
    __device__ NB nb(const Acc&, uint32_t, float4) const { return NB{}; }
    __device__ void begin(Acc& a, uint32_t, float4 Ai) const
    {
        a.nx = a.ny = 0.f;
        a.cnt = 0;
        a.f = Ai.z / sp.rest_density;
    }
    __device__ void pair(Acc& a, float4, NB, float dx, float dy, float r2, float hij) const
    {
        float gx, gy;
        m.grad(dx, dy, r2, hij, gx, gy);
        a.nx -= a.f * gx;
        a.ny -= a.f * gy;
        a.cnt++;
    }
    __device__ bool finish(Acc& a, uint32_t i, float4 Ai, bool) const
    {
        uint8_t st, insufficient = 0;
        const float n2 = a.nx * a.nx + a.ny * a.ny;
        if (a.cnt < 3u) {   // neighs.len() < 2 * D - 1
            st = 1;
            insufficient = 1;
        } else if (n2 < 0.00001f) {
            st = 0;
        } else {
            float dist = __uint_as_float(0x7f800000u);   // BoundaryHandler::distance_to_boundary (boundary_winchenbach2020.rs:320-325)
            for (int q = 0; q < sp.n_planes; q++) dist = fminf(dist, sdf_probe(planes, q, Ai.x, Ai.y));
            if (!boundary_is_fluid_surface && dist < Ai.w * 1.5f) {
                st = 0;
            } else {
                st = 2;
                const float nn = sqrtf(n2);
                nrm[i] = make_float2(a.nx / nn, a.ny / nn);
            }
        }
        state[i] = st;
        flag_insufficient[i] = insufficient;
        return false;
    }
};

// Op: surface detection, part 2 (simulation.rs:584-625): a particle whose normal cone (50 degrees) contains no
// neighbour is on the surface: level 0 (FluidSurface(0)), everything else NaN (FluidInterior).
template <class MathT>
struct OpLevelCone {
    typedef MathT Math;
    static constexpr bool HAS_EPILOGUE = false, SKIP_SELF = false, EXTENDED = true;
    typedef NBNone NB;
    MathT m;
    const float4* __restrict__ pm;
    const float4* __restrict__ pm_cell;   // pre-step positions when pm holds the advected ones, else nullptr
    const float2* __restrict__ nrm;
    const uint8_t* __restrict__ state;
    float* __restrict__ level;
    uint32_t* __restrict__ when;   // propagation sweep that gave the particle its value: 0 surface, LVL_UNASSIGNED none yet
    uint32_t* __restrict__ mark;   // two frontier-mark words per particle: [i] read by even sweeps, [i + mark_stride] by odd ones
    uint32_t mark_stride;
    uint8_t* __restrict__ flag_surface;
    float* __restrict__ stash;   // fill_stash_with == SurfaceDistanceFirst (simulation.rs:886-893), else nullptr
    float k, threshold, max_surface_distance;
    // is_neighbor_in_level_estimation_range (simulation.rs:698-723): FromDistribution / FromDistribution2 only -- neighbours
    // beyond particle_radius(i) * maximum_range do not count; range_factor2 = maximum_range^2 / rest_density / pi, or < 0: off
    float range_factor, sp_rest_density;
    struct Acc {
        float nx, ny, r2max;
        uint32_t st;
        bool hit;
    };
    __device__ float krange() const { return k; }
    __device__ bool skip() const { return false; }
    __device__ bool lane_skip(uint32_t) const { return false; }
    __device__ void init(Acc&) const {}
    __device__ void epilogue(Acc&, bool, uint32_t) const {}
    __device__ float2 cell_pos(uint32_t i, const float4& Ai) const
    {
        if (!pm_cell) return make_float2(Ai.x, Ai.y);
        const float4 q = pm_cell[i];
        return make_float2(q.x, q.y);
    }
    __device__ float4 loadA(uint32_t j) const { return pm[j]; }
    __device__ NB nb(const Acc&, uint32_t, float4) const { return NB{}; }
    __device__ void begin(Acc& a, uint32_t i, float4 Ai) const
    {
        a.st = state[i];
        a.hit = false;
        a.nx = a.ny = 0.f;
        a.r2max = level_range_sq(Ai.z, range_factor, sp_rest_density);
        if (a.st == 2u) {
            const float2 n = nrm[i];
            a.nx = n.x;
            a.ny = n.y;
        }
    }
    __device__ void pair(Acc& a, float4, NB, float dx, float dy, float r2, float) const
    {
        if (a.st != 2u) return;
        if (r2 > a.r2max) return;
        // x_j - x_i = -(x_i - x_j) exactly; its norm_squared is r2
        const float dn = sqrtf(r2) + 0.000001f;
        const float ux = -dx / dn, uy = -dy / dn;
        if (ux * a.nx + uy * a.ny > threshold) a.hit = true;
    }
    __device__ bool finish(Acc& a, uint32_t i, float4, bool) const
    {
        const bool interior = a.st == 0u || (a.st == 2u && a.hit);
        const float v = interior ? __uint_as_float(0x7fc00000u) : 0.f;
        level[i] = v;
        when[i] = interior ? LVL_UNASSIGNED : 0u;
        mark[i] = 0u;
        mark[i + mark_stride] = 0u;
        flag_surface[i] = interior ? 0 : 1;
        if (stash) stash[i] = interior ? -max_surface_distance : v;
        return false;
    }
};

// Op: surface_detection_by_center_diff (simulation.rs:631-695, "Mass preserving multi-scale SPH"): phi = |x_i - weighted mean of
// the neighbours' positions| - weighted mean radius; surface if phi >= -0.85 mean radius.  Only reachable after advection (the
// reference refuses it before, simulation.rs:2029-2031).  BUILD sweep of the extended lists, like OpLevelNormal.
template <class MathT>
struct OpLevelCenterDiff {
    typedef MathT Math;
    static constexpr bool HAS_EPILOGUE = false, SKIP_SELF = false, EXTENDED = true;
    typedef NBNone NB;
    MathT m;
    const float4* __restrict__ pm;
    const float4* __restrict__ pm_cell;   // pre-step positions when pm holds the advected ones, else nullptr
    float* __restrict__ level;
    uint32_t* __restrict__ when;
    uint32_t* __restrict__ mark;
    uint32_t mark_stride;
    uint8_t* __restrict__ flag_surface;
    float* __restrict__ stash;   // fill_stash_with == SurfaceDistanceFirst, else nullptr
    float k, max_surface_distance, rest_density;
    struct Acc {
        float cx, cy, rad, wsum;
        uint32_t num;
    };
    __device__ float krange() const { return k; }
    __device__ bool skip() const { return false; }
    __device__ bool lane_skip(uint32_t) const { return false; }
    __device__ void init(Acc&) const {}
    __device__ void epilogue(Acc&, bool, uint32_t) const {}
    __device__ float2 cell_pos(uint32_t i, const float4& Ai) const
    {
        if (!pm_cell) return make_float2(Ai.x, Ai.y);
        const float4 q = pm_cell[i];
        return make_float2(q.x, q.y);
    }
    __device__ float4 loadA(uint32_t j) const { return pm[j]; }
    __device__ NB nb(const Acc&, uint32_t, float4) const { return NB{}; }
    __device__ void begin(Acc& a, uint32_t, float4) const
    {
        a.cx = a.cy = a.rad = a.wsum = 0.f;
        a.num = 0;
    }
    __device__ void pair(Acc& a, float4 Aj, NB, float, float, float r2, float hij) const
    {
        const float vol = Aj.z / rest_density;
        const float r = sqrtf(vol * SPH_FRAC_1_PI_F);
        const float w = m.w(r2, hij) * vol;
        a.cx += Aj.x * w;
        a.cy += Aj.y * w;
        a.rad += r * w;
        a.wsum += w;
        a.num++;
    }
    __device__ bool finish(Acc& a, uint32_t i, float4 Ai, bool) const
    {
        const float rad = a.rad / a.wsum;
        const float surface_level = -0.85f * rad;
        float phi;
        if (a.num < 5u) {
            phi = surface_level;
        } else {
            const float dx = Ai.x - a.cx / a.wsum, dy = Ai.y - a.cy / a.wsum;
            phi = sqrtf(dx * dx + dy * dy) - rad;
        }
        const bool surface = phi >= surface_level;
        const float v = surface ? phi : __uint_as_float(0x7fc00000u);
        level[i] = v;
        when[i] = surface ? 0u : LVL_UNASSIGNED;
        mark[i] = 0u;
        mark[i + mark_stride] = 0u;
        flag_surface[i] = surface ? 1 : 0;
        if (stash) stash[i] = surface ? v : -max_surface_distance;
        return false;
    }
};

// Op: one propagation sweep (propagate_level_set_from_surface_detection, simulation.rs:729-801): a particle without a
// value takes max_j (level_j - |x_ij|) over the neighbours that had one when the sweep started; valued particles keep
// theirs.  The reference re-evaluates every particle in every sweep (K ~ depth / range sweeps over N particles).  Here a
// value is final once written, so `when[i]` (the sweep that wrote it) replaces the reference's double buffer -- sweep t
// reads only neighbours with when < t -- and only the frontier works: a particle assigned in sweep t-1 marks its
// unassigned neighbours for sweep t while it walks its list, everyone else leaves after two loads.  Same values, same
// sweep count (a particle is assigned in the first sweep t in which a neighbour has when <= t-1, and then one of them
// has when == t-1 and has marked it).  Sweep 0 only lets the surface particles mark their neighbours.
// The marks are double-buffered by sweep parity: sweep t tests mark_cur[i] == t and writes t + 1 into mark_next.  With ONE
// array a candidate of sweep t could find its own mark already overwritten with t + 1 by an EARLIER block of the same launch
// (a neighbouring candidate that still saw it unassigned) and sit the sweep out; assigned one sweep late, it was invisible to
// the neighbours that needed it -- 0.5 % of the particles of BASELINE configs[4] (4M particles, blocks of one launch start
// milliseconds apart) ended up with a longer path, up to 10 spacings deeper than the reference's field.
struct NBLevel {
    uint32_t w;
    float lv;
    uint32_t j;
};
template <class MathT, bool SLAB>   // SLAB: the frontier form of a slab decomposition (mode 2 below) with its second pass
struct OpLevelPropagateT {
    typedef MathT Math;
    static constexpr bool HAS_EPILOGUE = false, SKIP_SELF = false, EXTENDED = true;
    typedef NBLevel NB;
    MathT m;
    const float4* __restrict__ pm;
    const float4* __restrict__ pm_cell;   // pre-step positions when pm holds the advected ones, else nullptr
    float* __restrict__ level;
    uint32_t* __restrict__ when;
    const uint32_t* __restrict__ mark_cur;   // marks for THIS sweep (written by sweep t - 1)
    uint32_t* __restrict__ mark_next;        // marks for sweep t + 1
    uint32_t* __restrict__ changed;   // one word per sweep of the batch
    float k;
    uint32_t t;
    float range_factor, sp_rest_density;   // see OpLevelCone
    // Values at or below -maximum_surface_distance never leave the step: the smoothing clamps them (and the unassigned) to that
    // bound before anything reads the field (simulation.rs:826-832).  A sweep whose assignments all lie there ends the loop: a
    // particle still unassigned has no neighbour from an earlier sweep (it would have been assigned then), so everything
    // later derives from this sweep's values and lies deeper still.  `changed` therefore means "assigned something above the bound".
    float useful_above;
    int plain;   // 1: no frontier marks -- every unassigned particle looks at its list in every sweep, like the reference does.
                 // 2: the frontier form on a slab decomposition.  A ghost cannot mark for its owner's rank, so an owned particle
                 //    whose candidacy would come from a ghost (assigned on another rank one sweep earlier) is never marked: every
                 //    unassigned HALO MEMBER (edge[i]: an owned particle the neighbour rank holds as a ghost -- the owned particles
                 //    with a ghost on their extended list are among them) therefore PROBES its list in every sweep.  A probing lane
                 //    marks nothing while it looks (it may find no assigned neighbour, and a mark from a particle that is not
                 //    assigned would start a cascade of idle candidates); if it is assigned, its unassigned neighbours are marked
                 //    by a second pass over its list (OpLevelMark, SECOND_PASS of k_sweep) -- they are the candidates of sweep t + 1,
                 //    exactly those the one-context form marks.
    const uint8_t* __restrict__ edge;   // mode 2
    struct Acc {
        float best, r2max;
        bool have, marker, late;
    };
    static constexpr bool SECOND_PASS = SLAB;
    __device__ float krange() const { return k; }
    __device__ bool skip() const { return false; }
    __device__ bool lane_skip(uint32_t i) const
    {
        if (plain == 1) return t == 0u || when[i] != LVL_UNASSIGNED;
        if (t == 0u) return when[i] != 0u;
        if (when[i] != LVL_UNASSIGNED) return true;
        return !(mark_cur[i] == t || (plain == 2 && edge[i] != 0));
    }
    __device__ void init(Acc&) const {}
    __device__ void epilogue(Acc&, bool, uint32_t) const {}
    __device__ float2 cell_pos(uint32_t i, const float4& Ai) const
    {
        if (!pm_cell) return make_float2(Ai.x, Ai.y);
        const float4 q = pm_cell[i];
        return make_float2(q.x, q.y);
    }
    __device__ float4 loadA(uint32_t j) const { return pm[j]; }
    __device__ NB nb(const Acc&, uint32_t j, float4) const
    {
        // loads only, and both unconditionally: a store to mark[] between the loads of a trip (the compiler cannot rule out
        // that it aliases when[] / level[]) or a level load that waits for `when` turns the four neighbours of a trip into
        // four dependent round trips -- the sweep is nothing but such chains on a few frontier lanes
        return NB{when[j], level[j], j};
    }
    __device__ void begin(Acc& a, uint32_t i, float4 Ai) const
    {
        a.best = 0.f;
        a.have = false;
        a.late = false;
        a.marker = plain == 0 || t == 0u || (plain == 2 && mark_cur[i] == t);   // (a marked lane has an assigned neighbour: it WILL be assigned)
        a.r2max = level_range_sq(Ai.z, range_factor, sp_rest_density);
    }
    __device__ void pair(Acc& a, float4, NB Bj, float, float, float r2, float) const
    {
        // a candidate of sweep t is assigned in sweep t; its unassigned neighbours are the candidates of sweep t+1.  (In the
        // candidate-walk fallback only accepted pairs arrive here, so only real neighbours are marked.)
        if (a.marker && Bj.w == LVL_UNASSIGNED) mark_next[Bj.j] = t + 1u;
        if (!(Bj.w < t)) return;
        if (r2 > a.r2max) return;
        const float est = Bj.lv - sqrtf(r2);
        a.best = a.have ? fmaxf(a.best, est) : est;
        a.have = true;
    }
    __device__ bool finish(Acc& a, uint32_t i, float4, bool) const
    {
        if (t > 0u && a.have) {
            level[i] = a.best;
            when[i] = t;
            if (a.best > useful_above) *changed = 1u;   // same value from every lane
            a.late = !a.marker;   // a probing halo member (mode 2) that found its neighbour: OpLevelMark marks for it
        }
        return false;
    }
    // ---- the second pass of a lane (k_sweep, SECOND_PASS): mark the unassigned neighbours of a probing lane that was assigned ----
    struct Second {
        typedef MathT Math;
        static constexpr bool HAS_EPILOGUE = false, SKIP_SELF = false, EXTENDED = true;
        typedef NBLevel NB;
        MathT m;
        const float4* __restrict__ pm;
        const float4* __restrict__ pm_cell;
        const uint32_t* __restrict__ when;
        uint32_t* __restrict__ mark_next;
        float k;
        uint32_t t;
        struct Acc {};
        __device__ float krange() const { return k; }
        __device__ bool skip() const { return false; }
        __device__ bool lane_skip(uint32_t) const { return false; }
        __device__ void init(Acc&) const {}
        __device__ void epilogue(Acc&, bool, uint32_t) const {}
        __device__ float2 cell_pos(uint32_t i, const float4& Ai) const
        {
            if (!pm_cell) return make_float2(Ai.x, Ai.y);
            const float4 q = pm_cell[i];
            return make_float2(q.x, q.y);
        }
        __device__ float4 loadA(uint32_t j) const { return pm[j]; }
        __device__ NB nb(const Acc&, uint32_t j, float4) const { return NB{when[j], 0.f, j}; }
        __device__ void begin(Acc&, uint32_t, float4) const {}
        __device__ void pair(Acc&, float4, NB Bj, float, float, float, float) const
        {
            // (the lane itself was assigned a moment ago: when[i] = t, so its own entry is not "unassigned" any more)
            if (Bj.w == LVL_UNASSIGNED) mark_next[Bj.j] = t + 1u;
        }
        __device__ bool finish(Acc&, uint32_t, float4, bool) const { return false; }
    };
    __device__ bool second_wanted(const Acc& a) const { return a.late; }
    __device__ Second second() const { return Second{m, pm, pm_cell, when, mark_next, k, t}; }
};

// ---- the propagation on a COMPACTED FRONTIER (one context) ---------------------------------------------
// The frontier form above still launches N / 256 blocks per sweep and reads 8 bytes of (when, mark) per particle to find the few
// thousand candidates of the sweep, and a candidate then walks its list on ONE lane in dependent trips of four: 104 launches x 13 us
// at N = 2^20, running alone.  Here the candidates of sweep t are the set bytes of a MAP with one byte per particle: a particle
// assigned in sweep t stores a 1 into the bytes of its unassigned list neighbours in the map of sweep t + 1 (plain byte stores, no
// atomics: every writer writes the same value), and sweep t + 1 is a launch of N / 8 lanes, each reading eight bytes of the map.
// The map is TRANSPOSED -- particle j = q S + r sits at byte ((r + q P) mod S) 64 + q, S = 2^k >= N / 64, P odd -- so that consecutive particles (the
// frontier is a band: the free surface of a dam break is two or three rows of cells) land eight lanes apart: a wave then holds ~8
// candidates, one per group of G = 8 lanes, and the group works on its candidate together -- lane l takes the index quads l, l + G,
// ... of the explicit list, so a list of up to 32 neighbours is ONE round of gathers (when, level, position), the eight partial
// maxima meet through __shfl_xor.  A lane's chain is three dependent loads (map word; own record + list word + index quad;
// neighbours) instead of eleven, and no value-returning atomic sits in it (on this device an agent-scope atomic is performed at
// the memory side, 2-3 us per round trip: a first form with a queue, atomicMax marks and an atomic tail took 16 us per sweep).
// The values are those of the other forms bit for bit: a particle is assigned once, in the first sweep in which a neighbour inside its
// range had a value, with the exact maximum over the neighbours that had one when the sweep started (fmaxf is order-independent on
// finite values; `when < t` is the sweep's snapshot).  Candidates without an explicit index list (more than NLX_CAP neighbours:
// they walk their candidates) take the generic path through sweep_particle on one lane of their group; sweep 0 (the surface
// particles mark their neighbours) is the generic sweep over all particles with the same op.
struct LevelFrontier {
    uint64_t* __restrict__ cur;    // the map of this sweep (eight particles' bytes per word), cleared by its readers
    uint8_t* __restrict__ next;    // the map of sweep t + 1
    uint32_t lg_s, n_words;        // S = 1 << lg_s; words per map = 8 S
    // particle j = q S + r0 sits at byte r 64 + q with r = (r0 + q P) mod S: the skew by an odd P per q keeps particles that are a
    // multiple of S / 2^k apart (the ends of the cell rows of a 1024-wide lattice: the vertical part of the band) out of each other's group
    static constexpr uint32_t SKEW = 0x9E3779B1u;
    __device__ __forceinline__ uint32_t slot_of(uint32_t j) const
    {
        const uint32_t q = j >> lg_s, m = (1u << lg_s) - 1u;
        return (((j + q * SKEW) & m) << 6) | q;
    }
    __device__ __forceinline__ uint32_t particle_of(uint32_t slot) const
    {
        const uint32_t q = slot & 63u, m = (1u << lg_s) - 1u;
        return (q << lg_s) | (((slot >> 6) - q * SKEW) & m);
    }
    __device__ __forceinline__ void push(uint32_t j) const { next[slot_of(j)] = 1; }
};
template <class MathT>
struct OpLevelPropagateQ {
    typedef MathT Math;
    static constexpr bool HAS_EPILOGUE = false, SKIP_SELF = false, EXTENDED = true;
    typedef NBLevel NB;
    MathT m;
    const float4* __restrict__ pm;
    const float4* __restrict__ pm_cell;
    float* __restrict__ level;
    uint32_t* __restrict__ when;
    LevelFrontier q;
    uint32_t* __restrict__ changed;
    float k;
    uint32_t t;
    float range_factor, sp_rest_density, useful_above;   // see OpLevelPropagateT
    struct Acc {
        float best, r2max;
        bool have;
    };
    __device__ float krange() const { return k; }
    __device__ bool skip() const { return false; }
    __device__ bool lane_skip(uint32_t i) const { return t == 0u ? when[i] != 0u : when[i] != LVL_UNASSIGNED; }
    __device__ void init(Acc&) const {}
    __device__ void epilogue(Acc&, bool, uint32_t) const {}
    __device__ float2 cell_pos(uint32_t i, const float4& Ai) const
    {
        if (!pm_cell) return make_float2(Ai.x, Ai.y);
        const float4 c = pm_cell[i];
        return make_float2(c.x, c.y);
    }
    __device__ float4 loadA(uint32_t j) const { return pm[j]; }
    __device__ NB nb(const Acc&, uint32_t j, float4) const { return NB{when[j], level[j], j}; }
    __device__ void begin(Acc& a, uint32_t, float4 Ai) const
    {
        a.best = 0.f;
        a.have = false;
        a.r2max = level_range_sq(Ai.z, range_factor, sp_rest_density);
    }
    __device__ void pair(Acc& a, float4, NB Bj, float, float, float r2, float) const
    {
        if (Bj.w == LVL_UNASSIGNED) q.push(Bj.j);
        if (!(Bj.w < t)) return;
        if (r2 > a.r2max) return;
        const float est = Bj.lv - sqrtf(r2);
        a.best = a.have ? fmaxf(a.best, est) : est;
        a.have = true;
    }
    __device__ bool finish(Acc& a, uint32_t i, float4, bool) const
    {
        if (t > 0u && a.have) {
            level[i] = a.best;
            when[i] = t;
            if (a.best > useful_above) *changed = 1u;
        }
        return false;
    }
};

#ifdef LEVEL_FRONTIER_STATS
__device__ unsigned long long g_lf_stats[8];   // launches with work, waves with work, rounds, candidates, pushes, second trips
extern "C" void sph_debug_frontier_stats(unsigned long long* out)
{
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lf_stats), sizeof(g_lf_stats));
    unsigned long long z[8] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_lf_stats), z, sizeof(z));
}
#define LF_STAT(K, V) atomicAdd(&g_lf_stats[K], (unsigned long long)(V))
#else
#define LF_STAT(K, V)
#endif
template <class Op>
__device__ __forceinline__ void level_frontier_body(const Op& op, const SweepCommon& c)
{
    constexpr int G = 8;   // lanes per candidate == map bytes per lane == the transposition's lane stride
    const uint32_t tid = blockIdx.x * 256u + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u, gl = lane & (uint32_t)(G - 1), grp = lane & ~(uint32_t)(G - 1);
    uint64_t w = tid < op.q.n_words ? op.q.cur[tid] : 0ull;
    if (w) op.q.cur[tid] = 0ull;   // (this map is the "next" of sweep t + 1)
    const uint32_t t = op.t;
    const float NEG_INF = __uint_as_float(0xff800000u);
#ifdef LEVEL_FRONTIER_STATS
    const bool wave_has_work = __any(w != 0ull);
    if (lane == 0 && wave_has_work) LF_STAT(1, 1);
#endif
    while (__any(w != 0ull)) {
        if (lane == 0) LF_STAT(2, 1);
        // every group of G lanes takes ONE candidate per round: the lowest set byte of its lowest lane that has one
        const uint64_t have = __ballot(w != 0ull);
        const uint32_t gm = (uint32_t)(have >> grp) & 0xffu;
        const uint32_t src = grp + (gm ? (uint32_t)__ffs(gm) - 1u : 0u);
        const uint32_t lb = w ? ((uint32_t)__ffsll((unsigned long long)w) - 1u) >> 3 : 0u;
        const uint32_t mine = op.q.particle_of(tid * 8u + lb);
        if (lane == src && w) w &= ~(0xffull << (lb * 8u));
        const uint32_t i = __shfl(mine, src, 64);
        const bool in = gm != 0u && i < c.n;
        // own record, list word and this lane's first index quad in one round (the quad is read before the list's length is known:
        // the index-list array holds NLX_GROUPS quads for every particle)
        uint32_t wi = 0u;
        uint4 lw = make_uint4(0, 0, 0, 0), quad = make_uint4(0, 0, 0, 0);
        float4 Ai = make_float4(0.f, 0.f, 0.f, 0.f);
        if (in) {
            wi = op.when[i];
            lw = c.nl[i];
            Ai = op.pm[i];
            quad = c.nlx[(size_t)gl * c.n + i];
        }
        // (a particle may be a candidate of two consecutive sweeps: a neighbour saw it unassigned while its own sweep assigned it)
        const bool cand = in && wi == LVL_UNASSIGNED;
        if (gl == 0 && in) LF_STAT(6, 1);
        if (gl == 0 && cand) LF_STAT(3, 1);
        const bool fast = cand && (lw.w & NL_IDX) != 0u;
        const uint32_t ni = fast ? (lw.w & 0xffffu) : 0u;
        const float r2max = level_range_sq(Ai.z, op.range_factor, op.sp_rest_density);
        float best = NEG_INF;
        for (uint32_t g0 = 0; __any(g0 * 4u < ni); g0 += (uint32_t)G) {
            const uint32_t g = g0 + gl;
            const bool gv = g * 4u < ni;
            if (g0 && gv) quad = c.nlx[(size_t)g * c.n + i];
            if (g0 && gv) LF_STAT(5, 1);
            const uint32_t nv = gv ? min(ni - g * 4u, 4u) : 0u;   // valid entries of the quad
            const uint32_t j0 = nv ? quad.x : i;
            const uint32_t jj[4] = {j0, nv > 1u ? quad.y : j0, nv > 2u ? quad.z : j0, nv > 3u ? quad.w : j0};
            uint32_t wj[4];
            float lv[4];
            float4 A[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                wj[k] = op.when[jj[k]];
                lv[k] = op.level[jj[k]];
                A[k] = op.pm[jj[k]];
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if ((uint32_t)k >= nv) continue;
                // the unassigned neighbours are the candidates of sweep t + 1
                if (wj[k] == LVL_UNASSIGNED) {
                    op.q.push(jj[k]);
                    LF_STAT(4, 1);
                }
                const float dx = Ai.x - A[k].x, dy = Ai.y - A[k].y;
                const float r2 = dx * dx + dy * dy;
                if (wj[k] < t && !(r2 > r2max)) best = fmaxf(best, lv[k] - sqrtf(r2));
            }
        }
#pragma unroll
        for (int d = G / 2; d >= 1; d >>= 1) best = fmaxf(best, __shfl_xor(best, d, 64));
        if (fast && gl == 0u && best > NEG_INF) {
            op.level[i] = best;
            op.when[i] = t;
            if (best > op.useful_above) *op.changed = 1u;
        }
        if (cand && !fast && gl == 0u) {   // no explicit index list: the generic path on one lane
            typename Op::Acc acc;
            op.init(acc);
            sweep_particle<Op, false>(op, c, acc, i, Ai, lw);
        }
    }
}

template <class Op>
__global__ __launch_bounds__(256) void k_level_frontier(Op op, SweepCommon c)
{
    level_frontier_body(op, c);
}

// (Measured negatives, round 3 -- profiles/r3_variants.md: (1) the whole propagation as ONE persistent launch, 256 resident
//  workgroups, a progress word per workgroup as grid barrier, every shared word moved with agent-scope (sc1) loads and stores:
//  67 us per sweep, 7.8 ms per step against 2.2 -- a frontier lane's chain is five dependent round trips, and an sc1 round trip
//  beside the step's streaming sweeps is 1-3 us where a cached load behind a kernel boundary is 0.3-0.5; (2) a sweep kernel of
//  its own with per-tile frontier marks and the whole list of a candidate gathered in batches of 4 / 8 / 16 / 32: 28-38 us per
//  sweep against 14 through this skeleton, whatever the batch, the register count, the stream priority or the tile ownership.)

// fill_stash_with (simulation.rs:886-893, 769-779): the level field as it stands, interior -> -maximum_surface_distance
__global__ __launch_bounds__(256) void k_fill_stash(uint32_t n, const float* __restrict__ level, float* __restrict__ stash, float max_surface_distance)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float v = level[i];
    stash[i] = isnan(v) ? -max_surface_distance : v;
}

// Op: smooth_level_estimation_field (simulation.rs:803-857): Shepard-normalised SPH average of the clamped field over
// the k = 2 lists of the step, evaluated at the ADVECTED positions (the lists are those of the start of the step).
struct NBSmooth {
    float x, y, mr, dist;
    float m, rho;   // EXACT policy only: the reference's `dist * mass[j] / density[j] * w_ij` is ((dist m) / rho) w, not dist (m / rho) w
};
// EXT (level_estimation_after_advection with the extended range, simulation.rs:2678-2722): `self.neighs` then holds the
// extended lists of the ADVECTED positions -- pm = pm_new = advected, pm_cell = pre-step (cells), lists = nl_ext / nlx_ext.
template <class MathT, bool EXT>
struct OpLevelSmoothT {
    typedef MathT Math;
    static constexpr bool HAS_EPILOGUE = false, SKIP_SELF = false, EXTENDED = EXT;
    typedef NBSmooth NB;
    MathT m;
    const float4* __restrict__ pm;       // positions the lists were built from (walk predicate)
    const float4* __restrict__ pm_cell;  // positions the particles are sorted by when they differ from pm, else nullptr
    float k_ext;
    const float4* __restrict__ pm_new;   // advected positions
    const uint32_t* __restrict__ orig;
    const float* __restrict__ mrho;
    const float* __restrict__ level_in;
    float* __restrict__ level_out;
    float* __restrict__ level_old;
    DeviceStatus* status;
    float max_surface_distance;
    const float* __restrict__ rho;       // (EXACT policy: see NBSmooth)
    struct Acc {
        float x, y, level, weight;
    };
    __device__ float krange() const { return EXT ? k_ext : 2.f; }
    __device__ float2 cell_pos(uint32_t i, const float4& Ai) const
    {
        if (!pm_cell) return make_float2(Ai.x, Ai.y);
        const float4 q = pm_cell[i];
        return make_float2(q.x, q.y);
    }
    __device__ bool skip() const { return false; }
    __device__ bool lane_skip(uint32_t) const { return false; }
    __device__ void init(Acc&) const {}
    __device__ void epilogue(Acc&, bool, uint32_t) const {}
    __device__ float4 loadA(uint32_t j) const { return pm[j]; }
    __device__ NB nb(const Acc&, uint32_t j, float4) const
    {
        const float4 q = pm_new[j];
        const float lj = level_in[j];
        const float dist = isnan(lj) ? -max_surface_distance : fmaxf(lj, -max_surface_distance);
        if constexpr (MathT::EXACT) return NB{q.x, q.y, mrho[j], dist, pm[j].z, rho[j]};
        return NB{q.x, q.y, mrho[j], dist, 0.f, 0.f};
    }
    __device__ void begin(Acc& a, uint32_t i, float4) const
    {
        const float4 q = pm_new[i];
        a.x = q.x;
        a.y = q.y;
        a.level = a.weight = 0.f;
    }
    __device__ void pair(Acc& a, float4, NB Bj, float, float, float, float hij) const
    {
        const float dx = a.x - Bj.x, dy = a.y - Bj.y;
        const float w = m.w(dx * dx + dy * dy, hij);
        if constexpr (MathT::EXACT) {
            a.level += Bj.dist * Bj.m / Bj.rho * w;
            a.weight += Bj.m / Bj.rho * w;
        } else {
            a.level += Bj.dist * Bj.mr * w;
            a.weight += Bj.mr * w;
        }
    }
    __device__ bool finish(Acc& a, uint32_t i, float4, bool) const
    {
        if (!isfinite(a.weight) || a.weight <= 0.f) {
            raise_error(status, SPH_ERR_LEVEL_WEIGHT, orig[i]);
            return false;
        }
        const float v = a.level / a.weight;
        if (!isfinite(v)) raise_error(status, SPH_ERR_LEVEL_WEIGHT, orig[i]);
        level_old[i] = v;
        level_out[i] = v;
        return false;
    }
};

// LevelEstimationState::target_mass + classify_particle (simulation.rs:213-237, adaptivity/mod.rs:32-59)
__device__ __forceinline__ float level_target_mass(float lv, float max_surface_distance, float rest_density, int sizing_function, float radius_fine,
                                                   float radius_base)
{
    const float lvl = fmaxf(lv, -max_surface_distance);
    const float interp = lvl / -max_surface_distance;
    // DimensionUtils2d::radius_to_sphere_volume (sph_kernels.rs:208-211): PI * r^2
    const float mass_fine = (SPH_PI_F * radius_fine * radius_fine) * rest_density;
    const float mass_base = (SPH_PI_F * radius_base * radius_base) * rest_density;
    if (sizing_function == SPH_SIZING_MASS) return mass_fine * (1.f - interp) + mass_base * interp;
    if (sizing_function == SPH_SIZING_RADIUS) {
        const float r = radius_fine * (1.f - interp) + radius_base * interp;
        return (SPH_PI_F * r * r) * rest_density;
    }
    const float e = 1.f / 2.f;
    const float r = radius_fine * (1.f - powf(interp, e)) + radius_base * powf(interp, e);
    return (SPH_PI_F * r * r) * rest_density;
}

__global__ __launch_bounds__(256) void k_classify(uint32_t n, const float4* __restrict__ pm, const float* __restrict__ level,
                                                   uint8_t* __restrict__ size_class, const uint8_t* __restrict__ owned,
                                                   const uint32_t* __restrict__ orig, DeviceStatus* status, float max_surface_distance,
                                                   float rest_density, int sizing_function, float radius_fine, float radius_base)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n || (owned && !owned[i])) return;
    const float lv = level[i];
    if (isnan(lv)) {   // LevelEstimationState::level() of FluidInterior: unreachable!()
        raise_error(status, SPH_ERR_INVALID_ARGUMENT, orig[i]);
        return;
    }
    const float target = level_target_mass(lv, max_surface_distance, rest_density, sizing_function, radius_fine, radius_base);
    const float mrel = pm[i].z / target;
    uint8_t cls;
    if (mrel <= 0.5f) cls = 0;
    else if (mrel <= 1.f / 1.1f) cls = 1;
    else if (mrel < 1.1f) cls = 2;
    else if (mrel < 2.0f) cls = 3;
    else cls = 4;
    size_class[i] = cls;
}

// ------------------------------------------------------------------------------------------------
// stop decision of a slab decomposition: the ranks' totals of iteration `iter` (solver_reduce_decide, multi) were all-reduced
// (RCCL, in stream); every rank takes the same decision here (stopping rule of iisph_pressure_iterations, simulation.rs:1453-1479)
// ------------------------------------------------------------------------------------------------
__global__ void k_solver_decide(const double* __restrict__ tot, SolverCtrl* ctrl, int iter, SolveP q, float rest_density, float dt, const uint32_t* __restrict__ gate)
{
    if (gate && *gate == 0u) return;   // chained solves: the solve before this one has not ended
    if (threadIdx.x == 0) solver_decide_multi(tot, ctrl, iter, q, rest_density, dt, true);
}
// an empty slab's control block at the start of a (possibly gated) solve
__global__ void k_ctrl_reset(SolverCtrl* ctrl, const uint32_t* __restrict__ gate)
{
    if (gate && *gate == 0u) return;
    if (threadIdx.x == 0 && blockIdx.x == 0) *ctrl = SolverCtrl{};
}
// slab decomposition: the decision on iteration `iter` (and a paced solve's progress word) by a launch of its own, right behind the
// all-reduce of its totals -- where no sweep A's block 0 can take it: a split sweep A (its interior runs BESIDE the all-reduce, its
// edge launch only after the interior), an empty slab (no sweep at all)
__global__ void k_solver_progress(const double* __restrict__ tot, SolverCtrl* ctrl, int iter, SolveP q, float rest_density, float dt, const uint32_t* __restrict__ gate)
{
    if (gate && *gate == 0u) return;
    if (threadIdx.x == 0) solver_decide_multi(tot, ctrl, iter, q, rest_density, dt, true);
}
void launch_solver_progress(hipStream_t s, Profiler* prof, const SweepArgs& a, int iter, int residual_density, float max_avg_error, uint32_t max_iters)
{
    ProfScope ps(prof, "solver_progress", s);
    hipLaunchKernelGGL(k_solver_progress, dim3(1), dim3(64), 0, s, a.solver_tot, a.ctrl, iter, solve_params(a, residual_density, max_avg_error, max_iters, 2, true),
                       a.sp.rest_density, a.sp.dt, a.gate);
}
void launch_ctrl_reset(hipStream_t s, SolverCtrl* ctrl, const uint32_t* gate) { hipLaunchKernelGGL(k_ctrl_reset, dim3(1), dim3(64), 0, s, ctrl, gate); }

// ------------------------------------------------------------------------------------------------
// per-particle maps
// ------------------------------------------------------------------------------------------------
// IISPH2 after the solve: pressure /= sqrt(omega) (simulation.rs:2358-2360), and the p / rho^2 payload of the
// pressure-acceleration sweep that follows
__global__ __launch_bounds__(256) void k_iisph2_scale(uint32_t n, const SolverCtrl* __restrict__ ctrl, float* __restrict__ p0, float* __restrict__ p1,
                                                       float* __restrict__ pt0, float* __restrict__ pt1, const float* __restrict__ omega,
                                                       const float* __restrict__ rho)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float* p = ctrl->cur ? p1 : p0;
    float* pt = ctrl->cur ? pt1 : pt0;
    const float v = p[i] / sqrtf(omega[i]);
    const float r = rho[i];
    p[i] = v;
    pt[i] = v / (r * r);
}
void launch_iisph2_scale(hipStream_t s, Profiler* prof, const SweepArgs& a)
{
    ProfScope ps(prof, "iisph2_scale", s);
    if (a.n) hipLaunchKernelGGL(k_iisph2_scale, dim3((a.n + 255) / 256), dim3(256), 0, s, a.n, a.ctrl, a.p0, a.p1, a.pt0, a.pt1, a.omega, a.rho);
}

// ------------------------------------------------------------------------------------------------
// constrain_neighborhood_count (simulation.rs:2145-2177): a particle with more than `target` neighbours takes the
// (count - target)-th largest "fringe" value 2 |x_ij| - 2 h_j as its smoothing length.  Selection without a per-lane
// array: every pass replays the list and finds the largest fringe value below the lane's threshold and how often it occurs;
// the rank reached so far is carried in `consumed`.  Passes are launched until no lane is pending (1-3 in practice, the
// wanted rank is count - 19).
// ------------------------------------------------------------------------------------------------
#define CON_DONE 0xffffffffu
__global__ __launch_bounds__(256) void k_constrain_init(uint32_t n, uint32_t target, const float4* __restrict__ pm, const uint32_t* __restrict__ ncount,
                                                         float* __restrict__ thr, uint32_t* __restrict__ consumed, float* __restrict__ h_new,
                                                         uint8_t* __restrict__ flag, const uint8_t* __restrict__ owned)
{
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const bool over = ncount[i] > target && (!owned || owned[i]);
    thr[i] = INFINITY;
    consumed[i] = over ? 0u : CON_DONE;
    h_new[i] = pm[i].w;
    flag[i] = over ? 1 : 0;
}

template <class MathT>
struct OpConstrain {
    typedef MathT Math;
    static constexpr bool HAS_EPILOGUE = false, SKIP_SELF = false, EXTENDED = false;
    __device__ constexpr float krange() const { return 2.f; }
    typedef NBNone NB;
    MathT m;
    const float4* __restrict__ pm;
    const uint32_t* __restrict__ orig;
    const uint32_t* __restrict__ ncount;
    float* __restrict__ thr;
    uint32_t* __restrict__ consumed;
    float* __restrict__ h_new;
    uint32_t* __restrict__ pending;
    DeviceStatus* status;
    uint32_t target;
    struct Acc {
        float thr, best;
        uint32_t cnt;
    };
    __device__ bool skip() const { return false; }
    __device__ bool lane_skip(uint32_t i) const { return consumed[i] == CON_DONE; }
    __device__ void init(Acc&) const {}
    __device__ void epilogue(Acc&, bool, uint32_t) const {}
    __device__ float4 loadA(uint32_t j) const { return pm[j]; }
    __device__ NB nb(const Acc&, uint32_t, float4) const { return NB{}; }
    __device__ void begin(Acc& a, uint32_t i, float4) const
    {
        a.thr = thr[i];
        a.best = -INFINITY;
        a.cnt = 0;
    }
    __device__ void pair(Acc& a, float4 Aj, NB, float, float, float r2, float) const
    {
        const float f = 2.f * sqrtf(r2) - Aj.w * 2.f;   // 2 |x_ij| - support_radius_single(j), simulation.rs:2155-2158
        if (f < a.thr) {
            if (f > a.best) {
                a.best = f;
                a.cnt = 1;
            } else if (f == a.best) {
                a.cnt++;
            }
        }
    }
    __device__ bool finish(Acc& a, uint32_t i, float4 Ai, bool) const
    {
        const uint32_t k = ncount[i] - target;   // index into the descending order (simulation.rs:2161)
        const uint32_t have = consumed[i];
        if (a.cnt == 0u || have + a.cnt > k) {
            const float hn = a.best;
            h_new[i] = hn;
            consumed[i] = CON_DONE;
            if (!(hn < Ai.w)) raise_error(status, SPH_ERR_CONSTRAIN_NOT_SMALLER, orig[i]);
            else if (!(hn >= 0.f)) raise_error(status, SPH_ERR_CONSTRAIN_NEGATIVE, orig[i]);
        } else {
            consumed[i] = have + a.cnt;
            thr[i] = a.best;
            *pending = 1u;
        }
        return false;
    }
};

// h2_next <- h2, h2 <- constrained value: the mem::swap of simulation.rs:2172 on the (x, y, m, h) records
__global__ __launch_bounds__(256) void k_constrain_apply(uint32_t n, float4* __restrict__ pm, const float* __restrict__ h_new, float* __restrict__ h2_next)
{
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float4 a = pm[i];
    h2_next[i] = a.w;
    a.w = h_new[i];
    pm[i] = a;
}

// ------------------------------------------------------------------------------------------------
// Two ops over one replay of the list: same neighbours, same geometry, one gather of the (x, y, m, h) records and one kernel
// gradient per pair instead of two (the compiler merges the identical m.grad calls).  Used for a_ii + constant_field together
// with the non-pressure acceleration when the latter directly follows (every solver mode except HybridDFSPH with the
// non-pressure forces behind the divergence solve): 67 -> ~47 us per step at N = 1M.
// ------------------------------------------------------------------------------------------------
template <class A, class B>
struct OpFuse {
    static constexpr bool TILE = true;
    typedef typename A::Math Math;
    static constexpr bool HAS_EPILOGUE = false, SKIP_SELF = A::SKIP_SELF && B::SKIP_SELF, EXTENDED = false;
    // (k_sweep_off: A knows about its own term -- OpAiiConst --, B's pair of the particle with itself is zero)
    static constexpr bool OFF16 = OpOff16<A>::value && OpOff16<B>::value && OpOffSelf<A>::value && B::SKIP_SELF, OFF16_SELF = true;
    // (OFF_WAVES: not here -- 79 VGPRs = 6 waves; bounded to 7 it spills 10 registers and runs 35.3 instead of 35.7 us (A/B on one box, four
    //  runs each: inside the noise), to 8 it spills 165 and runs twice as long)
    __device__ constexpr float krange() const { return 2.f; }
    A a;
    B b;
    Math m;
    struct NB {
        typename A::NB x;
        typename B::NB y;
    };
    struct Acc {
        typename A::Acc x;
        typename B::Acc y;
    };
    __device__ bool skip() const { return false; }
    __device__ bool lane_skip(uint32_t) const { return false; }
    __device__ void init(Acc& c) const
    {
        a.init(c.x);
        b.init(c.y);
    }
    __device__ void epilogue(Acc&, bool, uint32_t) const {}
    __device__ float4 loadA(uint32_t j) const { return a.loadA(j); }
    __device__ NB nb(const Acc& c, uint32_t j, float4 Aj) const { return NB{a.nb(c.x, j, Aj), b.nb(c.y, j, Aj)}; }
    __device__ void begin(Acc& c, uint32_t i, float4 Ai) const
    {
        a.begin(c.x, i, Ai);
        b.begin(c.y, i, Ai);
    }
    __device__ void pair(Acc& c, float4 Aj, NB n, float dx, float dy, float r2, float hij) const
    {
        a.pair(c.x, Aj, n.x, dx, dy, r2, hij);
        b.pair(c.y, Aj, n.y, dx, dy, r2, hij);
    }
    __device__ void begin_off(Acc& c, uint32_t i, float4 Ai) const { a.begin_off(c.x, i, Ai); }
    __device__ void pair_off(Acc& c, float4 Aj, NB n, float dx, float dy, float r2, float hij, int off) const
    {
        a.pair_off(c.x, Aj, n.x, dx, dy, r2, hij, off);
        b.pair(c.y, Aj, n.y, dx, dy, r2, hij);
    }
    __device__ bool finish(Acc& c, uint32_t i, float4 Ai, bool wall) const
    {
        const bool w = a.finish(c.x, i, Ai, wall);
        b.finish(c.y, i, Ai, wall);
        return w;
    }
};

// HybridDFSPH after the divergence solve: v += dt * a^p   (simulation.rs:2547-2560)
__global__ __launch_bounds__(256) void k_vel_add_pacc(uint32_t n, float dt, float2* __restrict__ vel, const float4* __restrict__ pacc,
                                                       const uint32_t* __restrict__ orig, DeviceStatus* status)
{
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float2 v = vel[i];
    const float4 rec = pacc[i];
    const float2 a = make_float2(rec.z, rec.w);
    v.x += dt * a.x;
    v.y += dt * a.y;
    vel[i] = v;
    if (!isfinite(v.x) || !isfinite(v.y)) raise_error(status, SPH_ERR_VELOCITY_NOT_FINITE, orig[i]);
}

// mode 0: v += dt a^p ; x += dt v                     (IISPH / OnlyDivergence, simulation.rs:2433-2445, 2486-2499)
// mode 1: x += dt v + dt^2 a^p ; v += dt a^p * min(dt*factor, 1)   (HybridDFSPH, simulation.rs:2644-2646)
__global__ __launch_bounds__(256) void k_integrate(uint32_t n, float dt, float vfactor, int mode, const float4* __restrict__ pm,
                                                    float4* __restrict__ pm_out, float2* __restrict__ vel, const float4* __restrict__ pacc,
                                                    const uint32_t* __restrict__ orig, DeviceStatus* status)
{
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float4 p = pm[i];
    float2 v = vel[i];
    const float4 rec = pacc[i];
    const float2 a = make_float2(rec.z, rec.w);
    if (mode == 0) {
        v.x += dt * a.x;
        v.y += dt * a.y;
        p.x += dt * v.x;
        p.y += dt * v.y;
        if (!isfinite(v.x) || !isfinite(v.y)) raise_error(status, SPH_ERR_VELOCITY_NOT_FINITE, orig[i]);
    } else {
        p.x += dt * v.x + dt * dt * a.x;
        p.y += dt * v.y + dt * dt * a.y;
        v.x += dt * a.x * vfactor;
        v.y += dt * a.y * vfactor;
        if (!isfinite(p.x) || !isfinite(p.y)) raise_error(status, SPH_ERR_POSITION_NOT_FINITE, orig[i]);
    }
    pm_out[i] = p;
    vel[i] = v;
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
static SweepCommon common_of(const SweepArgs& a, bool ext)
{
    return SweepCommon{a.g, ext ? a.t_ext : a.t, a.n, (a.n + SWEEP_THREADS - 1) / SWEEP_THREADS, a.cell_start, ext ? a.nl_ext : a.nl, ext ? a.nlx_ext : a.nlx, a.owned, a.ring1,
                       0, nullptr, nullptr, nullptr, 0u, 0u, ext ? nullptr : a.nloff, ext ? nullptr : a.nlh, ext ? nullptr : a.nloff_out, ext ? nullptr : a.nlh_out, nullptr};
}

// SPH_TILE: bit 0 = the BUILD sweep (density), bit 1 = the replay sweeps through the LDS-staged form (k_sweep_tile) in
// uniform-h scenes.  Measured on MI355X: profiles/r2_variants.md.
static int g_tile_mode = -1;   // sph_set_sweep_variant (process-wide override of the contexts' SPH_TILE option)
#ifdef SPH_LAB
static int tile_mode(const SweepArgs& a) { return g_tile_mode >= 0 ? g_tile_mode : (a.opt_tile ? a.opt_tile : SPH_TILE_DEFAULT); }
#else
static int tile_mode(const SweepArgs&) { return 0; }   // (the product ships the gather forms only)
#endif
// Options::jacobi_generic: the Jacobi update of uniform scenes through OpJacobi as well (measurement: what OpJacobiU is worth)
// SPH_ACCEL_GENERIC (read once per step by the step driver, which then passes no record buffers: the tests compare both forms in one
// process): sweep A of such scenes through OpPressureAccel, the solves' p / rho^2 as a field of its own
static bool jacobi_on_records(const SweepArgs& a)   // sweep B gathers {x, y, a^p} whole (OpJacobiU)
{
    return !a.exact && a.uniform_h && a.h_mode == SPH_H_FROM_MASS && a.sp.opdisc != SPH_OP_WINCHENBACH2020 && !a.opt_jacobi_generic &&
           a.n <= (1u << 28);   // (the record sweeps address their records with 32-bit byte offsets: load_record)
}
static bool solve_on_records(const SweepArgs& a)    // ... and sweep A {x, y, p / rho^2, p} (OpPressureAccelU): not IISPH2
{
    return jacobi_on_records(a) && a.rec0 != nullptr;   // (Options::accel_generic / ::slab_records: the step driver leaves rec0 null)
}
bool sweep_a_on_records(const SweepArgs& a) { return solve_on_records(a); }
// ... and the source-term sweep {x, y, v} (OpSourceU): one context, the record kept current by the writers of the velocities
bool source_term_on_records(const SweepArgs& a) { return solve_on_records(a) && a.xv != nullptr; }
extern "C" int sph_set_sweep_variant(int mode)
{
    if (mode < 0 || mode > 3) return SPH_ERR_INVALID_ARGUMENT;
#ifndef SPH_LAB
    if (mode != 0) return SPH_ERR_UNSUPPORTED;   // the LDS-staged forms are laboratory forms (libsph_lab.so, -DSPH_LAB)
#endif
    g_tile_mode = mode;
    return SPH_OK;
}

// one sweep kernel; under Profiler mode 3 it stamps the device clock into the launch's slot (sweep_stamp)
template <class K, class Op>
static void launch_sweep_kernel(K kernel, dim3 grid, hipStream_t s, const SweepArgs& a, const Op& op, SweepCommon c)
{
    c.ts = a.prof ? a.prof->take_slot() : nullptr;
    hipEvent_t e0, e1;
    if (a.prof && a.prof->take_events(&e0, &e1)) {   // Profiler mode 4: the dispatch's own start / end timestamps
        hipExtLaunchKernelGGL(kernel, grid, dim3(SWEEP_THREADS), 0, s, e0, e1, 0, op, c);
        return;
    }
    hipLaunchKernelGGL(kernel, grid, dim3(SWEEP_THREADS), 0, s, op, c);
}

template <class Op, bool BUILD>
static void launch_sweep(hipStream_t s, const SweepArgs& a, const Op& op)
{
    if (a.n == 0) return;
    if (a.part) {   // one of the two launches of a split sweep (slab decomposition, sph_step.hip)
        SweepCommon c = common_of(a, Op::EXTENDED);
        c.part = a.part;
        c.edge = a.edge;
        c.elist_a = a.elist_a;
        c.elist_b = a.elist_b;
        c.n_ea = a.n_ea;
        c.n_eb = a.n_eb;
        if (a.part == 2) c.nblocks = (a.n_ea + a.n_eb + SWEEP_THREADS - 1) / SWEEP_THREADS;
        if (c.nblocks == 0) c.nblocks = 1;   // (the launch's prologue still runs)
        if constexpr (!BUILD && OpOff16<Op>::value) {
            if (a.nloff) {
                launch_sweep_kernel(k_sweep_off<Op>, dim3(((c.nblocks + 7) / 8) * 8), s, a, op, c);
                return;
            }
        }
        launch_sweep_kernel(k_sweep<Op, BUILD>, dim3(((c.nblocks + 7) / 8) * 8), s, a, op, c);
        return;
    }
    const uint32_t nblocks = (a.n + SWEEP_THREADS - 1) / SWEEP_THREADS;
    const uint32_t grid = ((nblocks + 7) / 8) * 8;  // XCD remap needs a multiple of 8
    if constexpr (!BUILD && OpOff16<Op>::value) {
        if (a.nloff) {
            launch_sweep_kernel(k_sweep_off<Op>, dim3(grid), s, a, op, common_of(a, false));
            return;
        }
    }
#ifdef SPH_LAB
    if constexpr (Op::Math::UNIFORM && !Op::EXTENDED && OpTile<Op>::value) {
        if (tile_mode(a) & (BUILD ? 1 : 2)) {
            launch_sweep_kernel(k_sweep_tile<Op, BUILD>, dim3(grid), s, a, op, common_of(a, Op::EXTENDED));
            return;
        }
    }
#endif
    launch_sweep_kernel(k_sweep<Op, BUILD>, dim3(grid), s, a, op, common_of(a, Op::EXTENDED));
}

size_t sweep_list_bytes(uint32_t n) { return (size_t)n * sizeof(uint4); }
size_t sweep_index_list_bytes(uint32_t n) { return (size_t)n * sizeof(uint4) * NLX_GROUPS; }
size_t sweep_offset_list_bytes(uint32_t n) { return (size_t)n * sizeof(uint2) * NLOFF_GROUPS; }
// every scene but the EXACT policy's: the gradient sweeps of the step replay them where a particle has a mask list (3 x 3 stencil)
// (not with the LDS-staged BUILD form or forced index lists, which do not write them)
// Measured (bench.py other_configs, round 4): configs[2] (1M particles, 4:1) 0.634 -> 0.586 ms/step with them, configs[4]'s scene (4M, 50:1)
// 0.914 -> 0.972 WITHOUT them -- at 4M the step's arrays no longer sit in the 256 MB Infinity Cache, its generic sweeps are bound by HBM bytes
// and the list is 9 bytes longer than the mask word -- so multi-resolution scenes take them up to 2 M particles; uniform scenes (one
// gathered record per neighbour) gain at 1M and at 8M alike.
bool sweeps_want_offset_lists(const SweepArgs& a)
{
    return !a.exact && !(tile_mode(a) & 1) && SPH_FORCE_IDX == 0 && (jacobi_on_records(a) || a.n <= (1u << 21)) && a.n <= (1u << 28);   // (2^28: load_record's 32-bit byte offsets)
}
bool sweep_forces_index_lists() { return SPH_FORCE_IDX != 0; }
uint32_t solver_reduce_blocks(uint32_t n) { return (n + SWEEP_THREADS - 1) / SWEEP_THREADS; }

static MathUniform uniform_math(float h)
{
    MathUniform m;
    m.h = h;
    m.nf = 10.f / (SPH_SEVEN_PI * (h * h));
    m.inv2h = 1.f / (2.f * h);
    m.nf2 = 2.f * m.nf;
    m.nf6 = 6.f * m.nf * m.inv2h;
    return m;
}

static MathExact exact_math(const SweepArgs& a)
{
    MathExact m;
    m.h = 0.f;
    m.wall_pl = a.wall_pl;
    m.wall_cnt = a.wall_cnt;
    m.wall_n = a.n;
    return m;
}

// math mode dispatch: EXACT (diagnostic) / UNIFORM (all h identical) / FAST
#define SPH_DISPATCH(OPNAME, BUILD, ...)                                                   \
    if (a.exact) {                                                                         \
        OPNAME<MathExact> op{exact_math(a), __VA_ARGS__};                                  \
        launch_sweep<OPNAME<MathExact>, BUILD>(s, a, op);                                  \
    } else if (a.uniform_h) {                                                              \
        OPNAME<MathUniform> op{uniform_math(a.h_uniform), __VA_ARGS__};                    \
        launch_sweep<OPNAME<MathUniform>, BUILD>(s, a, op);                                \
    } else {                                                                               \
        OPNAME<MathFast> op{MathFast{0.f}, __VA_ARGS__};                                   \
        launch_sweep<OPNAME<MathFast>, BUILD>(s, a, op);                                   \
    }

template <class M>
using OpDensityMass = OpDensity<M, false>;
template <class M>
using OpDensityDist = OpDensity<M, true>;

void launch_density(hipStream_t s, Profiler* prof, const SweepArgs& a)
{
    ProfScope ps(prof, "density", s, true);
    if (a.h_mode != SPH_H_FROM_MASS) {
        SPH_DISPATCH(OpDensityDist, true, a.pm, a.orig, a.rho, a.mrho, a.lam_sum, a.lam_grad, a.ncount, a.planes, a.lam_lut, a.dlam_lut, a.status, a.sp,
                     a.h_mode, a.h2_next, a.lam_prev, a.owned)
        return;
    }
    SPH_DISPATCH(OpDensityMass, true, a.pm, a.orig, a.rho, a.mrho, a.lam_sum, a.lam_grad, a.ncount, a.planes, a.lam_lut, a.dlam_lut, a.status, a.sp,
                 a.h_mode, a.h2_next, a.lam_prev, a.owned)
}

// the density sweep again, over the recorded lists (after constrain_neighborhood_count changed the smoothing lengths)
void launch_density_replay(hipStream_t s, Profiler* prof, const SweepArgs& a)
{
    ProfScope ps(prof, "density_replay", s, true);
    SPH_DISPATCH(OpDensityMass, false, a.pm, a.orig, a.rho, a.mrho, a.lam_sum, a.lam_grad, a.ncount, a.planes, a.lam_lut, a.dlam_lut, a.status, a.sp,
                 a.h_mode, a.h2_next, a.lam_prev, a.owned)
}

void launch_constrain_init(hipStream_t s, Profiler* prof, const SweepArgs& a, uint32_t target, float* thr, uint32_t* consumed, float* h_new, uint8_t* flag)
{
    ProfScope ps(prof, "constrain_init", s);
    if (a.n) hipLaunchKernelGGL(k_constrain_init, dim3((a.n + 255) / 256), dim3(256), 0, s, a.n, target, a.pm, a.ncount, thr, consumed, h_new, flag, a.owned);
}

void launch_constrain_pass(hipStream_t s, Profiler* prof, const SweepArgs& a, uint32_t target, float* thr, uint32_t* consumed, float* h_new,
                           uint32_t* pending)
{
    ProfScope ps(prof, "constrain_pass", s, true);
    SPH_DISPATCH(OpConstrain, false, a.pm, a.orig, a.ncount, thr, consumed, h_new, pending, a.status, target)
}

void launch_constrain_apply(hipStream_t s, Profiler* prof, const SweepArgs& a, float4* pm, const float* h_new, float* h2_next)
{
    ProfScope ps(prof, "constrain_apply", s);
    if (a.n) hipLaunchKernelGGL(k_constrain_apply, dim3((a.n + 255) / 256), dim3(256), 0, s, a.n, pm, h_new, h2_next);
}

void launch_aii_const(hipStream_t s, Profiler* prof, const SweepArgs& a)
{
    ProfScope ps(prof, "aii_constfield", s, true);
    SPH_DISPATCH(OpAiiConst, false, a.pm, a.orig, a.rho, a.mrho, a.lam_sum, a.lam_grad, a.aii, a.constf, a.status, a.sp,
                 a.sp_check_aii ? reinterpret_cast<float2*>(a.pacc) : nullptr)   // (check_aii borrows the a^p buffer before the solve: a^p for the field e_i)
}

template <class M>
static void launch_fused_aii_np(hipStream_t s, const SweepArgs& a, const M& math)
{
    typedef OpFuse<OpAiiConst<M>, OpNonPressure<M>> Op;
    Op op{OpAiiConst<M>{math, a.pm, a.orig, a.rho, a.mrho, a.lam_sum, a.lam_grad, a.aii, a.constf, a.status, a.sp, nullptr},
          OpNonPressure<M>{math, a.pm, a.orig, a.rho, a.vel, a.vel_tmp, a.status, a.sp, source_term_on_records(a) ? a.xv : nullptr}, math};
    launch_sweep<Op, false>(s, a, op);
}

// constant_field + a_ii and the non-pressure acceleration in one sweep (see OpFuse)
void launch_aii_const_non_pressure(hipStream_t s, Profiler* prof, const SweepArgs& a)
{
    ProfScope ps(prof, "aii_nonpressure", s, true);
    if (a.exact) launch_fused_aii_np(s, a, exact_math(a));
    else if (a.uniform_h) launch_fused_aii_np(s, a, uniform_math(a.h_uniform));
    else launch_fused_aii_np(s, a, MathFast{0.f});
}

void launch_check_aii(hipStream_t s, Profiler* prof, const SweepArgs& a)
{
    ProfScope ps(prof, "check_aii", s, true);
    SPH_DISPATCH(OpCheckAii, false, a.pm, a.orig, a.rho, a.mrho, a.lam_grad, a.aii, reinterpret_cast<const float2*>(a.pacc), a.status, a.sp)
}

void launch_non_pressure(hipStream_t s, Profiler* prof, const SweepArgs& a)
{
    ProfScope ps(prof, "non_pressure_accel", s, true);
    SPH_DISPATCH(OpNonPressure, false, a.pm, a.orig, a.rho, a.vel, a.vel_tmp, a.status, a.sp, source_term_on_records(a) ? a.xv : nullptr)
}

template <class M>
using OpSourcePlain = OpSource<M, false>;
template <class M>
using OpSourceOmega = OpSource<M, true>;

void launch_source_term(hipStream_t s, Profiler* prof, const SweepArgs& a, int kind, int residual_density)
{
    ProfScope ps(prof, "source_term", s, true);
    float4* rec1 = solve_on_records(a) ? a.rec1 : nullptr;
    if (source_term_on_records(a) && a.xv_ok && (kind == 0 || kind == 1)) {
        OpSourceU<MathUniform> op{{uniform_math(a.h_uniform), a.pm, a.orig, a.rho, a.mrho, a.vel, a.lam_grad, a.aii, a.src, a.p1, a.pt1, rec1, a.dens_err,
                                   (SolverPartial*)a.partials, a.ctrl, a.status, a.sp, kind, residual_density, nullptr, nullptr, a.gate},
                                  a.xv};
        launch_sweep<OpSourceU<MathUniform>, false>(s, a, op);
        return;
    }
    if (kind == 3) {
        SPH_DISPATCH(OpSourceOmega, false, a.pm, a.orig, a.rho, a.mrho, a.vel, a.lam_grad, a.aii, a.src, a.p1, a.pt1, rec1, a.dens_err,
                     (SolverPartial*)a.partials, a.ctrl, a.status, a.sp, kind, residual_density, a.omega, a.size_class, a.gate)
        return;
    }
    SPH_DISPATCH(OpSourcePlain, false, a.pm, a.orig, a.rho, a.mrho, a.vel, a.lam_grad, a.aii, a.src, a.p1, a.pt1, rec1, a.dens_err,
                 (SolverPartial*)a.partials, a.ctrl, a.status, a.sp, kind, residual_density, nullptr, nullptr, a.gate)
}

void launch_pressure_accel(hipStream_t s, Profiler* prof, const SweepArgs& a0, int iter, int residual_density, float max_avg_error, uint32_t max_iters, int multi,
                           int part)
{
    ProfScope ps(prof, part == 2 ? "pressure_accel_edge" : "pressure_accel", s, true);
    SweepArgs a = a0;
    a.part = part;
    // (split sweep: neither launch publishes the paced solve's progress -- the interior runs BESIDE the all-reduce of the totals, the
    //  edge launch only after the interior: the step driver queues launch_solver_progress right behind the all-reduce instead)
    const SolveP q = solve_params(a, residual_density, max_avg_error, max_iters, part ? 3 : multi, part == 0);   // (part != 0: launch_solver_progress decides)
    if (solve_on_records(a) && iter >= 1) {   // (iter < 0: IISPH2's extra sweep -- never on records)
        OpPressureAccelU<MathUniform> op{{uniform_math(a.h_uniform), a.pm, a.orig, a.rho, a.p0, a.p1, a.pt0, a.pt1, a.lam_grad, a.pacc, a.ctrl,
                                          (const SolverPartial*)a.partials, solver_reduce_blocks(a.n), a.solver_tot, a.status, a.sp, q, iter, a.owned, a.gate},
                                         (iter & 1) ? a.rec1 : a.rec0};
        launch_sweep<OpPressureAccelU<MathUniform>, false>(s, a, op);
        return;
    }
    SPH_DISPATCH(OpPressureAccel, false, a.pm, a.orig, a.rho, a.p0, a.p1, a.pt0, a.pt1, a.lam_grad, a.pacc, a.ctrl, (const SolverPartial*)a.partials,
                 solver_reduce_blocks(a.n), a.solver_tot, a.status, a.sp, q, iter, a.owned, a.gate)
}

// Split sweep A (slab decomposition with neighbours): this rank's totals of iteration `iter` -- what block 0 of the unsplit sweep
// A(iter + 1) adds up -- by a launch of its own on the stream of the collectives, so that the all-reduce runs under the sweep.
__global__ __launch_bounds__(SWEEP_THREADS) void k_solver_totals(const SolverPartial* __restrict__ partials, uint32_t nparts, SolverCtrl* ctrl,
                                                                 double* __restrict__ tot, int iter, SolveP q, float rest_density, float dt,
                                                                 const DeviceStatus* status, const uint32_t* __restrict__ gate)
{
    if (gate && *gate == 0u) return;
    if (ctrl->slot_done[iter & 1] != 0u) return;   // the solve ended before: the totals are stale, nobody reads them (solver_decide_multi)
    solver_reduce_decide(partials, nparts, ctrl, tot, iter, q, rest_density, dt, status);
}
void launch_solver_totals(hipStream_t s, Profiler* prof, const SweepArgs& a, int iter, int residual_density, float max_avg_error, uint32_t max_iters)
{
    ProfScope ps(prof, "solver_totals", s);
    hipLaunchKernelGGL(k_solver_totals, dim3(1), dim3(SWEEP_THREADS), 0, s, (const SolverPartial*)a.partials, solver_reduce_blocks(a.n), a.ctrl, a.solver_tot, iter,
                       solve_params(a, residual_density, max_avg_error, max_iters, 1, false), a.sp.rest_density, a.sp.dt, a.status, a.gate);
}

// The same totals as block 0 of the launch that packs the halo members' values for the iteration's ghost exchange (slab
// decomposition): both wait for sweep B, both precede the exchange -- one launch instead of two (a kernel boundary is ~4 us, and an
// iteration of 1M particles per rank is ~45 us of sweeps).  Blocks 1.. pack: entries [0, cnt0) go to the left neighbour's staging
// buffer, [cnt0, cnt0 + cnt1) to the right one's (k_pack_field of sph_slabs.hip).
static_assert(RANK_TOTALS_THREADS == 1024, "k_pack_totals' block 0 is rank_totals_block's workgroup");
#define PACK_THREADS 1024   // block 0 adds up one partial per 256 particles: 32768 of them at 8M particles per rank -- on the iteration's critical path
__global__ __launch_bounds__(PACK_THREADS) void k_pack_totals(const SolverPartial* __restrict__ partials, uint32_t nparts, SolverCtrl* ctrl, double* __restrict__ tot,
                                                              int iter, float rest_density, float dt, const DeviceStatus* status,
                                                              const uint32_t* __restrict__ gate, const uint32_t* __restrict__ src_idx, uint32_t cnt0, uint32_t cnt1,
                                                              int words, int stride, int off, const float* __restrict__ field, float* __restrict__ out0,
                                                              float* __restrict__ out1)
{
    if (blockIdx.x == 0) {
        if (gate && *gate == 0u) return;
        if (ctrl->slot_done[iter & 1] != 0u) return;   // (see k_solver_totals)
        rank_totals_block(partials, nparts, tot, &status->error);   // (sph_device.h: the same reduction the push transport's fused launch runs)
        return;
    }
    const uint32_t k = (blockIdx.x - 1u) * PACK_THREADS + threadIdx.x;
    if (k >= cnt0 + cnt1) return;
    const uint32_t i = src_idx[k];
    float* out = k < cnt0 ? out0 + (size_t)k * words : out1 + (size_t)(k - cnt0) * words;
    for (int w = 0; w < words; w++) out[w] = field[(size_t)i * stride + off + w];
}
void launch_pack_and_totals(hipStream_t s, Profiler* prof, const SweepArgs& a, int iter, int residual_density, float max_avg_error, uint32_t max_iters,
                            const uint32_t* src_idx, uint32_t cnt0, uint32_t cnt1, int words, int stride, int off, const float* field, float* out0, float* out1)
{
    (void)residual_density;
    (void)max_avg_error;
    (void)max_iters;
    ProfScope ps(prof, "ghost_pack_totals", s);
    const uint32_t nb = 1u + (cnt0 + cnt1 + PACK_THREADS - 1u) / PACK_THREADS;
    hipLaunchKernelGGL(k_pack_totals, dim3(nb), dim3(PACK_THREADS), 0, s, (const SolverPartial*)a.partials, solver_reduce_blocks(a.n), a.ctrl, a.solver_tot, iter,
                       a.sp.rest_density, a.sp.dt, a.status, a.gate, src_idx, cnt0, cnt1, words, stride, off, field, out0, out1);
}

void launch_solver_tail(hipStream_t s, Profiler* prof, const SweepArgs& a, int tail, float4* pm_out, int decide_iter, int residual_density,
                        float max_avg_error, uint32_t max_iters, SolverCtrl* handoff_host, uint32_t* gate_out)
{
    ProfScope ps(prof, "solver_tail", s);
    if (a.n && tail != TAIL_NONE)
        hipLaunchKernelGGL(k_solver_tail, dim3((a.n + 255) / 256), dim3(256), 0, s, a.n, tail, a.sp.dt, a.sp.hyb_vfactor, a.pm, pm_out, a.vel, a.pacc, a.orig,
                           a.owned, a.ctrl, tail >= TAIL_VX ? a.hdr_partials : nullptr, a.status, a.gate, a.solver_tot, decide_iter,
                           solve_params(a, residual_density, max_avg_error, max_iters, 1, false), a.sp.rest_density, handoff_host, gate_out,
                           tail >= TAIL_VX ? a.inc : IncClassifyP{}, tail == TAIL_VEL && source_term_on_records(a) ? a.xv : nullptr);
}

// Chained solves (HybridDFSPH, one context): the host does not wait between the divergence solve and the density solve.  Behind
// the first solve's tail this kernel hands its control block to the host (mapped memory, read after the step's final wait) and
// opens the gate of the second solve's launches only if the first one ended within its queued iterations.
__global__ void k_solver_handoff(const SolverCtrl* __restrict__ ctrl, SolverCtrl* __restrict__ saved_host, uint32_t* __restrict__ gate)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const SolverCtrl v = *ctrl;
    *saved_host = v;
    __threadfence_system();
    *gate = v.done ? 1u : 0u;
}
void launch_solver_handoff(hipStream_t s, Profiler* prof, SolverCtrl* ctrl, SolverCtrl* saved_host, uint32_t* gate)
{
    ProfScope ps(prof, "solver_handoff", s);
    hipLaunchKernelGGL(k_solver_handoff, dim3(1), dim3(64), 0, s, ctrl, saved_host, gate);
}

void launch_jacobi_update(hipStream_t s, Profiler* prof, const SweepArgs& a, int iter, int residual_density, float max_avg_error, uint32_t max_iters, int multi)
{
    ProfScope ps(prof, "jacobi_update", s, true);
    const float* pin = (iter & 1) ? a.p1 : a.p0;
    float* pout = (iter & 1) ? a.p0 : a.p1;
    float* ptout = (iter & 1) ? a.pt0 : a.pt1;
    const SolveP q = solve_params(a, residual_density, max_avg_error, max_iters, multi, false);
    if (jacobi_on_records(a)) {
        float4* recout = solve_on_records(a) ? ((iter & 1) ? a.rec0 : a.rec1) : nullptr;
        OpJacobiU<MathUniform> op{{uniform_math(a.h_uniform), a.pm, a.orig, a.rho, a.mrho, a.pacc, a.lam_grad, a.aii, a.src, pin, pout, ptout, recout, a.dens_err,
                                   (SolverPartial*)a.partials, a.ctrl, a.status, a.sp, iter, residual_density, q, a.solver_tot, a.gate}};
        launch_sweep<OpJacobiU<MathUniform>, false>(s, a, op);
        return;
    }
    SPH_DISPATCH(OpJacobi, false, a.pm, a.orig, a.rho, a.mrho, a.pacc, a.lam_grad, a.aii, a.src, pin, pout, ptout, (float4*)nullptr, a.dens_err, (SolverPartial*)a.partials, a.ctrl,
                 a.status, a.sp, iter, residual_density, q, a.solver_tot, a.gate)
}

void launch_solver_decide(hipStream_t s, Profiler* prof, const SweepArgs& a, int iter, int residual_density, float max_avg_error,
                          uint32_t max_iters)
{
    ProfScope ps(prof, "solver_decide", s);
    hipLaunchKernelGGL(k_solver_decide, dim3(1), dim3(64), 0, s, a.solver_tot, a.ctrl, iter, solve_params(a, residual_density, max_avg_error, max_iters, 1, false),
                       a.sp.rest_density, a.sp.dt, a.gate);
}

void launch_vel_add_pacc(hipStream_t s, Profiler* prof, const SweepArgs& a)
{
    ProfScope ps(prof, "vel_add_pacc", s);
    hipLaunchKernelGGL(k_vel_add_pacc, dim3((a.n + 255) / 256), dim3(256), 0, s, a.n, a.sp.dt, a.vel, a.pacc, a.orig, a.status);
}

void launch_integrate(hipStream_t s, Profiler* prof, const SweepArgs& a, float4* pm_out, int mode)
{
    ProfScope ps(prof, "integrate", s);
    hipLaunchKernelGGL(k_integrate, dim3((a.n + 255) / 256), dim3(256), 0, s, a.n, a.sp.dt, a.sp.hyb_vfactor, mode, a.pm, pm_out, a.vel, a.pacc,
                       a.orig, a.status);
}

// ---- level estimation (simulation.rs:862-927, 803-857; adaptivity/mod.rs:32-59) ---------------------------------
// a particle whose list word is neither a valid mask list nor an index list walks its candidates with the neighbour predicate
// in every sweep -- which must not happen when the lists are replayed at OTHER positions than they were built from
__global__ __launch_bounds__(256) void k_require_recorded_lists(uint32_t n, const uint4* __restrict__ nl, const uint32_t* __restrict__ orig,
                                                                 const uint8_t* __restrict__ owned, DeviceStatus* status)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n || (owned && !owned[i])) return;
    if (!(nl[i].w & (NL_OK | NL_IDX))) raise_error(status, SPH_ERR_UNSUPPORTED, orig[i]);
}

void launch_level_detect(hipStream_t s, Profiler* prof, const SweepArgs& a, const LevelArgs& l)
{
    if (l.replay_step_lists) {
        // level estimation after advection WITHOUT the extended range: the step's own lists (a.nl_ext / a.nlx_ext point at them),
        // geometry of the advected positions
        if (a.n) hipLaunchKernelGGL(k_require_recorded_lists, dim3((a.n + 255) / 256), dim3(256), 0, s, a.n, a.nl_ext, a.orig, a.owned, a.status);
        if (l.center_diff) {
            ProfScope ps(prof, "level_center_diff", s, true);
            SPH_DISPATCH(OpLevelCenterDiff, false, a.pm, l.pm_cell, l.level, l.when, l.mark, a.n, l.flag_surface, l.stash_first, l.k, l.max_surface_distance,
                         a.sp.rest_density)
            return;
        }
        {
            ProfScope ps(prof, "level_normal", s, true);
            SPH_DISPATCH(OpLevelNormal, false, a.pm, l.pm_cell, l.nrm, l.state, l.flag_insufficient, a.planes, a.sp, l.k, l.boundary_is_fluid_surface)
        }
        {
            ProfScope ps(prof, "level_cone", s, true);
            SPH_DISPATCH(OpLevelCone, false, a.pm, l.pm_cell, l.nrm, l.state, l.level, l.when, l.mark, a.n, l.flag_surface, l.stash_first, l.k, l.threshold, l.max_surface_distance,
                         l.maximum_range, a.sp.rest_density)
        }
        return;
    }
    if (l.center_diff) {
        ProfScope ps(prof, "level_center_diff", s, true);
        SPH_DISPATCH(OpLevelCenterDiff, true, a.pm, l.pm_cell, l.level, l.when, l.mark, a.n, l.flag_surface, l.stash_first, l.k, l.max_surface_distance,
                     a.sp.rest_density)
        return;
    }
    {
        ProfScope ps(prof, "level_normal", s, true);
        SPH_DISPATCH(OpLevelNormal, true, a.pm, l.pm_cell, l.nrm, l.state, l.flag_insufficient, a.planes, a.sp, l.k, l.boundary_is_fluid_surface)
    }
    {
        ProfScope ps(prof, "level_cone", s, true);
        SPH_DISPATCH(OpLevelCone, false, a.pm, l.pm_cell, l.nrm, l.state, l.level, l.when, l.mark, a.n, l.flag_surface, l.stash_first, l.k, l.threshold, l.max_surface_distance,
                     l.maximum_range, a.sp.rest_density)
    }
}

template <class M>
using OpLevelPropagate = OpLevelPropagateT<M, false>;
template <class M>
using OpLevelPropagateSlab = OpLevelPropagateT<M, true>;
template <class M>
static void launch_level_frontier(hipStream_t s, const SweepArgs& a, const LevelArgs& l, const M& math, uint32_t t, uint32_t* changed)
{
    // the map of sweep t at fmap[(t & 1) x 64 S ..]
    const uint32_t S = 1u << l.fmap_lg_s;
    const size_t map_bytes = (size_t)S * 64u;
    LevelFrontier q{(uint64_t*)(l.fmap + ((t & 1u) ? map_bytes : 0u)), l.fmap + ((t & 1u) ? 0u : map_bytes), l.fmap_lg_s, S * 8u};
    OpLevelPropagateQ<M> op{math, a.pm, l.pm_cell, l.level, l.when, q, changed, l.k, t, l.maximum_range, a.sp.rest_density, -l.max_surface_distance};
    if (t == 0u) {   // the surface particles mark their neighbours: the generic sweep over all particles
        (void)hipMemsetAsync(l.fmap, 0, 2 * map_bytes, s);
        launch_sweep<OpLevelPropagateQ<M>, false>(s, a, op);
        return;
    }
    SweepCommon c = common_of(a, true);
    hipLaunchKernelGGL((k_level_frontier<OpLevelPropagateQ<M>>), dim3((S * 8u + 255u) / 256u), dim3(256), 0, s, op, c);
}

void launch_level_propagate(hipStream_t s, Profiler* prof, const SweepArgs& a, const LevelArgs& l, uint32_t t, uint32_t* changed)
{
    ProfScope ps(prof, "level_propagate", s, true);
    if (l.fmap && a.n) {   // one context: the compacted frontier
        if (a.exact) launch_level_frontier(s, a, l, exact_math(a), t, changed);
        else if (a.uniform_h) launch_level_frontier(s, a, l, uniform_math(a.h_uniform), t, changed);
        else launch_level_frontier(s, a, l, MathFast{0.f}, t, changed);
        return;
    }
    const uint32_t* mark_cur = l.mark + ((t & 1u) ? a.n : 0u);
    uint32_t* mark_next = l.mark + ((t & 1u) ? 0u : a.n);
    if (l.plain_propagate == 2) {
        SPH_DISPATCH(OpLevelPropagateSlab, false, a.pm, l.pm_cell, l.level, l.when, mark_cur, mark_next, changed, l.k, t, l.maximum_range, a.sp.rest_density,
                     -l.max_surface_distance, l.plain_propagate, l.edge)
        return;
    }
    SPH_DISPATCH(OpLevelPropagate, false, a.pm, l.pm_cell, l.level, l.when, mark_cur, mark_next, changed, l.k, t, l.maximum_range, a.sp.rest_density,
                 -l.max_surface_distance, l.plain_propagate, l.edge)
}

void launch_fill_stash(hipStream_t s, Profiler* prof, const SweepArgs& a, const LevelArgs& l, float* stash)
{
    ProfScope ps(prof, "level_stash", s);
    if (a.n) hipLaunchKernelGGL(k_fill_stash, dim3((a.n + 255) / 256), dim3(256), 0, s, a.n, l.level, stash, l.max_surface_distance);
}

template <class M>
using OpLevelSmooth = OpLevelSmoothT<M, false>;
template <class M>
using OpLevelSmoothExt = OpLevelSmoothT<M, true>;

// largest displacement |x_new - x_old| (float bits, atomicMax on non-negative floats)
__global__ __launch_bounds__(256) void k_max_disp(uint32_t n, const float4* __restrict__ a, const float4* __restrict__ b, uint32_t* __restrict__ out)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    float d = 0.f;
    if (i < n) {
        const float4 p = a[i], q = b[i];
        const float dx = p.x - q.x, dy = p.y - q.y;
        d = sqrtf(dx * dx + dy * dy);
        if (!(d >= 0.f)) d = __uint_as_float(0x7f800000u);   // NaN -> inf: the caller reports non-finite positions
    }
    d = wave_max(d);
    if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(d));
}
void launch_max_disp(hipStream_t s, Profiler* prof, uint32_t n, const float4* pm_old, const float4* pm_new, uint32_t* out)
{
    ProfScope ps(prof, "level_max_disp", s);
    (void)hipMemsetAsync(out, 0, 4, s);
    if (n) hipLaunchKernelGGL(k_max_disp, dim3((n + 255) / 256), dim3(256), 0, s, n, pm_old, pm_new, out);
}

void launch_level_smooth(hipStream_t s, Profiler* prof, const SweepArgs& a, const LevelArgs& l, const float4* pm_new, const float* in, float* out)
{
    ProfScope ps(prof, "level_smooth", s, true);
    if (l.pm_cell && !l.replay_step_lists) {   // after advection, extended lists of the advected positions
        SPH_DISPATCH(OpLevelSmoothExt, false, pm_new, l.pm_cell, l.k, pm_new, a.orig, a.mrho, in, out, l.level_old, a.status, l.max_surface_distance, a.rho)
        return;
    }
    SPH_DISPATCH(OpLevelSmooth, false, a.pm, nullptr, 2.f, pm_new, a.orig, a.mrho, in, out, l.level_old, a.status, l.max_surface_distance, a.rho)
}

void launch_classify(hipStream_t s, Profiler* prof, uint32_t n, const float4* pm, const float* level, uint8_t* size_class, const uint8_t* owned,
                     const uint32_t* orig, DeviceStatus* status, const sph_params* p)
{
    ProfScope ps(prof, "classify", s);
    if (n)
        hipLaunchKernelGGL(k_classify, dim3((n + 255) / 256), dim3(256), 0, s, n, pm, level, size_class, owned, orig, status, p->maximum_surface_distance,
                           p->rest_density, p->sizing_function, p->particle_radius_fine, p->particle_radius_base);
}
