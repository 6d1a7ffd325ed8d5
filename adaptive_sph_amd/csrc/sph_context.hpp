// Context of one simulation on one GPU + helpers shared by sph_api.hip and sph_step.hip.
#pragma once

#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <string>
#include <vector>

#include "sph_internal.hpp"

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    hipError_t ensure(size_t need)
    {
        if (need <= bytes) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
        size_t grow = need + need / 4 + 256;
        hipError_t e = hipMalloc(&p, grow);
        if (e == hipSuccess) bytes = grow;
        return e;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    template <class T>
    T* as() const { return (T*)p; }
};

// a DevBuf of function scope: released on every path out of the function (early error returns included)
struct TmpBuf : DevBuf {
    TmpBuf() = default;
    TmpBuf(const TmpBuf&) = delete;
    TmpBuf& operator=(const TmpBuf&) = delete;
    ~TmpBuf() { release(); }
};

// Measurement / test switches.  Read ONCE per context, at sph_create, from the environment (options_from_env, sph_api.hip: the
// variable names are listed there and in README.md) -- nothing in the step path calls getenv, and two contexts of one process may
// run different forms side by side (the bit-identity tests do).  Defaults = the product's behaviour.
// Read from the environment once, at sph_create.  The PRODUCT library (libsph_hip.so) reads the math policy, the transport's behaviour and
// the debug aids (marked P); everything else is a laboratory switch and exists only in the -DSPH_LAB build of the same sources
// (libsph_lab.so, adaptive_sph_amd/build.py: build_lab; options_from_env in sph_api.hip) -- the product runs the defaults below.
struct Options {
    int exact = 0;              // SPH_HIP_EXACT=1        P EXACT math policy (the reference's operations; diagnostics)
    int paced = 1;              // SPH_PACED=0              predicted queue + waits instead of pacing against the device's decisions
    int pace_lead = 0;          // SPH_PACE_LEAD=<k>        undecided iterations allowed in the queue (0: by particle count)
    int pace_pred = 0xffff;     // SPH_PACE_PRED=0|1|2      unpaced head of a solve (0xffff: by particle count)
    int chain = -1;             // SPH_CHAIN=0|1            predicted queue: never / always chain the two solves of HybridDFSPH (-1: while the count repeats)
    int overlap = -1;           // SPH_OVERLAP=0|1          slabs: never / always split sweep A around the exchange (-1: by slab size)
    int accel_generic = 0;      // SPH_ACCEL_GENERIC        sweep A through OpPressureAccel (no pressure records)
    int jacobi_generic = 0;     // SPH_JACOBI_GENERIC       sweep B through OpJacobi
    int source_generic = 0;     // SPH_SOURCE_GENERIC       the source-term sweep through OpSource (no {x, y, v} records)
    int slab_general = 0;       // SPH_SLAB_GENERAL         slabs: always the general maintenance path (no fused refresh)
    int slab_level_plain = 0;   // SPH_SLAB_LEVEL_PLAIN     slabs: level propagation without frontier marks
    int level_serial = 0;       // SPH_LEVEL_SERIAL         level estimation on the main stream
    int level_batch8 = 0;       // SPH_LEVEL_BATCH8         propagation sweeps in fixed batches of 8
    int offset_lists = 1;       // SPH_OFFSET_LISTS=0       the Jacobi sweeps of uniform scenes replay the mask words instead of the 16-bit offset lists
    int level_queue = 1;        // SPH_LEVEL_QUEUE=0        one context: propagation sweeps over all particles (frontier marks) instead of the compacted frontier
    int no_fuse = 0;            // SPH_NO_FUSE              a_ii / constant field and the non-pressure forces in two sweeps
    int event_wait = 0;         // SPH_EVENT_WAIT           wait on events instead of spinning on mapped words
    int loopback_sync = 0;      // SPH_LOOPBACK_SYNC=1      loopback transport with host waits
    int side_stream_normal = 0; // SPH_SIDE_STREAM_NORMAL   side stream at normal priority
    int force_slab_mode = 0;    // SPH_FORCE_SLAB_MODE      sph_dist_configure(0, 1, ..) turns the slab driver on (one-rank check of that path)
    int tile = 0;               // SPH_TILE=<bits>          LDS-staged sweeps (sph_set_sweep_variant overrides, process-wide)
    int ahead_build = 1;        // SPH_AHEAD_BUILD=0        one context: never queue the next step's cell sort behind the integrating tail
    int inc_sort = 1;           // SPH_INC_SORT=0 | k       the cell sort queued ahead is always the radix sort (never the incremental merge); k > 1: the merge up to n / k movers (default n / 3: at 4M the merge still beats the radix sort with a third of the particles changing cell)
    int slab_paced = 1;         // SPH_SLAB_PACED=0         slabs: predicted queue instead of pacing
    int slab_records = 1;       // SPH_SLAB_RECORDS=0       slabs: sweep A through the generic form (p / rho^2 as a field of its own)
    int debug_sync = 0;         // SPH_DEBUG_SYNC=<mask>    synchronise and name the phases (fault hunting)
    int debug_counts = 0;       // SPH_DEBUG_COUNTS         print the fused refresh's counts
    int comm_delay_us = 0;      // SPH_DEBUG_COMM_DELAY_US  loopback transport: every exchange / all-reduce occupies its stream that long
    int hip_trace = 0;          // SPH_HIP_TRACE=1          host-side timeline of the step
    int side_cus = 0;           // SPH_SIDE_CUS=<k>         (lab) the side stream (level-set propagation) owns k CUs of every XCD through a CU mask (0: no mask)
    int main_exclude = 0;       // SPH_MAIN_EXCLUDE=1       (lab) ... and the main stream is masked OFF those CUs
};
Options options_from_env();   // sph_api.hip

struct sph_ctx {
    int device = 0;
    Options opt;
    uint64_t cap = 0, n = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;   // side stream: the level estimation before advection runs under the step's own sweeps
    hipEvent_t ev_fork = nullptr;    // main -> side stream dependency
    uint32_t level_seq = 0;          // sequence number of the side stream's publishes (lvl_changed[63])
    int n_planes = 0;
    BoundaryP bnd_h{};   // planes, or one Sdf2D polygon (sph_set_boundary_polygon)
    float time = 0.f;
    uint64_t step_number = 0;
    uint64_t n_waits = 0;    // host waits on the device since the last sph_dist_get_stats(reset)
    bool poisoned = false;   // a guard fired inside a step: state undefined until sph_upload (sph_ffi.h)
    std::string err;
    Profiler prof;
    int exact = 0;
    std::vector<uint32_t> export_cnt, export_off;   // sph_download_neighbors: host staging of the counts / offsets, kept across calls
    DevBuf export_d_off, export_d_idx;              // ... and its device-side CSR (a 200 MB hipMalloc + hipFree per call otherwise); freed by sph_destroy and by sph_upload

    // persistent SoA (ping-pong across the per-step reorder)
    DevBuf pm[2], vel[2], orig[2], lvl[2], lvlold[2];
    int cur = 0;   // which of the (vel, orig, lvl, lvlold) ping-pong set is live
    int pcur = 0;  // which pm buffer is live; the other one holds the sorted PRE-step positions after a step
    DevBuf vel_tmp;
    // per-step
    DevBuf key[2], val[2], sort_scratch, cxy, cell_start, cs_scratch, nl, nl_ok, mrho, pt0, pt1, prec0, prec1, xv;
    bool uniform_h = false;
    float h_uniform = 0.f;
    DevBuf wall_pl, wall_cnt;   // EXACT policy only (MathExact, sph_device.h)
    DevBuf rho, lam_sum, lam_grad, constf, aii, src, p0, p1, pacc, dens_err, stat, ncount;
    DevBuf planes_d, lam_lut, dlam_lut, hdr_partials, hdr_out, ctrl, status, n_tiles, red_partials, scratch;
    // mapped pinned host memory: written by kernels directly (no D2H copy launches)
    HeaderOut* hdr_host = nullptr;
    volatile uint32_t* hint_word = nullptr;   // the kernel queued LAST stores hint_seq here as its final store: the next wait spins on it
    uint32_t hint_seq = 0;
    uint32_t publish_seq = 0;   // sequence number of the last k_publish (the host spins on its arrival in ctrl_host)
    bool publish_folded = false;   // the k_header_ahead queued last also publishes: launch_publish has nothing to launch
    SolverCtrl* ctrl_host = nullptr;
    DeviceStatus* status_host = nullptr;
    HeaderOut* hdr_host_dev = nullptr;
    SolverCtrl* ctrl_host_dev = nullptr;
    DeviceStatus* status_host_dev = nullptr;
    volatile uint32_t* prog_host = nullptr;   // paced solves: the device's last stop decision (behind ctrl_host[1]; SweepArgs::prog_host)
    uint32_t* prog_host_dev = nullptr;
    uint32_t solve_epoch = 0;
    bool paced_step = false;   // this step's solves are paced (SPH_PACED, read once per step)
    hipEvent_t ev_sync = nullptr;

    // ---- slab decomposition (multi-GPU): this context owns x in [cut_lo, cut_hi) -------------------
    struct Dist {
        bool on = false;
        int rank = 0, nranks = 1;
        float cut_lo = 0.f, cut_hi = 0.f;
        uint32_t n_tot = 0;          // particles in the arrays incl. ghosts (== n when not distributed)
        bool have_flags = false;     // `owned` describes the current arrays (after a step)
        uint32_t n_halo[2] = {0, 0}, n_ghost[2] = {0, 0};   // [left, right]
        // Ghost width PER CUT: H = the largest smoothing length among the particles of both ranks within the region a global-width
        // layer would span around the cut ([left, right]; agreed with the x-neighbour), layer width = base_k * H + slack.  A fine
        // region of a strongly multi-resolution scene then pays for its own support, not for the coarsest particle anywhere.
        float hcut[2] = {0.f, 0.f};     // of the current ghost layer (the next fused refresh predicts with it)
        float halo_w[2] = {0.f, 0.f};   // the widths the current layer was selected with
        DevBuf owned;                // u8 per slot: 1 owned, 0 ghost
        // fused refresh (sph_step.hip, slab_refresh_fused): class byte per slot of the previous step's arrays, per-block class
        // counts / offsets; `pre_*` describe the arrays between the refresh and the cell sort, which drops the slots that left
        DevBuf cls, blk;
        int overlap_env = -1;        // SPH_OVERLAP as read at the start of the step (-1: unset)
        bool pre = false;            // the arrays still hold slots that left this rank (class >= SC_GONE_FROM in cls[0 .. pre_cls_n))
        uint32_t pre_n = 0;          // slots in the arrays before the cell sort (live + gone)
        uint32_t pre_cls_n = 0;      // slots cls describes (the previous step's arrays)
        uint32_t pre_own = 0;        // pre-sort slots below this index are owned (if live), the ghosts follow
        DevBuf tot_table;            // the ranks' solver totals side by side (RCCL transport: all-gather by send / receive), 2 slots
        DevBuf ring1, ring1_src;     // u8 per slot / per ghost ordinal: ghost within one support radius of the cut
        DevBuf halo_idx, halo_pos, halo_src, ghost_dst;      // index lists / maps (u32)
        DevBuf send[2], recv[2];     // staging, [left, right]
        DevBuf counts;               // device counters
        DevBuf solver_tot;           // 4 doubles: normal, singular, negative, sum_err (all-reduced)
        uint32_t* counts_host = nullptr;       // mapped pinned
        uint32_t* counts_host_dev = nullptr;
        void* nccl = nullptr;        // ncclComm_t
        // Split sweep A (sph_step.hip, exchange_and_sweep_a): the sweep over the particles that have no ghost in reach runs on a
        // stream of its own, beside the ghost exchange and the all-reduce of the iteration on the main stream
        DevBuf edge;                 // u8 per slot: 1 = halo member or ghost
        hipStream_t xstream = nullptr;
        hipEvent_t ev_x[3] = {nullptr, nullptr, nullptr};   // [0] sweep B done (main), [1] interior done (side)
        // loopback transport without host waits (LocalComm): my staging is packed / my copies from the neighbours are done / my
        // totals are published; the group's totals meet in mapped host memory owned by member 0 (two parities x ranks x 8 doubles)
        hipEvent_t ev_pack = nullptr, ev_copied = nullptr, ev_tot = nullptr;
        double* gtot = nullptr;
        uint32_t gtot_seq = 0;
        bool ghosts_ok = true;       // false: this rank ran out of room for its ghost layer (the step goes on without it and ends in SPH_ERR_CAPACITY on every rank)
        void* tgroup = nullptr;      // ThreadGroup*: in-process transport with one host thread per rank (sph_comm_init_threads)
        void* ipc = nullptr;         // IpcState*: peer-mapped push transport on top of the shared-memory one (sph_comm_init_ipc)
        void* shm = nullptr;         // ShmSegment*: processes of one node without RCCL (sph_comm_init_shm)
        size_t shm_bytes = 0;
        std::string shm_name;        // non-empty on the rank that created the segment (it unlinks the name)
        int rebalance_every = 0;     // move the cuts to equal particle counts every so many steps (0: static cuts)
        DevBuf hist;                 // x histogram of the owned particles (rebalancing)
        uint32_t rebalances = 0;     // how often the cuts moved
        uint64_t stat_exchanges = 0, stat_bytes_sent = 0, stat_bytes_recv = 0, stat_allreduces = 0, stat_step0 = 0;   // sph_dist_get_stats
    } dist;

    GridP grid{};
    bool grid_valid = false;
    GridP fgrid{};          // the grid the particles are sorted by (== grid in uniform scenes)
    int tile_ts = 0, tile_tsx = 0, tile_tsy = 0;
    DevBuf tile_raw, tile_h, tile_h_ext, nlx;
    DevBuf nloff, nlh;   // relative-offset lists + header words (uniform scenes whose solves run on records; sph_sweeps.hip: k_sweep_off)
    // Neighbour build AHEAD (one context; uniform scenes and multi-resolution scenes on their fine grid; sph_step.hip: plan_ahead_build, queue_ahead_build): the NEXT step's cell sort, reorder and
    // cell-range table are queued behind this step's integrating tail, on a grid predicted from this step's bounding box plus a
    // margin -- the device works on them while the host finishes the step, returns, and enters the next one (the step boundary was
    // ~20-36 us of idle queue).  They write buffers of their own (post-step downloads still see this step's order, keys and ranges);
    // the next step adopts them by swapping pointers if nothing touched the state and the real bounding box fits the predicted grid.
    DevBuf akey[2], aval[2], acxy, acell_start, pm2;
    DevBuf atile_raw, atile_h;   // ... and, in a multi-resolution scene, the tiles' h bounds on the predicted grid
    // the build queued ahead as a merge of the particles that stay in their cells with the few that do not (sph_sort.hip:
    // incremental_cell_sort): per-cell list heads (epoch-tagged, never cleared), list links, block sums, the mover counter
    DevBuf inc_head, inc_next, inc_bsum, inc_movers;
    uint32_t inc_epoch = 0;     // tag of the current call's list heads
    int inc_radix_streak = 0;   // ahead builds in a row that took the radix sort because the last known mover count was large
    bool inc_count_valid = false;   // the mapped mover-count word was written by a merge queued SINCE the state was last replaced (upload, edits,
                                    // adaptivity, math policy): a build still in flight at such a call may write its stale count after the host's reset (advisor r5)
    struct Ahead {
        bool valid = false;
        GridP g{};
        float h_max = 0.f, h_min = 0.f, rest_density = 0.f;
        int tile_ts = 0, tile_tsx = 0, tile_tsy = 0;   // multi-resolution scenes: the tiles of the predicted grid
        uint64_t n = 0;
    } ahead;
    DevBuf hdr_ahead_partials;   // per sweep block: next step's header terms from the integrating final sweep
    bool hdr_ahead = false;      // hdr_host already holds the header of the state on the device (no k_header needed)
    float hdr_ahead_rest_density = 0.f;
    DevBuf h2n[2];     // ParticleVec::h2_next (FromDistribution* support-length estimation), ping-pong across the reorder
    DevBuf lam_prev;   // lambda_sum of the previous step in this step's order (estimate_h_next_from_distribution)
    // level estimation (simulation.rs:539-927), sorted order
    DevBuf lvl_queue;   // the propagation's compacted frontier: two transposed candidate maps, one byte per particle (sph_sweeps.hip: k_level_frontier)
    DevBuf lvl_tmp, lvl_nrm, lvl_state, lvl_when, lvl_mark, flag_surface, flag_insufficient, stash, nl_ext, nlx_ext;
    DevBuf con_thr, con_consumed, con_h, flag_reduced;   // constrain_neighborhood_count
    bool have_reduced = false;
    bool lists_after = false;        // the cache holds the extended lists of the advected positions (level_estimation_after_advection)
    float lists_after_k = 0.f, lists_after_slack = 0.f;
    float h_max_step = 0.f;   // largest smoothing length of the current step (all ranks)
    float last_dmax = 0.f;    // slab decomposition, level estimation after advection: largest displacement of the previous step (all ranks)
    float slab_slack_w = 0.f; // ... and the extra ghost width this step's layer was built with for it
    DevBuf szc[2];     // ParticleVec::particle_size_class (u8), persistent: IISPH2's omega reads the class of the previous step
    DevBuf omega;      // IISPH2 (simulation.rs:2262-2311)
    DevBuf split_patterns;            // SplitPatterns::pos_s of every pattern, concatenated float2 (sph_set_split_patterns)
    uint32_t n_split_patterns = 0;
    bool have_level = false;            // the level-estimation outputs above are those of the last step
    DevBuf lvl_changed_d;               // per-sweep "assigned something" words of a batch (device), published once per batch
    uint32_t* lvl_changed = nullptr;    // mapped pinned host copy
    uint32_t last_level_sweeps = 0;     // effective propagation sweeps of the previous step (length of the next first batch)
    uint32_t* lvl_changed_dev = nullptr;
    uint32_t pressure_cur = 0;
    uint32_t last_div_iters = 2, last_dens_iters = 2;
    uint32_t prev_dens_iters = 0;
    uint32_t prev_div_iters = 0;   // the step before: the solves are chained only while the divergence solve's count repeats
    hipEvent_t ev[8];

    int fail(int code, const char* fmt, ...)
    {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        err = buf;
        return code;
    }
};

#define HIPCHK(ctx, call)                                                                                  \
    do {                                                                                                   \
        hipError_t e_ = (call);                                                                            \
        if (e_ != hipSuccess) return (ctx)->fail(SPH_ERR_DEVICE, "%s failed: %s", #call, hipGetErrorString(e_)); \
    } while (0)


// ---- shared helpers (sph_step.hip) -------------------------------------------------------------
int wait_stream(sph_ctx* c);
const char* status_message(uint32_t code);
SweepArgs make_args(sph_ctx* c, const StepP& sp);

// ---- small launch wrappers owned by sph_api.hip -----------------------------------------------
void launch_header(sph_ctx* c, uint32_t n, float rest_density, int from_mass, HeaderOut* out_dev, const uint8_t* owned = nullptr);
void launch_publish(sph_ctx* c);
void launch_profile_calibration(sph_ctx* c);   // Profiler mode 1: one spin kernel of known duration per step (sph_api.hip)
void launch_check_neighborhood(sph_ctx* c, const SweepArgs& a);
void dist_release(sph_ctx* c);  // sph_step.hip
