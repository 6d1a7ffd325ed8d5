// Internal host-side interface between the translation units of libsph_hip.so.
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "sph_device.h"
#include "sph_ffi.h"

// ---- HIP-event profiler (measurement hook; sph_profile_* in the C ABI) ------------------------
struct Profiler {
    int mode = 0;  // 0 off, 1 every kernel, 2 only names starting with "density" (sampled: every 8th step),
                   // 3 (diagnostic) the SWEEPS only, stamped on the device's own clock: the first blocks stamp the constant 100 MHz
                   //   counter at their start, every block at its end, into the launch's slot of a ring (atomic min / max).  The stamps'
                   //   same-address atomics make a sweep ~30 us longer, so this mode does not TIME a sweep; it places the dispatch: first
                   //   wave 3 us behind rocprofv3's start, last wave 0.8 us before its end (profiles/r4_event_calibration.md)
                   // 4 every kernel like mode 1, but a scope that holds ONE sweep launch is timed by that dispatch's own start / end
                   //   timestamps (hipExtLaunchKernelGGL's start / stop events: the completion signal's, what rocprofv3 --kernel-trace
                   //   reports) instead of a marker bracket around it -- no marker excess to calibrate away (bench.py's roofline figures)
    uint64_t step_index = 0;
    struct Rec {
        std::string name;
        uint64_t launches = 0;
        double total_ms = 0;
        std::vector<float> samples;   // per-launch ms
    };
    struct Pending {
        int rec;
        hipEvent_t a, b;
        int slot;   // >= 0: device timestamps in ts_dev[slot] / ts_dev[TS_RING + slot] instead of events
    };
    std::vector<Rec> recs;
    std::vector<Pending> pending;
    std::vector<hipEvent_t> pool;

    int find(const char* name);
    bool wants(const char* name) const;
    hipEvent_t get_event();
    void begin(const char* name, hipStream_t s, bool single_launch = false);
    void end(hipStream_t s);
    // mode 3, inside a single-launch scope: the launch's slot for its device timestamps (nullptr: not wanted)
    unsigned long long* take_slot();
    // mode 4, inside a single-launch scope: the event pair the launch itself carries (false: not wanted, launch plainly)
    bool take_events(hipEvent_t* a, hipEvent_t* b);
    enum { TS_RING = 1 << 16 };
    unsigned long long* ts_dev = nullptr;   // [0, TS_RING): first start (min), [TS_RING, 2 TS_RING): last end (max)
    uint32_t ts_next = 0;
    void collect();  // after a stream sync: fold pending event pairs into recs
    void reset();
    ~Profiler();
    int cur = -1;
    int depth = 0;   // open scopes (only the outermost is timed)
    hipEvent_t cur_a = nullptr, cur_b = nullptr;
    bool ext_open = false;
    int ext_slot = -1;
    bool kev_open = false, kev_taken = false;
};

struct ProfScope {
    Profiler* p;
    hipStream_t s;
    bool on;
    // (mode 3 times the stamped sweeps ONLY: one timing-enabled event anywhere in the process switches the queue to per-dispatch
    //  profiling, and the sweeps themselves then run ~2 us longer -- bench.py takes the other kernels from a mode-1 pass)
    ProfScope(Profiler* p_, const char* name, hipStream_t s_, bool single_launch = false)
        : p(p_), s(s_), on(p_ && p_->mode && p_->wants(name) && (p_->mode != 3 || single_launch))
    {
        if (on) p->begin(name, s, single_launch);
    }
    ~ProfScope()
    {
        if (on) p->end(s);
    }
};

// ---- step header: per-step scalars reduced on the device, read once by the host ---------------
struct HeaderOut {
    float min_x, min_y, max_x, max_y, h_max, h_min, min_cfl;
    uint32_t pad;
};

// ---- Jacobi control block (device resident; copied to the host at sync points) ----------------
struct SolverCtrl {
    uint32_t done;    // stop decision taken (iisph_pressure_iterations break)
    uint32_t iters;   // num_pressure_iters at the break
    uint32_t cur;     // index of the pressure buffer holding the current iterate
    uint32_t peer_error;   // slab decomposition: a device-side guard fired on some rank (the solve was ended on every rank)
    uint32_t normal, singular, negative;
    float sum_err, max_err;
    uint32_t slot_done[2];   // stop decision as seen by the launches of one iteration (written by sweep A(k) into slot k & 1, sph_sweeps.hip)
    uint32_t seq;            // host copy only: sequence number of the k_publish that wrote it (wait_publish)
};

// error word layout: first failing guard wins (atomicCAS from 0)
struct DeviceStatus {
    uint32_t error;  // SPH_ERR_* or 0
    uint32_t info;   // particle index (host order) or auxiliary value
};

// ---- sph_sort.hip ------------------------------------------------------------------------------
// Stable LSD radix sort of (key,val) pairs on `bits` key bits.  Returns 0 if the result is in
// (keyA,valA), 1 if in (keyB,valB).
// the cell sort's key source: cell index of a record; slots [0, n_gone) whose class byte is >= gone_from get the key ncells
struct CellKeyGen {
    const float4* pm;
    GridP g;
    const uint8_t* gone;
    uint32_t n_gone, gone_from;
    // 1: a record outside the grid gets the key of the nearest cell instead of an index beyond the cell table -- the grid of a build
    // queued AHEAD is a prediction (sph_step.hip: queue_ahead_build); such a build is never adopted (the real bounding box does not
    // fit its grid), but its kernels have run by then
    int clamp = 0;
};
// keygen != nullptr: the first pass computes the keys (keyA) and the identity values (valA) itself
int radix_sort_pairs(hipStream_t s, Profiler* prof, uint32_t* keyA, uint32_t* valA, uint32_t* keyB, uint32_t* valB,
                     uint32_t n, int bits, uint32_t* hist_scratch /* >= 256 * nblocks + 256 */, const CellKeyGen* keygen = nullptr);
size_t radix_sort_scratch_elems(uint32_t n);
// The same stable sort as a MERGE, for an array that is sorted by the cells q.cxy_cur (grid q.cur, ranges cell_start_cur) and whose
// particles mostly stay in their cells -- with the reorder of the per-particle arrays in the same pass: sorted keys of the positions
// pm_new in grid q.nxt (same cell size, clamped keys), the arrays of `io` in the new order and the new cell-range table
// [q.nxt.ncells + 1], bit for bit what radix_sort_pairs + launch_reorder + launch_cell_start produce (the permutation itself is never
// stored).  `classified`: the kernel that computed pm_new has run inc_classify_particle (sph_device.h) for every particle already.
// Scratch: q.nk[n], q.mv[n] bytes, q.next[n], q.head[q.nxt.ncells] (zeroed once when allocated, never cleared: q.epoch must differ
// from call to call and from 0), bsum[incremental_sort_block_sums(ncells)], movers (one zeroed word; the call leaves the number of
// movers in *movers_host).
struct ReorderIO {   // what launch_reorder moves (same meaning, same optional members)
    const float4* pm_in;
    const float2* vel_in;
    const uint32_t* orig_in;
    const float *lvl_in, *lvlold_in;
    float4* pm_out;
    float2* vel_out;
    uint32_t* orig_out;
    float *lvl_out, *lvlold_out;
    uint32_t* cxy_out;
    const float* h2n_in;
    float* h2n_out;
    const float* lam_in;
    float* lam_prev_out;
    const uint8_t* szc_in;
    uint8_t* szc_out;
};
size_t incremental_sort_block_sums(uint32_t ncells);
void incremental_cell_sort_reorder(hipStream_t s, Profiler* prof, uint32_t n, const float4* pm_new, const IncClassifyP& q, bool classified,
                                   const uint32_t* cell_start_cur, uint32_t* key_out, uint32_t* cell_start_out, const ReorderIO& io, uint32_t* bsum, uint32_t* movers,
                                   uint32_t* movers_host);

// ... for a slab rank behind its fused refresh: slots [0, n_prev) are last step's sorted array (those of class >= kg.gone_from in
// kg.gone[0 .. kg.n_gone) left the rank and are not placed), [n_prev, n) this step's arrivals; sorted keys and the permutation
// (slot -> current index) of the live slots, and the cell-range table -- what radix_sort_pairs (with the key `ncells` for the slots
// that left) + launch_cell_start produce for them
void incremental_cell_sort_perm(hipStream_t s, Profiler* prof, uint32_t n, uint32_t n_prev, const CellKeyGen& kg, const IncClassifyP& q, const uint32_t* cell_start_cur,
                                uint32_t* key_out, uint32_t* perm_out, uint32_t* cell_start_out, uint32_t* bsum, uint32_t* movers, uint32_t* movers_host);

void launch_reorder(hipStream_t s, Profiler* prof, uint32_t n, GridP g, const uint32_t* sorted_key, const uint32_t* perm,
                    const float4* pm_in, const float2* vel_in, const uint32_t* orig_in, const float* lvl_in,
                    const float* lvlold_in, float4* pm_out, float2* vel_out, uint32_t* orig_out, float* lvl_out,
                    float* lvlold_out, uint32_t* cxy, const float* h2n_in = nullptr, float* h2n_out = nullptr,
                    const float* lam_in = nullptr, float* lam_prev_out = nullptr, void* cell_start_scratch = nullptr,
                    const uint8_t* szc_in = nullptr, uint8_t* szc_out = nullptr);
size_t cell_start_scratch_bytes();
void launch_cell_start(hipStream_t s, Profiler* prof, const uint32_t* sorted_key, uint32_t n, uint32_t ncells,
                       uint32_t* cell_start /* [ncells+1] */, void* scratch /* cell_start_scratch_bytes() */,
                       bool count_zeroed = false /* launch_reorder(.., scratch) already cleared the work-list counter */);
void launch_tile_hmax(hipStream_t s, Profiler* prof, uint32_t n, const float4* pm, GridP g, int ts, int tsx, int tsy, uint32_t* raw,
                      uint32_t* out /* dilated by one tile */, uint32_t* out_ext = nullptr /* dilated by d_ext tiles */, int d_ext = 2);

// ---- sph_sweeps.hip ----------------------------------------------------------------------------
struct SweepArgs {
    GridP g;
    StepP sp;
    uint32_t n;
    int exact;       // EXACT math mode
    int uniform_h;   // all h bit-identical
    float h_uniform;
    // grid structure
    const uint32_t* cell_start;
    const uint32_t* orig;
    // particle state (sorted order)
    const float4* pm;
    float2* vel;
    float2* vel_tmp;
    float* rho;
    float* lam_sum;
    float2* lam_grad;
    float2* wall_pl = nullptr;    // EXACT policy: the boundary handler's per-SDF gradient entries and their count (MathExact, sph_device.h)
    uint8_t* wall_cnt = nullptr;
    float* constf;
    float* aii;
    float* src;
    float* p0;
    float* p1;
    float4* pacc;       // {x, y, a^p} per particle (OpPressureAccel writes it, OpJacobiU gathers it whole)
    float* dens_err;
    float* stat;
    uint32_t* ncount;
    uint4* nl;          // neighbour list words (sph_sweeps.hip)
    uint4* nlx;         // explicit index lists (multi-resolution scenes)
    HeaderOut* hdr_partials;   // per-block partials of the NEXT step's header, written by the integrating final sweep (or nullptr)
    IncClassifyP inc;          // the integrating tail also classifies its particles for the incremental cell sort queued behind it (head == nullptr: no)
    int h_mode;         // support_length_estimation (SPH_H_*)
    int sp_check_aii;   // SimulationParams::check_aii
    float* h2_next;     // FromDistribution*: the estimate for the next step is written here by the density sweep
    float* omega;       // IISPH2
    const uint8_t* size_class;
    const float* lam_prev;
    const uint2* nloff = nullptr;   // relative-offset lists of this step (sph_sweeps.hip: k_sweep_off) and their header bytes; nullptr: the
    const uint8_t* nlh = nullptr;   //   gradient sweeps of a uniform scene replay the mask words
    uint2* nloff_out = nullptr;     // the density BUILD sweep writes them (set for that launch only)
    uint8_t* nlh_out = nullptr;
    uint4* nl_ext;      // list words / index lists of the extended-range lists (level estimation)
    uint4* nlx_ext;
    TileP t;            // stencil bound per tile (multi-resolution scenes; ts = 0: uniform)
    TileP t_ext;        // the same bound over a wider tile neighbourhood, for the extended-range lists
    float* partials;    // per-block solver statistics
    const uint8_t* owned;  // slab decomposition: 1 owned, 0 ghost (nullptr: everything is owned)
    const uint8_t* ring1;  // slab decomposition: 1 = ghost within one support radius of the cut (runs the RING1 ops)
    // slab decomposition, split sweep (SweepCommon): 1 = halo member or ghost; the halo members' slots, the ghosts' slots
    const uint8_t* edge = nullptr;
    const uint32_t* elist_a = nullptr;
    const uint32_t* elist_b = nullptr;
    uint32_t n_ea = 0, n_eb = 0;
    int part = 0;
    double* solver_tot; // multi-rank: all-reduced solver totals (RCCL: this rank's own row; the others' rows in tot_table)
    const double* tot_table = nullptr;   // RCCL transport: every rank's totals side by side, summed by the readers (SolveP, sph_sweeps.hip)
    int tot_nr = 0, tot_self = 0;
    float* mrho;        // m / rho
    float* pt0;         // p / rho^2 for pressure buffer 0 / 1
    float* pt1;
    float4* rec0 = nullptr;   // {x, y, p / rho^2, p} for pressure buffer 0 / 1: written INSTEAD of pt0 / pt1 by the solves of uniform-h scenes
    float4* rec1 = nullptr;   // on one context (OpPressureAccelU gathers one record per neighbour); nullptr: slabs, IISPH2
    float4* xv = nullptr;     // {x, y, vx, vy}: what the source-term sweep of such a scene gathers per neighbour (OpSourceU) -- written by whoever
    bool xv_ok = false;       // writes the velocities behind the step's sort (the non-pressure forces, the divergence solve's tail); xv_ok: it
                              // holds the current positions and velocities (set by the step driver behind those launches); nullptr: slabs, IISPH2
    // boundary
    const BoundaryP* planes;
    const float* lam_lut;
    const float* dlam_lut;
    // control
    SolverCtrl* ctrl;
    const uint32_t* gate = nullptr;   // chained solves: the launches of the second solve leave at once while this word is 0 (k_solver_handoff)
    DeviceStatus* status;
    // paced solves (one context): the stop decision of every iteration is also stored in mapped host memory as
    // (epoch << 16) | (stop << 15) | iteration, so that the host queues iterations against the device's progress (sph_step.hip)
    uint32_t* prog_host = nullptr;
    uint32_t prog_epoch = 0;
    int opt_tile = 0, opt_jacobi_generic = 0;   // Options::tile / ::jacobi_generic of the context (sph_context.hpp)
    Profiler* prof = nullptr;                   // host side only: the context's profiler (launch_sweep asks it for a timestamp slot)
};

// the solves of this step keep p / rho^2 inside the 16-byte records {x, y, p / rho^2, p} (rec0 / rec1) that sweep A gathers whole, not
// in pt0 / pt1: what a slab decomposition exchanges for its ghosts per iteration is then word 2 of those records
bool sweep_a_on_records(const SweepArgs& a);
bool source_term_on_records(const SweepArgs& a);   // the step driver sets SweepArgs::xv_ok behind the launches that write the record
size_t sweep_list_bytes(uint32_t n);
size_t sweep_index_list_bytes(uint32_t n);   // explicit index lists (multi-resolution scenes)
size_t sweep_offset_list_bytes(uint32_t n);  // relative-offset lists (uniform scenes whose solves run on records)
bool sweeps_want_offset_lists(const SweepArgs& a);
bool sweep_forces_index_lists();             // build variant SPH_FORCE_IDX
uint32_t solver_reduce_blocks(uint32_t n);
void launch_density(hipStream_t s, Profiler* prof, const SweepArgs& a);
void launch_density_replay(hipStream_t s, Profiler* prof, const SweepArgs& a);
void launch_constrain_init(hipStream_t s, Profiler* prof, const SweepArgs& a, uint32_t target, float* thr, uint32_t* consumed, float* h_new, uint8_t* flag);
void launch_constrain_pass(hipStream_t s, Profiler* prof, const SweepArgs& a, uint32_t target, float* thr, uint32_t* consumed, float* h_new, uint32_t* pending);
void launch_constrain_apply(hipStream_t s, Profiler* prof, const SweepArgs& a, float4* pm, const float* h_new, float* h2_next);
void launch_aii_const(hipStream_t s, Profiler* prof, const SweepArgs& a);
void launch_aii_const_non_pressure(hipStream_t s, Profiler* prof, const SweepArgs& a);   // + the non-pressure acceleration, one sweep
void launch_non_pressure(hipStream_t s, Profiler* prof, const SweepArgs& a);                // vel -> vel_tmp
void launch_source_term(hipStream_t s, Profiler* prof, const SweepArgs& a, int kind, int residual_density);  // kind: 0 div, 1 full, 2 only-density; includes Jacobi iteration 0
// sweep A of iteration iter >= 1 (+ the stop decision of iteration iter - 1, taken by its block 0); iter < 0: a^p from the solve's final pressures
// part (slab decomposition): 0 one launch; 1 the lanes without a ghost in reach; 2 the halo members and the ghosts (SweepCommon) --
// with part != 0 the rank's totals come from launch_solver_totals instead of this sweep's block 0
void launch_pressure_accel(hipStream_t s, Profiler* prof, const SweepArgs& a, int iter, int residual_density, float max_avg_error, uint32_t max_iters, int multi,
                           int part = 0);
void launch_solver_totals(hipStream_t s, Profiler* prof, const SweepArgs& a, int iter, int residual_density, float max_avg_error, uint32_t max_iters);
// ... as block 0 of the launch that packs the halo members' values for the iteration's ghost exchange (sph_slabs.hip: refresh_ghosts)
void launch_pack_and_totals(hipStream_t s, Profiler* prof, const SweepArgs& a, int iter, int residual_density, float max_avg_error, uint32_t max_iters,
                            const uint32_t* src_idx, uint32_t cnt0, uint32_t cnt1, int words, int stride, int off, const float* field, float* out0, float* out1);
void launch_ctrl_reset(hipStream_t s, SolverCtrl* ctrl, const uint32_t* gate);
void launch_solver_progress(hipStream_t s, Profiler* prof, const SweepArgs& a, int iter, int residual_density, float max_avg_error, uint32_t max_iters);   // paced slabs (no-op unless a.prog_host)
void launch_solver_handoff(hipStream_t s, Profiler* prof, SolverCtrl* ctrl, SolverCtrl* saved_host, uint32_t* gate);
void launch_solver_tail(hipStream_t s, Profiler* prof, const SweepArgs& a, int tail, float4* pm_out, int decide_iter = -1, int residual_density = 0,
                        float max_avg_error = 0.f, uint32_t max_iters = 0, SolverCtrl* handoff_host = nullptr, uint32_t* gate_out = nullptr);
// decide_iter >= 0 (slabs): the stop decision of that iteration is taken here; gate_out: the tail also does k_solver_handoff's job   // integrate map of the solver mode, once the solve is done
void launch_jacobi_update(hipStream_t s, Profiler* prof, const SweepArgs& a, int iter, int residual_density, float max_avg_error, uint32_t max_iters, int multi);
void launch_solver_decide(hipStream_t s, Profiler* prof, const SweepArgs& a, int iter, int residual_density, float max_avg_error,
                          uint32_t max_iters);
// level estimation (sorted order)
struct LevelArgs {
    float k;                       // level_estimation_range / ETA
    float threshold;               // cos(50 degrees)
    float max_surface_distance;
    int boundary_is_fluid_surface;
    float2* nrm;
    uint8_t* state;
    uint8_t* flag_surface;
    uint8_t* flag_insufficient;
    float maximum_range;   // is_neighbor_in_level_estimation_range (FromDistribution / FromDistribution2), < 0: rule off
    uint8_t* size_class;
    float* level;                  // the level field (detection output, propagated in place)
    uint32_t* when;                // propagation sweep that assigned the value (sph_sweeps.hip)
    uint32_t* mark;                // 2 n words: frontier marks, double-buffered by sweep parity 
    uint8_t* fmap = nullptr;       // one context: the propagation on a compacted frontier (sph_sweeps.hip: k_level_frontier) -- two candidate
    uint32_t fmap_lg_s = 0;        //   maps of 64 S bytes, S = 1 << fmap_lg_s >= n / 64 (transposed: particle j at byte (j mod S) 64 + j / S); nullptr: the sweep forms
    float* level_old;
    float* stash_first;            // stash filled right after the detection (SurfaceDistanceFirst) or nullptr
    int center_diff;               // surface_detection_by_center_diff instead of the empty-angle detector
    int plain_propagate;           // slab decomposition: 1 = propagation without frontier marks, 2 = frontier form with probing halo members (OpLevelPropagate)
    const uint8_t* edge;           // mode 2: the slab's halo-member / ghost flags (sph_slabs.hip)
    int replay_step_lists;         // after advection without the extended range: the step's k = 2 lists at the advected positions
    const float4* pm_cell;         // level estimation after advection: the pre-step positions (cells); a.pm = advected. Else nullptr
};
void launch_max_disp(hipStream_t s, Profiler* prof, uint32_t n, const float4* pm_old, const float4* pm_new, uint32_t* out);
void launch_tile_redilate(hipStream_t s, Profiler* prof, int tsx, int tsy, int d, const uint32_t* raw, uint32_t* out);
void launch_level_detect(hipStream_t s, Profiler* prof, const SweepArgs& a, const LevelArgs& l);
void launch_level_propagate(hipStream_t s, Profiler* prof, const SweepArgs& a, const LevelArgs& l, uint32_t t, uint32_t* changed);
void launch_fill_stash(hipStream_t s, Profiler* prof, const SweepArgs& a, const LevelArgs& l, float* stash);
void launch_level_smooth(hipStream_t s, Profiler* prof, const SweepArgs& a, const LevelArgs& l, const float4* pm_new, const float* in, float* out);
void launch_classify(hipStream_t s, Profiler* prof, uint32_t n, const float4* pm, const float* level, uint8_t* size_class, const uint8_t* owned,
                     const uint32_t* orig, DeviceStatus* status, const sph_params* p);
// reduce the per-block header partials of the integrating final sweep into `out_dev` (skipped while the solve is not done)
void launch_header_ahead(sph_ctx* c, uint32_t nblocks, HeaderOut* out_dev, bool publish = false);   // publish: also k_publish's job (the next launch_publish is a no-op)
// IISPH2: p /= sqrt(omega) on the current pressure buffer (+ p / rho^2), simulation.rs:2358-2360
void launch_iisph2_scale(hipStream_t s, Profiler* prof, const SweepArgs& a);
void launch_check_aii(hipStream_t s, Profiler* prof, const SweepArgs& a);   // after launch_aii_const when check_aii is set
void launch_vel_add_pacc(hipStream_t s, Profiler* prof, const SweepArgs& a);               // v += dt a^p
void launch_integrate(hipStream_t s, Profiler* prof, const SweepArgs& a, float4* pm_out, int mode);  // 0: v+=dt a; x+=dt v   1: hybrid

// ---- regather of the persistent state into a new HOST-index order (sparse edits, merging, splitting; sph_api.hip) ----
struct EditSrc {
    uint32_t obj;      // object: < n_old = the particle with that OLD host index, else a default particle of an EXTEND
    uint32_t set_idx;  // index into the override records, or 0xffffffff
};
struct EditSet {
    uint32_t fields;   // SPH_EDIT_F_* of the values to take from this record
    float mass, px, py, vx, vy, h2, h2_next, lvl, lvlold;
};
struct sph_ctx;
// final index f holds object d_src[f].obj (+ the overrides of d_sets[d_src[f].set_idx]); both arrays live on the DEVICE.
// Slot f = host index f afterwards, like a fresh upload; per-step outputs, lists and the header are invalidated.
int regather_host_order(sph_ctx* c, uint32_t n_new, const EditSrc* d_src, const EditSet* d_sets);
// out[i] = sum of in[0 .. i) ; *total (device word) = sum of all.  scratch: >= (n / 2048 + 2) words
void device_exclusive_scan_u32(hipStream_t s, const uint32_t* in, uint32_t* out, uint32_t n, uint32_t* scratch, uint32_t* total);
