// The step driver: sequences the sweeps exactly like single_step_without_adaptivity
// (/root/reference/src/simulation/simulation.rs:1980-2730) -- the citations sit next to each call --
// for a GROUP of contexts (ranks) and a transport between them:
//
//   single GPU            group of one context, no transport                      (sph_step)
//   one process per GPU   group of one context, RCCL transport over xGMI          (sph_step after sph_comm_init)
//   loopback              group of k contexts in one process, copies as transport (sph_group_step; used to
//                         verify the slab algorithm against the single-context result on one GPU)
//
// (The slab maintenance itself is sph_slabs.hip, the transports sph_transport.hip; what they share with this file: sph_dist.hpp.)
// Multi-rank decomposition: 1-D slabs along x, rank r owns x in [cut_r, cut_r+1).
// Per step: owned particles that left the slab migrate to the x-neighbour; owned particles within one
// support radius (2 h_max, all-reduced) of a cut are sent to that neighbour as ghosts; every sweep whose
// output is read through neighbour gathers (rho, m/rho, v, p/rho^2, a^p) is followed by a ghost refresh
// of exactly that field; CFL dt and the Jacobi residual statistics are all-reduced so that every rank
// takes the same stop decision.  Ghost lanes are idle in the sweeps (their values come from their owner).
#include <atomic>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <array>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <vector>

#include "sph_dist.hpp"

// ------------------------------------------------------------------------------------------------
// helpers shared with sph_api.hip
// ------------------------------------------------------------------------------------------------
SweepArgs make_args(sph_ctx* c, const StepP& sp)
{
    SweepArgs a{};
    const int k = c->cur;
    a.g = c->fgrid;
    a.sp = sp;
    a.n = c->dist.on ? c->dist.n_tot : (uint32_t)c->n;
    a.exact = c->exact;
    a.cell_start = c->cell_start.as<uint32_t>();
    a.orig = c->orig[k].as<uint32_t>();
    a.pm = c->pm[c->pcur].as<float4>();
    a.vel = c->vel[k].as<float2>();
    a.vel_tmp = c->vel_tmp.as<float2>();
    a.rho = c->rho.as<float>();
    a.lam_sum = c->lam_sum.as<float>();
    a.lam_grad = c->lam_grad.as<float2>();
    a.wall_pl = c->wall_pl.as<float2>();
    a.wall_cnt = c->wall_cnt.as<uint8_t>();
    a.constf = c->constf.as<float>();
    a.aii = c->aii.as<float>();
    a.src = c->src.as<float>();
    a.p0 = c->p0.as<float>();
    a.p1 = c->p1.as<float>();
    a.pacc = c->pacc.as<float4>();
    a.dens_err = c->dens_err.as<float>();
    a.stat = c->stat.as<float>();
    a.ncount = c->ncount.as<uint32_t>();
    a.nl = c->nl.as<uint4>();
    a.nlx = c->nlx.as<uint4>();
    a.hdr_partials = c->hdr_ahead_partials.as<HeaderOut>();
    a.h_mode = SPH_H_FROM_MASS;   // set by the step driver
    a.h2_next = c->h2n[k].as<float>();
    a.omega = c->omega.as<float>();
    a.size_class = c->szc[k].as<uint8_t>();
    a.lam_prev = c->lam_prev.as<float>();
    a.nl_ext = c->nl_ext.as<uint4>();
    a.nlx_ext = c->nlx_ext.as<uint4>();
    a.t = TileP{c->tile_ts, c->tile_tsx, c->tile_tsy, c->tile_h.as<uint32_t>()};
    a.t_ext = TileP{c->tile_ts, c->tile_tsx, c->tile_tsy, c->tile_h_ext.as<uint32_t>()};
    a.partials = c->red_partials.as<float>();
    a.mrho = c->mrho.as<float>();
    a.pt0 = c->pt0.as<float>();
    a.pt1 = c->pt1.as<float>();
    if (!c->dist.on || c->opt.slab_records) {   // (Options::slab_records = 0: slab decompositions keep p / rho^2 as a field of its own)
        a.rec0 = c->prec0.as<float4>();
        a.rec1 = c->prec1.as<float4>();
    }
    if (!c->dist.on && !c->opt.source_generic) a.xv = c->xv.as<float4>();   // (a slab's ghosts get velocities, not records)
    a.uniform_h = c->uniform_h ? 1 : 0;
    a.h_uniform = c->h_uniform;
    a.planes = c->planes_d.as<BoundaryP>();
    a.lam_lut = c->lam_lut.as<float>();
    a.dlam_lut = c->dlam_lut.as<float>();
    a.ctrl = c->ctrl.as<SolverCtrl>();
    a.status = c->status.as<DeviceStatus>();
    a.owned = c->dist.on ? c->dist.owned.as<uint8_t>() : nullptr;
    a.ring1 = c->dist.on ? c->dist.ring1.as<uint8_t>() : nullptr;
    if (c->dist.on) {
        a.edge = c->dist.edge.as<uint8_t>();
        a.elist_a = c->dist.halo_src.as<uint32_t>();
        a.n_ea = c->dist.n_halo[0] + c->dist.n_halo[1];
        a.elist_b = c->dist.ghost_dst.as<uint32_t>();
        a.n_eb = c->dist.ghosts_ok ? c->dist.n_ghost[0] + c->dist.n_ghost[1] : 0u;
    }
    a.solver_tot = c->dist.solver_tot.as<double>();
    if (c->dist.on && c->dist.nccl) {   // RCCL transport: the totals are an all-gather, summed by their readers (sph_transport.hip)
        a.tot_table = c->dist.tot_table.as<double>();
        a.tot_nr = c->dist.nranks;
        a.tot_self = c->dist.rank;
    }
    a.opt_tile = c->opt.tile;
    a.opt_jacobi_generic = c->opt.jacobi_generic;
    a.prof = &c->prof;
    return a;
}

const char* status_message(uint32_t code)
{
    switch (code) {
    case SPH_ERR_DENSITY_NOT_FINITE: return "assertion failed: p_density.is_finite()";
    case SPH_ERR_DENSITY_TOO_SMALL: return "assertion failed: *p_density > 0.0001";
    case SPH_ERR_AII_NOT_FINITE: return "assertion failed: (*p_aii).is_finite()";
    case SPH_ERR_AII_NEGATIVE: return "AII should not be negative!";
    case SPH_ERR_AP_NOT_FINITE: return "'!a_p.is_finite()' failed. Pressure values probably have exploded!";
    case SPH_ERR_PRESSURE_NOT_FINITE: return "'!p_pressure_next_iter.is_finite()' failed.";
    case SPH_ERR_TOO_MANY_NEIGHBORS: return "exceeded maximum allowed number of 20000 neighbors";
    case SPH_ERR_VELOCITY_NOT_FINITE: return "Assertion 'p_velocity[d].is_finite()' failed!";
    case SPH_ERR_POSITION_NOT_FINITE: return "Assertion 'p_position[d].is_finite()' failed!";
    case SPH_ERR_VISCOSITY_NOT_FINITE: return "Assertion 'viscosity_accel[d].is_finite()' failed!";
    case SPH_ERR_CHECK_NEIGHBORHOOD: return "neighbour list differs from the brute-force definition";
    case SPH_ERR_CHECK_AII: return "a_ii value not equal with a tolerance of 0.01";
    case SPH_ERR_LEVEL_WEIGHT: return "weight is <=0 in smooth_level_estimation_field";
    case SPH_ERR_VOLUME_ESTIMATE: return "assertion failed: volume_estimate >= 0.";
    case SPH_ERR_CONSTRAIN_NOT_SMALLER: return "assertion failed: *p_h_next < smoothing_length_single(&particles.h2, i, simulation_params)";
    case SPH_ERR_CONSTRAIN_NEGATIVE: return "assertion failed: *p_h_next >= 0.";
    case SPH_ERR_UNSUPPORTED: return "a particle's neighbour list exceeds what is recorded (it re-walks its candidates): replaying the step's lists at the advected positions is not covered for it";
    case SPH_ERR_CAPACITY: return "a slab context ran out of room for the particles handed to it or for its ghost layer (it needs owned + arrivals + 2 x ghosts slots)";
    default: return "device-side guard failed";
    }
}

// Wait for everything queued on the context's stream.  Busy-polls an event instead of
// hipStreamSynchronize: the blocking wait's wake-up latency would otherwise be paid at every wait point.
static int wait_word(sph_ctx* c, volatile uint32_t* word, uint32_t want);
// Only for a wait that DIRECTLY follows the launch that armed the hint (k_header_final, k_copy_counts): that kernel announces its
// end in mapped memory as its last store, and the host spins there instead of polling an event.  Every other wait drops the hint.
static int wait_hinted(sph_ctx* c)
{
    if (c->hint_word) {
        volatile uint32_t* w = c->hint_word;
        c->hint_word = nullptr;
        return wait_word(c, w, c->hint_seq);
    }
    return wait_stream(c);
}
int wait_stream(sph_ctx* c)
{
    c->hint_word = nullptr;
    c->n_waits++;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipEventRecord(c->ev_sync, c->stream));
    auto t0 = std::chrono::steady_clock::now();
    for (uint64_t spins = 0;; spins++) {
        hipError_t e = hipEventQuery(c->ev_sync);
        if (e == hipSuccess) return SPH_OK;
        if (e != hipErrorNotReady) return c->fail(SPH_ERR_DEVICE, "hipEventQuery failed: %s", hipGetErrorString(e));
        if ((spins & 0xfffu) == 0xfffu && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 120.0)
            return c->fail(SPH_ERR_DEVICE, "device did not finish the queued work within 120 s");
    }
}

// Wait for the k_publish just queued: its sequence number arrives in mapped host memory as the kernel's last store, behind
// everything queued before it.  Falls back to the event wait if it does not show up (or SPH_EVENT_WAIT is set).
static int wait_word(sph_ctx* c, volatile uint32_t* word, uint32_t want)
{
    if (c->opt.event_wait) return wait_stream(c);
    c->n_waits++;
    auto t0 = std::chrono::steady_clock::now();
    for (uint64_t spins = 0; *word != want; spins++) {
        if ((spins & 0xffffu) == 0xffffu && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 2.0)
            return wait_stream(c);   // e.g. a faulted kernel: let the event path report it
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    return SPH_OK;
}
static int wait_publish(sph_ctx* c)
{
    c->hint_word = nullptr;
    return wait_word(c, &((volatile SolverCtrl*)c->ctrl_host)->seq, c->publish_seq);
}

// a few device words -> mapped host memory, the sequence number last (same idea as k_publish): the result of a host-value
// collective reaches the host without a copy engine round trip and without polling an event
__global__ void k_publish_words(const uint32_t* __restrict__ src, uint32_t n, uint32_t* __restrict__ dst_host, uint32_t* __restrict__ seq_host,
                                uint32_t seq)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        for (uint32_t k = 0; k < n; k++) dst_host[k] = src[k];
        __threadfence_system();
        *(volatile uint32_t*)seq_host = seq;
    }
}
int publish_and_wait(sph_ctx* c, const void* dev_words, uint32_t n_words)   // -> counts_host[16 ..), sequence in counts_host[63]
{
    c->publish_seq++;
    if (c->publish_seq == 0u) c->publish_seq = 1u;
    hipLaunchKernelGGL(k_publish_words, dim3(1), dim3(64), 0, c->stream, (const uint32_t*)dev_words, n_words, c->dist.counts_host_dev + 16,
                       c->dist.counts_host_dev + 63, c->publish_seq);
    return wait_word(c, (volatile uint32_t*)c->dist.counts_host + 63, c->publish_seq);
}

// sum of a batch's 0/1 flags -> mapped host memory, the sequence number last (same idea as k_publish)
__global__ void k_publish_count(const uint32_t* __restrict__ flags, uint32_t n, uint32_t* __restrict__ dst_host, uint32_t seq_slot, uint32_t seq)
{
    uint32_t v = 0;
    for (uint32_t k = threadIdx.x; k < n; k += 64) v += flags[k] ? 1u : 0u;
    v = wave_sum_u32(v);
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        dst_host[0] = v;
        __threadfence_system();
        ((volatile uint32_t*)dst_host)[seq_slot] = seq;
    }
}

static int ilog2_ceil(uint32_t v)
{
    int b = 0;
    while ((1ull << b) < (uint64_t)v) b++;
    return b;
}

// host-side timeline of one step (SPH_HIP_TRACE=1): where the CPU thread spends its time
struct HostTrace {
    bool on = false;
    std::chrono::steady_clock::time_point t0;
    double acc[8] = {0};
    int steps = 0;
    void start(bool enabled)
    {
        on = enabled;
        if (on) t0 = std::chrono::steady_clock::now();
    }
    void mark(int k)
    {
        if (!on) return;
        auto t1 = std::chrono::steady_clock::now();
        acc[k] += std::chrono::duration<double, std::micro>(t1 - t0).count();
        t0 = t1;
    }
    void end_step()
    {
        if (!on) return;
        if (++steps % 20 == 0)
            fprintf(stderr, "[sph trace] per step us: decomposition %.1f | header %.1f | neighbourhood launches %.1f | first sweeps %.1f | div solve %.1f | dens solve %.1f\n",
                    acc[0] / steps, acc[1] / steps, acc[2] / steps, acc[3] / steps, acc[4] / steps, acc[5] / steps);
    }
};
static thread_local HostTrace g_trace;   // (diagnostic, SPH_HIP_TRACE=1; per host thread: contexts stepped from different threads do not share it)

// SPH_DEBUG_SYNC=1 (fault hunting): synchronise and name the phase just queued
void dbg_sync(sph_ctx* c, const char* what, int id)
{
    if (!(c->opt.debug_sync & (1 << id))) return;   // bit per sync point
    (void)hipSetDevice(c->device);
    const hipError_t e = hipStreamSynchronize(c->stream);
    fprintf(stderr, "[sph debug] rank %d step %llu: %s -> %s\n", c->dist.rank, (unsigned long long)c->step_number, what, hipGetErrorString(e));
}

// (wait for everything queued on every member's stream; declared in sph_dist.hpp)
int wait_all_hinted(Group& G)   // see wait_hinted
{
    for (auto c : G.m) {
        int rc = wait_hinted(c);
        if (rc) return rc;
    }
    return SPH_OK;
}
int wait_all(Group& G)
{
    for (auto c : G.m) {
        int rc = wait_stream(c);
        if (rc) return rc;
    }
    return SPH_OK;
}

// ------------------------------------------------------------------------------------------------
// the group step
// ------------------------------------------------------------------------------------------------
// collective error check at a wait point: returns the first error of this process, or (multi-rank) a generic
// failure if another rank failed -- every rank must leave the step together or the next collective hangs
int agree(Group& G, int local_rc)
{
    if (!G.multi()) return local_rc;
    std::vector<int> v(G.m.size(), local_rc);
    int rc = G.comm->allreduce_max_i32(G, v);
    if (rc) return rc;
    if (local_rc) return local_rc;
    if (v[0]) return G.m[0]->fail(v[0], "another rank of the slab decomposition reported status %d", v[0]);
    return SPH_OK;
}

// publish ctrl + status of every member to the host and wait; device-side guards -> status code.
//   SYNC_AGREE  one rank, or a slab wait point that agrees on the guards with a collective of its own (level estimation)
//   SYNC_DEFER  slab decomposition, inside a solve: the guards are NOT turned into a status here -- a rank that left alone would
//               leave the others in the next collective.  A guard that fired ends the solve on every rank through the
//               all-reduced totals (solver_decide_multi) and the step runs to its end
//   SYNC_FINAL  slab decomposition, the step's last wait: the guards were agreed on the device (Comm::agree_guards_queued put
//               the maximum over the ranks into ctrl->peer_error before this publish), so every rank reports together without
//               another round trip
int sync_ctrl(Group& G, int mode)
{
    int rc = SPH_OK;
    if (!G.multi()) mode = SYNC_AGREE;
    for (auto c : G.m) {
        (void)hipSetDevice(c->device);
        launch_publish(c);
    }
    uint32_t peer = 0;
    for (auto c : G.m) {
        int r = wait_publish(c);
        if (r && !rc) rc = r;
        if (mode == SYNC_DEFER) continue;
        if (!r && c->status_host->error) {
            uint32_t code = c->status_host->error, info = c->status_host->info;
            (void)hipMemsetAsync(c->status.p, 0, sizeof(DeviceStatus), c->stream);
            c->status_host->error = 0;
            r = c->fail((int)code, "%s (particle i=%u)", status_message(code), info);
            if (!rc) rc = r;
        }
        if (!r && c->ctrl_host->peer_error > peer) peer = c->ctrl_host->peer_error;
    }
    if (mode == SYNC_FINAL) {
        if (!rc && peer) rc = G.m[0]->fail(peer > 1 ? (int)peer : SPH_ERR_DEVICE, "another rank of the slab decomposition reported status %u", peer);
        return rc;
    }
    // (SYNC_DEFER: rc can only be a wait that failed, i.e. a lost device -- nothing a collective on that device could agree on)
    if (mode == SYNC_DEFER) return rc;
    return agree(G, rc);
}

static float* sel_rho(Member& m) { return m.a.rho; }
static float* sel_vel(Member& m) { return (float*)m.a.vel; }
static float* sel_pt0(Member& m) { return m.a.pt0; }
static float* sel_pt1(Member& m) { return m.a.pt1; }
static float* sel_rec0(Member& m) { return (float*)m.a.rec0; }
static float* sel_rec1(Member& m) { return (float*)m.a.rec1; }
static float* sel_lv_level(Member& m) { return m.lv_level; }
static float* sel_lv_when(Member& m) { return m.lv_when; }
static float* sel_lv_pmnew(Member& m) { return m.lv_pmnew; }

// iisph_pressure_iterations (simulation.rs:1377-1516).  Iteration 0 was folded into the source-term sweep (closed form, see
// OpSource).  An iteration is two launches: sweep B(k) (Jacobi update + per-block residual partials) and sweep A(k + 1), whose
// block 0 first reduces those partials and takes the stop decision of iteration k while the other blocks already compute a^p
// from the new pressures -- if the decision is "stop", that launch WAS the solve's last pressure-acceleration sweep
// (simulation.rs:1499-1509) and everything queued behind it returns at once.  Iterations are queued speculatively up to the
// predicted count, followed by the solve's tail (the integrate map of the solver mode, k_solver_tail), which runs only once the
// decision is taken.  One host wait per chunk.
// Slab decomposition: block 0 of A(k + 1) adds up the rank's totals; the all-reduce and the decision (k_solver_decide) follow
// behind that sweep, i.e. the decision on iteration k is taken one sweep late and the reduction kernel between B and A is gone.
// The queueing state of one solve.
struct SolveQ {
    float max_avg_error;
    int residual_density;
    uint32_t max_iters;
    int tail;
    bool density_solver;
    int tot_slot = 0;   // slab decomposition: which totals buffer the solve all-reduces (1: the gated solve of a chained pair)
    uint32_t k = 1, upto = 2;
    void extend() { upto = k + std::max(1u, k / 4u); }   // a quarter more iterations per wait, at least one: a skipped
                                                         // iteration costs two empty launches (~9 us), a wait the round trip to the host
                                                         // (measured on the driver window, 5 runs each: k/4 1.279-1.295 ms per step, k/2 1.284-1.292, k 1.300-1.311)
};
// sweep A of iteration k: a^p from pressure buffer k & 1, and the stop decision of iteration k - 1
static int solve_sweep_a(Group& G, std::vector<Member>& M, const SolveQ& q, uint32_t k)
{
    const int multi = G.multi() ? 1 : 0;
    for (auto& m : M) {
        (void)hipSetDevice(m.c->device);
        if (m.n) launch_pressure_accel(m.c->stream, &m.c->prof, m.a, (int)k, q.residual_density, q.max_avg_error, q.max_iters, multi);
        else if (!multi) {   // a context without particles: nothing to iterate on (n_normal == 0)
            SolverCtrl z{};
            z.done = 1u;
            z.slot_done[0] = z.slot_done[1] = 1u;
            z.cur = k & 1u;
            (void)hipMemcpyAsync(m.c->ctrl.p, &z, sizeof z, hipMemcpyHostToDevice, m.c->stream);
        } else (void)hipMemsetAsync(m.c->dist.solver_tot.as<double>() + 8 * q.tot_slot, 0, 48, m.c->stream);   // an empty slab contributes zeros
    }
    // slab decomposition: the totals of iteration k - 1 are all-reduced behind this sweep; the decision is evaluated by the
    // blocks of B(k) themselves (OpJacobi::prologue), or by k_solver_decide before a solve's tail
    if (multi) return G.comm->allreduce_solver(G, q.tot_slot);
    return SPH_OK;
}
// Slab decomposition with neighbours: sweep A in two launches, so that the iteration's communication runs under compute.  The
// interior -- every lane without a ghost in reach, most of the slab -- is swept on the context's SIDE stream as soon as sweep B is
// done; the main stream meanwhile packs the halo members' p / rho^2, exchanges, unpacks, adds up this rank's totals of the previous
// iteration (k_solver_totals: what block 0 of the one-launch sweep does) and all-reduces them; then it waits for the interior and
// sweeps the halo members and the first ghost ring (a list launch, a few per cent of the slab).  The collectives stay on the main
// stream on purpose: a dependency between two streams costs ~9.5 us on this platform (scripts/ubench/xstream_hop.hip), and this
// way the two such hops (B -> interior, interior -> edge) sit beside the communication, not in its chain.  Same collectives in the
// same order as the one-launch form, same arithmetic per particle; every rank decides for itself (nothing collective depends on
// it): SPH_OVERLAP=0 never splits, =1 always (tests), default: slabs of at least SPLIT_MIN_PARTICLES.
#ifndef SPLIT_MIN_PARTICLES
#define SPLIT_MIN_PARTICLES 786432u   // the interior sweep (22 us per 1M particles) must outlast the two hops by a margin
#endif
static bool split_sweep_a(const Group& G, const std::vector<Member>& M)
{
    const int env = G.m[0]->dist.overlap_env;   // SPH_OVERLAP, read once per step (the tests switch it between two runs of one process)
    if (env == 0 || G.comm == nullptr || G.m[0]->dist.nranks <= 1 || G.m[0]->dist.xstream == nullptr) return false;
    if (env > 0) return true;
    uint32_t n_max = 0;
    for (auto& m : M) n_max = std::max(n_max, m.n);
    return n_max >= SPLIT_MIN_PARTICLES;
}
// the ghosts' p / rho^2 of pressure buffer ka & 1, sweep A(ka), and the all-reduce of iteration ka - 1's totals
static int exchange_and_sweep_a(Group& G, std::vector<Member>& M, const SolveQ& q, uint32_t ka)
{
    int rc;
    // the ghosts' p / rho^2 of this iteration's pressure buffer: a field of its own, or word 2 of the 16-byte pressure records the
    // solves of a uniform scene keep (sweep_a_on_records: parameters and all-reduced values, the same on every rank)
    const bool recs = sweep_a_on_records(M[0].a);
    float* (*sel)(Member&) = recs ? ((ka & 1u) ? sel_rec1 : sel_rec0) : ((ka & 1u) ? sel_pt1 : sel_pt0);
    const int stride = recs ? 4 : 1, off = recs ? 2 : 0;
    const TotalsJob job{(int)ka - 1, q.residual_density, q.max_avg_error, q.max_iters};   // this rank's totals of iteration ka - 1 (what block 0 of the single-rank sweep adds up)
    if (!split_sweep_a(G, M)) {
        if (!G.multi()) return solve_sweep_a(G, M, q, ka);
        // slab decomposition, one launch per sweep: this rank's totals of iteration ka - 1 first (k_solver_totals: what block 0 of
        // the single-rank sweep does), so that their all-reduce can travel WITH the ghosts' p / rho^2 -- one collective call per
        // iteration (one grouped RCCL launch) -- then the sweep with the reduced totals already in place
        for (auto& m : M) {
            sph_ctx* c = m.c;
            (void)hipSetDevice(c->device);
            if (!m.n) HIPCHK(c, hipMemsetAsync(c->dist.solver_tot.as<double>() + 8 * q.tot_slot, 0, 48, c->stream));   // an empty slab contributes zeros
        }
        if ((rc = refresh_ghosts(G, M, sel, 1, "pt", q.tot_slot, nullptr, stride, off, &job))) return rc;   // (block 0 of the pack launch adds up the totals)
        for (auto& m : M) {
            (void)hipSetDevice(m.c->device);
            if (m.n) launch_pressure_accel(m.c->stream, &m.c->prof, m.a, (int)ka, q.residual_density, q.max_avg_error, q.max_iters, 2);   // (block 0 takes the decision on iteration ka - 1 from the all-reduced totals)
            else launch_solver_progress(m.c->stream, &m.c->prof, m.a, (int)ka - 1, q.residual_density, q.max_avg_error, q.max_iters);   // an empty slab has no sweep to take it
        }
        return SPH_OK;
    }
    // The COLLECTIVES of the two forms are the same call in the same place -- the totals first, then ONE exchange-and-all-reduce --
    // because the split decision below is rank-local under a per-rank transport (each rank sees its own particle count): slab counts
    // that straddle SPLIT_MIN_PARTICLES must not leave one rank in a grouped send/receive and its neighbour in an all-reduce.
    for (auto& m : M) {   // the interior, beside everything below
        sph_ctx* c = m.c;
        auto& d = c->dist;
        (void)hipSetDevice(c->device);
        HIPCHK(c, hipEventRecord(d.ev_x[0], c->stream));
        HIPCHK(c, hipStreamWaitEvent(d.xstream, d.ev_x[0], 0));
        if (m.n) launch_pressure_accel(d.xstream, &c->prof, m.a, (int)ka, q.residual_density, q.max_avg_error, q.max_iters, 1, 1);
        HIPCHK(c, hipEventRecord(d.ev_x[1], d.xstream));
        if (!m.n) HIPCHK(c, hipMemsetAsync(c->dist.solver_tot.as<double>() + 8 * q.tot_slot, 0, 48, c->stream));   // an empty slab contributes zeros
    }
    if ((rc = refresh_ghosts(G, M, sel, 1, "pt", q.tot_slot, nullptr, stride, off, &job))) return rc;
    for (auto& m : M) {
        sph_ctx* c = m.c;
        (void)hipSetDevice(c->device);
        launch_solver_progress(c->stream, &c->prof, m.a, (int)ka - 1, q.residual_density, q.max_avg_error, q.max_iters);   // the decision on iteration ka - 1 (paced: the host learns of it while the interior still sweeps)
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->dist.ev_x[1], 0));
        if (m.n) launch_pressure_accel(c->stream, &c->prof, m.a, (int)ka, q.residual_density, q.max_avg_error, q.max_iters, 1, 2);
    }
    return SPH_OK;
}
static int solve_begin(Group& G, std::vector<Member>& M, SolveQ& q, uint32_t predicted_iters)
{
    int rc;
    // iteration 0 wrote pressure buffer 1
    if ((rc = exchange_and_sweep_a(G, M, q, 1))) return rc;
    q.k = 1;
    q.upto = predicted_iters > 2 ? predicted_iters : 2;
    return SPH_OK;
}
// iterations k .. upto, the (late) decision on the last one and the solve's tail
// `handoff`: the solve is the first of a chained pair -- its tail (or, where no tail runs, k_solver_handoff) hands the control block
// over and sets the second solve's gate.  `publish_next`: the caller waits right behind this (sync_ctrl): the header kernel of an
// integrating tail publishes as well.
static int solve_queue(Group& G, std::vector<Member>& M, SolveQ& q, bool handoff = false, bool publish_next = false)
{
    int rc;
    const int multi = G.multi() ? 1 : 0;
    for (; q.k <= q.upto && q.k <= q.max_iters; q.k++) {
        const uint32_t k = q.k;
        for (auto& m : M) {
            (void)hipSetDevice(m.c->device);
            if (m.n) launch_jacobi_update(m.c->stream, &m.c->prof, m.a, (int)k, q.residual_density, q.max_avg_error, q.max_iters, multi);
        }
        // (no exchange of a^p: the first ghost ring computes its own in sweep A)
        if ((rc = exchange_and_sweep_a(G, M, q, k + 1))) return rc;
    }
    // (slab decomposition: the decision on the last queued iteration was taken behind its all-reduce, like every other -- by the last
    //  sweep A's block 0, or the launch that stands in for it: exchange_and_sweep_a)
    for (auto& m : M) {
        (void)hipSetDevice(m.c->device);
        SolverCtrl* hand_host = handoff ? m.c->ctrl_host_dev + 1 : nullptr;
        uint32_t* gate_out = handoff ? (uint32_t*)(m.c->ctrl.as<SolverCtrl>() + 2) : nullptr;
        if (!m.n || q.tail == 0 /* TAIL_NONE */) {
            if (handoff) launch_solver_handoff(m.c->stream, &m.c->prof, m.c->ctrl.as<SolverCtrl>(), hand_host, gate_out);
            continue;
        }
        launch_solver_tail(m.c->stream, &m.c->prof, m.a, q.tail, m.c->pm[m.c->pcur ^ 1].as<float4>(), -1, q.residual_density,
                           q.max_avg_error, q.max_iters, hand_host, gate_out);
        if (q.tail == 1 /* TAIL_VEL */ && source_term_on_records(m.a)) m.a.xv_ok = true;   // (the tail rewrote the {x, y, v} record with the velocities it left)
        if (q.tail >= 2 /* TAIL_VX, TAIL_HYBRID */ && m.a.hdr_partials) launch_header_ahead(m.c, (m.n + 255u) / 256u, m.c->hdr_host_dev, publish_next && !multi);
    }
    return SPH_OK;
}
static void solve_stats(Member& m, const SolveQ& q, const SolverCtrl& h)
{
    m.c->pressure_cur = h.cur;
    sph_solver_stats* st = q.density_solver ? &m.st.density_solver : &m.st.div_solver;
    st->iters = h.iters;
    st->converged = 1;
    st->normal_count = h.normal;
    st->singular_count = h.singular;
    st->negative_count = h.negative;
    st->avg_error = h.normal > 0 ? h.sum_err / (float)h.normal : NAN;
    st->max_error = h.max_err;
}

// ---- paced solve (one context) ---------------------------------------------------------------------------------------------
// The iteration counts of a violent scene jump by factors from step to step (configs[1], steps 5-24: 7 3 6 11 17 22 21 18 6 4 15 ...):
// queueing "the previous step's count" either falls short (a host wait, then a quarter more) or overshoots (two empty launches of
// ~3.8 us per iteration not needed): together ~0.14 of the driver window's 1.2 ms per step.  Paced, the host needs no prediction to be
// right: block 0 of sweep A(k + 1) stores its decision on iteration k in mapped host memory when that sweep STARTS (SolveP::prog),
// and the host -- which runs far ahead of the device anyway -- queues iteration k + 1 only once fewer than `lead` undecided
// iterations are in the queue.  The sweep that took the decision still runs for its ~23 us per 1M particles while the next two
// launches travel, so the device never idles, and a solve ends with at most `lead` - 1 iterations queued in vain.  The first
// iterations (the smaller of the last two counts) are queued without looking: a cushion against a host thread that is late once.
// The tail is queued when the host has SEEN "stop": no gate, no re-queueing, and between the two solves of HybridDFSPH no wait.

// Small scenes: an iteration of n particles takes ~46 us x n / 2^20 on the device, the host's answer to a decision ~10-20 us: the
// lead grows as the sweeps shrink (1 from ~0.45M particles up), and the unpaced head is the smaller of the last two counts; large
// scenes queue nothing unpaced (measured on configs[1]'s driver window: 1.148 ms/step, against 1.164 with the head and 1.201-1.207
// with the predicted queue; profiles/r3_variants.md section 5).  SPH_PACE_LEAD / SPH_PACE_PRED override both.
static uint32_t pace_lead(const sph_ctx* c, uint32_t n)
{
    if (c->opt.pace_lead > 0) return (uint32_t)c->opt.pace_lead;
    return std::min(8u, std::max(1u, (450000u + n - 1u) / std::max(n, 1u)));
}
static uint32_t pace_prediction(const sph_ctx* c, uint32_t n, uint32_t last, uint32_t prev)
{
    const int mode = c->opt.pace_pred;   // 0: nothing unpaced; 1: min of the last two counts; 2: the last count
    const int md = mode == 0xffff ? (n >= 450000u ? 0 : 1) : mode;
    if (md == 0) return 2u;
    if (md == 2 || prev == 0u) return last;
    return std::min(last, prev);
}
// Slab decompositions pace the same way with lead 1 and no unpaced head beyond the two iterations every solve runs: block 0 of the
// sweep A behind the iteration's exchange-and-all-reduce publishes what the stop rule makes of the ALL-REDUCED totals
// (solver_publish_progress_multi) -- the same word on every rank -- and iteration k + 1 is queued only once iteration k is known to
// continue: every rank queues exactly the same launches and collectives, none in vain (the predicted queue this replaces re-sent the
// ghosts and all-reduced the totals of every over-predicted iteration, and waited for the host on every shortfall).
static int solve_paced(Group& G, std::vector<Member>& M, SolveQ& q, uint32_t predicted_iters)
{
    const bool multi = G.multi();
    size_t lead_member = 0;   // whose progress word the host watches: the first member with particles (all members publish the same decisions)
    while (lead_member + 1 < M.size() && M[lead_member].n == 0) lead_member++;
    Member& m = M[lead_member];
    sph_ctx* c = m.c;
    const uint32_t lead = multi ? 1u : pace_lead(c, m.n);
    int rc;
    // one epoch for the whole group (a member's word is only compared with the epoch its own launches carry)
    sph_ctx* c0 = M[0].c;
    c0->solve_epoch = c0->solve_epoch >= 0xffffu ? 1u : c0->solve_epoch + 1u;
    const uint32_t epoch = c0->solve_epoch;
    for (auto& mm : M) {
        mm.c->solve_epoch = epoch;
        mm.a.prog_host = mm.c->prog_host_dev;
        mm.a.prog_epoch = epoch;
    }
    if ((rc = solve_begin(G, M, q, multi ? 2u : predicted_iters))) return rc;
    auto iteration = [&](uint32_t k) -> int {
        for (auto& mm : M) {
            (void)hipSetDevice(mm.c->device);
            if (mm.n) launch_jacobi_update(mm.c->stream, &mm.c->prof, mm.a, (int)k, q.residual_density, q.max_avg_error, q.max_iters, multi ? 1 : 0);
        }
        return exchange_and_sweep_a(G, M, q, k + 1);
    };
    for (; q.k <= q.upto && q.k <= q.max_iters; q.k++)
        if ((rc = iteration(q.k))) return rc;
    auto t0 = std::chrono::steady_clock::now();
    int last_seen = -2;
    for (uint64_t spins = 0;; spins++) {
        const uint32_t w = *c->prog_host;
        const bool seen = (w >> 16) == epoch;
        if (seen && (w & 0x8000u)) break;
        const int decided = seen ? (int)(w & 0x7fffu) : -1;   // iterations 0 .. decided are decided; A(q.k) is queued and decides q.k - 1
        if (q.k <= q.max_iters && (int)q.k - 1 - decided < (int)lead) {
            if ((rc = iteration(q.k))) return rc;
            q.k++;
            continue;
        }
        if (decided != last_seen) {
            last_seen = decided;
            t0 = std::chrono::steady_clock::now();
        } else if ((spins & 0xffffu) == 0xffffu && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 2.0) {
            // no decision for 2 s: a faulted kernel (the event path reports it) -- or a decision that never comes
            if ((rc = wait_stream(c))) return rc;
            const uint32_t w2 = *c->prog_host;
            if ((w2 >> 16) == epoch && (w2 & 0x8000u)) break;
            if ((w2 >> 16) == epoch && (int)(w2 & 0x7fffu) != decided) continue;
            return c->fail(SPH_ERR_DEVICE, "pressure solve: the device finished its queue without a stop decision (iteration %u queued)", q.k - 1);
        }
    }
    q.upto = q.k - 1;   // solve_queue() has only the tail left to queue
    for (auto& mm : M) mm.a.prog_host = nullptr;
    return SPH_OK;
}

// iisph_pressure_iterations (simulation.rs:1377-1516) with a host wait of its own
static int pressure_iterations(Group& G, std::vector<Member>& M, float max_avg_error, int residual_density, uint32_t max_iters,
                               uint32_t predicted_iters, int tail, bool density_solver, bool final_solve,
                               const std::function<int()>* behind_tail = nullptr /* paced: queued behind the solve's tail, in front of the wait */)
{
    int rc;
    const int multi = G.multi() ? 1 : 0;
    SolveQ q{max_avg_error, residual_density, max_iters, tail, density_solver};
    // (a slab decomposition paces whether or not THIS rank has particles: every rank must queue the same collectives -- an empty
    //  slab's progress word is published by a launch of its own, launch_solver_progress)
    if ((multi || M[0].n > 0) && M[0].c->paced_step) {
        sph_ctx* c0 = M[0].c;
        const uint32_t head = density_solver ? pace_prediction(c0, M[0].n, c0->last_dens_iters, c0->prev_dens_iters) : pace_prediction(c0, M[0].n, c0->last_div_iters, c0->prev_div_iters);
        if ((rc = solve_paced(G, M, q, head))) return rc;
        if ((rc = solve_queue(G, M, q, false, true))) return rc;   // (the tail)
        if (behind_tail && (rc = (*behind_tail)())) return rc;
        if (multi && final_solve && (rc = G.comm->agree_guards_queued(G))) return rc;
        if ((rc = sync_ctrl(G, multi ? (final_solve ? SYNC_FINAL : SYNC_DEFER) : SYNC_AGREE))) return rc;
        for (auto& m : M) solve_stats(m, q, *m.c->ctrl_host);
        return SPH_OK;
    }
    if ((rc = solve_begin(G, M, q, predicted_iters))) return rc;
    for (;;) {
        if ((rc = solve_queue(G, M, q, false, true))) return rc;
        if (multi && final_solve && (rc = G.comm->agree_guards_queued(G))) return rc;
        if ((rc = sync_ctrl(G, multi ? (final_solve ? SYNC_FINAL : SYNC_DEFER) : SYNC_AGREE))) return rc;
        if (M[0].c->ctrl_host->done) break;
        if (q.k > max_iters) break;  // cannot happen: the decision on iteration max_iters is always "stop"
        q.extend();   // the prediction (the previous step's count) fell short
    }
    for (auto& m : M) solve_stats(m, q, *m.c->ctrl_host);
    return SPH_OK;
}

static int group_step_inner(Group& G, const sph_params* p, sph_step_stats* outs, bool* started)
{
    int rc = SPH_OK;
    g_trace.start(G.m[0]->opt.hip_trace != 0);
    std::vector<Member> M(G.m.size());
    for (size_t i = 0; i < G.m.size(); i++) {
        M[i].c = G.m[i];
        M[i].n = (uint32_t)G.m[i]->n;
        memset(&M[i].st, 0, sizeof(sph_step_stats));
        M[i].wall0 = std::chrono::steady_clock::now();
    }
    sph_ctx* c0 = G.m[0];
    // ---- parameter combinations this build does not cover are refused, never approximated ---------------
    if (c0->n_planes == 0) return c0->fail(SPH_ERR_NO_BOUNDARY, "not implemented: NoBoundaryHandler::iisph_aii");
    const bool h_from_mass_mode = p->support_length_estimation == SPH_H_FROM_MASS;
    const bool level_on = p->level_estimation_method != SPH_LEVEL_NONE;
    const bool level_after = level_on && p->level_estimation_after_advection;
    if (level_on && !level_after && p->level_estimation_method == SPH_LEVEL_CENTER_DIFF)   // simulation.rs:2029-2031
        return c0->fail(SPH_ERR_INVALID_ARGUMENT, "center diff level estimation method needs density values");
    if (!p->level_estimation_after_advection && !p->use_extended_range_for_level_estimation)
        return c0->fail(SPH_ERR_INVALID_ARGUMENT, "assertion failed: simulation_params.use_extended_range_for_level_estimation");
    if (!G.multi() && c0->n == 0) return c0->fail(SPH_ERR_INVALID_ARGUMENT, "called `Option::unwrap()` on a `None` value (no particles)");
    // Level estimation before advection builds the lists at k = level_estimation_range / ETA and filter_down(2) only REMOVES
    // entries (neighborhood_search.rs:56-70; simulation.rs:2018-2070 -- also with level_estimation_method None): below k = 2 the
    // whole step runs on the narrower k-range lists.  The sweeps here are built for the SPH support (k = 2): refuse, never widen.
    if (!p->level_estimation_after_advection && !(p->level_estimation_range / SPH_ETA >= 2.f))
        return c0->fail(SPH_ERR_UNSUPPORTED, "level_estimation_range / 1.9 = %g < 2: the reference then steps on lists narrower than the SPH support, which this build does not cover",
                        (double)(p->level_estimation_range / SPH_ETA));
    for (auto c : G.m)
        if (c->poisoned) return c->fail(SPH_ERR_POISONED, "an earlier step failed inside the step: the particle state is undefined until sph_upload");
    *started = true;   // from here on a failure leaves the state half-stepped
    // measurement / test switches of the solves, read ONCE per step (never inside the iteration path)
    const bool no_records = c0->opt.accel_generic != 0;   // sweep A through the generic form
    // (the progress word carries the iteration in 15 bits: a solve that may run longer keeps the predicted queue)
    // Slab decompositions pace too (Options::slab_paced; lead 1: nothing is ever queued in vain, so every rank queues the same
    // collectives) -- parameters only: the same on every rank
    const bool paced = c0->opt.paced != 0 && p->max_iters <= 0x7fffu && (!G.multi() || c0->opt.slab_paced != 0);   // 0: predicted queue + waits
    for (auto c : G.m) c->paced_step = paced;
    for (auto c : G.m) c->dist.overlap_env = c0->opt.overlap;
    for (auto c : G.m) c->publish_folded = false;

    // ---- slab maintenance part 1 needs no global scalar: partition + migrate (multi-rank) -----------------
    // (the ghost layer needs the all-reduced h_max, so the header of the owned particles comes first)
    const bool tev = c0->prof.mode == 1;
    for (auto& m : M) {
        (void)hipSetDevice(m.c->device);
        if (tev) (void)hipEventRecord(m.c->ev[0], m.c->stream);
    }

    // ---- step header (h from mass, simulation.rs:1998-2003; CFL term; h_max; bounding box) and, on slabs, the hand-over of the
    // particles that left their slab -------------------------------------------------------------------------------------
    // per rank: -h_max, h_min, CFL term, -status, and the bounding box of the owned particles (min x, -max x, min y, -max y): all min-reduced
    std::vector<std::vector<float>> red(M.size(), std::vector<float>(8));
    int hdr_rc = SPH_OK, setup_rc = SPH_OK;
    bool slab_fused = false;   // the ghost layer is already in place (slab_refresh_fused)
    // Ghost width in smoothing lengths of the largest particle: two rings of one support radius (2 h_max each), or the extended
    // range of the level estimation.  Level estimation AFTER advection looks for neighbours at the advected positions among the
    // ghosts selected BEFORE the step: the layer is widened by twice the largest displacement the step may produce -- 4 x the
    // previous step's (all-reduced) largest displacement, at least 2 h_max -- and the step checks afterwards that it sufficed.
    const float slab_slack_k = (level_on && p->level_estimation_after_advection && G.multi())
                                   ? fmaxf(2.f, c0->h_max_step > 0.f ? 4.f * c0->last_dmax / c0->h_max_step : 0.f) : 0.f;
    const float halo_base_k = fmaxf(4.f, level_on ? p->level_estimation_range / SPH_ETA : 0.f);
    // The tail of the previous step's last solve already reduced this step's header into hdr_host (k_solver_tail,
    // k_header_ahead): nothing to launch, nothing to wait for -- unless the host touched the state or the smoothing lengths are
    // not the mass-derived ones.  On a slab the header describes the particles the rank owned at the END of that step; the ones
    // about to migrate are some other rank's soon, but every value below is reduced over ALL ranks (or replaced by the cuts).
    auto header_ready = [&](sph_ctx* c) { return h_from_mass_mode && c->hdr_ahead && c->hdr_ahead_rest_density == p->rest_density; };
    const bool ahead_usable = !G.multi() && header_ready(c0);   // (every call that touches the state clears hdr_ahead)
    auto fill_red = [&]() {
        for (size_t i = 0; i < M.size(); i++) {
            const HeaderOut h = *M[i].c->hdr_host;
            const bool any = M[i].c->n > 0;
            red[i][0] = any ? -h.h_max : 0.f;
            red[i][1] = any ? h.h_min : INFINITY;
            red[i][2] = any ? h.min_cfl : INFINITY;
            red[i][3] = -(float)hdr_rc;   // the agreement on this wait point rides in the same all-reduce (min of -status)
            red[i][4] = any ? h.min_x : INFINITY;
            red[i][5] = any ? -h.max_x : INFINITY;
            red[i][6] = any ? h.min_y : INFINITY;
            red[i][7] = any ? -h.max_y : INFINITY;
            M[i].c->hdr_ahead = false;
        }
    };
    if (G.multi()) {
        bool rebalanced = false;
        const int every = c0->dist.rebalance_every;
        if (every > 0 && c0->step_number > 0 && c0->step_number % (uint64_t)every == 0 && c0->h_max_step > 0.f)
            if ((rc = rebalance_cuts(G, M, &rebalanced))) return rc;
        // header of the particles each rank owns NOW (before the hand-over: the global values do not care who owns a particle)
        bool launched = false;
        for (auto& m : M) {
            sph_ctx* c = m.c;
            if (header_ready(c)) continue;
            (void)hipSetDevice(c->device);
            const uint32_t n_prev = c->dist.have_flags ? c->dist.n_tot : (uint32_t)c->n;
            launch_header(c, n_prev, p->rest_density, h_from_mass_mode ? 1 : 2, c->hdr_host_dev, c->dist.have_flags ? c->dist.owned.as<uint8_t>() : nullptr);
            launched = true;
        }
        if (launched) hdr_rc = wait_all_hinted(G);   // (first step, after uploads, FromDistribution*: the values must be on the host before the all-reduce)
        fill_red();
        // particles follow the cuts to the x-neighbour; after a re-balance a particle may have to cross several slabs:
        // repeat until nobody moved (the all-reduced count), at most once per rank.  The header all-reduce rides in the round
        // trip of the first partition's counts.
        // ordinary steps: the fused refresh (one round trip, no partition sort); it reduces `red` whether or not it applies
        const bool no_fused = c0->opt.slab_general != 0;   // measurement / test aid: always the general path
        const bool attempt = h_from_mass_mode && !rebalanced && !no_fused;     // (parameters and all-reduced values: the same on every rank)
        if (attempt) {
            if ((rc = slab_refresh_fused(G, M, red, halo_base_k, slab_slack_k, &slab_fused))) return rc;
            if (!slab_fused && (hdr_rc || red[0][3] < 0.f)) {
                if (hdr_rc) return hdr_rc;
                return c0->fail((int)-red[0][3], "another rank of the slab decomposition reported status %d", (int)-red[0][3]);
            }
        }
        std::vector<int> moved(M.size(), 0);
        for (int round = 0; !slab_fused; round++) {
            if ((rc = partition_and_migrate(G, M, &moved, round == 0 && !attempt ? &red : nullptr))) return rc;
            // one hand-over moves a particle to the x-neighbour; one that is further past the cut than the narrowest slab (after a
            // re-balance; or because it crossed a slab in one step, which the fused refresh answers with this path) may need another:
            // until no rank has such a migrant (the all-reduced flag)
            if ((rc = G.comm->allreduce_max_i32(G, moved))) return rc;
            if (moved[0] == 0 || round + 1 >= c0->dist.nranks) break;
        }
    } else {
        if (!header_ready(c0)) {
            launch_header(c0, (uint32_t)c0->n, p->rest_density, h_from_mass_mode ? 1 : 2, c0->hdr_host_dev);
            if ((hdr_rc = wait_all_hinted(G))) return hdr_rc;
        }
        fill_red();
    }
    g_trace.mark(0);
    if (hdr_rc) return hdr_rc;
    if (red[0][3] < 0.f)
        return c0->fail((int)-red[0][3], "another rank of the slab decomposition reported status %d", (int)-red[0][3]);
    const float h_max_g = -red[0][0], h_min_g = red[0][1], min_cfl_g = red[0][2];
    // (identical on every rank: all-reduced values only)
    if (!(h_max_g > 0.f) || !std::isfinite(h_max_g))
        return c0->fail(SPH_ERR_POSITION_NOT_FINITE, "particle positions or smoothing lengths are not finite");
    const HeaderOut gbox{red[0][4], red[0][6], -red[0][5], -red[0][7], h_max_g, h_min_g, min_cfl_g, 0};   // global bounding box
    // CFL (simulation.rs:2190-2191)
    const float cfl_dt = p->cfl_factor * sqrtf(min_cfl_g);
    const float dt = fminf(p->max_dt, cfl_dt);

    std::vector<HeaderOut> boxes(M.size(), gbox);   // bounding box of what each member sorts: owned + ghosts
    if (G.multi()) {
        // Checks every rank evaluates on the SAME all-reduced numbers (no agreement needed): the cell grid of the global box
        // bounds every rank's own grid.
        int status_in = SPH_OK;
        if (!std::isfinite(gbox.min_x) || !std::isfinite(gbox.max_x) || !std::isfinite(gbox.min_y) || !std::isfinite(gbox.max_y))
            return c0->fail(SPH_ERR_POSITION_NOT_FINITE, "particle positions are not finite");
        {
            const float cs = h_max_g * 2.f;
            const long long gx = (long long)floorf(gbox.max_x / cs) - (long long)floorf(gbox.min_x / cs) + 3, gy = (long long)floorf(gbox.max_y / cs) - (long long)floorf(gbox.min_y / cs) + 3;
            if (gx <= 0 || gy <= 0 || gx >= 65536 || gy >= 65536 || gx * gy >= (1ll << 27))
                return c0->fail(SPH_ERR_UNSUPPORTED, "cell grid of cell size %g is too large for this build", (double)cs);
        }
        // ghost layer with the real width: one support radius of the largest particle anywhere
        // (the extended lists of the level estimation reach level_estimation_range / ETA smoothing lengths)
        // Two rings of one support radius (2 h_max) each: ghosts of the first ring compute their pressure acceleration here
        // (their neighbours are all inside the second), so a Jacobi iteration exchanges p / rho^2 only.
        for (auto& m : M) m.c->slab_slack_w = slab_slack_k * h_max_g;
        if (!slab_fused && (rc = build_ghost_layer(G, M, halo_base_k, slab_slack_k * h_max_g, h_max_g, status_in))) return rc;
        // bounding box of owned + ghosts from the cuts: an owned particle lies between them, a ghost within the cut's layer width beyond one
        for (size_t i = 0; i < M.size(); i++) {
            const auto& d = M[i].c->dist;
            if (d.rank > 0) boxes[i].min_x = fmaxf(gbox.min_x, d.cut_lo - d.halo_w[0]);
            if (d.rank + 1 < d.nranks) boxes[i].max_x = fminf(gbox.max_x, d.cut_hi + d.halo_w[1]);
            if (!(boxes[i].max_x >= boxes[i].min_x)) boxes[i].max_x = boxes[i].min_x;   // an empty slab beyond the fluid
        }
    }
    g_trace.mark(1);

    auto setup_member = [&](Member& m) -> int {
        sph_ctx* c = m.c;
        (void)hipSetDevice(c->device);
        const HeaderOut hdr = boxes[(size_t)(&m - M.data())];
        const uint32_t n = m.n;
        // CellGrid (neighborhood_search.rs:261-275) with cell = support radius of the largest particle: the grid the
        // reference's convention defines (sph_grid, cell_index).  Uniform scenes sort by it.  Multi-resolution scenes sort
        // by a finer grid `fg` (cell = support of the smallest particle, doubled until the table fits) and give every
        // particle its own stencil width (TileP, sph_device.h), so a fine particle far from any coarse one still looks
        // at 3 x 3 small cells instead of 3 x 3 large ones.
        auto make_grid = [&](float cs, GridP& out) -> bool {
            out = GridP{};
            out.cs = cs;
            if (!n) {
                out.sx = out.sy = 1;
                out.ncells = 1;
                return true;
            }
            out.minx = (int)floorf(hdr.min_x / cs) - 1;
            out.miny = (int)floorf(hdr.min_y / cs) - 1;
            const long long sx = (long long)((int)floorf(hdr.max_x / cs) + 2) - out.minx;
            const long long sy = (long long)((int)floorf(hdr.max_y / cs) + 2) - out.miny;
            if (sx <= 0 || sy <= 0 || sx >= 65536 || sy >= 65536 || sx * sy >= (1ll << 27)) return false;
            out.sx = (int)sx;
            out.sy = (int)sy;
            out.ncells = (uint32_t)sx * (uint32_t)sy;
            return true;
        };
        if (n && (!std::isfinite(hdr.min_x) || !std::isfinite(hdr.max_x) || !std::isfinite(hdr.min_y) || !std::isfinite(hdr.max_y)))
            return c->fail(SPH_ERR_POSITION_NOT_FINITE, "particle positions are not finite");
        GridP g{};
        const bool coarse_ok = make_grid(h_max_g * 2.f, g);
        // (constrain_neighborhood_count changes individual smoothing lengths after the lists are built)
        c->uniform_h = (h_min_g == h_max_g) && !p->constrain_neighborhood_count;
        c->h_uniform = h_max_g;
        c->h_max_step = h_max_g;
        GridP fg = g;
        c->tile_ts = 0;
        // (a narrow h distribution -- FromDistribution* support lengths wander by a few percent -- keeps the one-cell stencil
        //  of the coarse grid: the fine grid only pays once 3 x 3 coarse cells hold several times the needed candidates)
        if (!c->uniform_h && h_max_g >= 1.75f * h_min_g) {
            float cs = h_min_g * 2.f;
            bool ok = false;
            for (int k = 0; k < 24 && cs < g.cs; k++, cs *= 2.f)
                if ((ok = make_grid(cs, fg))) break;
            if (ok) {
                int ts = (int)ceilf(g.cs / fg.cs);
                while ((float)ts * fg.cs < g.cs) ts++;
                c->tile_ts = ts;
                c->tile_tsx = (fg.sx + ts - 1) / ts;
                c->tile_tsy = (fg.sy + ts - 1) / ts;
            } else {
                fg = g;   // the finest grid that fits is the coarse one: a one-cell tile, 3 x 3 stencils
                if (coarse_ok) {
                    c->tile_ts = 1;
                    c->tile_tsx = fg.sx;
                    c->tile_tsy = fg.sy;
                }
            }
        }
        if (!coarse_ok)
            return c->fail(SPH_ERR_UNSUPPORTED, "cell grid of cell size %g is too large for this build", (double)g.cs);
        // (what the last build left behind -- the grid the arrays are sorted by, their cells, the cell ranges -- if nothing touched the
        //  state since: a slab rank's sort below is then a merge)
        const GridP prev_grid = c->fgrid;
        const bool prev_valid = c->grid_valid;
        c->grid = g;
        c->fgrid = fg;
        c->grid_valid = true;
        g = fg;   // everything below (keys, sort, cell ranges, sweeps) works on the sorting grid

        StepP sp{};
        sp.rest_density = p->rest_density;
        sp.viscosity = p->viscosity;
        sp.gravity = p->gravity;
        sp.jacobi_omega = p->jacobi_omega;
        sp.dt = dt;
        sp.sdf_eps = p->sdf_gradient_eps;
        sp.pull_x = p->pull_fluid_to[0];
        sp.pull_y = p->pull_fluid_to[1];
        sp.hyb_vfactor = fminf(dt * p->hybrid_dfsph_factor, 1.f);
        sp.viscosity_type = p->viscosity_type;
        sp.penalty = p->boundary_penalty_term;
        sp.opdisc = p->operator_discretization;
        sp.has_pull = p->has_pull_fluid_to;
        sp.n_planes = c->n_planes;
        m.sp = sp;

        // ---- neighbourhood: cell index -> radix sort -> reorder -> cell ranges ---------------------------
        // (replaces build_neighborhood_list + filter_down, simulation.rs:2018-2070; same neighbour set)
        hipStream_t s = c->stream;
        Profiler* prof = &c->prof;
        int k = c->cur;
        HIPCHK(c, c->cs_scratch.ensure(cell_start_scratch_bytes()));
        // (fused slab refresh: the arrays still hold the slots that left this rank -- they sort behind the last cell and stay there)
        const bool pre = c->dist.on && c->dist.pre;
        const uint32_t n_sort = pre ? m.n_sort : n;
        c->dist.pre = false;
        // A slab rank behind its fused refresh: slots [0, n_prev) are last step's sorted array (some of them left the rank), the
        // arrivals follow unsorted -- the stable sort is the merge of sph_sort.hip (incremental_cell_sort_perm) with the arrivals as
        // movers, if they and the particles that changed cell last time are few; same keys, permutation and cell ranges as the radix sort.
        const uint32_t n_prev = pre ? c->dist.pre_cls_n : 0u;
        uint32_t* movers_host = (uint32_t*)(c->ctrl_host + 2) + 1;
        bool merge = pre && c->opt.inc_sort && prev_valid && prev_grid.cs == g.cs && prev_grid.ncells > 0 && n_prev > 0 && n_prev <= n_sort && !c->exact &&
                     g.ncells <= n_sort + 4096u;   // (k_inc_scan adds up the preceding block sums per block: quadratic in ncells / 1024 -- a sparse grid takes the radix sort; advisor r4)
        if (merge) {
            const uint32_t limit = n_sort / (c->opt.inc_sort > 1 ? (uint32_t)c->opt.inc_sort : 3u);
            if (n_sort - n_prev > limit) merge = false;
            else if (c->inc_count_valid && *movers_host > limit) {
                merge = ++c->inc_radix_streak >= 8;
                if (merge) c->inc_radix_streak = 0;
            }
        }
        if (!merge) HIPCHK(c, c->cell_start.ensure(((size_t)g.ncells + 1) * sizeof(uint32_t)));
        // the build the previous step queued ahead (queue_ahead_build): adopted if nothing touched the state since (the header that
        // step left behind was still the one in force), the scene is what it predicted and the real bounding box fits its grid
        const GridP ag = c->ahead.g;
        // (a multi-resolution scene: the same smoothing lengths, hence the same cell size and tile side; its tiles then lie on the
        //  predicted grid -- another tiling bounds the neighbours' h as well, and the visiting order of a list does not depend on the
        //  stencil width it was gathered with)
        const bool adopt = ahead_usable && c->ahead.valid && !c->dist.on && c->ahead.n == c->n && n == (uint32_t)c->n && c->ahead.h_max == h_max_g &&
                           c->ahead.h_min == h_min_g && c->ahead.rest_density == p->rest_density && (c->uniform_h ? c->tile_ts == 0 : c->tile_ts > 0) &&
                           c->ahead.tile_ts == c->tile_ts && !c->exact && ag.cs == g.cs && g.minx >= ag.minx && g.miny >= ag.miny &&
                           g.minx + g.sx <= ag.minx + ag.sx && g.miny + g.sy <= ag.miny + ag.sy;
        c->ahead.valid = false;
        if (adopt) {
            if (c->tile_ts > 0) {
                c->tile_tsx = c->ahead.tile_tsx;
                c->tile_tsy = c->ahead.tile_tsy;
                std::swap(c->tile_raw, c->atile_raw);
                std::swap(c->tile_h, c->atile_h);
            }
            for (int q = 0; q < 2; q++) {
                std::swap(c->key[q], c->akey[q]);
                std::swap(c->val[q], c->aval[q]);
            }
            std::swap(c->cxy, c->acxy);
            std::swap(c->cell_start, c->acell_start);
            std::swap(c->pm[c->pcur ^ 1], c->pm2);
            c->cur = k ^ 1;
            c->pcur ^= 1;
            c->fgrid = ag;
            g = ag;
        } else if (n_sort) {
            // (the keys -- cell index, or one past the last cell for a slot that left -- are made by the sort's first pass)
            const CellKeyGen kg{c->pm[c->pcur].as<float4>(), g, pre ? c->dist.cls.as<uint8_t>() : nullptr, pre ? c->dist.pre_cls_n : 0u, (uint32_t)SC_GONE_FROM};
            if (merge) {
                const size_t head_before = c->inc_head.bytes;
                HIPCHK(c, c->inc_head.ensure((size_t)g.ncells * 8));
                if (c->inc_head.bytes != head_before) HIPCHK(c, hipMemsetAsync(c->inc_head.p, 0, c->inc_head.bytes, s));   // (epoch 0: no list)
                HIPCHK(c, c->inc_next.ensure((size_t)c->cap * 4));
                HIPCHK(c, c->inc_bsum.ensure(incremental_sort_block_sums(g.ncells) * 4));
                if (!c->inc_movers.p) {
                    HIPCHK(c, c->inc_movers.ensure(4));
                    HIPCHK(c, hipMemsetAsync(c->inc_movers.p, 0, 4, s));
                }
                HIPCHK(c, c->acell_start.ensure(((size_t)g.ncells + 1) * sizeof(uint32_t)));
                if (++c->inc_epoch == 0u) c->inc_epoch = 1u;
                const IncClassifyP q{prev_grid, g, c->cxy.as<uint32_t>(), c->key[1].as<uint32_t>(), c->val[1].as<uint8_t>(), c->inc_next.as<uint32_t>(),
                                     c->inc_head.as<unsigned long long>(), c->inc_epoch};
                incremental_cell_sort_perm(s, prof, n_sort, n_prev, kg, q, c->cell_start.as<uint32_t>(), c->key[0].as<uint32_t>(), c->val[0].as<uint32_t>(),
                                           c->acell_start.as<uint32_t>(), c->inc_bsum.as<uint32_t>(), c->inc_movers.as<uint32_t>(),
                                           (uint32_t*)(c->ctrl_host_dev + 2) + 1);
                c->inc_count_valid = true;   // (a count of the current state is on its way, behind any stale one on the same stream)
                std::swap(c->cell_start, c->acell_start);   // (the old table was read while the new one was written)
            } else {
                int res = radix_sort_pairs(s, prof, c->key[0].as<uint32_t>(), c->val[0].as<uint32_t>(), c->key[1].as<uint32_t>(),
                                           c->val[1].as<uint32_t>(), n_sort, ilog2_ceil(g.ncells + (pre ? 1u : 0u)), c->sort_scratch.as<uint32_t>(), &kg);
                if (res == 1) {  // keep the sorted keys in key[0] / val[0]
                    std::swap(c->key[0], c->key[1]);
                    std::swap(c->val[0], c->val[1]);
                }
            }
            if (n)
                launch_reorder(s, prof, n, g, c->key[0].as<uint32_t>(), c->val[0].as<uint32_t>(), c->pm[c->pcur].as<float4>(), c->vel[k].as<float2>(),
                               c->orig[k].as<uint32_t>(), c->lvl[k].as<float>(), c->lvlold[k].as<float>(), c->pm[c->pcur ^ 1].as<float4>(),
                               c->vel[k ^ 1].as<float2>(), c->orig[k ^ 1].as<uint32_t>(), c->lvl[k ^ 1].as<float>(), c->lvlold[k ^ 1].as<float>(),
                               c->cxy.as<uint32_t>(), c->h2n[k].as<float>(), c->h2n[k ^ 1].as<float>(), c->lam_sum.as<float>(),
                               c->lam_prev.as<float>(), c->cs_scratch.p, c->szc[k].as<uint8_t>(), c->szc[k ^ 1].as<uint8_t>());
            c->cur = k ^ 1;
            c->pcur ^= 1;
        }
        if (!adopt && !(merge && n_sort)) launch_cell_start(s, prof, c->key[0].as<uint32_t>(), n, g.ncells, c->cell_start.as<uint32_t>(), c->cs_scratch.p, n > 0);
        if (c->tile_ts > 0) {
            const size_t nt = (size_t)c->tile_tsx * (size_t)c->tile_tsy;
            // the extended-range lists reach k * h_max with k = level_estimation_range / ETA > 2: a larger particle may sit
            // ceil(k / 2) tiles away (tile side >= 2 h_max)
            const bool want_ext = p->level_estimation_method != SPH_LEVEL_NONE;
            const int d_ext = want_ext ? (int)ceilf(fmaxf(p->level_estimation_range / SPH_ETA, 2.f) * 0.5f) : 1;
            if (want_ext) HIPCHK(c, c->tile_h_ext.ensure(nt * 4));
            if (!adopt) {
                HIPCHK(c, c->tile_raw.ensure(nt * 4));
                HIPCHK(c, c->tile_h.ensure(nt * 4));
                launch_tile_hmax(s, prof, n, c->pm[c->pcur].as<float4>(), g, c->tile_ts, c->tile_tsx, c->tile_tsy, c->tile_raw.as<uint32_t>(),
                                 c->tile_h.as<uint32_t>(), want_ext ? c->tile_h_ext.as<uint32_t>() : nullptr, d_ext);
            } else if (want_ext) {   // (adopted: the build queued ahead made the tiles' bounds; it never runs with the level estimation on)
                launch_tile_redilate(s, prof, c->tile_tsx, c->tile_tsy, d_ext, c->tile_raw.as<uint32_t>(), c->tile_h_ext.as<uint32_t>());
            }
        }
        if (c->exact || !c->uniform_h || sweep_forces_index_lists()) HIPCHK(c, c->nlx.ensure(sweep_index_list_bytes(n ? n : 1)));
        if (c->dist.on) {
            const int mrc = slab_maps_after_sort(c, n, pre, s);
            if (mrc) return mrc;
        }
        if (tev) (void)hipEventRecord(c->ev[1], s);
        m.a = make_args(c, sp);
        if (p->pressure_solver_method == SPH_SOLVER_IISPH2 || no_records) m.a.rec0 = m.a.rec1 = m.a.xv = nullptr;   // (IISPH2's rescaling works on p and p / rho^2)
        m.a.h_mode = p->support_length_estimation;
        m.a.sp_check_aii = p->check_aii;
        m.st.n_particles = c->n;
        m.st.dt = dt;
        return SPH_OK;
    };
    for (auto& m : M)
        if (!setup_rc) setup_rc = setup_member(m);
    if (setup_rc) return setup_rc;
    for (auto& m : M) dbg_sync(m.c, "neighbourhood (sort, reorder, cell ranges, maps)", 3);   // (slabs: the grid checks were taken on all-reduced numbers above, identically on every rank)
    g_trace.mark(2);

    // ---- level estimation on the extended-range lists (simulation.rs:2018-2046, 862-927; after advection: 2678-2707) ------
    LevelArgs lv{};
    SweepArgs lv_args{};   // the arguments of a propagation queued on the side stream (level_estimation_finish)
    float lv_slack = 0.f;
    // `pm_old` == nullptr: the positions the particles are sorted by (before advection).  Else: `al.pm` holds the ADVECTED
    // positions, pm_old the pre-step ones, and the extended lists are gathered from the cells of the pre-step positions with
    // every search range widened by 2 x the largest displacement (TileP::slack).
    // `side`: detection and propagation are queued on the context's SECOND stream and the call returns without waiting
    // (level estimation BEFORE advection: nothing of it is read until the smoothing at the end of the step, so the ~100
    // latency-bound propagation sweeps run under the step's own sweeps); level_estimation_finish() collects it.
    struct LevelPending {
        bool on = false;
        uint32_t t = 1, effective = 0;
        int B = 8;
    } lvp;
    auto level_estimation = [&](const float4* pm_geo, const float4* pm_old, bool side) -> int {
        const auto t_lvl0 = std::chrono::steady_clock::now();
        Member& m = M[0];
        sph_ctx* c = m.c;
        hipStream_t ls = side ? c->stream2 : c->stream;
        if (side) {   // everything queued so far (sort, cell ranges, tile bounds) comes first
            HIPCHK(c, hipEventRecord(c->ev_fork, c->stream));
            HIPCHK(c, hipStreamWaitEvent(c->stream2, c->ev_fork, 0));
        }
        const size_t n = m.n ? m.n : 1;
        HIPCHK(c, c->lvl_tmp.ensure((c->cap ? c->cap : 1) * 4));   // swapped with lvl[cur] at the end of the step: a persistent array, capacity-sized
        HIPCHK(c, c->lvl_nrm.ensure(n * 8));
        HIPCHK(c, c->lvl_when.ensure(n * 4));
        HIPCHK(c, c->lvl_mark.ensure(n * 8));   // two parity buffers (OpLevelPropagate)
        HIPCHK(c, c->stash.ensure(n * 4));
        for (DevBuf* b : {&c->lvl_state, &c->flag_surface, &c->flag_insufficient}) HIPCHK(c, b->ensure(n));
        HIPCHK(c, c->nl_ext.ensure(sweep_list_bytes((uint32_t)n)));
        HIPCHK(c, c->nlx_ext.ensure(sweep_index_list_bytes((uint32_t)n)));
        HIPCHK(c, c->lvl_changed_d.ensure(1024 * sizeof(uint32_t)));   // [0, 1000): flags of a batch; 1022, 1023: see below
        uint32_t* chg = c->lvl_changed_d.as<uint32_t>();
        if (!pm_old) {
            m.a = make_args(c, m.sp);
            m.a.h_mode = p->support_length_estimation;
            m.a.sp_check_aii = p->check_aii;
        }
        m.a.nl_ext = c->nl_ext.as<uint4>();   // (allocated just above when this is the first level estimation of the context)
        m.a.nlx_ext = c->nlx_ext.as<uint4>();
        lv.k = p->level_estimation_range / SPH_ETA;                    // simulation.rs:2036
        lv.threshold = cosf(50.f * (SPH_PI_F / 180.f));                // simulation.rs:544
        lv.max_surface_distance = p->maximum_surface_distance;
        lv.boundary_is_fluid_surface = p->boundary_is_fluid_surface;
        lv.maximum_range = (p->support_length_estimation == SPH_H_FROM_DISTRIBUTION || p->support_length_estimation == SPH_H_FROM_DISTRIBUTION2)
                               ? p->maximum_range : -1.f;   // simulation.rs:705-721
        lv.nrm = c->lvl_nrm.as<float2>();
        lv.state = c->lvl_state.as<uint8_t>();
        lv.flag_surface = c->flag_surface.as<uint8_t>();
        lv.flag_insufficient = c->flag_insufficient.as<uint8_t>();
        lv.size_class = c->szc[c->cur].as<uint8_t>();
        lv.level = c->lvl[c->cur].as<float>();
        lv.when = c->lvl_when.as<uint32_t>();
        lv.mark = c->lvl_mark.as<uint32_t>();
        lv.level_old = c->lvlold[c->cur].as<float>();
        lv.stash_first = p->fill_stash_with == SPH_STASH_SURFACE_DISTANCE_FIRST ? c->stash.as<float>() : nullptr;
        lv.pm_cell = pm_old;
        lv.center_diff = p->level_estimation_method == SPH_LEVEL_CENTER_DIFF;
        lv.replay_step_lists = pm_old != nullptr && !p->use_extended_range_for_level_estimation;   // simulation.rs:2680: no rebuild
        // the propagation on a compacted frontier (explicit index lists: the extended-range lists; the step's own lists of a uniform
        // scene are mask words, which the sweep forms replay)
        lv.fmap = nullptr;
        if (c->opt.level_queue && !lv.replay_step_lists) {
            uint32_t lg = 0;
            while ((64ull << lg) < n) lg++;
            HIPCHK(c, c->lvl_queue.ensure((size_t)128 << lg));
            lv.fmap = c->lvl_queue.as<uint8_t>();
            lv.fmap_lg_s = lg;
        }
        if (lv.replay_step_lists) {
            m.a.nl_ext = m.a.nl;
            m.a.nlx_ext = m.a.nlx;
        }
        SweepArgs al = m.a;
        lv_slack = 0.f;
        if (pm_old && m.n && lv.replay_step_lists) al.pm = pm_geo;
        if (pm_old && m.n && !lv.replay_step_lists) {
            al.pm = pm_geo;
            launch_max_disp(ls, &c->prof, m.n, pm_old, pm_geo, chg + 1022);
            HIPCHK(c, hipMemcpyAsync(c->lvl_changed, chg + 1022, sizeof(uint32_t), hipMemcpyDeviceToHost, ls));
            if ((rc = wait_stream(c))) return rc;
            float dmax;
            memcpy(&dmax, (const void*)c->lvl_changed, 4);
            if (!std::isfinite(dmax)) return c->fail(SPH_ERR_POSITION_NOT_FINITE, "Assertion 'p_position[d].is_finite()' failed!");
            lv_slack = 2.f * dmax;
            al.t_ext.slack = lv_slack;
            if (c->tile_ts > 0) {
                // a neighbour may now sit ceil((k h_max + slack) / tile side) tiles away
                const float tile_side = (float)c->tile_ts * c->fgrid.cs;
                const int d = (int)ceilf((lv.k * c->h_max_step + lv_slack) / tile_side);
                launch_tile_redilate(ls, &c->prof, c->tile_tsx, c->tile_tsy, d < 1 ? 1 : d, c->tile_raw.as<uint32_t>(), c->tile_h_ext.as<uint32_t>());
            }
        }
        if (m.n) {
            // (lab, SPH_SIDE_CUS: the side stream owns a few CUs -- the heavy detection sweeps then run on the MAIN stream, and only the
            //  propagation's ~100 small dependent launches go to the side stream, behind an event)
            const bool masked_side = side && c->opt.side_cus > 0;
            hipStream_t ds = masked_side ? c->stream : ls;
            if (p->fill_stash_with == SPH_STASH_NONE) (void)hipMemsetAsync(c->stash.p, 0, n * 4, ds);
            // (the CenterDiff detector leaves flag_insufficient_neighs alone: its default, false)
            if (lv.center_diff) (void)hipMemsetAsync(c->flag_insufficient.p, 0, n, ds);
            launch_level_detect(ds, &c->prof, al, lv);
            if (masked_side) {
                HIPCHK(c, hipEventRecord(c->ev_fork, c->stream));
                HIPCHK(c, hipStreamWaitEvent(c->stream2, c->ev_fork, 0));
            }
            // propagate until a sweep assigns nothing (`while changed`, simulation.rs:740-800).  Sweeps are queued in batches and the
            // host learns once per batch how many of them assigned something (a sweep behind the last effective one has no
            // candidates and costs a scan).  The first batch is as long as the previous step's propagation + 1 -- the fluid's
            // depth hardly changes from step to step -- so a step usually waits once instead of once per 8 sweeps.
            launch_level_propagate(ls, &c->prof, al, lv, 0u, chg + 1023);   // surface particles mark their neighbours
            uint32_t t = 1, effective = 0;
            int B = (int)std::min<uint32_t>(std::max<uint32_t>(c->last_level_sweeps + 1u, 8u), 1000u);
            if (c->opt.level_batch8) B = 8;   // measurement aid: the fixed batches of 8
            for (bool done = false; !done; B = 8) {
                // the flags live in device memory (a store to mapped host memory from every assigning lane made each sweep
                // wait for PCIe at its end); their sum goes to the host once per batch
                (void)hipMemsetAsync(chg, 0, (size_t)B * sizeof(uint32_t), ls);
                for (int b = 0; b < B; b++, t++) {
                    launch_level_propagate(ls, &c->prof, al, lv, t, chg + b);
                    if (t == 1u && p->fill_stash_with == SPH_STASH_SURFACE_DISTANCE_MIDDLE)   // num_iter == 1, simulation.rs:769-779
                        launch_fill_stash(ls, &c->prof, al, lv, c->stash.as<float>());
                }
                c->level_seq++;
                if (c->level_seq == 0u) c->level_seq = 1u;
                hipLaunchKernelGGL(k_publish_count, dim3(1), dim3(64), 0, ls, chg, (uint32_t)B, c->lvl_changed_dev, 63u, c->level_seq);
                if (side) {   // collected by level_estimation_finish()
                    lvp.on = true;
                    lvp.t = t;
                    lvp.B = B;
                    lvp.effective = effective;
                    lv_args = al;
                    break;
                }
                if ((rc = wait_word(c, (volatile uint32_t*)c->lvl_changed + 63, c->level_seq))) return rc;
                const uint32_t changed = c->lvl_changed[0];   // "nobody assigned anything" is final: the flags are ones, then zeros
                effective += changed;
                done = changed < (uint32_t)B;
            }
            if (!side) c->last_level_sweeps = effective;
        }
        c->have_level = true;
        m.st.ms_level_estimation += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_lvl0).count();
        return SPH_OK;
    };
    // the rest of a propagation that was queued on the side stream: wait for its first batch, continue in batches of 8 if the
    // prediction (the previous step's sweep count + 1) fell short
    auto level_estimation_finish = [&]() -> int {
        if (!lvp.on) return SPH_OK;
        lvp.on = false;
        const auto t_lvl0 = std::chrono::steady_clock::now();
        Member& m = M[0];
        sph_ctx* c = m.c;
        hipStream_t ls = c->stream2;
        uint32_t* chg = c->lvl_changed_d.as<uint32_t>();
        uint32_t t = lvp.t, effective = lvp.effective;
        int B = lvp.B;
        for (;;) {
            if ((rc = wait_word(c, (volatile uint32_t*)c->lvl_changed + 63, c->level_seq))) return rc;
            const uint32_t changed = c->lvl_changed[0];
            effective += changed;
            if (changed < (uint32_t)B) break;
            B = 8;
            (void)hipMemsetAsync(chg, 0, (size_t)B * sizeof(uint32_t), ls);
            for (int b = 0; b < B; b++, t++) launch_level_propagate(ls, &c->prof, lv_args, lv, t, chg + b);
            c->level_seq++;
            if (c->level_seq == 0u) c->level_seq = 1u;
            hipLaunchKernelGGL(k_publish_count, dim3(1), dim3(64), 0, ls, chg, (uint32_t)B, c->lvl_changed_dev, 63u, c->level_seq);
        }
        c->last_level_sweeps = effective;
        // (the host saw the side stream's last publish: everything queued there has finished before the main stream goes on)
        m.st.ms_level_estimation += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_lvl0).count();
        return SPH_OK;
    };
    // the same on a slab decomposition (before advection): ghost lanes idle, the ghosts' (level, when) refreshed from their
    // owners after the detection and after every sweep, propagation without frontier marks, the stop decision all-reduced
    std::vector<LevelArgs> LV(M.size());
    std::vector<SweepArgs> AL(M.size());   // the sweep arguments the level estimation runs with (advected geometry when `after`)
    float slab_lv_slack = 0.f;
    // `after` (level_estimation_after_advection, simulation.rs:2678-2707): detection and propagation on the lists of the ADVECTED
    // positions -- the ghosts' advected records come from their owners first; the extended lists are gathered from the cells of
    // the pre-step positions with every range widened by twice the largest displacement over ALL ranks, which the ghost layer
    // was sized for at the start of the step (slab_slack_k) and which is checked here
    auto level_estimation_slabs = [&](bool after) -> int {
        const auto t_lvl0 = std::chrono::steady_clock::now();
        const bool replay_step_lists = after && !p->use_extended_range_for_level_estimation;   // simulation.rs:2680: no rebuild
        slab_lv_slack = 0.f;
        if (after) {
            for (auto& m : M) m.lv_pmnew = (float*)m.c->pm[m.c->pcur ^ 1].as<float4>();
            if ((rc = refresh_ghosts(G, M, sel_lv_pmnew, 4, "pm_new"))) return rc;
            if (!replay_step_lists) {
                for (auto& m : M) {
                    sph_ctx* c = m.c;
                    (void)hipSetDevice(c->device);
                    HIPCHK(c, c->lvl_changed_d.ensure(1024 * sizeof(uint32_t)));
                    uint32_t* chg = c->lvl_changed_d.as<uint32_t>();
                    (void)hipMemsetAsync(chg + 1022, 0, sizeof(uint32_t), c->stream);
                    launch_max_disp(c->stream, &c->prof, m.n, c->pm[c->pcur].as<float4>(), c->pm[c->pcur ^ 1].as<float4>(), chg + 1022);
                    HIPCHK(c, hipMemcpyAsync(c->lvl_changed, chg + 1022, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
                }
                if ((rc = agree(G, wait_all(G)))) return rc;
                std::vector<std::vector<float>> dm(M.size(), std::vector<float>(1));
                for (size_t i = 0; i < M.size(); i++) {
                    float d;
                    memcpy(&d, (const void*)M[i].c->lvl_changed, 4);
                    dm[i][0] = -d;   // (NaN stays NaN through fminf only by luck: tested below on the local value too)
                    if (!std::isfinite(d)) dm[i][0] = -INFINITY;
                }
                if ((rc = G.comm->allreduce_min_f32(G, dm))) return rc;
                const float dmax = -dm[0][0];
                if (!std::isfinite(dmax)) return c0->fail(SPH_ERR_POSITION_NOT_FINITE, "Assertion 'p_position[d].is_finite()' failed!");
                for (auto& m : M) m.c->last_dmax = dmax;
                slab_lv_slack = 2.f * dmax;
                // (all-reduced values only: every rank takes the same branch)
                if (slab_lv_slack > c0->slab_slack_w)
                    return c0->fail(SPH_ERR_UNSUPPORTED, "level estimation after advection on a slab decomposition: the particles moved %g in this step, the ghost layer "
                                    "was widened for %g (4 x the previous step's largest displacement, at least 2 h_max)", (double)dmax, (double)(0.5f * c0->slab_slack_w));
            }
        }
        for (size_t i = 0; i < M.size(); i++) {
            Member& m = M[i];
            sph_ctx* c = m.c;
            (void)hipSetDevice(c->device);
            const size_t n = m.n ? m.n : 1;
            HIPCHK(c, c->lvl_tmp.ensure((c->cap ? c->cap : 1) * 4));   // swapped with lvl[cur] at the end of the step: a persistent array, capacity-sized
            HIPCHK(c, c->lvl_nrm.ensure(n * 8));
            HIPCHK(c, c->lvl_when.ensure(n * 4));
            HIPCHK(c, c->lvl_mark.ensure(n * 8));   // two parity buffers (OpLevelPropagate)
            HIPCHK(c, c->stash.ensure(n * 4));
            for (DevBuf* b : {&c->lvl_state, &c->flag_surface, &c->flag_insufficient}) HIPCHK(c, b->ensure(n));
            HIPCHK(c, c->nl_ext.ensure(sweep_list_bytes((uint32_t)n)));
            HIPCHK(c, c->nlx_ext.ensure(sweep_index_list_bytes((uint32_t)n)));
            HIPCHK(c, c->lvl_changed_d.ensure(1024 * sizeof(uint32_t)));
            if (!after) {
                m.a = make_args(c, m.sp);
                m.a.h_mode = p->support_length_estimation;
                m.a.sp_check_aii = p->check_aii;
            }
            m.a.nl_ext = c->nl_ext.as<uint4>();
            m.a.nlx_ext = c->nlx_ext.as<uint4>();
            LevelArgs& l = LV[i];
            l = LevelArgs{};
            l.k = p->level_estimation_range / SPH_ETA;
            l.threshold = cosf(50.f * (SPH_PI_F / 180.f));
            l.max_surface_distance = p->maximum_surface_distance;
            l.boundary_is_fluid_surface = p->boundary_is_fluid_surface;
            l.maximum_range = (p->support_length_estimation == SPH_H_FROM_DISTRIBUTION || p->support_length_estimation == SPH_H_FROM_DISTRIBUTION2)
                                  ? p->maximum_range : -1.f;   // simulation.rs:705-721
            l.nrm = c->lvl_nrm.as<float2>();
            l.state = c->lvl_state.as<uint8_t>();
            l.flag_surface = c->flag_surface.as<uint8_t>();
            l.flag_insufficient = c->flag_insufficient.as<uint8_t>();
            l.size_class = c->szc[c->cur].as<uint8_t>();
            l.level = c->lvl[c->cur].as<float>();
            l.when = c->lvl_when.as<uint32_t>();
            l.mark = c->lvl_mark.as<uint32_t>();
            l.level_old = c->lvlold[c->cur].as<float>();
            l.stash_first = p->fill_stash_with == SPH_STASH_SURFACE_DISTANCE_FIRST ? c->stash.as<float>() : nullptr;
            // frontier form with probing halo members (OpLevelPropagate, mode 2); SPH_SLAB_LEVEL_PLAIN=1: every unassigned particle in
            // every sweep (measurement / tests; parameters only: the same on every rank)
            l.plain_propagate = c0->opt.slab_level_plain ? 1 : 2;
            l.edge = c->dist.edge.as<uint8_t>();
            l.pm_cell = after ? c->pm[c->pcur].as<float4>() : nullptr;
            l.center_diff = p->level_estimation_method == SPH_LEVEL_CENTER_DIFF;
            l.replay_step_lists = replay_step_lists;
            if (replay_step_lists) {
                m.a.nl_ext = m.a.nl;
                m.a.nlx_ext = m.a.nlx;
            }
            SweepArgs& al = AL[i];
            al = m.a;
            if (after) {
                al.pm = c->pm[c->pcur ^ 1].as<float4>();
                al.t_ext.slack = slab_lv_slack;
                if (!replay_step_lists && c->tile_ts > 0) {
                    const float tile_side = (float)c->tile_ts * c->fgrid.cs;
                    const int dd = (int)ceilf((l.k * c->h_max_step + slab_lv_slack) / tile_side);
                    launch_tile_redilate(c->stream, &c->prof, c->tile_tsx, c->tile_tsy, dd < 1 ? 1 : dd, c->tile_raw.as<uint32_t>(), c->tile_h_ext.as<uint32_t>());
                }
            }
            m.lv_level = (float*)l.level;
            m.lv_when = (float*)l.when;
            if (m.n) {
                if (p->fill_stash_with == SPH_STASH_NONE) (void)hipMemsetAsync(c->stash.p, 0, n * 4, c->stream);
                // (flags / states of the ghost slots are never read; zero them so that downloads see defined bytes)
                (void)hipMemsetAsync(c->flag_surface.p, 0, n, c->stream);
                (void)hipMemsetAsync(c->flag_insufficient.p, 0, n, c->stream);
                launch_level_detect(c->stream, &c->prof, al, l);
            }
        }
        if ((rc = refresh_ghosts(G, M, sel_lv_level, 1, "level + when", -1, sel_lv_when))) return rc;   // (both fields in one exchange)
        const bool frontier = LV[0].plain_propagate == 2;
        if (frontier)   // sweep 0: the surface particles mark their unassigned neighbours (the candidates of sweep 1)
            for (size_t i = 0; i < M.size(); i++) {
                (void)hipSetDevice(M[i].c->device);
                if (M[i].n) launch_level_propagate(M[i].c->stream, &M[i].c->prof, AL[i], LV[i], 0u, M[i].c->lvl_changed_d.as<uint32_t>() + 1023);
            }
        const int B = 8;
        uint32_t t = 1;
        for (bool done = false; !done;) {
            for (auto& m : M) {
                (void)hipSetDevice(m.c->device);
                (void)hipMemsetAsync(m.c->lvl_changed_d.p, 0, B * sizeof(uint32_t), m.c->stream);
            }
            for (int b = 0; b < B; b++, t++) {
                for (size_t i = 0; i < M.size(); i++) {
                    Member& m = M[i];
                    sph_ctx* c = m.c;
                    (void)hipSetDevice(c->device);
                    if (!m.n) continue;
                    launch_level_propagate(c->stream, &c->prof, AL[i], LV[i], t, c->lvl_changed_d.as<uint32_t>() + b);
                    if (t == 1u && p->fill_stash_with == SPH_STASH_SURFACE_DISTANCE_MIDDLE)
                        launch_fill_stash(c->stream, &c->prof, AL[i], LV[i], c->stash.as<float>());
                }
                if ((rc = refresh_ghosts(G, M, sel_lv_level, 1, "level + when", -1, sel_lv_when))) return rc;
            }
            for (auto& m : M) {
                (void)hipSetDevice(m.c->device);
                HIPCHK(m.c, hipMemcpyAsync(m.c->lvl_changed, m.c->lvl_changed_d.p, B * sizeof(uint32_t), hipMemcpyDeviceToHost, m.c->stream));
            }
            if ((rc = agree(G, wait_all(G)))) return rc;
            // "nobody assigned anything" is final once it happens: the flags of a batch are ones followed by zeros, on every rank
            std::vector<int> last(M.size(), 0);
            for (size_t i = 0; i < M.size(); i++)
                for (int b = 0; b < B; b++)
                    if (M[i].c->lvl_changed[b]) last[i] = b + 1;
            if ((rc = G.comm->allreduce_max_i32(G, last))) return rc;
            done = last[0] < B;
        }
        for (auto& m : M) {
            m.c->have_level = true;
            m.st.ms_level_estimation += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_lvl0).count();
        }
        return SPH_OK;
    };
    if (level_on && G.multi() && !level_after) {
        if ((rc = level_estimation_slabs(false))) return rc;
        for (auto& m : M) dbg_sync(m.c, "level estimation (slabs)", 4);
    } else if (level_on && !level_after) {
        if ((rc = level_estimation(nullptr, nullptr, !c0->opt.level_serial))) return rc;   // (level_serial: measurement aid, everything on one stream)
    } else if (!level_on) {
        for (auto& m : M) m.c->have_level = false;
    }

    // ---- density + boundary lambda + neighbour count (simulation.rs:2072-2074, 2179-2180, 2204) -----------
    for (auto& m : M) {
        (void)hipSetDevice(m.c->device);
        // uniform scenes whose solves run on records: the density BUILD sweep also writes the mask words as 16-bit relative offsets, for
        // the two sweeps of the Jacobi iterations and the source-term sweeps (sph_sweeps.hip: k_sweep_off)
        m.a.nloff = nullptr;
        m.a.nlh = nullptr;
        const bool offsets = m.n && m.c->opt.offset_lists && sweeps_want_offset_lists(m.a);
        if (offsets) {
            // (sized by the particles the sweep visits, not by the context's capacity -- group g of particle i sits at [g n + i], the lists are
            //  rebuilt by every BUILD sweep, and DevBuf::ensure keeps a quarter of slack: a growing scene reallocates now and then)
            HIPCHK(m.c, m.c->nloff.ensure(sweep_offset_list_bytes(m.a.n ? m.a.n : 1u)));
            HIPCHK(m.c, m.c->nlh.ensure((size_t)(m.a.n ? m.a.n : 1u)));
            m.a.nloff_out = m.c->nloff.as<uint2>();
            m.a.nlh_out = m.c->nlh.as<uint8_t>();
        }
        if (m.n) launch_density(m.c->stream, &m.c->prof, m.a);
        m.a.nloff_out = nullptr;
        m.a.nlh_out = nullptr;
        if (offsets) {
            m.a.nloff = m.c->nloff.as<uint2>();
            m.a.nlh = m.c->nlh.as<uint8_t>();
        }
        if (m.n) launch_profile_calibration(m.c);   // (profiler modes 1 and 4 only)
        if (p->check_neighborhood && m.n) launch_check_neighborhood(m.c, m.a);
    }
    // ---- constrain_neighborhood_count (simulation.rs:2145-2177): h2 of over-populated particles shrinks AFTER the lists are
    // built; boundary terms (:2179), the CFL step (:2182-2191), the densities (:2204) follow with the new values
    for (auto& m : M) m.c->have_reduced = false;
    float dt_step = dt;
    if (p->constrain_neighborhood_count) {
        // (slab decomposition: every particle decides from its own list -- the ghosts' records are the pre-reduction ones its owner
        //  sees too; the batch loop ends when no rank has a pass pending (all-reduced), the reduced smoothing lengths travel to the
        //  ghosts before anything reads them, the CFL term of the new header is min-reduced over the ranks)
        const float onn = (SPH_ETA * 2.f) * (SPH_ETA * 2.f);   // optimal_neighbor_number, simulation.rs:386-388
        const uint32_t target = (uint32_t)onn + 5u;
        for (auto& m : M) {
            sph_ctx* c = m.c;
            (void)hipSetDevice(c->device);
            const size_t n = m.n ? m.n : 1;
            HIPCHK(c, c->con_thr.ensure(n * 4));
            HIPCHK(c, c->con_consumed.ensure(n * 4));
            HIPCHK(c, c->con_h.ensure(n * 4));
            HIPCHK(c, c->flag_reduced.ensure(n));
            HIPCHK(c, c->lvl_changed_d.ensure(64 * sizeof(uint32_t)));
            launch_constrain_init(c->stream, &c->prof, m.a, target, c->con_thr.as<float>(), c->con_consumed.as<uint32_t>(), c->con_h.as<float>(),
                                  c->flag_reduced.as<uint8_t>());
        }
        bool any = false;
        for (auto& m : M) any = any || m.n > 0;
        for (bool done = !any && !G.multi(); !done;) {
            const int B = 4;
            for (auto& m : M) {
                sph_ctx* c = m.c;
                (void)hipSetDevice(c->device);
                uint32_t* pend = c->lvl_changed_d.as<uint32_t>();
                (void)hipMemsetAsync(pend, 0, B * sizeof(uint32_t), c->stream);
                for (int b = 0; b < B && m.n; b++)
                    launch_constrain_pass(c->stream, &c->prof, m.a, target, c->con_thr.as<float>(), c->con_consumed.as<uint32_t>(), c->con_h.as<float>(), pend + b);
                HIPCHK(c, hipMemcpyAsync(c->lvl_changed, pend, B * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
            }
            if ((rc = sync_ctrl(G))) return rc;   // also surfaces the two assertions of :2163-2165
            std::vector<int> pending(M.size(), 0);
            for (size_t i = 0; i < M.size(); i++) pending[i] = M[i].c->lvl_changed[B - 1] ? 1 : 0;
            if (G.multi() && (rc = G.comm->allreduce_max_i32(G, pending))) return rc;
            done = true;
            for (int v : pending) done = done && !v;
        }
        for (auto& m : M) {
            sph_ctx* c = m.c;
            (void)hipSetDevice(c->device);
            if (m.n) launch_constrain_apply(c->stream, &c->prof, m.a, c->pm[c->pcur].as<float4>(), c->con_h.as<float>(), m.a.h2_next);
            m.lv_pmnew = (float*)c->pm[c->pcur].as<float4>();
        }
        if ((rc = refresh_ghosts(G, M, sel_lv_pmnew, 4, "pm"))) return rc;   // the ghosts' records with their owners' new h
        std::vector<std::vector<float>> cfl(M.size(), std::vector<float>(1, INFINITY));
        for (auto& m : M) {
            sph_ctx* c = m.c;
            (void)hipSetDevice(c->device);
            if (m.n) launch_header(c, m.n, p->rest_density, 0, c->hdr_host_dev, c->dist.on && c->dist.have_flags ? c->dist.owned.as<uint8_t>() : nullptr);
        }
        if ((rc = sync_ctrl(G))) return rc;
        for (size_t i = 0; i < M.size(); i++)
            if (M[i].n && M[i].c->n > 0) cfl[i][0] = M[i].c->hdr_host->min_cfl;
        if (G.multi() && (rc = G.comm->allreduce_min_f32(G, cfl))) return rc;
        if (!G.multi())
            for (size_t i = 1; i < M.size(); i++) cfl[0][0] = fminf(cfl[0][0], cfl[i][0]);
        dt_step = fminf(p->max_dt, p->cfl_factor * sqrtf(cfl[0][0]));
        for (auto& m : M) {
            sph_ctx* c = m.c;
            (void)hipSetDevice(c->device);
            m.sp.dt = dt_step;
            m.sp.hyb_vfactor = fminf(dt_step * p->hybrid_dfsph_factor, 1.f);
            m.a.sp = m.sp;
            m.st.dt = dt_step;
            if (m.n) launch_density_replay(c->stream, &c->prof, m.a);
            c->have_reduced = true;
        }
    }
    if ((rc = refresh_ghosts(G, M, sel_rho, 1, "rho"))) return rc;
    if (G.multi())
        for (auto& m : M) {
            sph_ctx* c = m.c;
            (void)hipSetDevice(c->device);
            slab_ghost_mrho(c, m.a);
        }
    // ---- constant_field + a_ii (simulation.rs:2235-2259) ---------------------------------------------------
    // the non-pressure acceleration directly follows in every mode but HybridDFSPH-with-forces-behind-the-divergence-solve:
    // then it shares the sweep (one replay of the lists, one gradient per pair)
    const bool np_first = p->pressure_solver_method != SPH_SOLVER_HYBRID_DFSPH || p->hybrid_dfsph_non_pressure_accel_before_divergence_free;
    const bool np_fused = np_first && !p->check_aii && !c0->opt.no_fuse;   // (no_fuse: measurement aid, the two sweeps apart)
    for (auto& m : M) {
        (void)hipSetDevice(m.c->device);
        if (m.n && np_fused) {
            launch_aii_const_non_pressure(m.c->stream, &m.c->prof, m.a);
            continue;
        }
        if (m.n) launch_aii_const(m.c->stream, &m.c->prof, m.a);
        // (slabs: the check reads the particle's own unit-pressure acceleration and the neighbours' m/rho, which the ghosts have)
        if (m.n && p->check_aii) launch_check_aii(m.c->stream, &m.c->prof, m.a);   // simulation.rs:1109-1123
    }
    auto non_pressure = [&]() -> int {  // update_velocity_with_non_pressure_accel: velocity_temp, then mem::swap
        for (auto& m : M) {
            sph_ctx* c = m.c;
            (void)hipSetDevice(c->device);
            if (m.n && !np_fused) launch_non_pressure(c->stream, &c->prof, m.a);
            std::swap(c->vel[c->cur], c->vel_tmp);
            m.a.vel = c->vel[c->cur].as<float2>();
            m.a.vel_tmp = c->vel_tmp.as<float2>();
            m.a.xv_ok = m.n && source_term_on_records(m.a);   // (the sweep wrote the {x, y, v'} record beside v')
        }
        return refresh_ghosts(G, M, sel_vel, 2, "vel");
    };
    auto begin_solve = [&](int kind, int residual_density) {
        for (auto& m : M) {
            (void)hipSetDevice(m.c->device);
            if (!m.n) launch_ctrl_reset(m.c->stream, m.c->ctrl.as<SolverCtrl>(), m.a.gate);   // else: reset by the source-term sweep
            if (m.n) launch_source_term(m.c->stream, &m.c->prof, m.a, kind, residual_density);  // + Jacobi iteration 0
        }
    };
    auto rec = [&](int e) {
        if (tev)
            for (auto& m : M) {
                (void)hipSetDevice(m.c->device);
                (void)hipEventRecord(m.c->ev[e], m.c->stream);
            }
    };
    enum { T_NONE = 0, T_VEL = 1, T_VX = 2, T_HYBRID = 3 };  // TAIL_* of sph_sweeps.hip
    g_trace.mark(3);
    // The NEXT step's neighbour build, queued behind this step's integrating tail (sph_context.hpp: Ahead): cell keys on a grid
    // predicted from this step's bounding box + 2 cells (a particle moves at most cfl_factor supports per step), cell sort,
    // reorder into buffers of their own, cell ranges.  One context, paced solves (the tail is queued once the stop decision was
    // seen), mass-derived smoothing lengths, uniform scenes and multi-resolution scenes on their fine grid.
    // Two calls: plan_ahead_build() in front of the step's last solve -- everything it decides is known by then, and the integrating
    // tail of that solve classifies the particles for the incremental sort while it holds their new positions -- and
    // queue_ahead_build() behind the tail.
    struct AheadPlan {
        bool on = false, incremental = false;
        GridP g{};
        int tile_ts = 0, tile_tsx = 0, tile_tsy = 0;
        IncClassifyP q{};
    } plan;
    auto plan_ahead_build = [&]() -> int {
        sph_ctx* c = c0;
        plan = AheadPlan{};
        // (uniform scenes, and multi-resolution scenes sorted by their fine grid: tile_ts > 0; a narrow h distribution on the coarse
        //  grid -- FromDistribution* -- is excluded with h_from_mass_mode)
        if (G.multi() || !paced || !c->opt.ahead_build || !h_from_mass_mode || p->constrain_neighborhood_count || (!c->uniform_h && c->tile_ts <= 0) ||
            (c->uniform_h && c->tile_ts != 0) || c->exact || M[0].n == 0)
            return SPH_OK;
        const uint32_t n = M[0].n;
        // the grid this step sorted by, moved to where the particles may be after it: the same cell size (the smoothing lengths are the
        // masses'), the bounding box of the step's start + 1 cell + a margin of two SUPPORTS of the largest particle (a particle moves
        // at most cfl_factor supports per step): two cells of a uniform scene, two tiles of a multi-resolution one
        const float cs = c->fgrid.cs;
        const int ts = c->tile_ts, margin = 2 * (ts > 0 ? ts : 1);
        GridP g{};
        g.cs = cs;
        g.minx = (int)floorf(boxes[0].min_x / cs) - 1 - margin;
        g.miny = (int)floorf(boxes[0].min_y / cs) - 1 - margin;
        const long long sx = (long long)((int)floorf(boxes[0].max_x / cs) + 2 + margin) - g.minx, sy = (long long)((int)floorf(boxes[0].max_y / cs) + 2 + margin) - g.miny;
        if (sx <= 0 || sy <= 0 || sx >= 65536 || sy >= 65536 || sx * sy >= (1ll << 27)) return SPH_OK;
        g.sx = (int)sx;
        g.sy = (int)sy;
        g.ncells = (uint32_t)sx * (uint32_t)sy;
        if (ts > 0) {
            plan.tile_ts = ts;
            plan.tile_tsx = (g.sx + ts - 1) / ts;
            plan.tile_tsy = (g.sy + ts - 1) / ts;
            const size_t nt = (size_t)plan.tile_tsx * (size_t)plan.tile_tsy;
            HIPCHK(c, c->atile_raw.ensure(nt * 4));
            HIPCHK(c, c->atile_h.ensure(nt * 4));
        }
        for (int q = 0; q < 2; q++) {
            HIPCHK(c, c->akey[q].ensure((size_t)c->cap * 4));
            HIPCHK(c, c->aval[q].ensure((size_t)c->cap * 4));
        }
        HIPCHK(c, c->acxy.ensure((size_t)c->cap * 4));
        HIPCHK(c, c->pm2.ensure((size_t)c->cap * sizeof(float4)));
        HIPCHK(c, c->acell_start.ensure(((size_t)g.ncells + 1) * 4));
        plan.on = true;
        plan.g = g;
        // The array is sorted by the cells of this step's start and a step moves few particles into another cell: the sort is a
        // merge of the ones that stay with the ones that do not (sph_sort.hip: incremental_cell_sort_reorder) -- the same keys, order
        // and cell ranges as the radix sort, whatever the number of movers; its cost grows with them, so a large count (the last one
        // the device reported: a step or two old) sends the build through the radix sort, and every eighth such build probes again.
        const uint32_t* movers_host = (const uint32_t*)(c->ctrl_host + 2) + 1;   // (second word of the mapped block whose first word is the paced solves' progress)
        bool incremental = c->opt.inc_sort && c->grid_valid && c->fgrid.cs == cs && c->fgrid.ncells > 0 && g.ncells <= n + 4096u;   // (see k_inc_scan: quadratic in ncells / 1024; a grid much sparser than one cell per particle takes the radix sort)
        if (incremental && c->inc_count_valid && *movers_host > n / (c->opt.inc_sort > 1 ? (uint32_t)c->opt.inc_sort : 3u)) {
            incremental = ++c->inc_radix_streak >= 8;
            if (incremental) c->inc_radix_streak = 0;
        }
        if (!incremental) return SPH_OK;
        hipStream_t s = c->stream;
        const size_t head_before = c->inc_head.bytes;
        HIPCHK(c, c->inc_head.ensure((size_t)g.ncells * 8));
        if (c->inc_head.bytes != head_before) HIPCHK(c, hipMemsetAsync(c->inc_head.p, 0, c->inc_head.bytes, s));   // (epoch 0: no list)
        HIPCHK(c, c->inc_next.ensure((size_t)c->cap * 4));
        HIPCHK(c, c->inc_bsum.ensure(incremental_sort_block_sums(g.ncells) * 4));
        if (!c->inc_movers.p) {
            HIPCHK(c, c->inc_movers.ensure(4));
            HIPCHK(c, hipMemsetAsync(c->inc_movers.p, 0, 4, s));
        }
        if (++c->inc_epoch == 0u) c->inc_epoch = 1u;
        plan.incremental = true;
        plan.q = IncClassifyP{c->fgrid, g, c->cxy.as<uint32_t>(), c->akey[1].as<uint32_t>(), c->aval[1].as<uint8_t>(), c->inc_next.as<uint32_t>(),
                              c->inc_head.as<unsigned long long>(), c->inc_epoch};
        M[0].a.inc = plan.q;   // (the integrating tail classifies: launch_solver_tail)
        return SPH_OK;
    };
    // (with the level estimation on, the build -- it moves the level arrays -- waits for the smoothing at the end of the step: the
    //  calls behind the tail return at once, the one behind the step's last wait queues it)
    bool ahead_deferred = level_on;
    const std::function<int()> queue_ahead_build = [&]() -> int {
        sph_ctx* c = c0;
        if (ahead_deferred) return SPH_OK;
        c->ahead.valid = false;
        if (!plan.on) return SPH_OK;
        const uint32_t n = M[0].n;
        const GridP g = plan.g;
        hipStream_t s = c->stream;
        const int k = c->cur;
        const float4* integrated = c->pm[c->pcur ^ 1].as<float4>();   // (the tail's output; the step's end flips pcur)
        if (plan.incremental) {
            const ReorderIO io{integrated, c->vel[k].as<float2>(), c->orig[k].as<uint32_t>(), c->lvl[k].as<float>(), c->lvlold[k].as<float>(), c->pm2.as<float4>(),
                               c->vel[k ^ 1].as<float2>(), c->orig[k ^ 1].as<uint32_t>(), c->lvl[k ^ 1].as<float>(), c->lvlold[k ^ 1].as<float>(), c->acxy.as<uint32_t>(),
                               c->h2n[k].as<float>(), c->h2n[k ^ 1].as<float>(), /* lambda sums of this step: only FromDistribution* reads them next step,
                               and a step in that mode does not adopt this build */ nullptr, nullptr, c->szc[k].as<uint8_t>(), c->szc[k ^ 1].as<uint8_t>()};
            incremental_cell_sort_reorder(s, &c->prof, n, integrated, plan.q, /* classified by the tail */ true, c->cell_start.as<uint32_t>(), c->akey[0].as<uint32_t>(),
                                          c->acell_start.as<uint32_t>(), io, c->inc_bsum.as<uint32_t>(), c->inc_movers.as<uint32_t>(),
                                          (uint32_t*)(c->ctrl_host_dev + 2) + 1);
            c->inc_count_valid = true;   // (a count of the current state is on its way, behind any stale one on the same stream)
        } else {
            const CellKeyGen kg{integrated, g, nullptr, 0u, (uint32_t)SC_GONE_FROM, 1};   // (clamped keys: the grid is a prediction)
            const int res = radix_sort_pairs(s, &c->prof, c->akey[0].as<uint32_t>(), c->aval[0].as<uint32_t>(), c->akey[1].as<uint32_t>(), c->aval[1].as<uint32_t>(), n,
                                             ilog2_ceil(g.ncells), c->sort_scratch.as<uint32_t>(), &kg);
            if (res == 1) {
                std::swap(c->akey[0], c->akey[1]);
                std::swap(c->aval[0], c->aval[1]);
            }
            launch_reorder(s, &c->prof, n, g, c->akey[0].as<uint32_t>(), c->aval[0].as<uint32_t>(), integrated, c->vel[k].as<float2>(), c->orig[k].as<uint32_t>(),
                           c->lvl[k].as<float>(), c->lvlold[k].as<float>(), c->pm2.as<float4>(), c->vel[k ^ 1].as<float2>(), c->orig[k ^ 1].as<uint32_t>(),
                           c->lvl[k ^ 1].as<float>(), c->lvlold[k ^ 1].as<float>(), c->acxy.as<uint32_t>(), c->h2n[k].as<float>(), c->h2n[k ^ 1].as<float>(),
                           c->lam_sum.as<float>(), c->lam_prev.as<float>(), c->cs_scratch.p, c->szc[k].as<uint8_t>(), c->szc[k ^ 1].as<uint8_t>());
            launch_cell_start(s, &c->prof, c->akey[0].as<uint32_t>(), n, g.ncells, c->acell_start.as<uint32_t>(), c->cs_scratch.p, true);
        }
        if (plan.tile_ts > 0)
            launch_tile_hmax(s, &c->prof, n, c->pm2.as<float4>(), g, plan.tile_ts, plan.tile_tsx, plan.tile_tsy, c->atile_raw.as<uint32_t>(), c->atile_h.as<uint32_t>(),
                             nullptr, 1);
        c->ahead.valid = true;
        c->ahead.g = g;
        c->ahead.h_max = h_max_g;
        c->ahead.h_min = h_min_g;
        c->ahead.tile_ts = plan.tile_ts;
        c->ahead.tile_tsx = plan.tile_tsx;
        c->ahead.tile_tsy = plan.tile_tsy;
        c->ahead.rest_density = p->rest_density;
        c->ahead.n = c->n;
        return SPH_OK;
    };

    switch (p->pressure_solver_method) {
    case SPH_SOLVER_IISPH:  // simulation.rs:2389-2446
        if ((rc = non_pressure())) return rc;
        rec(4);
        begin_solve(1, 1);
        if ((rc = plan_ahead_build())) return rc;
        if ((rc = pressure_iterations(G, M, p->iisph_max_avg_density_error, 1, p->max_iters, c0->last_dens_iters, T_VX, true, !level_on, &queue_ahead_build))) return rc;
        rec(5);
        break;
    case SPH_SOLVER_IISPH2:  // simulation.rs:2262-2387: omega rides in the source-term sweep; p /= sqrt(omega) before the last a^p
        if ((rc = non_pressure())) return rc;
        rec(4);
        begin_solve(3, 1);
        if ((rc = pressure_iterations(G, M, p->iisph_max_avg_density_error, 1, p->max_iters, c0->last_dens_iters, T_NONE, true, false))) return rc;
        for (auto& m : M) {
            (void)hipSetDevice(m.c->device);
            if (m.n) launch_iisph2_scale(m.c->stream, &m.c->prof, m.a);
        }
        if ((rc = refresh_ghosts(G, M, M[0].c->pressure_cur ? sel_pt1 : sel_pt0, 1, "pt"))) return rc;
        for (auto& m : M) {   // a^p again, from the rescaled pressures (simulation.rs:2362-2373), then v += dt a^p ; x += dt v
            (void)hipSetDevice(m.c->device);
            if (!m.n) continue;
            launch_pressure_accel(m.c->stream, &m.c->prof, m.a, -1, 1, 0.f, 0u, G.multi() ? 1 : 0);
            launch_solver_tail(m.c->stream, &m.c->prof, m.a, T_VX, m.c->pm[m.c->pcur ^ 1].as<float4>());
            if (m.a.hdr_partials) launch_header_ahead(m.c, (m.n + 255u) / 256u, m.c->hdr_host_dev);
        }
        if (G.multi() && !level_on && (rc = G.comm->agree_guards_queued(G))) return rc;
        if ((rc = sync_ctrl(G, G.multi() ? (level_on ? SYNC_DEFER : SYNC_FINAL) : SYNC_AGREE))) return rc;
        rec(5);
        break;
    case SPH_SOLVER_ONLY_DIVERGENCE:  // simulation.rs:2448-2500
        if ((rc = non_pressure())) return rc;
        rec(2);
        begin_solve(0, 0);
        if ((rc = plan_ahead_build())) return rc;
        if ((rc = pressure_iterations(G, M, p->hybrid_dfsph_max_avg_divergence_error, 0, p->max_iters, c0->last_div_iters, T_VX, false, !level_on, &queue_ahead_build))) return rc;
        rec(3);
        break;
    default: {  // HybridDFSPH, simulation.rs:2502-2670
        if (p->hybrid_dfsph_non_pressure_accel_before_divergence_free)
            if ((rc = non_pressure())) return rc;
        rec(2);
        begin_solve(0, 0);
        // Forces in front of the divergence solve (the default): the two solves are CHAINED -- the density solve is queued right
        // behind the divergence solve's tail, its launches gated on the device by "the divergence solve ended" (k_solver_handoff),
        // and the step waits once, at its end.  If the divergence solve needed more iterations than were queued, the gated
        // launches cost a few us each, and both are queued again from where the first one stands.  On a slab decomposition the
        // gated solve all-reduces into totals of its own; the exchanges queued behind a closed gate re-send what the ghosts
        // already hold.
        // Chained only while the divergence solve's iteration count repeats from step to step (SPH_CHAIN=1: always, =0: never):
        // in the first steps of a dam break it jumps by factors (4, 15, 17, 7, ...), and a short-fall there throws away a
        // density solve's worth of gated launches (measured: 1.38 vs 1.27 ms/step over steps 5-24 when always chained).
        // (parameters and all-reduced iteration counts only: every rank decides the same)
        if (paced && (G.multi() || M[0].n > 0)) {
            // both solves paced against the device's progress, ONE host wait at the end of the step (a slab decomposition: the
            // ghosts' velocities follow the first solve's tail, the guards are agreed in front of the wait)
            const bool multi = G.multi();
            SolveQ qd{p->hybrid_dfsph_max_avg_divergence_error, 0, p->max_iters, T_VEL, false};
            SolveQ qs{p->hybrid_dfsph_max_avg_density_error, 1, p->max_iters, T_HYBRID, true};
            if ((rc = solve_paced(G, M, qd, pace_prediction(c0, M[0].n, c0->last_div_iters, c0->prev_div_iters)))) return rc;
            if ((rc = solve_queue(G, M, qd, true))) return rc;   // the tail (v += dt a^p): it also leaves the solve's control block in ctrl_host[1]
            g_trace.mark(4);
            rec(3);
            if (multi && (rc = refresh_ghosts(G, M, sel_vel, 2, "vel"))) return rc;   // v += dt a^p happened in the tail: the forces / the source term read the neighbours' velocities
            if (!p->hybrid_dfsph_non_pressure_accel_before_divergence_free)
                if ((rc = non_pressure())) return rc;
            rec(4);
            begin_solve(p->hybrid_dfsph_density_source_term == SPH_ONLY_DENSITY ? 2 : 1, 1);
            if ((rc = plan_ahead_build())) return rc;
            if ((rc = solve_paced(G, M, qs, pace_prediction(c0, M[0].n, c0->last_dens_iters, c0->prev_dens_iters)))) return rc;
            if ((rc = solve_queue(G, M, qs, false, true))) return rc;
            if ((rc = queue_ahead_build())) return rc;
            const bool final_solve = !level_on;
            if (multi && final_solve && (rc = G.comm->agree_guards_queued(G))) return rc;
            if ((rc = sync_ctrl(G, multi ? (final_solve ? SYNC_FINAL : SYNC_DEFER) : SYNC_AGREE))) return rc;
            for (auto& m : M) {
                solve_stats(m, qd, m.c->ctrl_host[1]);
                solve_stats(m, qs, *m.c->ctrl_host);
            }
            g_trace.mark(5);
            rec(5);
            break;
        }
        const bool chain_wanted = c0->opt.chain >= 0 ? c0->opt.chain == 1 : c0->last_div_iters == c0->prev_div_iters;
        const bool chain = (G.multi() || M[0].n > 0) && p->hybrid_dfsph_non_pressure_accel_before_divergence_free && chain_wanted;
        if (chain) {
            const int multi = G.multi() ? 1 : 0;
            const bool final_solve = !level_on;
            SolveQ qd{p->hybrid_dfsph_max_avg_divergence_error, 0, p->max_iters, T_VEL, false};
            SolveQ qs{p->hybrid_dfsph_max_avg_density_error, 1, p->max_iters, T_HYBRID, true};
            qs.tot_slot = 1;
            const int kind = p->hybrid_dfsph_density_source_term == SPH_ONLY_DENSITY ? 2 : 1;
            auto set_gate = [&](bool on) {
                for (auto& m : M) {
                    m.a.gate = on ? (const uint32_t*)(m.c->ctrl.as<SolverCtrl>() + 2) : nullptr;
                    m.a.solver_tot = m.c->dist.solver_tot.as<double>() + (on && m.c->dist.solver_tot.p ? 8 : 0);
                    if (m.a.tot_table) m.a.tot_table = m.c->dist.tot_table.as<double>() + (on ? (size_t)m.c->dist.nranks * 8 : 0);
                }
            };
            if ((rc = solve_begin(G, M, qd, c0->last_div_iters))) return rc;
            for (bool div_done = false;;) {
                if (!div_done) {
                    set_gate(false);
                    if ((rc = solve_queue(G, M, qd, true))) return rc;   // its tail hands over and sets the gate
                    g_trace.mark(4);
                    rec(3);
                    if ((rc = refresh_ghosts(G, M, sel_vel, 2, "vel"))) return rc;  // v += dt a^p happened in the tail (or nothing did: the same values again)
                    rec(4);
                    set_gate(true);
                    begin_solve(kind, 1);   // the density solve from its start
                    if ((rc = solve_begin(G, M, qs, c0->last_dens_iters))) return rc;
                }
                if ((rc = solve_queue(G, M, qs, false, true))) return rc;
                if (multi && final_solve && (rc = G.comm->agree_guards_queued(G))) return rc;
                if ((rc = sync_ctrl(G, multi ? (final_solve ? SYNC_FINAL : SYNC_DEFER) : SYNC_AGREE))) return rc;
                if (!div_done) {
                    if (!M[0].c->ctrl_host[1].done) {   // the divergence solve fell short: nothing of the density solve ran (on any rank)
                        if (qd.k > p->max_iters) return c0->fail(SPH_ERR_DEVICE, "divergence solve: no decision after max_iters iterations");
                        qd.extend();
                        continue;
                    }
                    div_done = true;
                    for (auto& m : M) solve_stats(m, qd, m.c->ctrl_host[1]);
                }
                if (M[0].c->ctrl_host->done) break;
                if (qs.k > p->max_iters) break;
                qs.extend();
            }
            set_gate(false);
            for (auto& m : M) solve_stats(m, qs, *m.c->ctrl_host);
            g_trace.mark(5);
            rec(5);
            break;
        }
        if ((rc = pressure_iterations(G, M, p->hybrid_dfsph_max_avg_divergence_error, 0, p->max_iters, c0->last_div_iters, T_VEL, false, false))) return rc;
        g_trace.mark(4);
        rec(3);
        if ((rc = refresh_ghosts(G, M, sel_vel, 2, "vel"))) return rc;  // v += dt a^p happened in the final sweep
        if (!p->hybrid_dfsph_non_pressure_accel_before_divergence_free)
            if ((rc = non_pressure())) return rc;
        rec(4);
        begin_solve(p->hybrid_dfsph_density_source_term == SPH_ONLY_DENSITY ? 2 : 1, 1);
        if ((rc = pressure_iterations(G, M, p->hybrid_dfsph_max_avg_density_error, 1, p->max_iters, c0->last_dens_iters, T_HYBRID, true, !level_on))) return rc;
        g_trace.mark(5);
        rec(5);
        break;
    }
    }
    rec(6);
    if (p->viscosity_type == SPH_VISC_XSPH)  // simulation.rs:2673-2676
        return c0->fail(SPH_ERR_XSPH_TODO, "not yet implemented (XSPH velocity smoothing)");

    // ---- smooth_level_estimation_field (simulation.rs:2709-2722) ------------------------------------------------------------
    if (level_on && G.multi()) {
        if (level_after && (rc = level_estimation_slabs(true))) return rc;   // simulation.rs:2678-2707, at the advected positions
        const auto t_lvl0 = std::chrono::steady_clock::now();
        for (auto& m : M) {
            m.lv_level = m.c->lvl[m.c->cur].as<float>();
            m.lv_pmnew = (float*)m.c->pm[m.c->pcur ^ 1].as<float4>();
        }
        // the ghosts' level values and ADVECTED records (ghost lanes do not integrate) come from their owners
        if ((rc = refresh_ghosts(G, M, sel_lv_level, 1, "level"))) return rc;
        if (!level_after && (rc = refresh_ghosts(G, M, sel_lv_pmnew, 4, "pm_new"))) return rc;
        for (size_t i = 0; i < M.size(); i++) {
            Member& m = M[i];
            sph_ctx* c = m.c;
            (void)hipSetDevice(c->device);
            if (!m.n) continue;
            SweepArgs as = level_after ? AL[i] : m.a;
            as.pm = m.a.pm;   // (the smoothing takes the advected records as its own argument)
            launch_level_smooth(c->stream, &c->prof, as, LV[i], c->pm[c->pcur ^ 1].as<float4>(), c->lvl[c->cur].as<float>(), c->lvl_tmp.as<float>());
            std::swap(c->lvl[c->cur], c->lvl_tmp);
            // (no classify_particles here: the reference's step never calls it -- sph_classify is the host's call)
        }
        if ((rc = G.comm->agree_guards_queued(G))) return rc;
        if ((rc = sync_ctrl(G, SYNC_FINAL))) return rc;
        for (auto& m : M) m.st.ms_level_estimation += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_lvl0).count();
    }
    if (level_after && !G.multi()) {   // simulation.rs:2678-2707: lists of the advected positions, then detection + propagation there
        sph_ctx* c = M[0].c;
        if ((rc = level_estimation(c->pm[c->pcur ^ 1].as<float4>(), c->pm[c->pcur].as<float4>(), false))) return rc;
    }
    if (level_on && !G.multi()) {
        if ((rc = level_estimation_finish())) return rc;   // (before advection: queued on the side stream at the start of the step)
        const auto t_lvl0 = std::chrono::steady_clock::now();
        Member& m = M[0];
        sph_ctx* c = m.c;
        if (m.n) {
            SweepArgs as = m.a;
            as.t_ext.slack = lv_slack;
            launch_level_smooth(c->stream, &c->prof, as, lv, c->pm[c->pcur ^ 1].as<float4>(), c->lvl[c->cur].as<float>(), c->lvl_tmp.as<float>());
            std::swap(c->lvl[c->cur], c->lvl_tmp);
        }
        if ((rc = sync_ctrl(G))) return rc;
        m.st.ms_level_estimation += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_lvl0).count();
        // the next step's neighbour build, behind the smoothed level values (the device works on it while the host leaves the step)
        ahead_deferred = false;
        if ((rc = queue_ahead_build())) return rc;
    }

    for (size_t i = 0; i < M.size(); i++) {
        Member& m = M[i];
        sph_ctx* c = m.c;
        c->pcur ^= 1;  // integrated positions live in the other pm buffer; the old one keeps the pre-step snapshot
        c->lists_after = level_after && p->use_extended_range_for_level_estimation;
        c->lists_after_k = G.multi() ? LV[i].k : lv.k;
        c->lists_after_slack = G.multi() ? slab_lv_slack : lv_slack;
        // every solver mode ends in an integrating final sweep, which left the next step's header in hdr_host
        // (constrain_neighborhood_count left reduced smoothing lengths in the records: the next step's k_header restores them)
        c->hdr_ahead = h_from_mass_mode && m.n > 0 && m.a.hdr_partials != nullptr && !p->constrain_neighborhood_count;
        c->hdr_ahead_rest_density = p->rest_density;
        c->prev_div_iters = c->last_div_iters;
        c->prev_dens_iters = c->last_dens_iters;
        c->last_div_iters = m.st.div_solver.iters;
        c->last_dens_iters = m.st.density_solver.iters;
        c->time += dt_step;  // simulation.rs:2724-2725
        c->step_number += 1;
        m.st.time = c->time;
        m.st.step_number = c->step_number;
        m.st.ms_simulation_step = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - m.wall0).count();
        if (tev) {
            (void)hipSetDevice(c->device);
            float ms = 0.f;
            (void)hipEventSynchronize(c->ev[6]);
            if (hipEventElapsedTime(&ms, c->ev[0], c->ev[6]) == hipSuccess) m.st.ms_simulation_step = ms;
            if (hipEventElapsedTime(&ms, c->ev[0], c->ev[1]) == hipSuccess) m.st.ms_neighborhood = ms;
            const bool has_div = p->pressure_solver_method == SPH_SOLVER_ONLY_DIVERGENCE || p->pressure_solver_method == SPH_SOLVER_HYBRID_DFSPH;
            const bool has_dens = p->pressure_solver_method != SPH_SOLVER_ONLY_DIVERGENCE;
            if (has_div && hipEventElapsedTime(&ms, c->ev[2], c->ev[3]) == hipSuccess) m.st.ms_div_solver = ms;
            if (has_dens && hipEventElapsedTime(&ms, c->ev[4], c->ev[5]) == hipSuccess) m.st.ms_density_solver = ms;
        }
        if (c->prof.mode) c->prof.collect();
        c->prof.step_index++;
        if (outs) outs[i] = m.st;
    }
    g_trace.end_step();
    return SPH_OK;
}

static int group_step(Group& G, const sph_params* p, sph_step_stats* outs)
{
    bool started = false;
    const int rc = group_step_inner(G, p, outs, &started);
    if (rc && started)
        for (auto c : G.m) {
            c->poisoned = true;
            c->hdr_ahead = false;
            (void)hipSetDevice(c->device);
            if (c->stream2) (void)hipStreamSynchronize(c->stream2);   // a level estimation may still be running on the side stream
            if (c->dist.xstream) (void)hipStreamSynchronize(c->dist.xstream);   // ... or the interior half of a split sweep A
        }
    return rc;
}

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" int sph_step(sph_ctx* c, const sph_params* p, sph_step_stats* out)
{
    if (!c || !p) return SPH_ERR_INVALID_ARGUMENT;
    HIPCHK(c, hipSetDevice(c->device));
    Group G;
    G.m.push_back(c);
    int rc = comm_for_rank(c, &G.comm);
    if (rc) return rc;
    rc = group_step(G, p, out);
    if (rc) comm_abandon(c);   // (thread / shared-memory transports: the other ranks' next collective reports it instead of waiting)
    return rc;
}

// classify_particles (adaptivity/mod.rs:50-59) on the device-resident state: the level field and masses as they stand
extern "C" int sph_classify(sph_ctx* c, const sph_params* p)
{
    if (!c || !p) return SPH_ERR_INVALID_ARGUMENT;
    if (c->poisoned) return c->fail(SPH_ERR_POISONED, "an earlier step failed inside the step: the particle state is undefined until sph_upload");
    HIPCHK(c, hipSetDevice(c->device));
    const bool flags = c->dist.on && c->dist.have_flags;
    const uint32_t n = flags ? c->dist.n_tot : (uint32_t)c->n;
    launch_classify(c->stream, &c->prof, n, c->pm[c->pcur].as<float4>(), c->lvl[c->cur].as<float>(), c->szc[c->cur].as<uint8_t>(),
                    flags ? c->dist.owned.as<uint8_t>() : nullptr, c->orig[c->cur].as<uint32_t>(), c->status.as<DeviceStatus>(), p);
    Group G;
    G.m.push_back(c);
    int rc = sync_ctrl(G);   // surfaces the unreachable!() of a particle without a level value
    if (rc == SPH_ERR_INVALID_ARGUMENT) c->err = "internal error: entered unreachable code (LevelEstimationState::level of FluidInterior) -- " + c->err;
    return rc;
}

extern "C" int sph_group_step(sph_ctx** ctxs, int n, const sph_params* p, sph_step_stats* outs)
{
    if (!ctxs || n <= 0 || !p) return SPH_ERR_INVALID_ARGUMENT;
    Group G;
    for (int i = 0; i < n; i++) {
        if (!ctxs[i]) return SPH_ERR_INVALID_ARGUMENT;
        if (n > 1 && (!ctxs[i]->dist.on || ctxs[i]->dist.rank != i || ctxs[i]->dist.nranks != n))
            return ctxs[i]->fail(SPH_ERR_INVALID_ARGUMENT, "context %d is not configured as rank %d of %d (sph_dist_configure)", i, i, n);
        G.m.push_back(ctxs[i]);
    }
    if (n > 1) G.comm = comm_loopback();
    return group_step(G, p, outs);
}

extern "C" int sph_dist_configure(sph_ctx* c, int rank, int n_ranks, float cut_lo, float cut_hi)
{
    if (!c || rank < 0 || n_ranks < 1 || rank >= n_ranks) return SPH_ERR_INVALID_ARGUMENT;
    // SPH_FORCE_SLAB_MODE=1: run the slab driver and the RCCL collectives with ONE rank (a single-GPU check of that code path)
    c->dist.on = n_ranks > 1 || c->opt.force_slab_mode != 0;
    c->dist.rank = rank;
    c->dist.nranks = n_ranks;
    c->dist.cut_lo = cut_lo;
    c->dist.cut_hi = cut_hi;
    c->dist.have_flags = false;
    c->hdr_ahead = false;
    c->dist.n_tot = (uint32_t)c->n;
    if (c->dist.on) {
        HIPCHK(c, hipSetDevice(c->device));
        return ensure_dist_buffers(c, (uint32_t)c->n);
    }
    return SPH_OK;
}

extern "C" int sph_dist_set_rebalance(sph_ctx* c, int every_n_steps)
{
    if (!c || every_n_steps < 0) return SPH_ERR_INVALID_ARGUMENT;
    c->dist.rebalance_every = every_n_steps;
    return SPH_OK;
}

extern "C" int sph_dist_get_cuts(sph_ctx* c, float* cut_lo, float* cut_hi, uint32_t* n_rebalances)
{
    if (!c) return SPH_ERR_INVALID_ARGUMENT;
    if (cut_lo) *cut_lo = c->dist.cut_lo;
    if (cut_hi) *cut_hi = c->dist.cut_hi;
    if (n_rebalances) *n_rebalances = c->dist.rebalances;
    return SPH_OK;
}

extern "C" int sph_dist_get_stats(sph_ctx* c, sph_dist_stats* out, int reset)
{
    if (!c || !out) return SPH_ERR_INVALID_ARGUMENT;
    auto& d = c->dist;
    memset(out, 0, sizeof(*out));
    out->steps = c->step_number - d.stat_step0;
    out->exchanges = d.stat_exchanges;
    out->bytes_sent = d.stat_bytes_sent;
    out->bytes_received = d.stat_bytes_recv;
    out->allreduces = d.stat_allreduces;
    out->host_waits = c->n_waits;
    out->n_owned = c->n;
    out->n_halo[0] = d.n_halo[0];
    out->n_halo[1] = d.n_halo[1];
    out->n_ghost[0] = d.n_ghost[0];
    out->n_ghost[1] = d.n_ghost[1];
    comm_describe(c, &out->transport, &out->comm_ranks);
    if (reset) {
        d.stat_exchanges = d.stat_bytes_sent = d.stat_bytes_recv = d.stat_allreduces = 0;
        d.stat_step0 = c->step_number;
        c->n_waits = 0;
    }
    return SPH_OK;
}
