// What the step driver (sph_step.hip) and the transports between ranks (sph_transport.hip) share: the transport interface, the
// group of contexts a step is driven for, and the few wait / staging helpers both sides use.
#pragma once

#include <cstdint>
#include <vector>

#include "sph_context.hpp"

enum { RC_BAD = 12, RC_FAR = 13, RC_HL = 14, RC_HR = 15 };   // RC_HL / RC_HR: float bits of the largest h in the region of the left / right cut   // words of dist.counts: the fused refresh's "take the general path" flag being collected; "a migrant may need more than one hand-over"

// (sph_step.hip) a few device words -> mapped host memory, then wait for them: counts_host[16 ..), sequence number in counts_host[63]
int publish_and_wait(sph_ctx* c, const void* dev_words, uint32_t n_words);

struct Group;
struct Xfer {
    const void* send[2];  // to left, to right
    size_t send_bytes[2];
    void* recv[2];        // from left, from right.  IN: where the caller has room; OUT: where the data IS once the exchange is through
                          // (a transport may hand back its own buffer -- the peer-mapped inbox the neighbour wrote into -- instead of copying)
    size_t recv_bytes[2];
};

// A Jacobi iteration's ghost exchange with the packing and the unpacking INSIDE the transport's own launches (round 6; the push transport:
// pack + totals + push in ONE launch, wait + totals' sum + unpack in ONE -- four launches per iteration with sweeps B and A instead of six).
// One float per particle at field[slot * stride + off]; the halo members' slots in halo_src = [to the left | to the right neighbour], the
// ghosts' in ghost_dst = [from the left | from the right]; the rank's totals of iteration `iter` are added up in front of the push (what
// block 0 of k_pack_totals does: same reduction, same early exits).
struct FusedField {
    const uint32_t* halo_src;
    uint32_t n_halo[2];
    const uint32_t* ghost_dst;
    uint32_t n_ghost[2];
    bool scatter;   // false: this rank has no ghost slots (ghosts_ok == false): the messages are received and dropped
    float* field;
    int stride, off;
    const void* partials;   // SolverPartial[nparts]
    uint32_t nparts;
    SolverCtrl* ctrl;
    const uint32_t* gate;
    int iter;
};

struct RefreshCounts {
    uint32_t mig[2], halo[2];         // this rank: migrants to / halo members (that stay) towards [left, right]
    uint32_t in_mig[2], in_halo[2];   // the neighbours': migrants for me / their halo members towards me, from [left, right]
    float hreg[2], in_hreg[2];        // largest h among this rank's particles in the region of its [left, right] cut; the neighbours' figure for the same cut
};

struct Comm {
    virtual ~Comm() {}
    // reduce k host values per member element-wise over ALL ranks; every member's row receives the result
    virtual int allreduce_min_f32(Group& G, std::vector<std::vector<float>>& rows) = 0;
    virtual int allreduce_max_i32(Group& G, std::vector<int>& vals) = 0;
    virtual int allreduce_sum_u32(Group& G, std::vector<std::vector<uint32_t>>& rows) = 0;   // element-wise, REBALANCE_BINS words
    // the same over `n` words of DEVICE memory per member (the adaptivity apply's global child counts, sph_adapt.hip); default: through
    // the host, in chunks of what allreduce_sum_u32 takes -- a correctness path; RCCL reduces in place
    virtual int allreduce_sum_u32_dev(Group& G, std::vector<uint32_t*>& bufs, size_t n);
    // every member learns the (to-left, to-right) counts its x-neighbours are about to send it; `status` (optional): this
    // process's status of the phase, replaced by the maximum over ALL ranks in the same round trip
    virtual int neighbour_counts(Group& G, const std::vector<uint32_t>& to_left, const std::vector<uint32_t>& to_right,
                                 std::vector<uint32_t>& from_left, std::vector<uint32_t>& from_right, int* status = nullptr) = 0;
    // device buffers; ordered after everything queued on the members' streams
    virtual int exchange(Group& G, std::vector<Xfer>& x) = 0;
    // element-wise sum of the members' 6 device doubles (solver totals + error flag), result in every member's buffer
    virtual int allreduce_solver(Group& G, int slot) = 0;   // slot 1: the second solve of a chained pair keeps totals of its own
    // a Jacobi iteration's two collectives -- the ghosts' p / rho^2 and the totals of the iteration before -- as ONE call: a
    // transport that pays per launch (RCCL) sends both in one group
    virtual int exchange_and_allreduce_solver(Group& G, std::vector<Xfer>& x, int slot)
    {
        const int rc = exchange(G, x);
        return rc ? rc : allreduce_solver(G, slot);
    }
    // ... and the same with the packing / unpacking of ONE float field inside the transport's launches, where the transport has launches of
    // its own to put them in (the push transport); `bytes` = the largest message of the exchange
    virtual bool can_fuse_iteration(Group&, size_t /*bytes*/) { return false; }
    virtual int exchange_fused(Group&, const FusedField&, int /*slot*/) { return SPH_ERR_UNSUPPORTED; }
    virtual bool host_collectives_wait() const = 0;
    // ONE round trip for a decomposition phase: the class counts the phase's classify kernel left in dist.counts[base ..
    // base + 3] (1 = to the left neighbour, 2 = to the right, 3 = dropped / too narrow; class 0 is derived by the caller) reach
    // the host (counts_host[base ..)), the x-neighbours' counts arrive as from_left / from_right, `red` (optional) is
    // min-reduced element-wise over ALL ranks and `status` (optional) max-reduced -- on the device the status also becomes
    // SPH_ERR_UNSUPPORTED if this rank's class-3 count of the halo phase (base 4) is non-zero (slab narrower than two ghost layers).
    virtual int counts_round(Group& G, int base, std::vector<std::vector<float>>* red, int* status, std::vector<uint32_t>& to_left,
                             std::vector<uint32_t>& to_right, std::vector<uint32_t>& from_left, std::vector<uint32_t>& from_right) = 0;
    // ONE round trip of the fused refresh: the class totals k_slab_scan left in dist.counts[0 .. 4] reach the host, the x-neighbours'
    // (migrants, halo members) arrive, `red` is min-reduced and `status` / `fallback` are max-reduced over ALL ranks
    virtual int refresh_round(Group& G, std::vector<std::vector<float>>* red, int* status, int* fallback, std::vector<RefreshCounts>& rc) = 0;
    // queued, no wait: the maximum over all ranks of the device-side guard word lands in every rank's ctrl->peer_error, which the
    // next publish brings to the host (the step's one agreement on the guards, without a round trip of its own)
    virtual int agree_guards_queued(Group& G) = 0;
};

struct Group {
    std::vector<sph_ctx*> m;
    Comm* comm = nullptr;  // nullptr: one rank, nothing to exchange
    bool multi() const { return comm != nullptr; }
    bool comm_waits() const { return comm && comm->host_collectives_wait(); }   // its host-value collectives end with a wait on the members' streams
};


// (sph_step.hip) wait for everything queued on every member's stream
int wait_all_hinted(Group& G);
int wait_all(Group& G);

// (sph_transport.hip) the transport a step of `c` uses -- RCCL, the thread group or the shared-memory segment its context was
// given -- or an error if a slab context has none; the loopback transport of sph_group_step; what a failing rank tells the others
int comm_for_rank(sph_ctx* c, Comm** out);
Comm* comm_loopback();
void comm_describe(sph_ctx* c, uint32_t* transport, uint32_t* ranks);   // sph_dist_stats::transport / ::comm_ranks
void comm_abandon(sph_ctx* c);

// ---- the members of a group step and what the step driver and the slab maintenance (sph_slabs.hip) share ----
#include <chrono>
#include "sph_internal.hpp"
struct Member {
    sph_ctx* c;
    uint32_t n;  // particles in the arrays (owned + ghosts)
    uint32_t n_sort = 0;   // slots the cell sort looks at (fused slab refresh: n + the slots that left, which it drops); 0: n
    StepP sp;
    SweepArgs a;
    sph_step_stats st;
    std::chrono::steady_clock::time_point wall0;
    float *lv_level = nullptr, *lv_when = nullptr, *lv_pmnew = nullptr;   // level estimation fields whose ghosts are refreshed
};

// (sph_step.hip) collective error check at a wait point; publish ctrl + status of every member and wait (SYNC_*: see the definition)
int agree(Group& G, int local_rc);
enum { SYNC_AGREE = 0, SYNC_DEFER = 1, SYNC_FINAL = 2 };
int sync_ctrl(Group& G, int mode = SYNC_AGREE);
void dbg_sync(sph_ctx* c, const char* what, int id = 0);   // SPH_DEBUG_SYNC: synchronise and name the phase just queued

// (sph_slabs.hip) slab maintenance of a group step, multi-rank only
enum { SC_STAY = 0, SC_HALO_L = 1, SC_HALO_R = 2, SC_MIG_L = 3, SC_MIG_R = 4, SC_GHOST = 5, SC_GONE_FROM = 3 };   // classes of the fused refresh
int ensure_dist_buffers(sph_ctx* c, uint32_t n);
int partition_and_migrate(Group& G, std::vector<Member>& M, std::vector<int>* moved = nullptr, std::vector<std::vector<float>>* red = nullptr);
int rebalance_cuts(Group& G, std::vector<Member>& M, bool* applied);
// ghost width per cut = base_k * H_cut + slack_w, H_cut = largest h within base_k * h_max + slack_w of the cut (both ranks); first ring: 2 H_cut
int build_ghost_layer(Group& G, std::vector<Member>& M, float base_k, float slack_w, float h_max, int status_in);
int slab_refresh_fused(Group& G, std::vector<Member>& M, std::vector<std::vector<float>>& red, float base_k, float slack_k, bool* fused);
// refresh `field` (words floats per particle) of every member's ghosts from their owners; tot_slot >= 0: the all-reduce of that slot's
// solver totals rides in the same call
// `sel2` (one word per particle, like `sel` then): a second field in the SAME exchange -- the level estimation's (level, when) pair
// `stride` / `off` (floats): where the field's words sit inside a wider per-particle record (default: a plain array of `words` floats)
// `totals` (with tot_slot >= 0): this rank's solver totals of that iteration are added up by block 0 of the packing launch itself
struct TotalsJob {
    int iter, residual_density;
    float max_avg_error;
    uint32_t max_iters;
};
int refresh_ghosts(Group& G, std::vector<Member>& M, float* (*sel)(Member&), int words, const char* what, int tot_slot = -1, float* (*sel2)(Member&) = nullptr,
                   int stride = 0, int off = 0, const TotalsJob* totals = nullptr);
// after the cell sort: slot maps of halo members and ghosts, ownership flags, the split sweep's edge bytes, the ghosts' {x, y, a^p} records
int slab_maps_after_sort(sph_ctx* c, uint32_t n, bool pre, hipStream_t s);
// m / rho of the ghosts from their refreshed densities
void slab_ghost_mrho(sph_ctx* c, const SweepArgs& a);

