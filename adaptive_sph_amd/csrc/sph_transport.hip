// Transports between the ranks of a slab decomposition, behind one interface (Comm, sph_dist.hpp):
//   LocalComm    k contexts of ONE process, device copies ordered by events (sph_group_step: the verification form, and what a
//                single-process host drives several GPUs with)
//   RcclComm     one rank per process, ncclSend / ncclRecv grouped per x-neighbour pair over its xGMI link, ncclAllReduce for scalars
//   ThreadComm   all ranks in this process, one HOST THREAD per rank, collectives as rendezvous in host memory: the per-rank driver
//                code with everything that would hang RCCL turned into an error
//   ShmComm      the same protocol between PROCESSES of one node, rendezvous and staging in a POSIX shared-memory segment: the
//                functional transport where RCCL cannot serve the launch (several ranks on one GPU), and the multi-process check
// The reference has no counterpart (its only parallelism is rayon inside one process, concurrency.rs:110-204).
#include <atomic>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <array>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <vector>

#include <cerrno>
#include <fcntl.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>

#include "sph_dist.hpp"


// counts_round (RCCL): out[0] = to_left, out[1] = to_right, out[2] = out[3] = 0 (received below), out[4] = status,
// out[5 .. 8] = this rank's four class counts, out[9] = its "a migrant is far" word
__global__ void k_counts_stage(const uint32_t* __restrict__ counts, uint32_t* __restrict__ out, uint32_t status_in, int narrow_is_error,
                               const uint32_t* __restrict__ far_word)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    out[9] = *far_word;
    out[0] = counts[1];
    out[1] = counts[2];
    out[2] = out[3] = 0u;
    uint32_t st = status_in;
    if (narrow_is_error && counts[3] != 0u && st < (uint32_t)SPH_ERR_UNSUPPORTED) st = (uint32_t)SPH_ERR_UNSUPPORTED;
    out[4] = st;
    for (int k = 0; k < 4; k++) out[5 + k] = counts[k];
}


// Measurement hook of the loopback transport (scripts/gpu_split_sweep_timing.py): SPH_DEBUG_COMM_DELAY_US=<us> makes every ghost
// exchange and every all-reduce of the solver totals occupy its stream for that long before it completes -- a stand-in for the
// latency of a collective between GPUs, which one GPU cannot produce
__global__ void k_spin_us(uint32_t us)
{
    const uint64_t t0 = wall_clock64();   // 100 MHz
    while (wall_clock64() - t0 < (uint64_t)us * 100ull) __builtin_amdgcn_s_sleep(16);
}
// loopback all-reduce of the solver totals without a host wait (LocalComm::allreduce_solver)
__global__ void k_tot_publish(const double* __restrict__ tot, double* __restrict__ row)
{
    if (threadIdx.x < 6) row[threadIdx.x] = tot[threadIdx.x];
    __threadfence_system();
}
__global__ void k_tot_sum(double* __restrict__ tot, const double* __restrict__ table, int n)
{
    if (threadIdx.x >= 6) return;
    double s = 0.0;
    for (int j = 0; j < n; j++) s += ((const volatile double*)table)[8 * j + threadIdx.x];   // rank order
    tot[threadIdx.x] = s;
}
static void debug_comm_delay(sph_ctx* c)
{
    const int us = c->opt.comm_delay_us;
    if (us > 0) hipLaunchKernelGGL(k_spin_us, dim3(1), dim3(1), 0, c->stream, (uint32_t)us);
}

int Comm::allreduce_sum_u32_dev(Group& G, std::vector<uint32_t*>& bufs, size_t n)
{
    const size_t CH = 4096;   // (the shared-memory transport's and the RCCL staging's limit per call)
    std::vector<std::vector<uint32_t>> rows(G.m.size(), std::vector<uint32_t>(CH));
    for (size_t o = 0; o < n; o += CH) {
        const size_t len = std::min(CH, n - o);
        for (size_t i = 0; i < G.m.size(); i++) {
            sph_ctx* c = G.m[i];
            HIPCHK(c, hipSetDevice(c->device));
            rows[i].assign(CH, 0u);
            HIPCHK(c, hipMemcpy(rows[i].data(), bufs[i] + o, len * 4, hipMemcpyDeviceToHost));
        }
        int rc = allreduce_sum_u32(G, rows);
        if (rc) return rc;
        for (size_t i = 0; i < G.m.size(); i++) {
            sph_ctx* c = G.m[i];
            HIPCHK(c, hipSetDevice(c->device));
            HIPCHK(c, hipMemcpy(bufs[i] + o, rows[i].data(), len * 4, hipMemcpyHostToDevice));
        }
    }
    return SPH_OK;
}

// ---- loopback: all ranks are contexts of this process ---------------------------------------------
struct LocalComm : Comm {
    bool host_collectives_wait() const override { return false; }   // plain host arithmetic on values the caller already has
    int allreduce_min_f32(Group& G, std::vector<std::vector<float>>& rows) override
    {
        for (size_t k = 0; k < rows[0].size(); k++) {
            float v = rows[0][k];
            for (auto& r : rows) v = fminf(v, r[k]);
            for (auto& r : rows) r[k] = v;
        }
        return SPH_OK;
    }
    int allreduce_max_i32(Group& G, std::vector<int>& vals) override
    {
        int v = 0;
        for (int x : vals) v = x > v ? x : v;
        for (int& x : vals) x = v;
        return SPH_OK;
    }
    int allreduce_sum_u32(Group& G, std::vector<std::vector<uint32_t>>& rows) override
    {
        for (size_t k = 0; k < rows[0].size(); k++) {
            uint32_t v = 0;
            for (auto& r : rows) v += r[k];
            for (auto& r : rows) r[k] = v;
        }
        return SPH_OK;
    }
    int neighbour_counts(Group& G, const std::vector<uint32_t>& tl, const std::vector<uint32_t>& tr, std::vector<uint32_t>& fl,
                         std::vector<uint32_t>& fr, int* status) override
    {
        (void)status;   // one process: its own status is the maximum
        const size_t n = G.m.size();
        for (size_t i = 0; i < n; i++) {
            fl[i] = i > 0 ? tr[i - 1] : 0;
            fr[i] = i + 1 < n ? tl[i + 1] : 0;
        }
        return SPH_OK;
    }
    // SPH_LOOPBACK_SYNC=1: the exchanges and the solver all-reduce wait on the host (the first form of this transport, kept as the
    // reference the event-ordered form is tested against).  Default: no host wait -- the copies are ordered by events between the
    // members' streams, the totals meet in mapped host memory: what a host that drives k GPUs from one process runs.
    static bool host_synchronous(const Group& G) { return G.m[0]->opt.loopback_sync != 0; }
    int exchange(Group& G, std::vector<Xfer>& x) override
    {
        const size_t n = G.m.size();
        // the RCCL transport pairs every ncclSend with an ncclRecv of the same size: hold the loopback to the same rule,
        // so that the single-GPU verification also proves the pairing
        for (size_t i = 0; i + 1 < n; i++) {
            if (x[i].send_bytes[1] != x[i + 1].recv_bytes[0] || x[i + 1].send_bytes[0] != x[i].recv_bytes[1])
                return G.m[i]->fail(SPH_ERR_DEVICE, "halo exchange sizes of ranks %zu and %zu do not pair up (%zu->%zu, %zu<-%zu)", i, i + 1,
                                    x[i].send_bytes[1], x[i + 1].recv_bytes[0], x[i].recv_bytes[1], x[i + 1].send_bytes[0]);
        }
        if (n && (x[0].send_bytes[0] || x[0].recv_bytes[0] || x[n - 1].send_bytes[1] || x[n - 1].recv_bytes[1]))
            return G.m[0]->fail(SPH_ERR_DEVICE, "halo exchange across the outer edge of the slab row");
        const bool sync = host_synchronous(G);
        int rc;
        if (sync) {
            for (auto c : G.m) debug_comm_delay(c);
            if ((rc = wait_all(G))) return rc;
        } else {
            for (auto c : G.m) {   // "my staging buffers are packed"
                HIPCHK(c, hipSetDevice(c->device));
                HIPCHK(c, hipEventRecord(c->dist.ev_pack, c->stream));
            }
        }
        for (size_t i = 0; i < n; i++) {
            sph_ctx* c = G.m[i];
            c->dist.stat_exchanges++;
            c->dist.stat_bytes_sent += x[i].send_bytes[0] + x[i].send_bytes[1];
            c->dist.stat_bytes_recv += x[i].recv_bytes[0] + x[i].recv_bytes[1];
            HIPCHK(c, hipSetDevice(c->device));
            const bool from_l = i > 0 && x[i].recv_bytes[0], from_r = i + 1 < n && x[i].recv_bytes[1];
            if (!sync) {
                if (from_l) HIPCHK(c, hipStreamWaitEvent(c->stream, G.m[i - 1]->dist.ev_pack, 0));
                if (from_r) HIPCHK(c, hipStreamWaitEvent(c->stream, G.m[i + 1]->dist.ev_pack, 0));
                if (from_l || from_r) debug_comm_delay(c);
            }
            if (from_l) HIPCHK(c, hipMemcpyAsync(x[i].recv[0], x[i - 1].send[1], x[i].recv_bytes[0], hipMemcpyDefault, c->stream));
            if (from_r) HIPCHK(c, hipMemcpyAsync(x[i].recv[1], x[i + 1].send[0], x[i].recv_bytes[1], hipMemcpyDefault, c->stream));
            if (!sync) HIPCHK(c, hipEventRecord(c->dist.ev_copied, c->stream));
        }
        if (sync) return wait_all(G);  // senders may reuse their staging buffers afterwards
        // a sender packs again (always on its main stream) only after the neighbours' copies out of its staging buffers
        for (size_t i = 0; i < n; i++) {
            sph_ctx* c = G.m[i];
            HIPCHK(c, hipSetDevice(c->device));
            if (i > 0 && x[i].send_bytes[0]) HIPCHK(c, hipStreamWaitEvent(c->stream, G.m[i - 1]->dist.ev_copied, 0));
            if (i + 1 < n && x[i].send_bytes[1]) HIPCHK(c, hipStreamWaitEvent(c->stream, G.m[i + 1]->dist.ev_copied, 0));
        }
        return SPH_OK;
    }
    int counts_round(Group& G, int base, std::vector<std::vector<float>>* red, int* status, std::vector<uint32_t>& tl, std::vector<uint32_t>& tr,
                     std::vector<uint32_t>& fl, std::vector<uint32_t>& fr) override
    {
        int rc = wait_all(G);
        if (rc) return rc;
        for (size_t i = 0; i < G.m.size(); i++) {
            sph_ctx* c = G.m[i];
            HIPCHK(c, hipSetDevice(c->device));
            HIPCHK(c, hipMemcpy(c->dist.counts_host + base, c->dist.counts.as<uint32_t>() + base, 16, hipMemcpyDeviceToHost));
            HIPCHK(c, hipMemcpy(c->dist.counts_host + RC_FAR, c->dist.counts.as<uint32_t>() + RC_FAR, 4, hipMemcpyDeviceToHost));
            tl[i] = c->dist.counts_host[base + 1];
            tr[i] = c->dist.counts_host[base + 2];
            if (status && base == 4 && c->dist.counts_host[base + 3] && *status < SPH_ERR_UNSUPPORTED) *status = SPH_ERR_UNSUPPORTED;
        }
        if (red) allreduce_min_f32(G, *red);
        return neighbour_counts(G, tl, tr, fl, fr, nullptr);
    }
    int refresh_round(Group& G, std::vector<std::vector<float>>* red, int* status, int* fallback, std::vector<RefreshCounts>& rcs) override
    {
        (void)status;   // one process: its own status is the maximum
        int rc = wait_all(G);
        if (rc) return rc;
        const size_t n = G.m.size();
        for (size_t i = 0; i < n; i++) {
            sph_ctx* c = G.m[i];
            HIPCHK(c, hipSetDevice(c->device));
            uint32_t w[7];
            HIPCHK(c, hipMemcpy(w, c->dist.counts.p, 28, hipMemcpyDeviceToHost));
            rcs[i].halo[0] = w[0];
            rcs[i].halo[1] = w[1];
            rcs[i].mig[0] = w[2];
            rcs[i].mig[1] = w[3];
            if (w[4]) *fallback = 1;
            memcpy(rcs[i].hreg, w + 5, 8);
        }
        for (size_t i = 0; i < n; i++) {
            rcs[i].in_mig[0] = i > 0 ? rcs[i - 1].mig[1] : 0;
            rcs[i].in_halo[0] = i > 0 ? rcs[i - 1].halo[1] : 0;
            rcs[i].in_hreg[0] = i > 0 ? rcs[i - 1].hreg[1] : 0.f;
            rcs[i].in_mig[1] = i + 1 < n ? rcs[i + 1].mig[0] : 0;
            rcs[i].in_halo[1] = i + 1 < n ? rcs[i + 1].halo[0] : 0;
            rcs[i].in_hreg[1] = i + 1 < n ? rcs[i + 1].hreg[0] : 0.f;
        }
        if (red) allreduce_min_f32(G, *red);
        return SPH_OK;
    }
    int agree_guards_queued(Group&) override { return SPH_OK; }   // one process: sync_ctrl sees every member's guard word
    int allreduce_solver(Group& G, int slot) override
    {
        const size_t n = G.m.size();
        if (host_synchronous(G)) {
            for (auto c : G.m) debug_comm_delay(c);
            int rc = wait_all(G);
            if (rc) return rc;
            double tot[6] = {0, 0, 0, 0, 0, 0};
            std::vector<std::array<double, 6>> rows(n);
            for (size_t i = 0; i < n; i++) {
                HIPCHK(G.m[i], hipMemcpy(rows[i].data(), G.m[i]->dist.solver_tot.as<double>() + 8 * slot, 48, hipMemcpyDeviceToHost));
                for (int k = 0; k < 6; k++) tot[k] += rows[i][k];
            }
            for (auto c : G.m) HIPCHK(c, hipMemcpy(c->dist.solver_tot.as<double>() + 8 * slot, tot, 48, hipMemcpyHostToDevice));
            return SPH_OK;
        }
        // every member publishes its six doubles into its row of the group's table (mapped host memory: any device reaches it),
        // then adds up all rows in rank order -- the sum the host-synchronous form computes.  Two tables, alternating: a member is
        // at most one all-reduce ahead of the slowest (it needed everybody's row of the previous one).
        sph_ctx* c0 = G.m[0];
        if (!c0->dist.gtot) {
            HIPCHK(c0, hipHostMalloc((void**)&c0->dist.gtot, 2 * 64 * 8 * sizeof(double), hipHostMallocMapped | hipHostMallocPortable));
            memset(c0->dist.gtot, 0, 2 * 64 * 8 * sizeof(double));
        }
        if (n > 64) return c0->fail(SPH_ERR_INVALID_ARGUMENT, "loopback group of %zu members", n);
        double* table = c0->dist.gtot + (size_t)(c0->dist.gtot_seq++ & 1u) * 64 * 8;
        for (size_t i = 0; i < n; i++) {
            sph_ctx* c = G.m[i];
            HIPCHK(c, hipSetDevice(c->device));
            hipLaunchKernelGGL(k_tot_publish, dim3(1), dim3(64), 0, c->stream, c->dist.solver_tot.as<double>() + 8 * slot, table + 8 * i);
            HIPCHK(c, hipEventRecord(c->dist.ev_tot, c->stream));
        }
        for (size_t i = 0; i < n; i++) {
            sph_ctx* c = G.m[i];
            HIPCHK(c, hipSetDevice(c->device));
            for (size_t j = 0; j < n; j++)
                if (j != i) HIPCHK(c, hipStreamWaitEvent(c->stream, G.m[j]->dist.ev_tot, 0));
            debug_comm_delay(c);
            hipLaunchKernelGGL(k_tot_sum, dim3(1), dim3(64), 0, c->stream, c->dist.solver_tot.as<double>() + 8 * slot, (const double*)table, (int)n);
        }
        return SPH_OK;
    }
};

// ---- RCCL over xGMI: one rank per process ------------------------------------------------------------
#define NCCLCHK(ctx, call)                                                                               \
    do {                                                                                                 \
        ncclResult_t r_ = (call);                                                                        \
        if (r_ != ncclSuccess) return (ctx)->fail(SPH_ERR_DEVICE, "%s failed: %s", #call, ncclGetErrorString(r_)); \
    } while (0)

struct RcclComm : Comm {
    bool host_collectives_wait() const override { return true; }    // publish_and_wait behind the collective
    // small device scratch for host-value collectives
    static int host_allreduce(sph_ctx* c, void* host, size_t bytes, size_t count, ncclDataType_t dt, ncclRedOp_t op)
    {
        ncclComm_t nc = (ncclComm_t)c->dist.nccl;
        void* d = c->dist.counts.as<uint32_t>() + 16;  // scratch area behind the counters
        // staged through pinned memory: copies from / to pageable vectors are synchronous and several times slower
        uint8_t* stage = (uint8_t*)c->dist.counts_host + 64;
        if (bytes > 128) return c->fail(SPH_ERR_INVALID_ARGUMENT, "host all-reduce of %zu bytes", bytes);
        memcpy(stage, host, bytes);
        HIPCHK(c, hipMemcpyAsync(d, stage, bytes, hipMemcpyHostToDevice, c->stream));
        c->dist.stat_allreduces++;
        {
            ProfScope ps(&c->prof, "rccl_allreduce", c->stream);
            NCCLCHK(c, ncclAllReduce(d, d, count, dt, op, nc, c->stream));
        }
        int rc = publish_and_wait(c, d, (uint32_t)((bytes + 3) / 4));
        if (rc) return rc;
        memcpy(host, stage, bytes);
        return SPH_OK;
    }
    int allreduce_min_f32(Group& G, std::vector<std::vector<float>>& rows) override
    {
        return host_allreduce(G.m[0], rows[0].data(), rows[0].size() * 4, rows[0].size(), ncclFloat32, ncclMin);
    }
    int allreduce_max_i32(Group& G, std::vector<int>& vals) override
    {
        return host_allreduce(G.m[0], vals.data(), vals.size() * 4, vals.size(), ncclInt32, ncclMax);
    }
    int allreduce_sum_u32(Group& G, std::vector<std::vector<uint32_t>>& rows) override
    {
        sph_ctx* c = G.m[0];
        ncclComm_t nc = (ncclComm_t)c->dist.nccl;
        HIPCHK(c, c->dist.hist.ensure(rows[0].size() * 4));
        void* d = c->dist.hist.p;   // the histogram's own device buffer is the staging area
        HIPCHK(c, hipMemcpyAsync(d, rows[0].data(), rows[0].size() * 4, hipMemcpyHostToDevice, c->stream));
        NCCLCHK(c, ncclAllReduce(d, d, rows[0].size(), ncclUint32, ncclSum, nc, c->stream));
        HIPCHK(c, hipMemcpyAsync(rows[0].data(), d, rows[0].size() * 4, hipMemcpyDeviceToHost, c->stream));
        return wait_stream(c);
    }
    int allreduce_sum_u32_dev(Group& G, std::vector<uint32_t*>& bufs, size_t n) override
    {
        sph_ctx* c = G.m[0];
        c->dist.stat_allreduces++;
        NCCLCHK(c, ncclAllReduce(bufs[0], bufs[0], n, ncclUint32, ncclSum, (ncclComm_t)c->dist.nccl, c->stream));
        return wait_stream(c);
    }
    int neighbour_counts(Group& G, const std::vector<uint32_t>& tl, const std::vector<uint32_t>& tr, std::vector<uint32_t>& fl,
                         std::vector<uint32_t>& fr, int* status) override
    {
        sph_ctx* c = G.m[0];
        ncclComm_t nc = (ncclComm_t)c->dist.nccl;
        const int r = c->dist.rank, nr = c->dist.nranks;
        uint32_t* d = c->dist.counts.as<uint32_t>() + 16;  // [0]=to_left [1]=to_right [2]=from_left [3]=from_right [4]=status
        uint32_t* h = (uint32_t*)((uint8_t*)c->dist.counts_host + 64);   // pinned staging
        h[0] = tl[0];
        h[1] = tr[0];
        h[2] = h[3] = 0;
        h[4] = status ? (uint32_t)*status : 0u;
        HIPCHK(c, hipMemcpyAsync(d, h, 20, hipMemcpyHostToDevice, c->stream));
        if (status) {
            c->dist.stat_allreduces++;
            NCCLCHK(c, ncclAllReduce(d + 4, d + 4, 1, ncclUint32, ncclMax, nc, c->stream));
        }
        NCCLCHK(c, ncclGroupStart());
        if (r > 0) {
            NCCLCHK(c, ncclSend(d + 0, 1, ncclUint32, r - 1, nc, c->stream));
            NCCLCHK(c, ncclRecv(d + 2, 1, ncclUint32, r - 1, nc, c->stream));
        }
        if (r + 1 < nr) {
            NCCLCHK(c, ncclSend(d + 1, 1, ncclUint32, r + 1, nc, c->stream));
            NCCLCHK(c, ncclRecv(d + 3, 1, ncclUint32, r + 1, nc, c->stream));
        }
        NCCLCHK(c, ncclGroupEnd());
        int rc = publish_and_wait(c, d, 5);
        if (rc) return rc;
        // (publish_and_wait lands in counts_host[16 ..), the staging words above sit at counts_host[16 ..) too: same words)
        fl[0] = r > 0 ? h[2] : 0;
        fr[0] = r + 1 < nr ? h[3] : 0;
        if (status) *status = (int)h[4];
        return SPH_OK;
    }
    int counts_round(Group& G, int base, std::vector<std::vector<float>>* red, int* status, std::vector<uint32_t>& tl, std::vector<uint32_t>& tr,
                     std::vector<uint32_t>& fl, std::vector<uint32_t>& fr) override
    {
        sph_ctx* c = G.m[0];
        ncclComm_t nc = (ncclComm_t)c->dist.nccl;
        const int r = c->dist.rank, nr = c->dist.nranks;
        // device scratch behind the counters: [0 .. 7] the min-reduced floats, [8 .. 17] k_counts_stage's words
        uint32_t* d = c->dist.counts.as<uint32_t>() + 16;
        uint32_t* h = (uint32_t*)((uint8_t*)c->dist.counts_host + 64);   // pinned staging / publish destination (same words)
        const size_t nred = red ? (*red)[0].size() : 0;
        if (nred > 8) return c->fail(SPH_ERR_INVALID_ARGUMENT, "counts_round: %zu reduced values", nred);
        if (nred) {
            memcpy(h, (*red)[0].data(), nred * 4);
            HIPCHK(c, hipMemcpyAsync(d, h, nred * 4, hipMemcpyHostToDevice, c->stream));
            c->dist.stat_allreduces++;
            ProfScope ps(&c->prof, "rccl_allreduce", c->stream);
            NCCLCHK(c, ncclAllReduce(d, d, nred, ncclFloat32, ncclMin, nc, c->stream));
        }
        hipLaunchKernelGGL(k_counts_stage, dim3(1), dim3(64), 0, c->stream, c->dist.counts.as<uint32_t>() + base, d + 8, status ? (uint32_t)*status : 0u,
                           base == 4 ? 1 : 0, c->dist.counts.as<uint32_t>() + RC_FAR);
        if (status) {
            c->dist.stat_allreduces++;
            ProfScope ps(&c->prof, "rccl_allreduce", c->stream);
            NCCLCHK(c, ncclAllReduce(d + 12, d + 12, 1, ncclUint32, ncclMax, nc, c->stream));
        }
        {
            ProfScope ps(&c->prof, "rccl_sendrecv", c->stream);
            NCCLCHK(c, ncclGroupStart());
            if (r > 0) {
                NCCLCHK(c, ncclSend(d + 8, 1, ncclUint32, r - 1, nc, c->stream));
                NCCLCHK(c, ncclRecv(d + 10, 1, ncclUint32, r - 1, nc, c->stream));
            }
            if (r + 1 < nr) {
                NCCLCHK(c, ncclSend(d + 9, 1, ncclUint32, r + 1, nc, c->stream));
                NCCLCHK(c, ncclRecv(d + 11, 1, ncclUint32, r + 1, nc, c->stream));
            }
            NCCLCHK(c, ncclGroupEnd());
        }
        int rc = publish_and_wait(c, d, 18);
        if (rc) return rc;
        c->dist.counts_host[RC_FAR] = h[17];
        for (size_t k = 0; k < nred; k++) memcpy(&(*red)[0][k], &h[k], 4);
        tl[0] = h[8];
        tr[0] = h[9];
        fl[0] = r > 0 ? h[10] : 0;
        fr[0] = r + 1 < nr ? h[11] : 0;
        if (status) *status = (int)h[12];
        for (int k = 0; k < 4; k++) c->dist.counts_host[base + k] = h[13 + k];
        return SPH_OK;
    }
    int refresh_round(Group& G, std::vector<std::vector<float>>* red, int* status, int* fallback, std::vector<RefreshCounts>& rcs) override
    {
        sph_ctx* c = G.m[0];
        ncclComm_t nc = (ncclComm_t)c->dist.nccl;
        const int r = c->dist.rank, nr = c->dist.nranks;
        // k_slab_classify's last block staged the round behind the counters: d[0 .. 7] header values, d[8] = -status,
        // d[9] = -"general path" (ten floats, ONE min all-reduce), d[10 .. 12] / d[13 .. 15] my (migrants, halo members, largest h in the
        // cut's region) for the left / right neighbour, d[16 .. 18] / d[19 .. 21] theirs
        uint32_t* d = c->dist.counts.as<uint32_t>() + 16;
        const uint32_t* h = (const uint32_t*)((const uint8_t*)c->dist.counts_host + 64);   // publish destination
        const size_t nred = red ? (*red)[0].size() : 0;
        if (nred != 8) return c->fail(SPH_ERR_INVALID_ARGUMENT, "refresh_round: %zu reduced values", nred);
        {
            c->dist.stat_allreduces++;
            ProfScope ps(&c->prof, "rccl_allreduce", c->stream);
            NCCLCHK(c, ncclAllReduce(d, d, 10, ncclFloat32, ncclMin, nc, c->stream));
        }
        if (nr > 1) {
            ProfScope ps(&c->prof, "rccl_sendrecv", c->stream);
            NCCLCHK(c, ncclGroupStart());
            if (r > 0) {
                NCCLCHK(c, ncclSend(d + 10, 3, ncclUint32, r - 1, nc, c->stream));
                NCCLCHK(c, ncclRecv(d + 16, 3, ncclUint32, r - 1, nc, c->stream));
            }
            if (r + 1 < nr) {
                NCCLCHK(c, ncclSend(d + 13, 3, ncclUint32, r + 1, nc, c->stream));
                NCCLCHK(c, ncclRecv(d + 19, 3, ncclUint32, r + 1, nc, c->stream));
            }
            NCCLCHK(c, ncclGroupEnd());
        }
        int rc = publish_and_wait(c, d, 22);
        if (rc) return rc;
        float f[10];
        memcpy(f, h, sizeof f);
        for (size_t k = 0; k < 8; k++) (*red)[0][k] = f[k];
        *status = (int)-f[8];
        if (f[9] < 0.f) *fallback = 1;
        RefreshCounts& o = rcs[0];
        o.mig[0] = h[10];
        o.halo[0] = h[11];
        memcpy(&o.hreg[0], &h[12], 4);
        o.mig[1] = h[13];
        o.halo[1] = h[14];
        memcpy(&o.hreg[1], &h[15], 4);
        o.in_mig[0] = r > 0 ? h[16] : 0;
        o.in_halo[0] = r > 0 ? h[17] : 0;
        o.in_mig[1] = r + 1 < nr ? h[19] : 0;
        o.in_halo[1] = r + 1 < nr ? h[20] : 0;
        o.in_hreg[0] = o.in_hreg[1] = 0.f;
        if (r > 0) memcpy(&o.in_hreg[0], &h[18], 4);
        if (r + 1 < nr) memcpy(&o.in_hreg[1], &h[21], 4);
        return SPH_OK;
    }
    int agree_guards_queued(Group& G) override
    {
        sph_ctx* c = G.m[0];
        // out of place, straight from the guard word into the control block: max over the ranks of status.error (a rank whose
        // guard fired during a solve also raised peer_error through the totals -- its own status.error is in this maximum)
        c->dist.stat_allreduces++;
        {
            ProfScope ps(&c->prof, "rccl_allreduce", c->stream);
            NCCLCHK(c, ncclAllReduce(&c->status.as<DeviceStatus>()->error, &c->ctrl.as<SolverCtrl>()->peer_error, 1, ncclUint32, ncclMax,
                                     (ncclComm_t)c->dist.nccl, c->stream));
        }
        return SPH_OK;
    }
    int exchange(Group& G, std::vector<Xfer>& x) override
    {
        sph_ctx* c = G.m[0];
        ncclComm_t nc = (ncclComm_t)c->dist.nccl;
        const int r = c->dist.rank, nr = c->dist.nranks;
        c->dist.stat_exchanges++;
        c->dist.stat_bytes_sent += (r > 0 ? x[0].send_bytes[0] : 0) + (r + 1 < nr ? x[0].send_bytes[1] : 0);
        c->dist.stat_bytes_recv += (r > 0 ? x[0].recv_bytes[0] : 0) + (r + 1 < nr ? x[0].recv_bytes[1] : 0);
        ProfScope ps(&c->prof, "rccl_sendrecv", c->stream);
        NCCLCHK(c, ncclGroupStart());
        if (r > 0) {
            if (x[0].send_bytes[0]) NCCLCHK(c, ncclSend(x[0].send[0], x[0].send_bytes[0], ncclChar, r - 1, nc, c->stream));
            if (x[0].recv_bytes[0]) NCCLCHK(c, ncclRecv(x[0].recv[0], x[0].recv_bytes[0], ncclChar, r - 1, nc, c->stream));
        }
        if (r + 1 < nr) {
            if (x[0].send_bytes[1]) NCCLCHK(c, ncclSend(x[0].send[1], x[0].send_bytes[1], ncclChar, r + 1, nc, c->stream));
            if (x[0].recv_bytes[1]) NCCLCHK(c, ncclRecv(x[0].recv[1], x[0].recv_bytes[1], ncclChar, r + 1, nc, c->stream));
        }
        NCCLCHK(c, ncclGroupEnd());
        return SPH_OK;
    }
    // The solver totals over RCCL are an ALL-GATHER into the rank's table (row p = rank p's six doubles; this rank's own row stays in
    // solver_tot), summed in rank order by whoever reads them (SolveP::tot_table, sph_sweeps.hip) -- never reduced in place.
    int allreduce_solver(Group& G, int slot) override
    {
        std::vector<Xfer> none(1);
        memset(&none[0], 0, sizeof(Xfer));
        return exchange_and_allreduce_solver(G, none, slot);
    }
    // ONE grouped launch per Jacobi iteration: the x-neighbours' ghost values, and this rank's six totals to every rank / every
    // rank's totals into a table (an all-gather by point-to-point messages); a one-block kernel then adds the rows in rank order --
    // the same sum on every rank, as the all-reduce gave, without its launch.  Messages of 48 bytes: what is saved is a launch
    // latency per iteration, which at 8 ranks and ~1M particles per rank is what the iteration's time is made of (DESIGN.md multi-GPU).
    int exchange_and_allreduce_solver(Group& G, std::vector<Xfer>& x, int slot) override
    {
        sph_ctx* c = G.m[0];
        ncclComm_t nc = (ncclComm_t)c->dist.nccl;
        const int r = c->dist.rank, nr = c->dist.nranks;
        double* tot = c->dist.solver_tot.as<double>() + 8 * slot;
        double* table = c->dist.tot_table.as<double>() + (size_t)slot * nr * 8;
        c->dist.stat_exchanges++;
        c->dist.stat_bytes_sent += (r > 0 ? x[0].send_bytes[0] : 0) + (r + 1 < nr ? x[0].send_bytes[1] : 0) + (size_t)(nr - 1) * 48;
        c->dist.stat_bytes_recv += (r > 0 ? x[0].recv_bytes[0] : 0) + (r + 1 < nr ? x[0].recv_bytes[1] : 0) + (size_t)(nr - 1) * 48;
        if (nr > 1) {
            ProfScope ps(&c->prof, "rccl_sendrecv", c->stream);
            NCCLCHK(c, ncclGroupStart());
            if (r > 0) {
                if (x[0].send_bytes[0]) NCCLCHK(c, ncclSend(x[0].send[0], x[0].send_bytes[0], ncclChar, r - 1, nc, c->stream));
                if (x[0].recv_bytes[0]) NCCLCHK(c, ncclRecv(x[0].recv[0], x[0].recv_bytes[0], ncclChar, r - 1, nc, c->stream));
            }
            if (r + 1 < nr) {
                if (x[0].send_bytes[1]) NCCLCHK(c, ncclSend(x[0].send[1], x[0].send_bytes[1], ncclChar, r + 1, nc, c->stream));
                if (x[0].recv_bytes[1]) NCCLCHK(c, ncclRecv(x[0].recv[1], x[0].recv_bytes[1], ncclChar, r + 1, nc, c->stream));
            }
            for (int p = 0; p < nr; p++) {
                if (p == r) continue;
                NCCLCHK(c, ncclSend(tot, 6, ncclFloat64, p, nc, c->stream));
                NCCLCHK(c, ncclRecv(table + 8 * p, 6, ncclFloat64, p, nc, c->stream));
            }
            NCCLCHK(c, ncclGroupEnd());
        }
        return SPH_OK;   // (no kernel adds the rows up: the readers do)
    }
};

// ---- threads: one HOST THREAD per rank, all ranks in this process (and on whatever devices their contexts name) -----------------
// The verification transport for the per-rank driver code: every rank runs sph_step on its own thread with a group of ONE member,
// exactly as a rank of the RCCL transport does -- its own view of the counts, its own branches (an exchange only where it has
// something to send or receive, ...) -- and the collectives are rendezvous in host memory.  A collective that not every rank
// enters, or a send that no receive of the same size matches, is what would hang RCCL: here it is a time-out / an error message.
struct ThreadGroup {
    int n = 0;
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    uint64_t gen = 0;
    std::atomic<bool> broken{false};
    // a ghost / migrant exchange is point to point, as ncclSend / ncclRecv are: rank r meets only the x-neighbours it sends to or
    // receives from (a rank with nothing for either neighbour does not enter at all).  One channel per adjacent pair (r, r + 1).
    struct PairChan {
        std::mutex mu;
        std::condition_variable cv;
        int arrived = 0;
        uint64_t gen = 0;
        Xfer xf[2];   // [0] the lower rank's, [1] the upper rank's
    };
    std::unique_ptr<PairChan[]> pair;
    std::vector<std::array<double, 6>> tot;
    std::vector<std::vector<float>> f32rows;
    std::vector<std::vector<uint32_t>> u32rows;
    std::vector<int> i32vals;
    std::vector<std::array<uint32_t, 8>> words;
    std::vector<int> op;          // which collective each rank is in (a mismatch is reported, not waited out)
    explicit ThreadGroup(int k) : n(k), pair(new PairChan[(size_t)std::max(k - 1, 1)]), tot(k), f32rows(k), u32rows(k), i32vals(k), words(k), op(k) {}
    // the two ranks of a channel meet; false: the other one did not come (60 s) or somebody left with an error
    bool pair_barrier(PairChan& ch)
    {
        std::unique_lock<std::mutex> lk(ch.mu);
        if (broken) return false;
        const uint64_t g = ch.gen;
        if (++ch.arrived == 2) {
            ch.arrived = 0;
            ch.gen++;
            ch.cv.notify_all();
            return true;
        }
        const auto t_end = std::chrono::steady_clock::now() + std::chrono::seconds(60);
        while (ch.gen == g && !broken) {
            if (ch.cv.wait_until(lk, std::min(t_end, std::chrono::steady_clock::now() + std::chrono::milliseconds(50))) == std::cv_status::timeout &&
                std::chrono::steady_clock::now() >= t_end) {
                broken = true;
                break;
            }
        }
        if (broken) ch.cv.notify_all();
        return ch.gen != g && !broken;
    }
    // all ranks meet; false: somebody did not come (60 s) or left with an error
    bool barrier()
    {
        std::unique_lock<std::mutex> lk(mu);
        if (broken) return false;
        const uint64_t g = gen;
        if (++arrived == n) {
            arrived = 0;
            gen++;
            cv.notify_all();
            return true;
        }
        if (!cv.wait_for(lk, std::chrono::seconds(60), [&] { return gen != g || broken; })) broken = true;
        if (broken) cv.notify_all();
        return !broken;
    }
    void abandon()
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            broken = true;
            cv.notify_all();
        }
        for (int i = 0; i + 1 < n; i++) {   // (the pair waits poll `broken` every 50 ms as well)
            std::lock_guard<std::mutex> lk(pair[(size_t)i].mu);
            pair[(size_t)i].cv.notify_all();
        }
    }
};

struct ThreadComm : Comm {
    bool host_collectives_wait() const override { return true; }
    static ThreadGroup* grp(Group& G) { return (ThreadGroup*)G.m[0]->dist.tgroup; }
    // publish -> everybody is there -> consume -> everybody is done (the slots may be overwritten again)
    template <class Pub, class Con>
    static int meet(Group& G, int opcode, Pub pub, Con con)
    {
        sph_ctx* c = G.m[0];
        ThreadGroup* g = grp(G);
        const int r = c->dist.rank;
        g->op[(size_t)r] = opcode;
        pub(g, r);
        if (!g->barrier()) return c->fail(SPH_ERR_DEVICE, "thread transport: a rank did not enter collective %d (it would hang over RCCL)", opcode);
        int rc = SPH_OK;
        for (int k = 0; k < g->n; k++)
            if (g->op[(size_t)k] != opcode) rc = c->fail(SPH_ERR_DEVICE, "thread transport: rank %d is in collective %d, rank %d in %d", r, opcode, k, g->op[(size_t)k]);
        if (!rc) rc = con(g, r);
        if (!g->barrier() && !rc) rc = c->fail(SPH_ERR_DEVICE, "thread transport: a rank left collective %d early", opcode);
        return rc;
    }
    int allreduce_min_f32(Group& G, std::vector<std::vector<float>>& rows) override
    {
        return meet(G, 1, [&](ThreadGroup* g, int r) { g->f32rows[(size_t)r] = rows[0]; },
                    [&](ThreadGroup* g, int) {
                        for (size_t k = 0; k < rows[0].size(); k++) {
                            float v = rows[0][k];
                            for (int q = 0; q < g->n; q++) {
                                if (g->f32rows[(size_t)q].size() != rows[0].size()) return G.m[0]->fail(SPH_ERR_DEVICE, "thread transport: all-reduce sizes differ");
                                v = fminf(v, g->f32rows[(size_t)q][k]);
                            }
                            rows[0][k] = v;
                        }
                        return (int)SPH_OK;
                    });
    }
    int allreduce_max_i32(Group& G, std::vector<int>& vals) override
    {
        return meet(G, 2, [&](ThreadGroup* g, int r) { g->i32vals[(size_t)r] = vals[0]; },
                    [&](ThreadGroup* g, int) {
                        int v = vals[0];
                        for (int q = 0; q < g->n; q++) v = std::max(v, g->i32vals[(size_t)q]);
                        vals[0] = v;
                        return (int)SPH_OK;
                    });
    }
    int allreduce_sum_u32(Group& G, std::vector<std::vector<uint32_t>>& rows) override
    {
        std::vector<uint32_t> mine = rows[0];
        return meet(G, 3, [&](ThreadGroup* g, int r) { g->u32rows[(size_t)r] = mine; },
                    [&](ThreadGroup* g, int) {
                        for (size_t k = 0; k < rows[0].size(); k++) {
                            uint32_t v = 0;
                            for (int q = 0; q < g->n; q++) v += g->u32rows[(size_t)q][k];
                            rows[0][k] = v;
                        }
                        return (int)SPH_OK;
                    });
    }
    int neighbour_counts(Group& G, const std::vector<uint32_t>& tl, const std::vector<uint32_t>& tr, std::vector<uint32_t>& fl,
                         std::vector<uint32_t>& fr, int* status) override
    {
        return meet(G, 4, [&](ThreadGroup* g, int r) { g->words[(size_t)r] = {tl[0], tr[0], status ? (uint32_t)*status : 0u, 0, 0, 0, 0, 0}; },
                    [&](ThreadGroup* g, int r) {
                        fl[0] = r > 0 ? g->words[(size_t)r - 1][1] : 0;
                        fr[0] = r + 1 < g->n ? g->words[(size_t)r + 1][0] : 0;
                        if (status)
                            for (int q = 0; q < g->n; q++) *status = std::max(*status, (int)g->words[(size_t)q][2]);
                        return (int)SPH_OK;
                    });
    }
    int counts_round(Group& G, int base, std::vector<std::vector<float>>* red, int* status, std::vector<uint32_t>& tl, std::vector<uint32_t>& tr,
                     std::vector<uint32_t>& fl, std::vector<uint32_t>& fr) override
    {
        sph_ctx* c = G.m[0];
        int rc = wait_stream(c);
        if (rc) {
            grp(G)->abandon();
            return rc;
        }
        HIPCHK(c, hipMemcpy(c->dist.counts_host + base, c->dist.counts.as<uint32_t>() + base, 16, hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(c->dist.counts_host + RC_FAR, c->dist.counts.as<uint32_t>() + RC_FAR, 4, hipMemcpyDeviceToHost));
        tl[0] = c->dist.counts_host[base + 1];
        tr[0] = c->dist.counts_host[base + 2];
        if (status && base == 4 && c->dist.counts_host[base + 3] && *status < SPH_ERR_UNSUPPORTED) *status = SPH_ERR_UNSUPPORTED;
        if (red && (rc = allreduce_min_f32(G, *red))) return rc;
        return neighbour_counts(G, tl, tr, fl, fr, status);
    }
    int refresh_round(Group& G, std::vector<std::vector<float>>* red, int* status, int* fallback, std::vector<RefreshCounts>& rcs) override
    {
        sph_ctx* c = G.m[0];
        int rc = wait_stream(c);
        if (rc) {
            grp(G)->abandon();
            return rc;
        }
        uint32_t w[7];
        HIPCHK(c, hipMemcpy(w, c->dist.counts.p, 28, hipMemcpyDeviceToHost));
        RefreshCounts& o = rcs[0];
        o.halo[0] = w[0];
        o.halo[1] = w[1];
        o.mig[0] = w[2];
        o.mig[1] = w[3];
        if (w[4]) *fallback = 1;
        memcpy(o.hreg, w + 5, 8);
        rc = meet(G, 5, [&](ThreadGroup* g, int r) { g->words[(size_t)r] = {o.mig[0], o.halo[0], o.mig[1], o.halo[1], (uint32_t)*status, (uint32_t)*fallback, w[5], w[6]}; },
                  [&](ThreadGroup* g, int r) {
                      o.in_mig[0] = r > 0 ? g->words[(size_t)r - 1][2] : 0;
                      o.in_halo[0] = r > 0 ? g->words[(size_t)r - 1][3] : 0;
                      o.in_mig[1] = r + 1 < g->n ? g->words[(size_t)r + 1][0] : 0;
                      o.in_halo[1] = r + 1 < g->n ? g->words[(size_t)r + 1][1] : 0;
                      o.in_hreg[0] = o.in_hreg[1] = 0.f;
                      if (r > 0) memcpy(&o.in_hreg[0], &g->words[(size_t)r - 1][7], 4);        // the left rank's figure for ITS right cut = my left one
                      if (r + 1 < g->n) memcpy(&o.in_hreg[1], &g->words[(size_t)r + 1][6], 4);
                      for (int q = 0; q < g->n; q++) {
                          *status = std::max(*status, (int)g->words[(size_t)q][4]);
                          if (g->words[(size_t)q][5]) *fallback = 1;
                      }
                      return (int)SPH_OK;
                  });
        if (rc) return rc;
        return red ? allreduce_min_f32(G, *red) : SPH_OK;
    }
    int exchange(Group& G, std::vector<Xfer>& x) override
    {
        sph_ctx* c = G.m[0];
        ThreadGroup* g = grp(G);
        const int r = c->dist.rank;
        c->dist.stat_exchanges++;
        int rc = wait_stream(c);   // my staging buffers are packed
        if (rc) {
            g->abandon();
            return rc;
        }
        // point to point, like the grouped ncclSend / ncclRecv of the RCCL transport: one rendezvous per x-neighbour this rank has
        // something for or expects something from; RCCL pairs every send with a receive of the same size on the other side -- the
        // same rule, checked; a neighbour that does not come is what would hang RCCL
        for (int side = 0; side < 2; side++) {
            const int nb = side == 0 ? r - 1 : r + 1;
            if (nb < 0 || nb >= g->n) {
                if (x[0].send_bytes[side] || x[0].recv_bytes[side]) return c->fail(SPH_ERR_DEVICE, "halo exchange across the outer edge of the slab row (rank %d)", r);
                continue;
            }
            if (!x[0].send_bytes[side] && !x[0].recv_bytes[side]) continue;
            ThreadGroup::PairChan& ch = g->pair[(size_t)std::min(r, nb)];
            const int mine = r < nb ? 0 : 1;
            {
                std::lock_guard<std::mutex> lk(ch.mu);
                ch.xf[mine] = x[0];
            }
            if (!g->pair_barrier(ch))
                return c->fail(SPH_ERR_DEVICE, "thread transport: rank %d did not enter the exchange rank %d has %zu bytes to send to / %zu bytes to receive from it for (it would hang over RCCL)",
                               nb, r, x[0].send_bytes[side], x[0].recv_bytes[side]);
            const Xfer o = ch.xf[mine ^ 1];
            const int oside = side ^ 1;   // my left neighbour's right side and vice versa
            if (x[0].recv_bytes[side] != o.send_bytes[oside] || x[0].send_bytes[side] != o.recv_bytes[oside]) {
                rc = c->fail(SPH_ERR_DEVICE, "halo exchange sizes of ranks %d and %d do not pair up (rank %d: send %zu recv %zu; rank %d: send %zu recv %zu)", r, nb, r,
                             x[0].send_bytes[side], x[0].recv_bytes[side], nb, o.send_bytes[oside], o.recv_bytes[oside]);
                g->abandon();
                return rc;
            }
            c->dist.stat_bytes_sent += x[0].send_bytes[side];
            c->dist.stat_bytes_recv += x[0].recv_bytes[side];
            if (x[0].recv_bytes[side]) HIPCHK(c, hipMemcpyAsync(x[0].recv[side], o.send[oside], x[0].recv_bytes[side], hipMemcpyDefault, c->stream));
            if ((rc = wait_stream(c))) {
                g->abandon();
                return rc;
            }
            // (the sender may reuse its staging buffer once both are past this)
            if (!g->pair_barrier(ch)) return c->fail(SPH_ERR_DEVICE, "thread transport: rank %d left the exchange with rank %d early", nb, r);
        }
        return SPH_OK;
    }
    int allreduce_solver(Group& G, int slot) override
    {
        sph_ctx* c = G.m[0];
        c->dist.stat_allreduces++;
        int rc = wait_stream(c);
        if (rc) {
            grp(G)->abandon();
            return rc;
        }
        std::array<double, 6> mine{};
        HIPCHK(c, hipMemcpy(mine.data(), c->dist.solver_tot.as<double>() + 8 * slot, 48, hipMemcpyDeviceToHost));
        return meet(G, 7 + slot, [&](ThreadGroup* g, int r) { g->tot[(size_t)r] = mine; },
                    [&](ThreadGroup* g, int) -> int {
                        double t[6] = {0, 0, 0, 0, 0, 0};
                        for (int q = 0; q < g->n; q++)
                            for (int k = 0; k < 6; k++) t[k] += g->tot[(size_t)q][k];   // (rank order: the same sum on every rank)
                        HIPCHK(c, hipMemcpy(c->dist.solver_tot.as<double>() + 8 * slot, t, 48, hipMemcpyHostToDevice));
                        return SPH_OK;
                    });
    }
    int agree_guards_queued(Group& G) override
    {
        sph_ctx* c = G.m[0];
        int rc = wait_stream(c);
        if (rc) {
            grp(G)->abandon();
            return rc;
        }
        uint32_t e = 0;
        HIPCHK(c, hipMemcpy(&e, &c->status.as<DeviceStatus>()->error, 4, hipMemcpyDeviceToHost));
        return meet(G, 9, [&](ThreadGroup* g, int r) { g->i32vals[(size_t)r] = (int)e; },
                    [&](ThreadGroup* g, int) -> int {
                        uint32_t m = 0;
                        for (int q = 0; q < g->n; q++) m = std::max(m, (uint32_t)g->i32vals[(size_t)q]);
                        HIPCHK(c, hipMemcpy(&c->ctrl.as<SolverCtrl>()->peer_error, &m, 4, hipMemcpyHostToDevice));
                        return SPH_OK;
                    });
    }
};


// ---- processes of one node WITHOUT RCCL: rendezvous and staging in a POSIX shared-memory segment ---------------------------------
// ThreadComm's protocol between PROCESSES: every rank is a process of its own (launched like the RCCL form: one per GPU, or several
// on one GPU -- which RCCL refuses, "Duplicate GPU detected"), the collectives are rendezvous on atomics in a segment all ranks map,
// ghost and migrant records are staged through it (device -> segment by the sender, segment -> device by the receiver).  Host
// synchronous and PCIe-bound: the functional transport for boxes where RCCL cannot serve the launch (bench.py --gpus 2 on a one-GPU
// box, CI), and the multi-process check of the launcher glue -- never the fast path.  Same checks as the thread transport: a collective
// not entered by every rank (60 s), ranks in different collectives, a send whose size no receive pairs with.
#define SHM_MAX_RANKS 16
#define SHM_MAX_F32 16
#define SHM_MAX_U32 4096
struct ShmSegment {
    uint32_t magic;   // written last by the creating rank
    uint32_t n;
    uint64_t bytes_per_side, total_bytes;
    std::atomic<uint32_t> arrived, gen, broken;
    struct Pair {
        std::atomic<uint32_t> arrived, gen;
        uint64_t send_bytes[2][2], recv_bytes[2][2];   // [who: 0 the lower rank, 1 the upper][side]
    } pair[SHM_MAX_RANKS];
    int32_t op[SHM_MAX_RANKS];
    double tot[SHM_MAX_RANKS][8];
    float f32rows[SHM_MAX_RANKS][SHM_MAX_F32];
    uint32_t f32len[SHM_MAX_RANKS];
    uint32_t u32rows[SHM_MAX_RANKS][SHM_MAX_U32];
    uint32_t u32len[SHM_MAX_RANKS];
    int32_t i32vals[SHM_MAX_RANKS];
    uint32_t words[SHM_MAX_RANKS][8];
    // followed by the outboxes: rank r, side s at payload() + (2 r + s) * bytes_per_side
    uint8_t* outbox(int r, int side) { return reinterpret_cast<uint8_t*>(this) + ((sizeof(ShmSegment) + 4095) & ~(size_t)4095) + ((size_t)2 * r + side) * bytes_per_side; }
    static size_t size_for(int n, uint64_t per_side) { return ((sizeof(ShmSegment) + 4095) & ~(size_t)4095) + (size_t)2 * n * per_side; }
    static bool spin_until(std::atomic<uint32_t>& word, uint32_t old, std::atomic<uint32_t>& broken)
    {
        const auto t_end = std::chrono::steady_clock::now() + std::chrono::seconds(60);
        for (uint32_t k = 0; word.load(std::memory_order_acquire) == old; k++) {
            if (broken.load(std::memory_order_relaxed)) return false;
            if ((k & 63u) == 63u) {
                if (std::chrono::steady_clock::now() >= t_end) {
                    broken.store(1u);
                    return false;
                }
                struct timespec ts = {0, 20000};
                nanosleep(&ts, nullptr);
            }
        }
        return !broken.load(std::memory_order_relaxed);
    }
    bool barrier()
    {
        if (broken.load()) return false;
        const uint32_t g = gen.load(std::memory_order_acquire);
        if (arrived.fetch_add(1u, std::memory_order_acq_rel) + 1u == n) {
            arrived.store(0u, std::memory_order_relaxed);
            gen.fetch_add(1u, std::memory_order_release);
            return true;
        }
        return spin_until(gen, g, broken);
    }
    bool pair_barrier(Pair& ch)
    {
        if (broken.load()) return false;
        const uint32_t g = ch.gen.load(std::memory_order_acquire);
        if (ch.arrived.fetch_add(1u, std::memory_order_acq_rel) + 1u == 2u) {
            ch.arrived.store(0u, std::memory_order_relaxed);
            ch.gen.fetch_add(1u, std::memory_order_release);
            return true;
        }
        return spin_until(ch.gen, g, broken);
    }
};

struct ShmComm : Comm {
    bool host_collectives_wait() const override { return true; }
    static ShmSegment* seg(Group& G) { return (ShmSegment*)G.m[0]->dist.shm; }
    template <class Pub, class Con>
    static int meet(Group& G, int opcode, Pub pub, Con con)
    {
        sph_ctx* c = G.m[0];
        ShmSegment* g = seg(G);
        const int r = c->dist.rank;
        g->op[r] = opcode;
        pub(g, r);
        if (!g->barrier()) return c->fail(SPH_ERR_DEVICE, "shared-memory transport: a rank did not enter collective %d (it would hang over RCCL)", opcode);
        int rc = SPH_OK;
        for (int k = 0; k < (int)g->n; k++)
            if (g->op[k] != opcode) rc = c->fail(SPH_ERR_DEVICE, "shared-memory transport: rank %d is in collective %d, rank %d in %d", r, opcode, k, g->op[k]);
        if (!rc) rc = con(g, r);
        if (!g->barrier() && !rc) rc = c->fail(SPH_ERR_DEVICE, "shared-memory transport: a rank left collective %d early", opcode);
        return rc;
    }
    int allreduce_min_f32(Group& G, std::vector<std::vector<float>>& rows) override
    {
        const size_t len = rows[0].size();
        if (len > SHM_MAX_F32) return G.m[0]->fail(SPH_ERR_INVALID_ARGUMENT, "shared-memory transport: all-reduce of %zu floats", len);
        return meet(G, 1,
                    [&](ShmSegment* g, int r) {
                        g->f32len[r] = (uint32_t)len;
                        memcpy(g->f32rows[r], rows[0].data(), len * 4);
                    },
                    [&](ShmSegment* g, int) {
                        for (size_t k = 0; k < len; k++) {
                            float v = rows[0][k];
                            for (int q = 0; q < (int)g->n; q++) {
                                if (g->f32len[q] != len) return G.m[0]->fail(SPH_ERR_DEVICE, "shared-memory transport: all-reduce sizes differ");
                                v = fminf(v, g->f32rows[q][k]);
                            }
                            rows[0][k] = v;
                        }
                        return (int)SPH_OK;
                    });
    }
    int allreduce_max_i32(Group& G, std::vector<int>& vals) override
    {
        return meet(G, 2, [&](ShmSegment* g, int r) { g->i32vals[r] = vals[0]; },
                    [&](ShmSegment* g, int) {
                        int v = vals[0];
                        for (int q = 0; q < (int)g->n; q++) v = std::max(v, (int)g->i32vals[q]);
                        vals[0] = v;
                        return (int)SPH_OK;
                    });
    }
    int allreduce_sum_u32(Group& G, std::vector<std::vector<uint32_t>>& rows) override
    {
        const size_t len = rows[0].size();
        if (len > SHM_MAX_U32) return G.m[0]->fail(SPH_ERR_INVALID_ARGUMENT, "shared-memory transport: all-reduce of %zu words", len);
        return meet(G, 3,
                    [&](ShmSegment* g, int r) {
                        g->u32len[r] = (uint32_t)len;
                        memcpy(g->u32rows[r], rows[0].data(), len * 4);
                    },
                    [&](ShmSegment* g, int) {
                        for (size_t k = 0; k < len; k++) {
                            uint32_t v = 0;
                            for (int q = 0; q < (int)g->n; q++) v += g->u32rows[q][k];
                            rows[0][k] = v;
                        }
                        return (int)SPH_OK;
                    });
    }
    int neighbour_counts(Group& G, const std::vector<uint32_t>& tl, const std::vector<uint32_t>& tr, std::vector<uint32_t>& fl,
                         std::vector<uint32_t>& fr, int* status) override
    {
        return meet(G, 4,
                    [&](ShmSegment* g, int r) {
                        const uint32_t w[8] = {tl[0], tr[0], status ? (uint32_t)*status : 0u, 0, 0, 0, 0, 0};
                        memcpy(g->words[r], w, sizeof w);
                    },
                    [&](ShmSegment* g, int r) {
                        fl[0] = r > 0 ? g->words[r - 1][1] : 0;
                        fr[0] = r + 1 < (int)g->n ? g->words[r + 1][0] : 0;
                        if (status)
                            for (int q = 0; q < (int)g->n; q++) *status = std::max(*status, (int)g->words[q][2]);
                        return (int)SPH_OK;
                    });
    }
    int counts_round(Group& G, int base, std::vector<std::vector<float>>* red, int* status, std::vector<uint32_t>& tl, std::vector<uint32_t>& tr,
                     std::vector<uint32_t>& fl, std::vector<uint32_t>& fr) override
    {
        sph_ctx* c = G.m[0];
        int rc = wait_stream(c);
        if (rc) {
            comm_abandon(c);
            return rc;
        }
        HIPCHK(c, hipMemcpy(c->dist.counts_host + base, c->dist.counts.as<uint32_t>() + base, 16, hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(c->dist.counts_host + RC_FAR, c->dist.counts.as<uint32_t>() + RC_FAR, 4, hipMemcpyDeviceToHost));
        tl[0] = c->dist.counts_host[base + 1];
        tr[0] = c->dist.counts_host[base + 2];
        if (status && base == 4 && c->dist.counts_host[base + 3] && *status < SPH_ERR_UNSUPPORTED) *status = SPH_ERR_UNSUPPORTED;
        if (red && (rc = allreduce_min_f32(G, *red))) return rc;
        return neighbour_counts(G, tl, tr, fl, fr, status);
    }
    int refresh_round(Group& G, std::vector<std::vector<float>>* red, int* status, int* fallback, std::vector<RefreshCounts>& rcs) override
    {
        sph_ctx* c = G.m[0];
        int rc = wait_stream(c);
        if (rc) {
            comm_abandon(c);
            return rc;
        }
        uint32_t w[7];
        HIPCHK(c, hipMemcpy(w, c->dist.counts.p, 28, hipMemcpyDeviceToHost));
        RefreshCounts& o = rcs[0];
        o.halo[0] = w[0];
        o.halo[1] = w[1];
        o.mig[0] = w[2];
        o.mig[1] = w[3];
        if (w[4]) *fallback = 1;
        memcpy(o.hreg, w + 5, 8);
        rc = meet(G, 5,
                  [&](ShmSegment* g, int r) {
                      const uint32_t v[8] = {o.mig[0], o.halo[0], o.mig[1], o.halo[1], (uint32_t)*status, (uint32_t)*fallback, w[5], w[6]};
                      memcpy(g->words[r], v, sizeof v);
                  },
                  [&](ShmSegment* g, int r) {
                      o.in_mig[0] = r > 0 ? g->words[r - 1][2] : 0;
                      o.in_halo[0] = r > 0 ? g->words[r - 1][3] : 0;
                      o.in_mig[1] = r + 1 < (int)g->n ? g->words[r + 1][0] : 0;
                      o.in_halo[1] = r + 1 < (int)g->n ? g->words[r + 1][1] : 0;
                      o.in_hreg[0] = o.in_hreg[1] = 0.f;
                      if (r > 0) memcpy(&o.in_hreg[0], &g->words[r - 1][7], 4);
                      if (r + 1 < (int)g->n) memcpy(&o.in_hreg[1], &g->words[r + 1][6], 4);
                      for (int q = 0; q < (int)g->n; q++) {
                          *status = std::max(*status, (int)g->words[q][4]);
                          if (g->words[q][5]) *fallback = 1;
                      }
                      return (int)SPH_OK;
                  });
        if (rc) return rc;
        return red ? allreduce_min_f32(G, *red) : SPH_OK;
    }
    int exchange(Group& G, std::vector<Xfer>& x) override
    {
        sph_ctx* c = G.m[0];
        ShmSegment* g = seg(G);
        const int r = c->dist.rank;
        c->dist.stat_exchanges++;
        int rc = wait_stream(c);   // my staging buffers are packed
        if (rc) {
            comm_abandon(c);
            return rc;
        }
        for (int side = 0; side < 2; side++) {
            const int nb = side == 0 ? r - 1 : r + 1;
            if (nb < 0 || nb >= (int)g->n) {
                if (x[0].send_bytes[side] || x[0].recv_bytes[side]) return c->fail(SPH_ERR_DEVICE, "halo exchange across the outer edge of the slab row (rank %d)", r);
                continue;
            }
            if (!x[0].send_bytes[side] && !x[0].recv_bytes[side]) continue;
            if (x[0].send_bytes[side] > g->bytes_per_side) {
                comm_abandon(c);
                return c->fail(SPH_ERR_CAPACITY, "shared-memory transport: a message of %zu bytes does not fit the %llu-byte outbox (sph_comm_init_shm bytes_per_side)",
                               x[0].send_bytes[side], (unsigned long long)g->bytes_per_side);
            }
            ShmSegment::Pair& ch = g->pair[std::min(r, nb)];
            const int mine = r < nb ? 0 : 1;
            if (x[0].send_bytes[side]) HIPCHK(c, hipMemcpy(g->outbox(r, side), x[0].send[side], x[0].send_bytes[side], hipMemcpyDeviceToHost));
            for (int sd = 0; sd < 2; sd++) {
                ch.send_bytes[mine][sd] = x[0].send_bytes[sd];
                ch.recv_bytes[mine][sd] = x[0].recv_bytes[sd];
            }
            if (!g->pair_barrier(ch))
                return c->fail(SPH_ERR_DEVICE, "shared-memory transport: rank %d did not enter the exchange rank %d has %zu bytes to send to / %zu bytes to receive from it for (it would hang over RCCL)",
                               nb, r, x[0].send_bytes[side], x[0].recv_bytes[side]);
            const int oside = side ^ 1;   // my left neighbour's right side and vice versa
            const uint64_t o_send = ch.send_bytes[mine ^ 1][oside], o_recv = ch.recv_bytes[mine ^ 1][oside];
            if (x[0].recv_bytes[side] != o_send || x[0].send_bytes[side] != o_recv) {
                rc = c->fail(SPH_ERR_DEVICE, "halo exchange sizes of ranks %d and %d do not pair up (rank %d: send %zu recv %zu; rank %d: send %llu recv %llu)", r, nb, r,
                             x[0].send_bytes[side], x[0].recv_bytes[side], nb, (unsigned long long)o_send, (unsigned long long)o_recv);
                comm_abandon(c);
                return rc;
            }
            c->dist.stat_bytes_sent += x[0].send_bytes[side];
            c->dist.stat_bytes_recv += x[0].recv_bytes[side];
            if (x[0].recv_bytes[side]) HIPCHK(c, hipMemcpy(x[0].recv[side], g->outbox(nb, oside), x[0].recv_bytes[side], hipMemcpyHostToDevice));
            // (the sender may reuse its outbox once both are past this)
            if (!g->pair_barrier(ch)) return c->fail(SPH_ERR_DEVICE, "shared-memory transport: rank %d left the exchange with rank %d early", nb, r);
        }
        return SPH_OK;
    }
    int allreduce_solver(Group& G, int slot) override
    {
        sph_ctx* c = G.m[0];
        c->dist.stat_allreduces++;
        int rc = wait_stream(c);
        if (rc) {
            comm_abandon(c);
            return rc;
        }
        double mine[6];
        HIPCHK(c, hipMemcpy(mine, c->dist.solver_tot.as<double>() + 8 * slot, 48, hipMemcpyDeviceToHost));
        return meet(G, 7 + slot, [&](ShmSegment* g, int r) { memcpy(g->tot[r], mine, 48); },
                    [&](ShmSegment* g, int) -> int {
                        double t[6] = {0, 0, 0, 0, 0, 0};
                        for (int q = 0; q < (int)g->n; q++)
                            for (int k = 0; k < 6; k++) t[k] += g->tot[q][k];   // (rank order: the same sum on every rank)
                        HIPCHK(c, hipMemcpy(c->dist.solver_tot.as<double>() + 8 * slot, t, 48, hipMemcpyHostToDevice));
                        return SPH_OK;
                    });
    }
    int agree_guards_queued(Group& G) override
    {
        sph_ctx* c = G.m[0];
        int rc = wait_stream(c);
        if (rc) {
            comm_abandon(c);
            return rc;
        }
        uint32_t e = 0;
        HIPCHK(c, hipMemcpy(&e, &c->status.as<DeviceStatus>()->error, 4, hipMemcpyDeviceToHost));
        return meet(G, 9, [&](ShmSegment* g, int r) { g->i32vals[r] = (int)e; },
                    [&](ShmSegment* g, int) -> int {
                        uint32_t m = 0;
                        for (int q = 0; q < (int)g->n; q++) m = std::max(m, (uint32_t)g->i32vals[q]);
                        HIPCHK(c, hipMemcpy(&c->ctrl.as<SolverCtrl>()->peer_error, &m, 4, hipMemcpyHostToDevice));
                        return SPH_OK;
                    });
    }
};


// ---- peer-mapped PUSH transport: the ranks of one node as processes, device to device --------
// Every rank owns one device allocation -- its BOX: an inbox per x-neighbour side (two parities), a table of the ranks' solver totals
// (two slots x two parities), and the arrival flags -- exported with hipIpcGetMemHandle and mapped by the others
// (hipIpcOpenMemHandle: over xGMI between GPUs, plain device memory between two ranks of one GPU).  A message is PUSHED: the sender's
// kernel copies its staging buffer into the receiver's inbox and then stores the message's sequence number into the receiver's flag
// (system-scope release); the receiver's next kernel spins on its own flag (system-scope acquire) and the kernels behind it read the
// inbox in place (Xfer::recv is handed back pointing at it).  No collective library launch, no host wait, no copy on the receiving
// side; per exchange one push kernel and one wait kernel of one workgroup each (multi-block copies for the per-step record messages).
// Flow control without acknowledgements: a pair of ranks enters every exchange together (the pairing rule the other transports check),
// so rank A's push k + 2 into the parity push k used follows A's wait for B's signal k + 1, which B issued behind its kernels that
// read message k.  The totals all-reduce is the same between all ranks: push six doubles into everybody's table, wait for
// everybody's, add the rows in rank order.  Host-value collectives (counts, header, agreement) stay the shared-memory transport's.
struct IpcBox {   // layout of a rank's exported allocation (device memory)
    uint32_t data_seq[2];                 // [side]: sequence number of the last complete message in my inbox of that side
    uint32_t tot_seq[2][SHM_MAX_RANKS];   // [slot][rank]: ... of that rank's totals row
    double tot[2][2][SHM_MAX_RANKS][8];   // [slot][parity][rank][6 used]
    // followed by the inboxes: side s, parity q at payload() + (2 s + q) * bytes_per_side
    static size_t header_bytes() { return (sizeof(IpcBox) + 255) & ~(size_t)255; }
    static size_t size_for(uint64_t per_side) { return header_bytes() + 4 * (size_t)per_side; }
};
struct IpcState {
    uint8_t* mine = nullptr;                  // my box
    uint8_t* peer[SHM_MAX_RANKS] = {};        // every rank's box as mapped here (peer[rank] == mine for myself)
    uint64_t bytes_per_side = 0;
    uint32_t seq[2] = {0, 0};                 // messages exchanged with the [left, right] neighbour
    uint32_t tot_n[2] = {0, 0};               // all-reduces of the totals of slot 0 / 1
    uint64_t timeout_ticks = 2000000000ull;   // a wait gives up after 20 s of the 100 MHz clock (SPH_IPC_TIMEOUT_MS)
    static uint8_t* inbox(uint8_t* box, uint64_t per_side, int side, uint32_t parity) { return box + IpcBox::header_bytes() + ((size_t)2 * side + parity) * per_side; }
};

__device__ __forceinline__ void ipc_store_release(uint32_t* flag, uint32_t v) { __hip_atomic_store(flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
// Bounded (advisor r4): a peer that failed before its push -- a device fault, the SPH_ERR_CAPACITY exit of round() -- must not leave
// this GPU spinning inside a kernel nobody can stop.  After `timeout_ticks` of the constant 100 MHz clock the wait gives up and raises
// the context's sticky status word (SPH_ERR_DEVICE: the step's status check turns it into the error the other transports report for a
// collective a rank never entered); the kernels behind it then read a stale inbox, and the step's result is discarded with the error.
__device__ __forceinline__ bool ipc_wait(const uint32_t* flag, uint32_t want, uint64_t timeout_ticks)
{
    // (sequence numbers only grow; a 32-bit wrap is 4 G exchanges away)
    const uint64_t t0 = wall_clock64();
    uint32_t spins = 0;
    while ((int32_t)(__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - want) < 0) {
        __builtin_amdgcn_s_sleep(2);
        if ((++spins & 0x3ffu) == 0u && wall_clock64() - t0 > timeout_ticks) return false;
    }
    return true;
}
struct IpcPush {
    const uint4* src[2];       // staging of the message to the [left, right] neighbour (16-byte granules) or nullptr
    uint4* dst[2];             // the neighbour's inbox (peer-mapped)
    uint32_t granules[2];
    uint32_t* flag[2];         // the neighbour's data_seq word for me
    uint32_t seq[2];           // 0: this side does not take part
    // totals (nr == 0: none): my row into every rank's table, then its flag
    const double* tot;
    double* tot_dst[SHM_MAX_RANKS];
    uint32_t* tot_flag[SHM_MAX_RANKS];
    uint32_t tot_seq;
    int nr, self;
};
// one workgroup: small messages (a Jacobi iteration's ghost values: a few thousand floats) -- copy, fence, signal
__global__ __launch_bounds__(1024) void k_ipc_push(IpcPush j)
{
    for (int s = 0; s < 2; s++)
        for (uint32_t k = threadIdx.x; k < j.granules[s]; k += 1024u) j.dst[s][k] = j.src[s][k];
    if (j.nr && threadIdx.x < 6u * (uint32_t)j.nr) {
        const int r = (int)(threadIdx.x / 6u), k = (int)(threadIdx.x % 6u);
        if (r != j.self) j.tot_dst[r][k] = j.tot[k];
    }
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x < 2 && j.seq[threadIdx.x]) ipc_store_release(j.flag[threadIdx.x], j.seq[threadIdx.x]);
    if (j.nr && threadIdx.x >= 64 && threadIdx.x < 64u + (uint32_t)j.nr && (int)(threadIdx.x - 64u) != j.self) ipc_store_release(j.tot_flag[threadIdx.x - 64u], j.tot_seq);
}
// the copy of a large message (per-step migrant / ghost records) by many workgroups; k_ipc_push with granules = 0 signals behind it
__global__ __launch_bounds__(256) void k_ipc_copy(const uint4* __restrict__ src, uint4* __restrict__ dst, uint32_t granules)
{
    for (uint32_t k = blockIdx.x * 256u + threadIdx.x; k < granules; k += gridDim.x * 256u) dst[k] = src[k];
    __threadfence_system();
}
struct IpcWait {
    const uint32_t* flag[2];
    uint32_t seq[2];           // 0: nothing expected from that side
    const uint32_t* tot_flag;  // my tot_seq[slot] row (nr words), nullptr: no totals
    uint32_t tot_seq;
    const double* table;       // my table of this slot and parity
    double* tot;               // in: my row; out: the sum over the ranks in rank order
    int nr, self;
    DeviceStatus* status;      // raised when a flag does not arrive within timeout_ticks (100 MHz)
    uint64_t timeout_ticks;
};
__global__ __launch_bounds__(64) void k_ipc_wait(IpcWait j)
{
    const uint32_t t = threadIdx.x;
    bool ok = true;
    if (t < 2 && j.seq[t]) ok = ipc_wait(j.flag[t], j.seq[t], j.timeout_ticks);
    if (j.tot_flag && t >= 2 && t < 2u + (uint32_t)j.nr && (int)(t - 2u) != j.self) ok = ipc_wait(j.tot_flag + (t - 2u), j.tot_seq, j.timeout_ticks);
    // a wait that gave up: the sticky status word ends the step with SPH_ERR_DEVICE at its next status check; sph_step then poisons the
    // context AND abandons the group (comm_abandon: the shared-memory segment under this transport is marked broken), so every rank's
    // later steps fail at once instead of pairing mismatched sequence numbers -- the segment is recreated, as a failed RCCL communicator
    // would be (advisor r5).  Under rocprofv3 or a debugger a rank may stall for longer than the default: raise SPH_IPC_TIMEOUT_MS there.
    if (!ok && atomicCAS(&j.status->error, 0u, (uint32_t)SPH_ERR_DEVICE) == 0u) j.status->info = 0x1bc00000u | t;   // (info: which waiter gave up)
    __syncthreads();
    if (j.tot_flag && t < 6) {
        double s = 0.0;
        for (int r = 0; r < j.nr; r++) s += r == j.self ? j.tot[t] : __hip_atomic_load(j.table + 8 * r + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        j.tot[t] = s;
    }
}

// ---- the fused round of a Jacobi iteration (round 6; VERDICT r5 next 4: launches that are deterministic cost) ----------------------------
// k_ipc_pack_push: ONE workgroup adds up the rank's totals of the iteration (rank_totals_block: what block 0 of k_pack_totals does, with its
// early exits), gathers the halo members' values from the field straight into the x-neighbours' inboxes (no staging buffer, no packing
// launch), writes the totals row into every rank's table, fences, signals.  k_ipc_wait_unpack: ONE workgroup waits for the neighbours' and
// the ranks' flags (bounded, like k_ipc_wait), sums the totals in rank order and scatters the inboxes into the ghosts' slots (no unpacking
// launch).  The inboxes live in fine-grained memory a peer GPU writes: they are read with system-scope loads behind the acquiring wait.
struct IpcFusedPush {
    const SolverPartial* partials;
    uint32_t nparts;
    const SolverCtrl* ctrl;
    const uint32_t* gate;
    const uint32_t* status_error;
    int iter;
    const uint32_t* halo_src;
    uint32_t cnt[2];
    const float* field;
    int stride, off;
    float* dst[2];             // the neighbour's inbox (peer-mapped), nullptr: no neighbour on that side
    uint32_t* flag[2];
    uint32_t seq[2];
    double* tot;               // my row (local): written here, read by my own k_ipc_wait_unpack
    double* tot_dst[SHM_MAX_RANKS];
    uint32_t* tot_flag[SHM_MAX_RANKS];
    uint32_t tot_seq;
    int nr, self;
};
__global__ __launch_bounds__(RANK_TOTALS_THREADS) void k_ipc_pack_push(IpcFusedPush j)
{
    // (block-uniform conditions: every lane takes the same branch around the reduction's barrier)
    const bool stale = (j.gate && *j.gate == 0u) || j.ctrl->slot_done[j.iter & 1] != 0u;   // k_pack_totals' early exits: nobody reads the totals then
    if (!stale) rank_totals_block(j.partials, j.nparts, j.tot, j.status_error);
    __syncthreads();   // thread 0's totals are visible to the lanes that push them
    uint32_t base = 0;
    for (int s = 0; s < 2; s++) {
        if (j.dst[s])
            for (uint32_t k = threadIdx.x; k < j.cnt[s]; k += RANK_TOTALS_THREADS) j.dst[s][k] = j.field[(size_t)j.halo_src[base + k] * j.stride + j.off];
        base += j.cnt[s];
    }
    if (threadIdx.x < 6u * (uint32_t)j.nr) {
        const int r = (int)(threadIdx.x / 6u), k = (int)(threadIdx.x % 6u);
        if (r != j.self) j.tot_dst[r][k] = j.tot[k];
    }
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x < 2 && j.seq[threadIdx.x]) ipc_store_release(j.flag[threadIdx.x], j.seq[threadIdx.x]);
    if (threadIdx.x >= 64 && threadIdx.x < 64u + (uint32_t)j.nr && (int)(threadIdx.x - 64u) != j.self) ipc_store_release(j.tot_flag[threadIdx.x - 64u], j.tot_seq);
}
struct IpcFusedWait {
    IpcWait w;
    const uint32_t* inbox[2];   // my inboxes of this round ([from left, from right]); nullptr: nothing to scatter from that side
    const uint32_t* ghost_dst;
    uint32_t cnt[2];
    uint32_t* field;            // (the float field, moved as words)
    int stride, off;
};
__global__ __launch_bounds__(1024) void k_ipc_wait_unpack(IpcFusedWait f)
{
    const IpcWait& j = f.w;
    const uint32_t t = threadIdx.x;
    bool ok = true;
    if (t < 2 && j.seq[t]) ok = ipc_wait(j.flag[t], j.seq[t], j.timeout_ticks);
    if (j.tot_flag && t >= 2 && t < 2u + (uint32_t)j.nr && (int)(t - 2u) != j.self) ok = ipc_wait(j.tot_flag + (t - 2u), j.tot_seq, j.timeout_ticks);
    if (!ok && atomicCAS(&j.status->error, 0u, (uint32_t)SPH_ERR_DEVICE) == 0u) j.status->info = 0x1bc00000u | t;   // (see k_ipc_wait)
    __syncthreads();
    if (j.tot_flag && t < 6) {
        double s = 0.0;
        for (int r = 0; r < j.nr; r++) s += r == j.self ? j.tot[t] : __hip_atomic_load(j.table + 8 * r + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        j.tot[t] = s;
    }
    uint32_t base = 0;
    for (int s = 0; s < 2; s++) {
        if (f.inbox[s])
            for (uint32_t k = t; k < f.cnt[s]; k += 1024u)
                f.field[(size_t)f.ghost_dst[base + k] * f.stride + f.off] = __hip_atomic_load(f.inbox[s] + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        base += f.cnt[s];
    }
}

struct IpcComm : ShmComm {
    static IpcState* st(sph_ctx* c) { return (IpcState*)c->dist.ipc; }
    bool can_fuse_iteration(Group& G, size_t bytes) override { return G.m.size() == 1 && st(G.m[0]) && bytes <= st(G.m[0])->bytes_per_side && bytes <= (64u << 10); }
    int exchange_fused(Group& G, const FusedField& f, int slot) override
    {
        sph_ctx* c = G.m[0];
        IpcState* I = st(c);
        const int r = c->dist.rank, nr = c->dist.nranks;
        IpcFusedPush ps{};
        IpcFusedWait fw{};
        IpcWait& w = fw.w;
        ps.partials = (const SolverPartial*)f.partials;
        ps.nparts = f.nparts;
        ps.ctrl = f.ctrl;
        ps.gate = f.gate;
        ps.status_error = &c->status.as<DeviceStatus>()->error;
        ps.iter = f.iter;
        ps.halo_src = f.halo_src;
        ps.field = f.field;
        ps.stride = f.stride;
        ps.off = f.off;
        fw.ghost_dst = f.ghost_dst;
        fw.field = (uint32_t*)f.field;
        fw.stride = f.stride;
        fw.off = f.off;
        for (int side = 0; side < 2; side++) {
            ps.cnt[side] = f.n_halo[side];
            fw.cnt[side] = f.n_ghost[side];
            const int nb = side == 0 ? r - 1 : r + 1;
            if (nb < 0 || nb >= nr) {
                if (f.n_halo[side] || f.n_ghost[side]) return c->fail(SPH_ERR_DEVICE, "halo exchange across the outer edge of the slab row (rank %d)", r);
                continue;
            }
            if (!f.n_halo[side] && !f.n_ghost[side]) continue;
            const uint32_t seq = ++I->seq[side], parity = seq & 1u;
            const int oside = side ^ 1;   // I am my left neighbour's right side
            ps.dst[side] = (float*)IpcState::inbox(I->peer[nb], I->bytes_per_side, oside, parity);
            ps.flag[side] = &((IpcBox*)I->peer[nb])->data_seq[oside];
            ps.seq[side] = seq;
            w.flag[side] = &((IpcBox*)I->mine)->data_seq[side];
            w.seq[side] = seq;
            if (f.scatter) fw.inbox[side] = (const uint32_t*)IpcState::inbox(I->mine, I->bytes_per_side, side, parity);
            c->dist.stat_bytes_sent += (size_t)f.n_halo[side] * 4;
            c->dist.stat_bytes_recv += (size_t)f.n_ghost[side] * 4;
        }
        {
            const uint32_t seq = ++I->tot_n[slot], parity = seq & 1u;
            ps.tot = c->dist.solver_tot.as<double>() + 8 * slot;
            ps.tot_seq = seq;
            ps.nr = nr;
            ps.self = r;
            for (int q = 0; q < nr; q++) {
                IpcBox* b = (IpcBox*)I->peer[q];
                ps.tot_dst[q] = b->tot[slot][parity][r];
                ps.tot_flag[q] = &b->tot_seq[slot][r];
            }
            IpcBox* me = (IpcBox*)I->mine;
            w.tot_flag = me->tot_seq[slot];
            w.tot_seq = seq;
            w.table = &me->tot[slot][parity][0][0];
            w.tot = c->dist.solver_tot.as<double>() + 8 * slot;
            w.nr = nr;
            w.self = r;
            c->dist.stat_allreduces++;
        }
        c->dist.stat_exchanges++;
        {
            ProfScope p1(&c->prof, "ipc_pack_push", c->stream);
            hipLaunchKernelGGL(k_ipc_pack_push, dim3(1), dim3(RANK_TOTALS_THREADS), 0, c->stream, ps);
        }
        w.status = c->status.as<DeviceStatus>();
        w.timeout_ticks = I->timeout_ticks;
        ProfScope p2(&c->prof, "ipc_wait_unpack", c->stream);
        hipLaunchKernelGGL(k_ipc_wait_unpack, dim3(1), dim3(1024), 0, c->stream, fw);
        return SPH_OK;
    }
    // queue the push and the wait of one round: the ghost / record messages of `x` (nullptr: none) and, with slot >= 0, the totals
    int round(Group& G, std::vector<Xfer>* x, int slot)
    {
        sph_ctx* c = G.m[0];
        IpcState* I = st(c);
        const int r = c->dist.rank, nr = c->dist.nranks;
        IpcPush ps{};
        IpcWait w{};
        bool any = false;
        for (int side = 0; side < 2 && x; side++) {
            Xfer& xf = (*x)[0];
            const int nb = side == 0 ? r - 1 : r + 1;
            if (nb < 0 || nb >= nr) {
                if (xf.send_bytes[side] || xf.recv_bytes[side]) return c->fail(SPH_ERR_DEVICE, "halo exchange across the outer edge of the slab row (rank %d)", r);
                continue;
            }
            if (!xf.send_bytes[side] && !xf.recv_bytes[side]) continue;
            if (xf.send_bytes[side] > I->bytes_per_side || xf.recv_bytes[side] > I->bytes_per_side) {
                comm_abandon(c);
                return c->fail(SPH_ERR_CAPACITY, "peer-mapped transport: a message of %zu bytes does not fit the %llu-byte inbox (sph_comm_ipc_export bytes_per_side)",
                               std::max(xf.send_bytes[side], xf.recv_bytes[side]), (unsigned long long)I->bytes_per_side);
            }
            any = true;
            const uint32_t seq = ++I->seq[side], parity = seq & 1u;
            const int oside = side ^ 1;   // I am my left neighbour's right side
            const uint32_t gran = (uint32_t)((xf.send_bytes[side] + 15) / 16);
            uint4* dst = (uint4*)IpcState::inbox(I->peer[nb], I->bytes_per_side, oside, parity);
            if (gran > 16384u) {   // > 256 KB: many workgroups copy, the push kernel only signals
                hipLaunchKernelGGL(k_ipc_copy, dim3(std::min(1024u, (gran + 255u) / 256u)), dim3(256), 0, c->stream, (const uint4*)xf.send[side], dst, gran);
                ps.granules[side] = 0;
            } else {
                ps.src[side] = (const uint4*)xf.send[side];
                ps.dst[side] = dst;
                ps.granules[side] = gran;
            }
            ps.flag[side] = &((IpcBox*)I->peer[nb])->data_seq[oside];
            ps.seq[side] = seq;
            w.flag[side] = &((IpcBox*)I->mine)->data_seq[side];
            w.seq[side] = seq;
            c->dist.stat_bytes_sent += xf.send_bytes[side];
            c->dist.stat_bytes_recv += xf.recv_bytes[side];
            xf.recv[side] = IpcState::inbox(I->mine, I->bytes_per_side, side, parity);   // the kernels behind the wait read the inbox in place
        }
        if (slot >= 0) {
            any = true;
            const uint32_t seq = ++I->tot_n[slot], parity = seq & 1u;
            ps.tot = c->dist.solver_tot.as<double>() + 8 * slot;
            ps.tot_seq = seq;
            ps.nr = nr;
            ps.self = r;
            for (int q = 0; q < nr; q++) {
                IpcBox* b = (IpcBox*)I->peer[q];
                ps.tot_dst[q] = b->tot[slot][parity][r];
                ps.tot_flag[q] = &b->tot_seq[slot][r];
            }
            IpcBox* me = (IpcBox*)I->mine;
            w.tot_flag = me->tot_seq[slot];
            w.tot_seq = seq;
            w.table = &me->tot[slot][parity][0][0];
            w.tot = c->dist.solver_tot.as<double>() + 8 * slot;
            w.nr = nr;
            w.self = r;
            c->dist.stat_allreduces++;
        }
        if (!any) return SPH_OK;
        if (x) c->dist.stat_exchanges++;
        {
            ProfScope p1(&c->prof, "ipc_push", c->stream);
            hipLaunchKernelGGL(k_ipc_push, dim3(1), dim3(1024), 0, c->stream, ps);
        }
        w.status = c->status.as<DeviceStatus>();
        w.timeout_ticks = I->timeout_ticks;
        ProfScope p2(&c->prof, "ipc_wait", c->stream);
        hipLaunchKernelGGL(k_ipc_wait, dim3(1), dim3(64), 0, c->stream, w);
        return SPH_OK;
    }
    int exchange(Group& G, std::vector<Xfer>& x) override { return round(G, &x, -1); }
    int allreduce_solver(Group& G, int slot) override { return round(G, nullptr, slot); }
    int exchange_and_allreduce_solver(Group& G, std::vector<Xfer>& x, int slot) override { return round(G, &x, slot); }
};
static IpcComm g_ipc;

// this rank's box: allocated here, its IPC handle for the other ranks (the launcher all-gathers the n_ranks handles)
extern "C" int sph_comm_ipc_export(sph_ctx* c, uint64_t bytes_per_side, uint8_t handle_out[64])
{
    if (!c || !handle_out || bytes_per_side < 4096) return SPH_ERR_INVALID_ARGUMENT;
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is 64 bytes");
    if (!c->dist.on || !c->dist.shm) return c->fail(SPH_ERR_INVALID_ARGUMENT, "call sph_comm_init_shm before sph_comm_ipc_export (the host-value collectives stay the shared-memory transport's)");
    if (c->dist.nranks > SHM_MAX_RANKS) return c->fail(SPH_ERR_INVALID_ARGUMENT, "peer-mapped transport: at most %d ranks", SHM_MAX_RANKS);
    HIPCHK(c, hipSetDevice(c->device));
    IpcState* I = new IpcState();
    I->bytes_per_side = (bytes_per_side + 255) & ~(uint64_t)255;
    const size_t total = IpcBox::size_for(I->bytes_per_side);
    if (const char* e = getenv("SPH_IPC_TIMEOUT_MS")) I->timeout_ticks = (uint64_t)std::max(1, atoi(e)) * 100000ull;
    // FINE-GRAINED device memory (advisor r4): another GPU writes this box -- inbox, flags, totals -- while kernels of this one spin on
    // it and then read it in place.  HIP guarantees visibility of a peer's writes DURING a kernel only for fine-grained allocations
    // (a coarse-grained hipMalloc may serve the poll from this GPU's L2, which remote xGMI writes bypass).
    if (hipExtMallocWithFlags((void**)&I->mine, total, hipDeviceMallocFinegrained) != hipSuccess) {
        delete I;
        return c->fail(SPH_ERR_DEVICE, "peer-mapped transport: no fine-grained device memory (%zu bytes)", total);
    }
    HIPCHK(c, hipMemset(I->mine, 0, IpcBox::header_bytes()));
    HIPCHK(c, hipDeviceSynchronize());
    hipIpcMemHandle_t h;
    const hipError_t e = hipIpcGetMemHandle(&h, I->mine);
    if (e != hipSuccess) {
        (void)hipFree(I->mine);
        delete I;
        return c->fail(SPH_ERR_DEVICE, "hipIpcGetMemHandle failed: %s", hipGetErrorString(e));
    }
    memcpy(handle_out, &h, 64);
    c->dist.ipc = I;
    return SPH_OK;
}
// map the other ranks' boxes: `handles` = the n_ranks handles of sph_comm_ipc_export in rank order.  From here on the ghost / record
// exchanges and the totals all-reduce of this context are pushed device to device.
extern "C" int sph_comm_init_ipc(sph_ctx* c, const uint8_t* handles, int n_ranks)
{
    if (!c || !handles || !c->dist.ipc || n_ranks != c->dist.nranks) return SPH_ERR_INVALID_ARGUMENT;
    HIPCHK(c, hipSetDevice(c->device));
    IpcState* I = (IpcState*)c->dist.ipc;
    for (int r = 0; r < n_ranks; r++) {
        if (r == c->dist.rank) {
            I->peer[r] = I->mine;
            continue;
        }
        hipIpcMemHandle_t h;
        memcpy(&h, handles + (size_t)64 * r, 64);
        void* p = nullptr;
        const hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) return c->fail(SPH_ERR_DEVICE, "hipIpcOpenMemHandle (rank %d's box) failed: %s", r, hipGetErrorString(e));
        I->peer[r] = (uint8_t*)p;
    }
    return SPH_OK;
}

// the transports carry no state of their own (everything lives in the contexts / the thread group): one object each serves every group
static RcclComm g_rccl;
static LocalComm g_local;
static ThreadComm g_threads;
static ShmComm g_shm;

Comm* comm_loopback() { return &g_local; }
int comm_for_rank(sph_ctx* c, Comm** out)
{
    *out = nullptr;
    if (!c->dist.on) return SPH_OK;
    if (c->dist.tgroup) *out = &g_threads;
    else if (c->dist.ipc && ((IpcState*)c->dist.ipc)->peer[c->dist.rank]) *out = &g_ipc;   // (exported AND mapped)
    else if (c->dist.shm) *out = &g_shm;
    else if (c->dist.nccl) *out = &g_rccl;
    else return c->fail(SPH_ERR_INVALID_ARGUMENT, "slab context without a communicator: call sph_comm_init or use sph_group_step");
    return SPH_OK;
}
void comm_describe(sph_ctx* c, uint32_t* transport, uint32_t* ranks)
{
    *transport = 0;
    *ranks = 0;
    if (!c->dist.on) return;
    *transport = 1;
    *ranks = (uint32_t)c->dist.nranks;
    if (c->dist.tgroup) *transport = 3;
    else if (c->dist.ipc && ((IpcState*)c->dist.ipc)->peer[c->dist.rank]) {
        *transport = 5;
        uint32_t mapped = 0;
        for (int r = 0; r < SHM_MAX_RANKS; r++) mapped += ((IpcState*)c->dist.ipc)->peer[r] != nullptr;
        *ranks = mapped;
    } else if (c->dist.shm) *transport = 4;
    else if (c->dist.nccl) {
        *transport = 2;
        int cnt = 0;
        *ranks = ncclCommCount((ncclComm_t)c->dist.nccl, &cnt) == ncclSuccess ? (uint32_t)cnt : 0u;
    }
}
void comm_abandon(sph_ctx* c)
{
    if (c->dist.tgroup) ((ThreadGroup*)c->dist.tgroup)->abandon();   // the other ranks' next collective reports it instead of waiting
    if (c->dist.shm) ((ShmSegment*)c->dist.shm)->broken.store(1u);
}


extern "C" int sph_comm_unique_id(uint8_t id_out[128])
{
    if (!id_out) return SPH_ERR_INVALID_ARGUMENT;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) return SPH_ERR_DEVICE;
    memcpy(id_out, &id, 128);
    return SPH_OK;
}

extern "C" int sph_comm_init(sph_ctx* c, const uint8_t id[128], int rank, int n_ranks)
{
    if (!c || !id || rank < 0 || n_ranks < 1 || rank >= n_ranks) return SPH_ERR_INVALID_ARGUMENT;
    if (n_ranks == 1 && !c->dist.on) return SPH_OK;
    if (!c->dist.on || c->dist.rank != rank || c->dist.nranks != n_ranks)
        return c->fail(SPH_ERR_INVALID_ARGUMENT, "call sph_dist_configure(rank, n_ranks, cuts) before sph_comm_init");
    HIPCHK(c, hipSetDevice(c->device));
    ncclUniqueId uid;
    memcpy(&uid, id, 128);
    ncclComm_t nc;
    NCCLCHK(c, ncclCommInitRank(&nc, n_ranks, uid, rank));
    c->dist.nccl = nc;
    HIPCHK(c, c->dist.tot_table.ensure((size_t)n_ranks * 8 * 2 * sizeof(double)));   // two slots (chained solves) of one row per rank
    HIPCHK(c, hipMemset(c->dist.tot_table.p, 0, (size_t)n_ranks * 8 * 2 * sizeof(double)));
    return SPH_OK;
}

extern "C" int sph_thread_group_create(int n_ranks, void** out)
{
    if (!out || n_ranks < 1) return SPH_ERR_INVALID_ARGUMENT;
    *out = new ThreadGroup(n_ranks);
    return SPH_OK;
}
extern "C" void sph_thread_group_destroy(void* group)
{
    delete (ThreadGroup*)group;
}
extern "C" int sph_comm_init_threads(sph_ctx* c, void* group, int rank, int n_ranks)
{
    if (!c || !group || rank < 0 || n_ranks < 1 || rank >= n_ranks || ((ThreadGroup*)group)->n != n_ranks) return SPH_ERR_INVALID_ARGUMENT;
    if (!c->dist.on || c->dist.rank != rank || c->dist.nranks != n_ranks)
        return c->fail(SPH_ERR_INVALID_ARGUMENT, "call sph_dist_configure(rank, n_ranks, cuts) before sph_comm_init_threads");
    c->dist.tgroup = group;
    return SPH_OK;
}

// the shared-memory transport of processes on one node (ShmComm): rank 0 passes create = 1 and makes the segment, the others map it
// once it exists (the launcher orders the two: a barrier of its own between rank 0's call and the others')
extern "C" int sph_comm_init_shm(sph_ctx* c, const char* name, int rank, int n_ranks, uint64_t bytes_per_side, int create)
{
    if (!c || !name || name[0] != '/' || rank < 0 || n_ranks < 1 || n_ranks > SHM_MAX_RANKS || rank >= n_ranks || bytes_per_side < 4096) return SPH_ERR_INVALID_ARGUMENT;
    if (!c->dist.on || c->dist.rank != rank || c->dist.nranks != n_ranks)
        return c->fail(SPH_ERR_INVALID_ARGUMENT, "call sph_dist_configure(rank, n_ranks, cuts) before sph_comm_init_shm");
    const size_t total = ShmSegment::size_for(n_ranks, bytes_per_side);
    int fd = -1;
    if (create) {
        (void)shm_unlink(name);
        fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)total) != 0) {
            if (fd >= 0) close(fd);
            return c->fail(SPH_ERR_DEVICE, "shm_open / ftruncate(%s, %zu bytes) failed: %s", name, total, strerror(errno));
        }
    } else {
        fd = shm_open(name, O_RDWR, 0600);
        if (fd < 0) return c->fail(SPH_ERR_DEVICE, "shm_open(%s) failed: %s (rank 0 creates the segment first)", name, strerror(errno));
    }
    void* mem = mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (mem == MAP_FAILED) return c->fail(SPH_ERR_DEVICE, "mmap of %s (%zu bytes) failed: %s", name, total, strerror(errno));
    ShmSegment* g = (ShmSegment*)mem;
    if (create) {
        memset(mem, 0, sizeof(ShmSegment));   // (ftruncate zero-filled the rest)
        g->n = (uint32_t)n_ranks;
        g->bytes_per_side = bytes_per_side;
        g->total_bytes = total;
        std::atomic_thread_fence(std::memory_order_release);
        g->magic = 0x53504853u;
    } else if (g->magic != 0x53504853u || g->n != (uint32_t)n_ranks || g->bytes_per_side != bytes_per_side) {
        munmap(mem, total);
        return c->fail(SPH_ERR_INVALID_ARGUMENT, "%s is not the segment of this launch (%d ranks, %llu bytes per side)", name, n_ranks, (unsigned long long)bytes_per_side);
    }
    c->dist.shm = mem;
    c->dist.shm_bytes = total;
    c->dist.shm_name = create ? name : "";
    return SPH_OK;
}

void dist_release(sph_ctx* c)
{
    auto& d = c->dist;
    if (d.ipc) {
        IpcState* I = (IpcState*)d.ipc;
        (void)hipSetDevice(c->device);
        for (int r = 0; r < SHM_MAX_RANKS; r++)
            if (I->peer[r] && I->peer[r] != I->mine) (void)hipIpcCloseMemHandle(I->peer[r]);
        if (I->mine) (void)hipFree(I->mine);
        delete I;
        d.ipc = nullptr;
    }
    if (d.shm) {
        munmap(d.shm, d.shm_bytes);
        if (!d.shm_name.empty()) (void)shm_unlink(d.shm_name.c_str());   // (the creating rank; the others' mappings stay valid)
    }
    d.shm = nullptr;
    if (d.nccl) ncclCommDestroy((ncclComm_t)d.nccl);
    d.nccl = nullptr;
    DevBuf* all[] = {&d.owned, &d.ring1, &d.ring1_src, &d.halo_idx, &d.halo_pos, &d.halo_src, &d.ghost_dst, &d.send[0], &d.send[1], &d.recv[0], &d.recv[1], &d.counts, &d.solver_tot, &d.tot_table, &d.hist, &d.cls, &d.blk, &d.edge};
    for (auto b : all) b->release();
    if (d.xstream) {
        (void)hipStreamSynchronize(d.xstream);
        (void)hipStreamDestroy(d.xstream);
    }
    d.xstream = nullptr;
    for (auto& e : d.ev_x) {
        if (e) (void)hipEventDestroy(e);
        e = nullptr;
    }
    for (hipEvent_t* e : {&d.ev_pack, &d.ev_copied, &d.ev_tot}) {
        if (*e) (void)hipEventDestroy(*e);
        *e = nullptr;
    }
    if (d.gtot) (void)hipHostFree(d.gtot);
    d.gtot = nullptr;
    if (d.counts_host) (void)hipHostFree(d.counts_host);
    d.counts_host = nullptr;
}
