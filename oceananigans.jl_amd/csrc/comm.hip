// comm.hip -- the collectives of the slab-x decomposition behind the C ABI: RCCL (librccl, linked directly) over xGMI.
//
// What it replaces in the reference (all MPI.jl, src/DistributedComputations/):
//   * Distributed(...) communicator set-up                      distributed_architectures.jl:167-297
//   * fill_halo_event! west / east: pack -> Isend / Irecv! -> (async) -> Waitall -> unpack
//                                                               halo_communication.jl:210-229, 267-366, distributed_fields.jl:58-75,
//                                                               Fields/field_boundary_buffers.jl:276-308
//   * transpose_y_to_x! / transpose_x_to_y!: Alltoallv! with equal counts       distributed_transpose.jl:185-191
//   * MPI.Allreduce of scalars (Δt, max|u|)                      Simulations/simulation.jl:128-134
//
// MI355X shape: one process per GPU and one RCCL communicator per process.  Every exchange is a grouped ncclSend / ncclRecv on the
// communicator's OWN non-blocking stream, ordered against the caller's compute stream by two events (ready: the packed buffers
// are complete; done: the received buffers are complete) -- there is no host synchronisation anywhere (the reference calls
// sync_device! before every post and Waitall after).  Between *_begin and *_end the caller's stream is free to run the interior
// tendency kernels (interleave_communication_and_computation.jl:29-67).  The all-to-all of the slab transposes runs on the caller's
// stream: it sits on the critical path of the pressure solve.  xGMI is point to point, so the R-1 peers of an all-to-all and the 2
// neighbours of a halo exchange each use their own link; nothing here is a ring.
#include <rccl/rccl.h>

#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <thread>
#include <vector>

#include "ocn_internal.h"

using namespace ocn;

#define OCN_CHECK_NCCL(call)                                                               \
    do {                                                                                   \
        ncclResult_t r_ = (call);                                                          \
        if (r_ != ncclSuccess) {                                                           \
            ocn::set_error("%s failed: %s (%s:%d)", #call, ncclGetErrorString(r_), __FILE__, __LINE__); \
            return OCN_ERR_COMM;                                                           \
        }                                                                                  \
    } while (0)

namespace {

// An open ncclGroupStart() must be closed on EVERY path: an early return from inside the group would leave it open, every later NCCL
// call of the thread would be queued into it and never launched, and the caller would hang instead of seeing OCN_ERR_COMM.
struct NcclGroup {
    bool open = false;
    ncclResult_t start() { ncclResult_t r = ncclGroupStart(); open = (r == ncclSuccess); return r; }
    ncclResult_t end() { open = false; return ncclGroupEnd(); }
    ~NcclGroup() { if (open) (void)ncclGroupEnd(); }
};

struct Comm {
    ncclComm_t comm = nullptr;
    int rank = 0, nranks = 1, west = 0, east = 0;
    hipStream_t stream = nullptr;            // communication stream
    hipEvent_t ready = nullptr, done = nullptr;
    double *buf[4] = {nullptr, nullptr, nullptr, nullptr};  // send west, send east, recv west, recv east
    size_t cap = 0;                          // doubles per buffer
    double *plane[2] = {nullptr, nullptr};   // one-plane exchange: send, recv
    size_t plane_cap = 0;
    double *pbuf[4] = {nullptr, nullptr, nullptr, nullptr};  // pressure-plane exchange: send west, send east, recv west, recv east
    size_t pcap = 0;
    bool pending = false;
    size_t pending_count = 0;
    bool pending_wrap = false;  // the exchange in flight carries interior rows only (ocn_halo_exchange_begin_packed): unpack wraps
    double *scalar = nullptr;   // one double for ocn_comm_barrier's all-reduce (the strip buffers may hold strips written by a tendency launch)
    bool self_via_rccl = false;  // OCN_COMM_SELF_VIA_RCCL=1: a rank's transfers to itself go through ncclSend / ncclRecv too (tests)
    struct LocalGroup *local = nullptr;  // != NULL: the in-process transport below instead of RCCL (ocn_comm_init_local)
    // ocn_comm_init_replica: this process is rank 0 of `nranks` IDENTICAL ranks (an x-periodic flow of period Lx / nranks): what a peer
    // would send me is what I send to its mirror image, so every receive is an asynchronous device copy from one of my own send buffers.
    // Timing of ONE rank of an R-rank run on a one-GPU box with the R-rank schedules, kernels and drivers (tools/bench_dist_rank.py).
    bool replica = false;
    // ocn_comm_enable_stats: event pairs around every exchange (category, start, stop), summed by ocn_comm_stats
    bool stats = false;
    struct StatPair { int cat; hipEvent_t a, b; };
    std::vector<StatPair> pairs;
    double counts[8] = {};
};

// brackets a piece of work on `s` with timing events while the communicator's stats are enabled
struct StatScope {
    Comm *c; int cat; hipStream_t s; hipEvent_t a = nullptr;
    StatScope(Comm *c_, int cat_, hipStream_t s_) : c(c_), cat(cat_), s(s_)
    {
        if (c->stats && hipEventCreate(&a) == hipSuccess) (void)hipEventRecord(a, s);
    }
    ~StatScope()
    {
        if (!a) return;
        hipEvent_t b = nullptr;
        if (hipEventCreate(&b) == hipSuccess) {
            (void)hipEventRecord(b, s);
            c->pairs.push_back({cat, a, b});
        } else {
            (void)hipEventDestroy(a);
        }
    }
};

// ---- in-process transport (ocn_comm_init_local): the ranks are THREADS of one process sharing ONE GPU.  RCCL refuses two ranks on one
// device, so on a one-GPU box everything above the transport -- the send / recv schedules, the pack / unpack launches, the event ordering
// between the compute and the communication stream, the pressure-plane exchange, the C drivers -- could only run with one rank; with this
// transport it runs with R = 2, 4, 8 (tests/test_gpu_distributed.py).  A send copies into a staging buffer on the sender's stream and
// posts (buffer, event) into the mailbox of the (source, destination) pair; the matching receive -- the k-th of that pair, the same
// pairing rule as RCCL's -- takes it, makes its stream wait for the event and copies out.  Within a group all sends are posted before the
// first receive blocks, so ranks that issue the same sequence of groups cannot deadlock.  Not a product path: one GPU, host-blocking.
struct LocalMsg {
    double *stage;
    size_t count;
    hipEvent_t ready;
};
struct LocalGroup {
    int nranks = 0, joined = 0, left = 0;
    std::mutex m;
    std::condition_variable cv;
    std::map<std::pair<int, int>, std::deque<LocalMsg>> box;  // (source, destination) -> messages in issue order
    std::vector<std::pair<double *, size_t>> free_stage;      // staging buffers whose last reader has been enqueued on the null-ordered streams
    std::vector<hipEvent_t> free_events;
    // host barrier + all-reduce scratch
    int arrived = 0, generation = 0;
    std::vector<double> red;
    int red_count = 0;
};
std::mutex g_local_mutex;
std::map<long long, LocalGroup *> g_local_groups;

int local_send(Comm *c, const double *buf, size_t count, int dst, hipStream_t stream)
{
    LocalGroup *G = c->local;
    LocalMsg msg{nullptr, count, nullptr};
    {
        std::lock_guard<std::mutex> lk(G->m);
        for (size_t q = 0; q < G->free_stage.size(); ++q)
            if (G->free_stage[q].second >= count) {
                msg.stage = G->free_stage[q].first;
                msg.count = G->free_stage[q].second;
                G->free_stage.erase(G->free_stage.begin() + q);
                break;
            }
        if (!G->free_events.empty()) {
            msg.ready = G->free_events.back();
            G->free_events.pop_back();
        }
    }
    if (!msg.stage) {
        OCN_CHECK_HIP(hipMalloc(&msg.stage, (count ? count : 1) * sizeof(double)));
        msg.count = count;
    }
    if (!msg.ready) OCN_CHECK_HIP(hipEventCreateWithFlags(&msg.ready, hipEventDisableTiming));
    // (a recycled staging buffer re-entered the free list only after the copy that read it had completed: local_recv synchronises)
    OCN_CHECK_HIP(hipMemcpyAsync(msg.stage, buf, count * sizeof(double), hipMemcpyDeviceToDevice, stream));
    OCN_CHECK_HIP(hipEventRecord(msg.ready, stream));
    msg.count = count;  // (a recycled buffer may be larger; it re-enters the free list with this size)
    {
        std::lock_guard<std::mutex> lk(G->m);
        G->box[{c->rank, dst}].push_back(msg);
    }
    G->cv.notify_all();
    return OCN_SUCCESS;
}

int local_recv(Comm *c, double *buf, size_t count, int src, hipStream_t stream)
{
    LocalGroup *G = c->local;
    LocalMsg msg;
    {
        std::unique_lock<std::mutex> lk(G->m);
        auto &q = G->box[{src, c->rank}];
        if (!G->cv.wait_for(lk, std::chrono::seconds(120), [&] { return !q.empty(); })) {
            ocn::set_error("in-process transport: rank %d waited 120 s for a message from rank %d (mismatched schedules?)", c->rank, src);
            return OCN_ERR_COMM;
        }
        msg = q.front();
        q.pop_front();
    }
    if (msg.count != count) {
        ocn::set_error("in-process transport: rank %d expected %zu doubles from rank %d, the matching send has %zu", c->rank, count, src, msg.count);
        return OCN_ERR_COMM;
    }
    OCN_CHECK_HIP(hipStreamWaitEvent(stream, msg.ready, 0));
    OCN_CHECK_HIP(hipMemcpyAsync(buf, msg.stage, count * sizeof(double), hipMemcpyDeviceToDevice, stream));
    // the staging buffer may be reused once this copy has run: wait for it here (a test transport: simplicity over overlap)
    OCN_CHECK_HIP(hipStreamSynchronize(stream));
    {
        std::lock_guard<std::mutex> lk(G->m);
        G->free_stage.push_back({msg.stage, count});
        G->free_events.push_back(msg.ready);
    }
    return OCN_SUCCESS;
}

// all sends of the group first, then the receives (see above)
int local_run_ops(Comm *c, const ocn_comm_op *ops, int n, const double *const *send, double *const *recv, size_t count, hipStream_t stream)
{
    for (int q = 0; q < n; ++q)
        if (!ops[q].is_recv) {
            int st = local_send(c, send[ops[q].slot], count, ops[q].peer, stream);
            if (st != OCN_SUCCESS) return st;
        }
    for (int q = 0; q < n; ++q)
        if (ops[q].is_recv) {
            int st = local_recv(c, recv[ops[q].slot], count, ops[q].peer, stream);
            if (st != OCN_SUCCESS) return st;
        }
    return OCN_SUCCESS;
}

int local_barrier(Comm *c)
{
    LocalGroup *G = c->local;
    std::unique_lock<std::mutex> lk(G->m);
    const int gen = G->generation;
    if (++G->arrived == G->nranks) {
        G->arrived = 0;
        ++G->generation;
        G->cv.notify_all();
        return OCN_SUCCESS;
    }
    if (!G->cv.wait_for(lk, std::chrono::seconds(120), [&] { return G->generation != gen; })) {
        ocn::set_error("in-process transport: rank %d waited 120 s at a barrier", c->rank);
        return OCN_ERR_COMM;
    }
    return OCN_SUCCESS;
}

// all-reduce of a small device buffer through the host (Δt, max|u|, checksums): every rank adds its values, the last one publishes
int local_allreduce(Comm *c, double *buf, size_t count, int op, hipStream_t stream)
{
    LocalGroup *G = c->local;
    std::vector<double> mine(count);
    OCN_CHECK_HIP(hipMemcpyAsync(mine.data(), buf, count * sizeof(double), hipMemcpyDeviceToHost, stream));
    OCN_CHECK_HIP(hipStreamSynchronize(stream));
    int st = local_barrier(c);  // the previous reduction's result has been read by everybody
    if (st != OCN_SUCCESS) return st;
    {
        std::lock_guard<std::mutex> lk(G->m);
        if (G->red_count == 0) G->red = mine;
        else
            for (size_t q = 0; q < count; ++q)
                G->red[q] = op == 0 ? G->red[q] + mine[q] : op == 1 ? (mine[q] > G->red[q] ? mine[q] : G->red[q]) : (mine[q] < G->red[q] ? mine[q] : G->red[q]);
        ++G->red_count;
    }
    st = local_barrier(c);  // everybody has contributed
    if (st != OCN_SUCCESS) return st;
    std::vector<double> res;
    {
        std::lock_guard<std::mutex> lk(G->m);
        res = G->red;
    }
    st = local_barrier(c);  // everybody has read
    if (st != OCN_SUCCESS) return st;
    {
        std::lock_guard<std::mutex> lk(G->m);
        G->red_count = 0;
    }
    OCN_CHECK_HIP(hipMemcpyAsync(buf, res.data(), count * sizeof(double), hipMemcpyHostToDevice, stream));
    OCN_CHECK_HIP(hipStreamSynchronize(stream));  // `res` lives on this stack frame
    return OCN_SUCCESS;
}

int ensure(Comm *c, size_t n)
{
    if (n <= c->cap) return OCN_SUCCESS;
    for (double *&b : c->buf) {
        if (b) OCN_CHECK_HIP(hipFree(b));
        b = nullptr;
        OCN_CHECK_HIP(hipMalloc(&b, n * sizeof(double)));
    }
    c->cap = n;
    return OCN_SUCCESS;
}

// the strips of all fields follow one another: doubles per side
size_t strip_doubles(const ocn_grid *grid, const int32_t *locs, int n)
{
    GridDev g = to_dev(*grid);
    size_t tot = 0;
    for (int q = 0; q < n; ++q) {
        Lay L = make_lay(g, locs[q]);
        tot += (size_t)g.Hx * L.sy * L.sz;
    }
    return tot;
}

int make_tuple(const ocn_grid *grid, double *const *fields, const int32_t *locs, int n, FieldTuple &ft)
{
    OCN_REQUIRE(fields && locs && n >= 1 && n <= MAX_TUPLE, "halo exchange: 1..%d fields", MAX_TUPLE);
    ft.n = n;
    for (int q = 0; q < n; ++q) {
        OCN_REQUIRE(fields[q], "halo exchange: null field pointer");
        ft.f[q] = fields[q];
        ft.loc[q] = locs[q];
    }
    (void)grid;
    return OCN_SUCCESS;
}

// ---- the send / recv schedules as PURE HOST FUNCTIONS (ocn_comm_schedule exports them; tests/test_comm_schedule.py simulates every
// rank of R = 1, 2, 3, 8 and checks that the k-th send of rank a to rank b meets the k-th receive of b from a, with the buffers the
// choreography means).  RCCL matches point-to-point operations of one group per (sender, receiver) pair in issue order.
//   kind OCN_SCHED_STRIPS: slots 0 send_west, 1 send_east, 2 recv_west, 3 recv_east.  My west strip becomes the west neighbour's east
//     halo and vice versa.  When both neighbours are the same peer (R = 2) or myself (R = 1) the receives are posted in the order
//     (from east, from west) so that they pair up with that peer's (west, east) sends.
//   kind OCN_SCHED_PLANE_EAST / _WEST: slot 0 send, 1 recv.  The plane I need from the east is my east neighbour's WEST interior plane.
//   kind OCN_SCHED_ALL_TO_ALL: slot = chunk index d (send chunk d to rank d, receive chunk d from rank d).
//   kind OCN_SCHED_ALL_GATHER: send slot 0 (my chunk) to every peer, receive slot s from rank s: the all-gather as R - 1 direct
//     transfers, one per xGMI link, instead of a ring's R - 1 hops over one link each.
int build_schedule(int kind, int rank, int nranks, bool self_via_rccl, ocn_comm_op *ops, int cap, int *n)
{
    const int west = (rank + nranks - 1) % nranks, east = (rank + 1) % nranks;
    int m = 0;
    auto push = [&](int is_recv, int peer, int slot) {
        if (m < cap) ops[m] = ocn_comm_op{is_recv, peer, slot};
        ++m;
    };
    switch (kind) {
        case OCN_SCHED_STRIPS:
            if (nranks == 1 && !self_via_rccl) break;  // device copies
            push(0, west, 0);
            push(0, east, 1);
            if (nranks > 2) { push(1, west, 2); push(1, east, 3); }
            else            { push(1, east, 3); push(1, west, 2); }
            break;
        case OCN_SCHED_PLANE_EAST:
        case OCN_SCHED_PLANE_WEST: {
            if (nranks == 1 && !self_via_rccl) break;
            const bool e = kind == OCN_SCHED_PLANE_EAST;
            push(0, e ? west : east, 0);
            push(1, e ? east : west, 1);
            break;
        }
        case OCN_SCHED_ALL_TO_ALL:
            for (int d = 0; d < nranks; ++d) {
                if (d == rank && !self_via_rccl) continue;  // own chunk: a device copy
                push(0, d, d);
                push(1, d, d);
            }
            break;
        case OCN_SCHED_ALL_GATHER:
            for (int d = 0; d < nranks; ++d) {
                if (d == rank && !self_via_rccl) continue;  // own chunk: a device copy
                push(0, d, 0);
                push(1, d, d);
            }
            break;
        default:
            ocn::set_error("ocn_comm_schedule: unknown kind %d", kind);
            return OCN_ERR_INVALID_ARGUMENT;
    }
    *n = m;
    if (m > cap) {
        ocn::set_error("ocn_comm_schedule: %d operations do not fit the capacity %d", m, cap);
        return OCN_ERR_INVALID_ARGUMENT;
    }
    return OCN_SUCCESS;
}

// issue a schedule inside ONE group on `stream`; send[slot] / recv[slot] give the buffer of a slot, `count` doubles each
int run_schedule(Comm *c, int kind, const double *const *send, double *const *recv, size_t count, hipStream_t stream);

// grouped exchange with the two x neighbours on the communication stream.  My west strip becomes the west neighbour's east halo
// and vice versa.  When both neighbours are the same peer (R = 2) or myself (R = 1) the receives are posted in the order
// (from east, from west) so that they pair up with that peer's (west, east) sends.
int post_exchange(Comm *c, const double *sw, const double *se, double *rw, double *re, size_t count)
{
    if (c->nranks == 1 && !c->self_via_rccl) {
        // one rank: both neighbours are this rank; a device copy moves the strips at HBM speed (RCCL's self send / recv runs on a few
        // channels only)
        OCN_CHECK_HIP(hipMemcpyAsync(re, sw, count * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        OCN_CHECK_HIP(hipMemcpyAsync(rw, se, count * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        return OCN_SUCCESS;
    }
    const double *snd[4] = {sw, se, nullptr, nullptr};
    double *rcv[4] = {nullptr, nullptr, rw, re};
    return run_schedule(c, OCN_SCHED_STRIPS, snd, rcv, count, c->stream);
}

int run_schedule(Comm *c, int kind, const double *const *send, double *const *recv, size_t count, hipStream_t stream)
{
    ocn_comm_op ops[2 * OCN_COMM_MAX_RANKS];
    int n = 0;
    int st = build_schedule(kind, c->rank, c->nranks, c->self_via_rccl, ops, 2 * OCN_COMM_MAX_RANKS, &n);
    if (st != OCN_SUCCESS) return st;
    if (n == 0) return OCN_SUCCESS;
    if (c->replica) {
        // the k-th receive from peer s pairs with s's k-th send to me (RCCL's rule) = my k-th send to the mirror peer (R - s) mod R
        if (kind == OCN_SCHED_ALL_TO_ALL) {
            // chunks are addressed by absolute rank: the return exchange of a solve would have to deliver the other ranks' parts of the
            // solution, which nobody computes here (tried: zeros / own chunks -- the resulting pressure destabilises the run within two steps)
            ocn::set_error("replica transport: an all-to-all has no mirror image (use the transpose-free pressure solve, or a one-rank RCCL world at the local size)");
            return OCN_ERR_UNSUPPORTED;
        }
        int taken[OCN_COMM_MAX_RANKS] = {};
        for (int q = 0; q < n; ++q) {
            if (!ops[q].is_recv) continue;
            const int mirror = (c->nranks - ops[q].peer) % c->nranks;
            int seen = 0, slot = -1;
            for (int t = 0; t < n && slot < 0; ++t)
                if (!ops[t].is_recv && ops[t].peer == mirror && seen++ == taken[mirror]) slot = ops[t].slot;
            if (slot < 0) {
                ocn::set_error("replica transport: receive %d from rank %d has no matching send to rank %d", q, ops[q].peer, mirror);
                return OCN_ERR_COMM;
            }
            ++taken[mirror];
            OCN_CHECK_HIP(hipMemcpyAsync(recv[ops[q].slot], send[slot], count * sizeof(double), hipMemcpyDeviceToDevice, stream));
        }
        return OCN_SUCCESS;
    }
    if (c->local) return local_run_ops(c, ops, n, send, recv, count, stream);
    NcclGroup group;
    OCN_CHECK_NCCL(group.start());
    for (int q = 0; q < n; ++q) {
        if (ops[q].is_recv)
            OCN_CHECK_NCCL(ncclRecv(recv[ops[q].slot], count, ncclDouble, ops[q].peer, c->comm, stream));
        else
            OCN_CHECK_NCCL(ncclSend(send[ops[q].slot], count, ncclDouble, ops[q].peer, c->comm, stream));
    }
    OCN_CHECK_NCCL(group.end());
    return OCN_SUCCESS;
}

}  // namespace

extern "C" {

int ocn_comm_schedule(int32_t kind, int32_t rank, int32_t nranks, int32_t self_via_rccl, ocn_comm_op *ops, int32_t capacity, int32_t *n_ops)
{
    OCN_REQUIRE(ops && n_ops && capacity >= 0, "ocn_comm_schedule: null pointer");
    OCN_REQUIRE(nranks >= 1 && nranks <= OCN_COMM_MAX_RANKS && rank >= 0 && rank < nranks, "ocn_comm_schedule: rank %d of %d (at most %d ranks)",
                rank, nranks, OCN_COMM_MAX_RANKS);
    int n = 0;
    int st = build_schedule(kind, rank, nranks, self_via_rccl != 0, ops, capacity, &n);
    *n_ops = n;
    return st;
}

int ocn_comm_unique_id(void *id_out)
{
    OCN_REQUIRE(id_out, "ocn_comm_unique_id: null pointer");
    ncclUniqueId id;
    OCN_CHECK_NCCL(ncclGetUniqueId(&id));
    static_assert(sizeof(id) == OCN_COMM_UNIQUE_ID_BYTES, "ncclUniqueId size");
    memcpy(id_out, &id, sizeof(id));
    return OCN_SUCCESS;
}

int ocn_comm_init(ocn_comm_t *comm, int32_t rank, int32_t nranks, const void *unique_id)
{
    OCN_REQUIRE(comm && unique_id, "ocn_comm_init: null pointer");
    OCN_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "ocn_comm_init: rank %d of %d", rank, nranks);
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    Comm *c = new Comm();
    c->rank = rank;
    c->nranks = nranks;
    const char *sv = getenv("OCN_COMM_SELF_VIA_RCCL");
    c->self_via_rccl = sv && sv[0] == '1';
    c->west = (rank + nranks - 1) % nranks;  // neighbours wrap around (distributed_architectures.jl:386-429)
    c->east = (rank + 1) % nranks;
    ncclResult_t r = ncclCommInitRank(&c->comm, nranks, id, rank);  // on the current device (ocn_set_device / hipSetDevice first)
    if (r != ncclSuccess) {
        ocn::set_error("ncclCommInitRank(rank %d of %d) failed: %s", rank, nranks, ncclGetErrorString(r));
        delete c;
        return OCN_ERR_COMM;
    }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->ready, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->done, hipEventDisableTiming) != hipSuccess) {
        ocn::set_error("ocn_comm_init: stream / event creation failed");
        ncclCommDestroy(c->comm);
        delete c;
        return OCN_ERR_HIP;
    }
    *comm = c;
    return OCN_SUCCESS;
}

int ocn_comm_destroy(ocn_comm_t comm)
{
    Comm *c = static_cast<Comm *>(comm);
    if (!c) return OCN_SUCCESS;
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (double *b : c->buf)
        if (b) (void)hipFree(b);
    for (double *b : c->plane)
        if (b) (void)hipFree(b);
    for (double *b : c->pbuf)
        if (b) (void)hipFree(b);
    if (c->scalar) (void)hipFree(c->scalar);
    if (c->ready) (void)hipEventDestroy(c->ready);
    if (c->done) (void)hipEventDestroy(c->done);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->comm) ncclCommDestroy(c->comm);
    if (c->local) {  // the last rank to leave frees the group
        LocalGroup *G = c->local;
        bool last;
        {
            std::lock_guard<std::mutex> lk(G->m);
            last = (++G->left == G->nranks);
        }
        if (last) {
            std::lock_guard<std::mutex> lk(g_local_mutex);
            for (auto it = g_local_groups.begin(); it != g_local_groups.end(); ++it)
                if (it->second == G) {
                    g_local_groups.erase(it);
                    break;
                }
            (void)hipDeviceSynchronize();
            for (auto &b : G->free_stage) (void)hipFree(b.first);
            for (auto &kv : G->box)
                for (auto &msg : kv.second) (void)hipFree(msg.stage);
            for (hipEvent_t e : G->free_events) (void)hipEventDestroy(e);
            delete G;
        }
    }
    delete c;
    return OCN_SUCCESS;
}

// The in-process transport (see LocalGroup above): rank `rank` of `nranks` THREADS of this process that share the current device.  Every
// rank of a group passes the same `group_key` (any number the ranks agree on); the call returns when all ranks have joined.
int ocn_comm_init_local(ocn_comm_t *comm, int32_t rank, int32_t nranks, int64_t group_key)
{
    OCN_REQUIRE(comm, "ocn_comm_init_local: null pointer");
    OCN_REQUIRE(nranks >= 1 && nranks <= OCN_COMM_MAX_RANKS && rank >= 0 && rank < nranks, "ocn_comm_init_local: rank %d of %d", rank, nranks);
    LocalGroup *G;
    {
        std::lock_guard<std::mutex> lk(g_local_mutex);
        auto it = g_local_groups.find(group_key);
        if (it == g_local_groups.end()) {
            G = new LocalGroup();
            G->nranks = nranks;
            g_local_groups[group_key] = G;
        } else {
            G = it->second;
        }
    }
    OCN_REQUIRE(G->nranks == nranks, "ocn_comm_init_local: group %lld has %d ranks, not %d", (long long)group_key, G->nranks, nranks);
    Comm *c = new Comm();
    c->rank = rank;
    c->nranks = nranks;
    c->west = (rank + nranks - 1) % nranks;
    c->east = (rank + 1) % nranks;
    c->local = G;
    c->self_via_rccl = true;  // a rank's transfers to itself take the mailboxes too: the same code path as any other peer
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->ready, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->done, hipEventDisableTiming) != hipSuccess) {
        ocn::set_error("ocn_comm_init_local: stream / event creation failed");
        delete c;
        return OCN_ERR_HIP;
    }
    {
        std::unique_lock<std::mutex> lk(G->m);
        ++G->joined;
        G->cv.notify_all();
        if (!G->cv.wait_for(lk, std::chrono::seconds(120), [&] { return G->joined >= G->nranks; })) {
            ocn::set_error("ocn_comm_init_local: only %d of %d ranks joined group %lld within 120 s", G->joined, G->nranks, (long long)group_key);
            return OCN_ERR_COMM;
        }
    }
    *comm = c;
    return OCN_SUCCESS;
}

// Rank 0 of `nranks` identical ranks (see Comm::replica): no peer exists; the schedules, pack / unpack launches, stream ordering and
// drivers are those of an `nranks`-rank run, every transfer is a device copy.  A measurement tool, not a way to run a model.
int ocn_comm_init_replica(ocn_comm_t *comm, int32_t nranks)
{
    OCN_REQUIRE(comm, "ocn_comm_init_replica: null pointer");
    OCN_REQUIRE(nranks >= 1 && nranks <= OCN_COMM_MAX_RANKS, "ocn_comm_init_replica: 1..%d ranks", OCN_COMM_MAX_RANKS);
    Comm *c = new Comm();
    c->rank = 0;
    c->nranks = nranks;
    c->west = nranks - 1;
    c->east = 1 % nranks;
    c->replica = true;
    c->self_via_rccl = nranks == 1;  // one rank: its transfers to itself take the schedules too
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->ready, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->done, hipEventDisableTiming) != hipSuccess) {
        ocn::set_error("ocn_comm_init_replica: stream / event creation failed");
        delete c;
        return OCN_ERR_HIP;
    }
    *comm = c;
    return OCN_SUCCESS;
}

int ocn_comm_info(ocn_comm_t comm, int32_t *rank, int32_t *nranks, int32_t *rccl_version)
{
    Comm *c = static_cast<Comm *>(comm);
    OCN_REQUIRE(c, "ocn_comm_info: null communicator");
    int count = 0, ver = 0;
    if (c->local || c->replica) {
        count = c->nranks;  // rccl_version 0: the in-process transport
    } else {
        OCN_CHECK_NCCL(ncclCommCount(c->comm, &count));  // the number of ranks RCCL itself sees
        OCN_CHECK_NCCL(ncclGetVersion(&ver));
    }
    if (rank) *rank = c->rank;
    if (nranks) *nranks = count;
    if (rccl_version) *rccl_version = ver;
    return OCN_SUCCESS;
}

// fill_halo_event!(...; async = true) for the x direction of a field tuple: pack on `stream`, exchange on the communication stream.
int ocn_halo_exchange_begin(ocn_comm_t comm, const ocn_grid *grid, double *const *fields, const int32_t *locs, int32_t n, void *stream)
{
    Comm *c = static_cast<Comm *>(comm);
    OCN_REQUIRE(c, "ocn_halo_exchange_begin: null communicator");
    OCN_REQUIRE(!c->pending, "ocn_halo_exchange_begin: an exchange is already in flight (call ocn_halo_exchange_end first)");
    int st = validate_grid_any(grid);
    if (st != OCN_SUCCESS) return st;
    FieldTuple ft;
    st = make_tuple(grid, fields, locs, n, ft);
    if (st != OCN_SUCCESS) return st;
    const size_t count = strip_doubles(grid, locs, n);
    st = ensure(c, count);
    if (st != OCN_SUCCESS) return st;
    hipStream_t s = as_stream(stream);
    st = launch_halo_pack_x_fields(grid, ft, c->buf[0], c->buf[1], 0, s);
    if (st != OCN_SUCCESS) return st;
    OCN_CHECK_HIP(hipEventRecord(c->ready, s));
    OCN_CHECK_HIP(hipStreamWaitEvent(c->stream, c->ready, 0));
    {
        StatScope sc(c, 0, c->stream);
        st = post_exchange(c, c->buf[0], c->buf[1], c->buf[2], c->buf[3], count);
    }
    if (st != OCN_SUCCESS) return st;
    c->counts[5] += 1;
    OCN_CHECK_HIP(hipEventRecord(c->done, c->stream));
    c->pending = true;
    c->pending_count = count;
    return OCN_SUCCESS;
}

// The send buffers of the next ocn_halo_exchange_begin_packed of this tuple (device pointers, layout: field q at q * field_doubles, then
// h + Hx * parent row): a tendency launch writes the stepped values of the Hx westmost / eastmost columns straight into them.
int ocn_halo_exchange_buffers(ocn_comm_t comm, const ocn_grid *grid, const int32_t *locs, int32_t n, double **send_west, double **send_east,
                              int64_t *field_doubles)
{
    Comm *c = static_cast<Comm *>(comm);
    OCN_REQUIRE(c && locs && send_west && send_east && field_doubles, "ocn_halo_exchange_buffers: null pointer");
    OCN_REQUIRE(!c->pending, "ocn_halo_exchange_buffers: an exchange is in flight");
    int st = validate_grid(grid);
    if (st != OCN_SUCCESS) return st;
    OCN_REQUIRE(grid->ty == OCN_PERIODIC && grid->tz == OCN_PERIODIC, "ocn_halo_exchange_buffers: Periodic y and z (the receiver wraps the halo rows)");
    for (int q = 1; q < n; ++q) {
        GridDev g = to_dev(*grid);
        Lay L0 = make_lay(g, locs[0]), L = make_lay(g, locs[q]);
        OCN_REQUIRE(L.sy == L0.sy && L.sz == L0.sz, "ocn_halo_exchange_buffers: fields of one cross-section");
    }
    const size_t count = strip_doubles(grid, locs, n);
    st = ensure(c, count);
    if (st != OCN_SUCCESS) return st;
    *send_west = c->buf[0];
    *send_east = c->buf[1];
    *field_doubles = (int64_t)(count / (size_t)n);
    return OCN_SUCCESS;
}

// ocn_halo_exchange_begin without the pack launch: the send buffers (ocn_halo_exchange_buffers) were filled on `stream` -- interior rows
// only -- by the launch that produced the fields; ocn_halo_exchange_end then unpacks with periodically wrapped (j, k).
int ocn_halo_exchange_begin_packed(ocn_comm_t comm, const ocn_grid *grid, double *const *fields, const int32_t *locs, int32_t n, void *stream)
{
    Comm *c = static_cast<Comm *>(comm);
    OCN_REQUIRE(c, "ocn_halo_exchange_begin_packed: null communicator");
    OCN_REQUIRE(!c->pending, "ocn_halo_exchange_begin_packed: an exchange is already in flight (call ocn_halo_exchange_end first)");
    int st = validate_grid(grid);
    if (st != OCN_SUCCESS) return st;
    OCN_REQUIRE(grid->ty == OCN_PERIODIC && grid->tz == OCN_PERIODIC, "ocn_halo_exchange_begin_packed: Periodic y and z");
    FieldTuple ft;
    st = make_tuple(grid, fields, locs, n, ft);
    if (st != OCN_SUCCESS) return st;
    const size_t count = strip_doubles(grid, locs, n);
    OCN_REQUIRE(count <= c->cap && c->buf[0], "ocn_halo_exchange_begin_packed: call ocn_halo_exchange_buffers first");
    hipStream_t s = as_stream(stream);
    OCN_CHECK_HIP(hipEventRecord(c->ready, s));
    OCN_CHECK_HIP(hipStreamWaitEvent(c->stream, c->ready, 0));
    {
        StatScope sc(c, 0, c->stream);
        st = post_exchange(c, c->buf[0], c->buf[1], c->buf[2], c->buf[3], count);
    }
    if (st != OCN_SUCCESS) return st;
    c->counts[5] += 1;
    OCN_CHECK_HIP(hipEventRecord(c->done, c->stream));
    c->pending = true;
    c->pending_wrap = true;
    c->pending_count = count;
    return OCN_SUCCESS;
}

// synchronize_communication! (distributed_fields.jl:58-75): `stream` waits for the exchange (an event, not the host) and unpacks.
int ocn_halo_exchange_end(ocn_comm_t comm, const ocn_grid *grid, double *const *fields, const int32_t *locs, int32_t n, void *stream)
{
    Comm *c = static_cast<Comm *>(comm);
    OCN_REQUIRE(c, "ocn_halo_exchange_end: null communicator");
    OCN_REQUIRE(c->pending, "ocn_halo_exchange_end: no exchange in flight");
    FieldTuple ft;
    int st = make_tuple(grid, fields, locs, n, ft);
    if (st != OCN_SUCCESS) return st;
    OCN_REQUIRE(strip_doubles(grid, locs, n) == c->pending_count, "ocn_halo_exchange_end: not the tuple the exchange was started with");
    hipStream_t s = as_stream(stream);
    {
        StatScope sc(c, 1, s);  // what the caller's stream waits for the exchange
        OCN_CHECK_HIP(hipStreamWaitEvent(s, c->done, 0));
    }
    st = launch_halo_pack_x_fields(grid, ft, c->buf[2], c->buf[3], c->pending_wrap ? 2 : 1, s);
    if (st == OCN_SUCCESS) c->pending = c->pending_wrap = false;  // a failed unpack leaves the exchange pending: the caller may call _end again
    return st;
}

// ONE x-plane of `field` from a neighbour (the two synchronous fills inside the pressure projection, pressure_correction.jl:10-17,
// whose fields get their complete exchange in the following update_state!): side 0 ("east"): field[nx+1] <- east neighbour's
// field[1]; side 1 ("west"): field[0] <- west neighbour's field[nx].  In stream order on `stream`.
int ocn_halo_exchange_plane(ocn_comm_t comm, const ocn_grid *grid, double *field, int32_t loc, int32_t side, void *stream)
{
    Comm *c = static_cast<Comm *>(comm);
    OCN_REQUIRE(c && field, "ocn_halo_exchange_plane: null pointer");
    OCN_REQUIRE(!c->pending, "ocn_halo_exchange_plane: a strip exchange is in flight");
    int st = validate_grid_any(grid);
    if (st != OCN_SUCCESS) return st;
    GridDev g = to_dev(*grid);
    Lay L = make_lay(g, loc);
    const size_t count = (size_t)L.sy * L.sz;
    if (count > c->plane_cap) {
        for (double *&b : c->plane) {
            if (b) OCN_CHECK_HIP(hipFree(b));
            b = nullptr;
            OCN_CHECK_HIP(hipMalloc(&b, count * sizeof(double)));
        }
        c->plane_cap = count;
    }
    hipStream_t s = as_stream(stream);
    const bool east = side == 0;
    // the plane I need from the east is my east neighbour's WEST interior plane, and vice versa
    st = launch_halo_plane_x(grid, field, loc, east ? 0 : 1, c->plane[0], 0, s);
    if (st != OCN_SUCCESS) return st;
    if (c->nranks == 1 && !c->self_via_rccl)
        return launch_halo_plane_x(grid, field, loc, east ? 1 : 0, c->plane[0], 1, s);  // my own plane is the neighbour's
    {
        const double *snd[2] = {c->plane[0], nullptr};
        double *rcv[2] = {nullptr, c->plane[1]};
        StatScope sc(c, 4, s);
        st = run_schedule(c, east ? OCN_SCHED_PLANE_EAST : OCN_SCHED_PLANE_WEST, snd, rcv, count, s);
        if (st != OCN_SUCCESS) return st;
    }
    return launch_halo_plane_x(grid, field, loc, east ? 1 : 0, c->plane[1], 1, s);
}

// The x-halo payload of the correction-on-load stage (ocn_halo_pack_pressure / _unpack_pressure, capi.hip): in stream order on
// `stream`, own buffers (a strip exchange of the velocities may have been received into buf[] just before).
int ocn_halo_exchange_pressure(ocn_comm_t comm, const ocn_grid *grid, double *p, double *u, double dt_correct, void *stream)
{
    Comm *c = static_cast<Comm *>(comm);
    OCN_REQUIRE(c && p && u, "ocn_halo_exchange_pressure: null pointer");
    OCN_REQUIRE(!c->pending, "ocn_halo_exchange_pressure: a strip exchange is in flight (its unpack would overwrite the corrected u plane)");
    OCN_REQUIRE(grid != nullptr, "ocn_halo_exchange_pressure: null grid");
    GridDev g = to_dev(*grid);
    Lay L = make_lay(g, OCN_LOC_CCC);
    const size_t count = (size_t)L.sy * L.sz * (g.Hx + 1);
    if (count > c->pcap) {
        for (double *&b : c->pbuf) {
            if (b) OCN_CHECK_HIP(hipFree(b));
            b = nullptr;
            OCN_CHECK_HIP(hipMalloc(&b, count * sizeof(double)));
        }
        c->pcap = count;
    }
    int st = ocn_halo_pack_pressure(grid, p, u, dt_correct, c->pbuf[0], c->pbuf[1], stream);
    if (st != OCN_SUCCESS) return st;
    {
        StatScope sc(c, 3, as_stream(stream));
        st = ocn_comm_exchange_strips(comm, c->pbuf[0], c->pbuf[1], c->pbuf[2], c->pbuf[3], count, stream);
    }
    if (st != OCN_SUCCESS) return st;
    return ocn_halo_unpack_pressure(grid, p, u, c->pbuf[2], c->pbuf[3], stream);
}

// one strip per x neighbour in stream order (the wide halos of the split-explicit substepping, ocn_split_explicit_dist_*)
int ocn_comm_exchange_strips(ocn_comm_t comm, const double *send_west, const double *send_east, double *recv_west, double *recv_east,
                             size_t count, void *stream)
{
    Comm *c = static_cast<Comm *>(comm);
    OCN_REQUIRE(c && send_west && send_east && recv_west && recv_east, "ocn_comm_exchange_strips: null pointer");
    hipStream_t keep = c->stream;
    c->stream = as_stream(stream);  // post_exchange issues on c->stream
    int st = post_exchange(c, send_west, send_east, recv_west, recv_east, count);
    c->stream = keep;
    return st;
}

// Alltoallv! with equal counts (distributed_transpose.jl:188; transposable_field.jl:94-98): chunk d of `send` goes to rank d, chunk
// s of `recv` comes from rank s; `count` doubles per peer.  In stream order on `stream`.
int ocn_comm_all_to_all(ocn_comm_t comm, const double *send, double *recv, size_t count, void *stream)
{
    Comm *c = static_cast<Comm *>(comm);
    OCN_REQUIRE(c && send && recv, "ocn_comm_all_to_all: null pointer");
    OCN_REQUIRE(send != recv, "ocn_comm_all_to_all: in-place exchange is not supported");
    hipStream_t s = as_stream(stream);
    StatScope sc(c, 2, s);
    c->counts[6] += 1;
    // the chunk a rank keeps for itself (1 / R of the payload) is a device copy at HBM speed; the R - 1 others are one grouped
    // send / recv per peer, each over its own xGMI link
    if (!c->self_via_rccl)
        OCN_CHECK_HIP(hipMemcpyAsync(recv + (size_t)c->rank * count, send + (size_t)c->rank * count, count * sizeof(double),
                                     hipMemcpyDeviceToDevice, s));
    if (c->nranks == 1 && !c->self_via_rccl) return OCN_SUCCESS;
    OCN_REQUIRE(c->nranks <= OCN_COMM_MAX_RANKS, "ocn_comm_all_to_all: at most %d ranks", OCN_COMM_MAX_RANKS);
    const double *snd[OCN_COMM_MAX_RANKS];
    double *rcv[OCN_COMM_MAX_RANKS];
    for (int d = 0; d < c->nranks; ++d) {
        snd[d] = send + (size_t)d * count;
        rcv[d] = recv + (size_t)d * count;
    }
    return run_schedule(c, OCN_SCHED_ALL_TO_ALL, snd, rcv, count, s);
}

// MPI.Allgather with equal counts: chunk s of `recv` is rank s's `send` (count doubles each).  In stream order on `stream`.
int ocn_comm_all_gather(ocn_comm_t comm, const double *send, double *recv, size_t count, void *stream)
{
    Comm *c = static_cast<Comm *>(comm);
    OCN_REQUIRE(c && send && recv, "ocn_comm_all_gather: null pointer");
    hipStream_t s = as_stream(stream);
    StatScope sc(c, 2, s);
    c->counts[6] += 1;
    if (c->nranks == 1 && !c->self_via_rccl) {
        OCN_CHECK_HIP(hipMemcpyAsync(recv, send, count * sizeof(double), hipMemcpyDeviceToDevice, s));
        return OCN_SUCCESS;
    }
    // the GPUs of one node are fully connected by point-to-point xGMI links: R - 1 direct transfers of my chunk, one per link and all
    // at once, instead of the collective's ring (R - 1 hops, each bound by one link).  OCN_COMM_ALL_GATHER=collective selects
    // ncclAllGather (the in-process transport has only the direct form).
    // (read per call, not once per process: bench.py's conservative leg and the fast leg of one process differ in it; every rank of
    // a run must hold the same value when it calls)
    const char *form = getenv("OCN_COMM_ALL_GATHER");
    const bool collective = form && !strcmp(form, "collective");
    if (c->local || c->replica || !collective) {
        OCN_REQUIRE(c->nranks <= OCN_COMM_MAX_RANKS, "ocn_comm_all_gather: at most %d ranks", OCN_COMM_MAX_RANKS);
        if (!c->self_via_rccl)
            OCN_CHECK_HIP(hipMemcpyAsync(recv + (size_t)c->rank * count, send, count * sizeof(double), hipMemcpyDeviceToDevice, s));
        const double *snd[1] = {send};
        double *rcv[OCN_COMM_MAX_RANKS];
        for (int d = 0; d < c->nranks; ++d) rcv[d] = recv + (size_t)d * count;
        return run_schedule(c, OCN_SCHED_ALL_GATHER, snd, rcv, count, s);
    }
    OCN_CHECK_NCCL(ncclAllGather(send, recv, count, ncclDouble, c->comm, s));
    return OCN_SUCCESS;
}

// the exchanges of a distributed Poisson handle: direction 0 = y-local -> x-local (send -> recv), 1 = back.  The transpose-free
// pipeline (ocn_dist_poisson_pipeline = 3) has ONE exchange: direction 0 all-gathers the interface values, direction 1 does nothing.
int ocn_dist_poisson_exchange(ocn_dist_poisson_t handle, ocn_comm_t comm, int32_t direction, void *stream)
{
    double *yf, *xf, *snd, *rcv;
    int st = ocn_dist_poisson_buffers(handle, &yf, &xf, &snd, &rcv);
    if (st != OCN_SUCCESS) return st;
    int32_t nyt, r2c, fast;
    int64_t nel;
    st = ocn_dist_poisson_layout(handle, &nyt, &nel, &r2c);
    if (st != OCN_SUCCESS) return st;
    st = ocn_dist_poisson_pipeline(handle, &fast);
    if (st != OCN_SUCCESS) return st;
    if (fast == 3) {
        if (direction != 0) return OCN_SUCCESS;
        double *gs, *gr;
        int64_t per_rank;
        st = ocn_dist_poisson_gather_buffers(handle, &gs, &gr, &per_rank);
        if (st != OCN_SUCCESS) return st;
        return ocn_comm_all_gather(comm, gs, gr, (size_t)per_rank, stream);
    }
    Comm *c = static_cast<Comm *>(comm);
    OCN_REQUIRE(c, "ocn_dist_poisson_exchange: null communicator");
    const size_t per_peer = (size_t)nel * 2 / c->nranks;  // complex elements -> doubles, equal chunks
    OCN_REQUIRE((size_t)nel * 2 % c->nranks == 0, "ocn_dist_poisson_exchange: buffer not divisible by the number of ranks");
    // the slab pipelines alternate the roles of the two buffers (csrc/colfft.hip): forward send -> recv; backward recv -> send for the
    // periodic flavour (1) whose x pass works in place in recv, send -> recv again for the transposing paths (0) and the tridiagonal
    // flavour (2), which leave their result in send
    const bool swap = direction == 1 && fast == 1;
    const double *src = swap ? rcv : snd;
    double *dst = swap ? snd : rcv;
    return ocn_comm_all_to_all(comm, src, dst, per_peer, stream);
}

int ocn_comm_enable_stats(ocn_comm_t comm, int32_t enable)
{
    Comm *c = static_cast<Comm *>(comm);
    OCN_REQUIRE(c, "ocn_comm_enable_stats: null communicator");
    c->stats = enable != 0;
    if (c->stats) c->counts[5] = c->counts[6] = 0.0;  // the counts run from here
    return OCN_SUCCESS;
}

int ocn_comm_stats(ocn_comm_t comm, double *out_ms)
{
    Comm *c = static_cast<Comm *>(comm);
    OCN_REQUIRE(c && out_ms, "ocn_comm_stats: null pointer");
    for (int q = 0; q < 8; ++q) out_ms[q] = 0.0;
    int st = OCN_SUCCESS;
    const auto t0 = std::chrono::steady_clock::now();
    for (auto &p : c->pairs) {
        // the stop events complete in stream order: wait for each with a deadline (a stuck exchange must not hang the caller)
        while (st == OCN_SUCCESS) {
            const hipError_t e = hipEventQuery(p.b);
            if (e == hipSuccess) break;
            if (e != hipErrorNotReady) { ocn::set_error("ocn_comm_stats: %s", hipGetErrorString(e)); st = OCN_ERR_HIP; }
            else if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 120.0) {
                ocn::set_error("ocn_comm_stats: an exchange did not complete within 120 s");
                st = OCN_ERR_TIMEOUT;
            } else std::this_thread::sleep_for(std::chrono::microseconds(200));
        }
        if (st != OCN_SUCCESS) break;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess && p.cat >= 0 && p.cat < 5) out_ms[p.cat] += ms;
    }
    if (st == OCN_SUCCESS) {
        for (auto &p : c->pairs) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
        out_ms[5] = c->counts[5];
        out_ms[6] = c->counts[6];
    }
    c->pairs.clear();  // (after a timeout the unfinished events are leaked on purpose)
    c->counts[5] = c->counts[6] = 0.0;
    return st;
}

int ocn_comm_wait(ocn_comm_t comm, double seconds)
{
    Comm *c = static_cast<Comm *>(comm);
    OCN_REQUIRE(c, "ocn_comm_wait: null communicator");
    return ocn::wait_stream(c->stream, seconds, "ocn_comm_wait");
}

int ocn_comm_allreduce(ocn_comm_t comm, double *buf, size_t count, int32_t op, void *stream)
{
    Comm *c = static_cast<Comm *>(comm);
    OCN_REQUIRE(c && buf, "ocn_comm_allreduce: null pointer");
    OCN_REQUIRE(op >= 0 && op <= 2, "ocn_comm_allreduce: op 0 = sum, 1 = max, 2 = min");
    if (c->local) return local_allreduce(c, buf, count, op, as_stream(stream));
    if (c->replica) {  // identical ranks: max and min are the value itself, the sum is nranks times it
        if (op != 0 || c->nranks == 1) return OCN_SUCCESS;
        std::vector<double> h(count);
        OCN_CHECK_HIP(hipMemcpyAsync(h.data(), buf, count * sizeof(double), hipMemcpyDeviceToHost, as_stream(stream)));
        OCN_CHECK_HIP(hipStreamSynchronize(as_stream(stream)));
        for (double &x : h) x *= c->nranks;
        OCN_CHECK_HIP(hipMemcpyAsync(buf, h.data(), count * sizeof(double), hipMemcpyHostToDevice, as_stream(stream)));
        OCN_CHECK_HIP(hipStreamSynchronize(as_stream(stream)));
        return OCN_SUCCESS;
    }
    const ncclRedOp_t ops[3] = {ncclSum, ncclMax, ncclMin};
    OCN_CHECK_NCCL(ncclAllReduce(buf, buf, count, ncclDouble, ops[op], c->comm, as_stream(stream)));
    return OCN_SUCCESS;
}

// MPI.Barrier: a one-element all-reduce on the communication stream, then wait for it on the host
int ocn_comm_barrier(ocn_comm_t comm)
{
    Comm *c = static_cast<Comm *>(comm);
    OCN_REQUIRE(c, "ocn_comm_barrier: null communicator");
    if (!c->scalar) OCN_CHECK_HIP(hipMalloc(&c->scalar, sizeof(double)));
    OCN_REQUIRE(!c->pending, "ocn_comm_barrier: a halo exchange is in flight");
    if (c->local) {
        OCN_CHECK_HIP(hipStreamSynchronize(c->stream));
        return local_barrier(c);
    }
    if (c->replica) {
        OCN_CHECK_HIP(hipStreamSynchronize(c->stream));
        return OCN_SUCCESS;
    }
    OCN_CHECK_NCCL(ncclAllReduce(c->scalar, c->scalar, 1, ncclDouble, ncclSum, c->comm, c->stream));
    // host wait with a deadline (OCN_COMM_TIMEOUT_S, default 300): a rank whose peers never arrive gets OCN_ERR_TIMEOUT instead of hanging
    static const double deadline = getenv("OCN_COMM_TIMEOUT_S") ? atof(getenv("OCN_COMM_TIMEOUT_S")) : 300.0;
    return ocn::wait_stream(c->stream, deadline, "ocn_comm_barrier");
}

}  // extern "C"
