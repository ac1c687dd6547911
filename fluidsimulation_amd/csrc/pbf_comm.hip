// pbf_comm.hip -- neighbour point-to-point transport of the slab decomposition on RCCL
// (the reference is single-GPU and has no counterpart).
//
// A pbf_comm is one RCCL communicator rank.  pbf_comm_batch enqueues one grouped set of
// ncclSend / ncclRecv on the CALLER's HIP stream -- the stream the ctx kernels run on -- so the
// exchange is ordered with the pack / unpack kernels without host synchronisation and costs a
// few microseconds of host time per group.  Slab neighbours talk over their direct xGMI link;
// there is no ring and no data-path collective.
//
// RCCL is bound at run time (dlopen of librccl.so.1, the copy already in the process if there
// is one), so single-GPU users of libpbf_hip.so need no RCCL at all.

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <chrono>
#include <condition_variable>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "pbf.h"

namespace {

// the slice of the RCCL ABI used here (rccl.h: stable since NCCL 2.7)
struct NcclUniqueId {
    char internal[PBF_COMM_ID_BYTES];
};
typedef void* NcclComm;
typedef int NcclResult;  // 0 == ncclSuccess
constexpr int kNcclInt8 = 0;

struct Rccl {
    void* handle = nullptr;
    NcclResult (*GetUniqueId)(NcclUniqueId*) = nullptr;
    NcclResult (*CommInitRank)(NcclComm*, int, NcclUniqueId, int) = nullptr;
    NcclResult (*CommDestroy)(NcclComm) = nullptr;
    NcclResult (*Send)(const void*, size_t, int, int, NcclComm, hipStream_t) = nullptr;
    NcclResult (*Recv)(void*, size_t, int, int, NcclComm, hipStream_t) = nullptr;
    NcclResult (*GroupStart)() = nullptr;
    NcclResult (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(NcclResult) = nullptr;
    std::string error;
    bool ok = false;
};

Rccl g_rccl;
std::once_flag g_rccl_once;
thread_local std::string g_comm_err = "no error";

void load_rccl() {
    Rccl& r = g_rccl;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    // a copy that is already mapped (PyTorch-ROCm brings its own) serves everybody: one RCCL,
    // one HIP runtime per process
    for (const char* n : names) {
        r.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
        if (r.handle) break;
    }
    for (int i = 0; !r.handle && i < 3; ++i) r.handle = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
    if (!r.handle) {
        const char* e = dlerror();
        r.error = std::string("cannot load librccl.so.1: ") + (e ? e : "unknown error");
        return;
    }
#define BIND(field, sym)                                                \
    do {                                                                \
        *(void**)(&r.field) = dlsym(r.handle, sym);                     \
        if (!r.field) {                                                 \
            r.error = std::string("librccl lacks symbol ") + sym;       \
            return;                                                     \
        }                                                               \
    } while (0)
    BIND(GetUniqueId, "ncclGetUniqueId");
    BIND(CommInitRank, "ncclCommInitRank");
    BIND(CommDestroy, "ncclCommDestroy");
    BIND(Send, "ncclSend");
    BIND(Recv, "ncclRecv");
    BIND(GroupStart, "ncclGroupStart");
    BIND(GroupEnd, "ncclGroupEnd");
    BIND(GetErrorString, "ncclGetErrorString");
#undef BIND
    r.ok = true;
}

const Rccl* rccl() {
    std::call_once(g_rccl_once, load_rccl);
    return &g_rccl;
}

// ---- in-process loopback transport (tests): the ranks are threads of ONE process, each with
// its own ctx and stream, possibly all on one GPU.  Same contract as the RCCL path -- the k-th
// send from a to b pairs with b's k-th receive from a; a send buffer may be reused once the
// stream has passed the batch -- implemented with a mailbox per ordered rank pair, events and
// device-to-device copies.  A communicator id that starts with "PBFLOCAL" selects it.
struct LocalMsg {
    const void* buf = nullptr;
    uint64_t bytes = 0;
    hipEvent_t ready = nullptr;   // recorded on the sender's stream before the message is posted
    hipEvent_t done = nullptr;    // recorded on the receiver's stream after its copy
    bool consumed = false, failed = false;
};

struct LocalGroup {
    int world = 0;
    std::mutex mu;
    std::condition_variable cv;
    std::map<std::pair<int, int>, std::deque<std::shared_ptr<LocalMsg>>> box;  // (from, to)
    int joined = 0;
};

std::mutex g_local_mu;
std::map<std::string, std::shared_ptr<LocalGroup>> g_local_groups;

}  // namespace

struct pbf_comm {
    NcclComm comm = nullptr;
    std::shared_ptr<LocalGroup> local;   // set for the loopback transport
    int rank = 0, world = 1, device = 0;
    std::string err = "no error";
};

namespace {
int comm_fail(pbf_comm* c, int code, const std::string& msg) {
    if (c) c->err = msg;
    else g_comm_err = msg;
    return code;
}
bool is_local_id(const void* id) { return memcmp(id, "PBFLOCAL", 8) == 0; }

int local_batch(pbf_comm* c, hipStream_t s, const pbf_p2p* ops, uint32_t n_ops) {
    LocalGroup& G = *c->local;
    std::vector<std::shared_ptr<LocalMsg>> mine;
    // 1. post every send (never blocks)
    for (uint32_t i = 0; i < n_ops; ++i) {
        const pbf_p2p& op = ops[i];
        if (!op.bytes || !op.send) continue;
        auto m = std::make_shared<LocalMsg>();
        m->buf = op.buf;
        m->bytes = op.bytes;
        if (hipEventCreateWithFlags(&m->ready, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&m->done, hipEventDisableTiming) != hipSuccess ||
            hipEventRecord(m->ready, s) != hipSuccess)
            return comm_fail(c, PBF_ERR_HIP, "loopback: event setup failed");
        {
            std::lock_guard<std::mutex> lk(G.mu);
            G.box[{c->rank, op.peer}].push_back(m);
        }
        G.cv.notify_all();
        mine.push_back(m);
    }
    // 2. every receive: wait for the matching message, copy behind the sender's event
    int rc = PBF_OK;
    for (uint32_t i = 0; i < n_ops; ++i) {
        const pbf_p2p& op = ops[i];
        if (!op.bytes || op.send) continue;
        std::shared_ptr<LocalMsg> m;
        {
            std::unique_lock<std::mutex> lk(G.mu);
            auto& q = G.box[{op.peer, c->rank}];
            if (!G.cv.wait_for(lk, std::chrono::seconds(60), [&] { return !q.empty(); })) {
                rc = comm_fail(c, PBF_ERR_INVALID, "loopback: no matching send within 60 s");
                break;
            }
            m = q.front();
            q.pop_front();
        }
        bool ok = m->bytes == op.bytes;
        if (!ok) rc = comm_fail(c, PBF_ERR_INVALID, "loopback: message sizes disagree between the two ends");
        ok = ok && hipStreamWaitEvent(s, m->ready, 0) == hipSuccess &&
             hipMemcpyAsync(op.buf, m->buf, op.bytes, hipMemcpyDeviceToDevice, s) == hipSuccess &&
             hipEventRecord(m->done, s) == hipSuccess;
        if (!ok && rc == PBF_OK) rc = comm_fail(c, PBF_ERR_HIP, "loopback: copy failed");
        {
            std::lock_guard<std::mutex> lk(G.mu);
            m->consumed = true;
            m->failed = !ok;
        }
        G.cv.notify_all();
    }
    // 3. my sends: the stream must not run past the batch before the receivers have copied
    for (auto& m : mine) {
        bool failed;
        {
            std::unique_lock<std::mutex> lk(G.mu);
            if (!G.cv.wait_for(lk, std::chrono::seconds(60), [&] { return m->consumed; })) {
                if (rc == PBF_OK) rc = comm_fail(c, PBF_ERR_INVALID, "loopback: a send was not received within 60 s");
                continue;   // the receiver may still use the events: they leak; the run is lost anyway
            }
            failed = m->failed;
        }
        if (!failed && hipStreamWaitEvent(s, m->done, 0) != hipSuccess && rc == PBF_OK)
            rc = comm_fail(c, PBF_ERR_HIP, "loopback: hipStreamWaitEvent failed");
        // destroyed lazily by the runtime once the work recorded behind them has completed
        (void)hipEventDestroy(m->ready);
        (void)hipEventDestroy(m->done);
    }
    return rc;
}

std::string nccl_msg(const char* what, NcclResult r) {
    const Rccl* R = rccl();
    return std::string(what) + ": " + (R->GetErrorString ? R->GetErrorString(r) : "RCCL error") +
           " (" + std::to_string(r) + ")";
}
}  // namespace

extern "C" {

const char* pbf_comm_last_error(const pbf_comm* c) { return c ? c->err.c_str() : g_comm_err.c_str(); }

int pbf_comm_unique_id(void* id_out) {
    if (!id_out) return comm_fail(nullptr, PBF_ERR_INVALID, "null id");
    const Rccl* R = rccl();
    if (!R->ok) return comm_fail(nullptr, PBF_ERR_NO_DEVICE, R->error);
    NcclUniqueId id;
    NcclResult r = R->GetUniqueId(&id);
    if (r) return comm_fail(nullptr, PBF_ERR_HIP, nccl_msg("ncclGetUniqueId", r));
    memcpy(id_out, &id, sizeof(id));
    return PBF_OK;
}

int pbf_comm_local_id(void* id_out) {
    if (!id_out) return comm_fail(nullptr, PBF_ERR_INVALID, "null id");
    static std::mutex mu;
    static uint64_t counter = 0;
    std::lock_guard<std::mutex> lk(mu);
    memset(id_out, 0, PBF_COMM_ID_BYTES);
    memcpy(id_out, "PBFLOCAL", 8);
    const uint64_t v = ++counter;
    memcpy((char*)id_out + 8, &v, sizeof(v));
    return PBF_OK;
}

int pbf_comm_create(const void* id_in, int32_t rank, int32_t world, int32_t device, pbf_comm** out) {
    if (out) *out = nullptr;
    if (!id_in || !out || world < 1 || rank < 0 || rank >= world)
        return comm_fail(nullptr, PBF_ERR_INVALID, "bad argument");
    const bool local = is_local_id(id_in);
    const Rccl* R = local ? nullptr : rccl();
    if (!local && !R->ok) return comm_fail(nullptr, PBF_ERR_NO_DEVICE, R->error);
    if (device >= 0) {
        hipError_t e = hipSetDevice(device);
        if (e != hipSuccess) return comm_fail(nullptr, PBF_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(e));
    } else if (hipGetDevice(&device) != hipSuccess) {
        device = 0;
    }
    pbf_comm* c = new (std::nothrow) pbf_comm();
    if (!c) return comm_fail(nullptr, PBF_ERR_INVALID, "out of host memory");
    c->rank = rank;
    c->world = world;
    c->device = device;
    if (local) {
        const std::string key((const char*)id_in, PBF_COMM_ID_BYTES);
        std::lock_guard<std::mutex> lk(g_local_mu);
        auto& g = g_local_groups[key];
        if (!g) {
            g = std::make_shared<LocalGroup>();
            g->world = world;
        }
        if (g->world != world) {
            delete c;
            return comm_fail(nullptr, PBF_ERR_INVALID, "loopback: ranks disagree on the world size");
        }
        c->local = g;
        if (++g->joined == world) g_local_groups.erase(key);   // everybody holds a reference now
        *out = c;
        return PBF_OK;
    }
    NcclUniqueId id;
    memcpy(&id, id_in, sizeof(id));
    NcclResult r = R->CommInitRank(&c->comm, world, id, rank);
    if (r) {
        delete c;
        return comm_fail(nullptr, PBF_ERR_HIP, nccl_msg("ncclCommInitRank", r));
    }
    *out = c;
    return PBF_OK;
}

int pbf_comm_destroy(pbf_comm* c) {
    if (!c) return PBF_ERR_INVALID;
    if (c->comm) {
        const Rccl* R = rccl();
        if (R->ok) (void)R->CommDestroy(c->comm);
    }
    delete c;
    return PBF_OK;
}

int32_t pbf_comm_rank(const pbf_comm* c) { return c ? c->rank : -1; }
int32_t pbf_comm_world(const pbf_comm* c) { return c ? c->world : 0; }

int pbf_comm_batch(pbf_comm* c, void* hip_stream, const pbf_p2p* ops, uint32_t n_ops) {
    if (!c) return PBF_ERR_INVALID;
    if (n_ops && !ops) return comm_fail(c, PBF_ERR_INVALID, "null ops");
    const Rccl* R = c->local ? nullptr : rccl();
    if (R && !R->ok) return comm_fail(c, PBF_ERR_NO_DEVICE, R->error);
    uint32_t live = 0;
    for (uint32_t i = 0; i < n_ops; ++i) {
        if (!ops[i].bytes) continue;   // both ends skip an empty message (sizes are agreed beforehand)
        if (!ops[i].buf) return comm_fail(c, PBF_ERR_INVALID, "null buffer in a non-empty op");
        if (ops[i].peer < 0 || ops[i].peer >= c->world) return comm_fail(c, PBF_ERR_INVALID, "peer out of range");
        ++live;
    }
    if (!live) return PBF_OK;
    hipError_t he = hipSetDevice(c->device);
    if (he != hipSuccess) return comm_fail(c, PBF_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(he));
    hipStream_t s = (hipStream_t)hip_stream;
    if (c->local) return local_batch(c, s, ops, n_ops);
    NcclResult r = R->GroupStart();
    if (r) return comm_fail(c, PBF_ERR_HIP, nccl_msg("ncclGroupStart", r));
    NcclResult first_bad = 0;
    const char* where = "";
    for (uint32_t i = 0; i < n_ops; ++i) {
        const pbf_p2p& op = ops[i];
        if (!op.bytes) continue;
        r = op.send ? R->Send(op.buf, (size_t)op.bytes, kNcclInt8, op.peer, c->comm, s)
                    : R->Recv(op.buf, (size_t)op.bytes, kNcclInt8, op.peer, c->comm, s);
        if (r && !first_bad) {
            first_bad = r;
            where = op.send ? "ncclSend" : "ncclRecv";
        }
    }
    r = R->GroupEnd();   // always closes the group, also after a failed op
    if (first_bad) return comm_fail(c, PBF_ERR_HIP, nccl_msg(where, first_bad));
    if (r) return comm_fail(c, PBF_ERR_HIP, nccl_msg("ncclGroupEnd", r));
    return PBF_OK;
}

}  // extern "C"
