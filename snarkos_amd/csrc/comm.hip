// The one exchange step of the sharded verifier: an all-gather of 304-byte partial accumulators over
// RCCL / xGMI, called from C++ -- no Python, no torch on the data path.
//
// RCCL has no elliptic-curve reduction operator, hence gather + local add (vs_sum_partials) instead of a reduce.
// Two shapes:
//   * one process per GPU (torchrun, or one validator process per device): mi_comm_unique_id on rank 0, the 128 bytes
//     travel over the caller's own transport, every rank calls mi_comm_init_rank; the streaming driver's exchange
//     callback is the library's own mi_vs_exchange_rccl (user = the communicator handle);
//   * one process driving all GPUs (what snarkOS is: reference cli/src/commands/start.rs:616-642 builds one rayon and
//     one tokio pool in ONE process): mi_comm_init_local = ncclCommInitAll over the library's device list; the block
//     entry point (mi_vs_verify_block) gathers the shards' partials with one grouped ncclAllGather.
// librccl is loaded with dlopen on first use (the copy the process already has, e.g. PyTorch's, if there is one): a
// host without RCCL still loads this library, and every communicator call then returns MI_ERR_NO_DEVICE so the caller
// keeps its own exchange (in one process the partials are host values anyway: mi_vs_verify_block falls back to adding
// them directly).  Pure host code: no kernels here.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <atomic>
#include <map>
#include <memory>
#include <mutex>
#include <vector>
#include "api.hpp"

namespace mi {

struct RcclApi {
    void *lib = nullptr;
    bool ok = false;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
static RcclApi &rccl() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names)  // a copy the process has already mapped wins (two RCCLs in one process do not mix)
            if ((api.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
        if (!api.lib)
            for (const char *n : names)
                if ((api.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
        if (!api.lib) return;
#define MI_SYM(field, name) *(void **)(&api.field) = dlsym(api.lib, name)
        MI_SYM(GetUniqueId, "ncclGetUniqueId");
        MI_SYM(CommInitRank, "ncclCommInitRank");
        MI_SYM(CommInitAll, "ncclCommInitAll");
        MI_SYM(CommDestroy, "ncclCommDestroy");
        MI_SYM(AllGather, "ncclAllGather");
        MI_SYM(GroupStart, "ncclGroupStart");
        MI_SYM(GroupEnd, "ncclGroupEnd");
        MI_SYM(GetErrorString, "ncclGetErrorString");
#undef MI_SYM
        api.ok = api.GetUniqueId && api.CommInitRank && api.CommInitAll && api.CommDestroy && api.AllGather &&
                 api.GroupStart && api.GroupEnd;
    });
    return api;
}
static int32_t nccl_status(ncclResult_t r, const char *what) {
    if (r == ncclSuccess) return MI_OK;
    fprintf(stderr, "[mi355x_snark] RCCL error %d (%s) in %s\n", (int)r,
            rccl().GetErrorString ? rccl().GetErrorString(r) : "?", what);
    return MI_ERR_KERNEL;
}
#define MI_NCCL(expr)                                 \
    do {                                              \
        int32_t _s = nccl_status((expr), #expr);      \
        if (_s != MI_OK) return _s;                   \
    } while (0)

// One member per device this process contributes: its communicator, a stream of its own and staging buffers.
struct CommMember {
    Context *cx = nullptr;
    ncclComm_t comm = nullptr;
    hipStream_t st = nullptr;
    void *d_send = nullptr, *d_recv = nullptr;  // device staging
    uint8_t *h_recv = nullptr;                  // pinned
    size_t cap_send = 0, cap_recv = 0;
};
struct Comm {
    std::mutex mu;  // one collective at a time per communicator (RCCL's own requirement for a given comm)
    int world = 1;  // ranks over all processes
    std::vector<CommMember> members;
};
static std::mutex g_comm_mu;
static std::map<uint64_t, std::shared_ptr<Comm>> g_comms;
static uint64_t g_comm_next = 1;

static std::shared_ptr<Comm> comm_find(uint64_t id) {
    std::lock_guard<std::mutex> lk(g_comm_mu);
    auto it = g_comms.find(id);
    return it == g_comms.end() ? nullptr : it->second;
}
static void member_free(CommMember &m) {
    if (!m.cx) return;
    (void)hipSetDevice(m.cx->device);
    if (m.st) (void)hipStreamSynchronize(m.st);
    if (m.comm) (void)rccl().CommDestroy(m.comm);
    if (m.d_send) (void)hipFree(m.d_send);
    if (m.d_recv) (void)hipFree(m.d_recv);
    if (m.h_recv) (void)hipHostFree(m.h_recv);
    if (m.st) (void)hipStreamDestroy(m.st);
    m = CommMember();
}
static void member_reserve(CommMember &m, size_t send, size_t recv) {
    if (send > m.cap_send) {
        if (m.d_send) {
            MI_HIP(hipStreamSynchronize(m.st));
            MI_HIP(hipFree(m.d_send));
        }
        m.d_send = nullptr;
        MI_HIP(hipMalloc(&m.d_send, send + 4096));
        m.cap_send = send + 4096;
    }
    if (recv > m.cap_recv) {
        if (m.d_recv) {
            MI_HIP(hipStreamSynchronize(m.st));
            MI_HIP(hipFree(m.d_recv));
            MI_HIP(hipHostFree(m.h_recv));
        }
        m.d_recv = nullptr;
        m.h_recv = nullptr;
        MI_HIP(hipMalloc(&m.d_recv, recv + 4096));
        MI_HIP(hipHostMalloc((void **)&m.h_recv, recv + 4096, hipHostMallocDefault));
        m.cap_recv = recv + 4096;
    }
}
void comm_release_all() {
    std::lock_guard<std::mutex> lk(g_comm_mu);
    for (auto &kv : g_comms)
        for (auto &m : kv.second->members) member_free(m);
    g_comms.clear();
}

// all members contribute `bytes` each (mine[i] = host pointer of member i's block) and receive world * bytes
// (all[i] = host destination of member i; pass the same pointer for members whose copy is not needed: nullptr skips
// the copy-out).  One grouped ncclAllGather over the members' communicators.
int32_t comm_all_gather(Comm &c, const void *const *mine, size_t bytes, void *const *all) {
    RcclApi &api = rccl();
    std::lock_guard<std::mutex> lk(c.mu);
    try {
        for (size_t i = 0; i < c.members.size(); i++) {
            CommMember &m = c.members[i];
            MI_HIP(hipSetDevice(m.cx->device));
            member_reserve(m, bytes, bytes * c.world);
            MI_HIP(hipMemcpyAsync(m.d_send, mine[i], bytes, hipMemcpyHostToDevice, m.st));
        }
        MI_NCCL(api.GroupStart());
        for (auto &m : c.members) {
            MI_HIP(hipSetDevice(m.cx->device));
            const ncclResult_t r = api.AllGather(m.d_send, m.d_recv, bytes, ncclUint8, m.comm, m.st);
            if (r != ncclSuccess) {
                (void)api.GroupEnd();
                return nccl_status(r, "ncclAllGather");
            }
        }
        MI_NCCL(api.GroupEnd());
        for (size_t i = 0; i < c.members.size(); i++) {
            CommMember &m = c.members[i];
            if (!all[i]) continue;
            MI_HIP(hipSetDevice(m.cx->device));
            MI_HIP(hipMemcpyAsync(m.h_recv, m.d_recv, bytes * c.world, hipMemcpyDeviceToHost, m.st));
        }
        for (size_t i = 0; i < c.members.size(); i++) {
            CommMember &m = c.members[i];
            MI_HIP(hipSetDevice(m.cx->device));
            MI_HIP(hipStreamSynchronize(m.st));
            if (all[i]) memcpy(all[i], m.h_recv, bytes * c.world);
        }
    } catch (const HipError &h) {
        return status_of(h);
    }
    return MI_OK;
}
// the in-process communicator over the library's device list, made on first use (nullptr: RCCL missing or init failed)
std::shared_ptr<Comm> comm_local() {
    static std::mutex mu;
    static std::shared_ptr<Comm> local;
    static bool tried = false;
    std::lock_guard<std::mutex> lk(mu);
    if (tried) return local;
    tried = true;
    const char *e = getenv("MI_SNARK_EXCHANGE");
    if (e && !strcmp(e, "host")) return nullptr;
    RcclApi &api = rccl();
    const int n = n_contexts();
    if (!api.ok || n < 1) return nullptr;
    std::shared_ptr<Comm> c(new Comm());
    c->world = n;
    c->members.resize(n);
    std::vector<int> devs(n);
    std::vector<ncclComm_t> comms(n, nullptr);
    for (int i = 0; i < n; i++) {
        c->members[i].cx = context_at(i);
        devs[i] = c->members[i].cx->device;
    }
    if (nccl_status(api.CommInitAll(comms.data(), n, devs.data()), "ncclCommInitAll") != MI_OK) return nullptr;
    try {
        for (int i = 0; i < n; i++) {
            c->members[i].comm = comms[i];
            MI_HIP(hipSetDevice(devs[i]));
            MI_HIP(hipStreamCreateWithFlags(&c->members[i].st, hipStreamNonBlocking));
        }
    } catch (const HipError &h) {
        (void)status_of(h);
        for (auto &m : c->members) member_free(m);
        return nullptr;
    }
    {
        std::lock_guard<std::mutex> lk2(g_comm_mu);
        g_comms[g_comm_next++] = c;  // owned by the table: released at mi_shutdown
    }
    local = c;
    return local;
}

}  // namespace mi

using namespace mi;

extern "C" {

int32_t mi_comm_unique_id(uint8_t *id_out) {
    if (!id_out) return MI_ERR_BAD_ARGS;
    if (!rccl().ok) return MI_ERR_NO_DEVICE;
    ncclUniqueId id;
    MI_NCCL(rccl().GetUniqueId(&id));
    static_assert(sizeof id == MI_COMM_ID_BYTES, "ncclUniqueId size");
    memcpy(id_out, &id, sizeof id);
    return MI_OK;
}

int32_t mi_comm_init_rank(const uint8_t *id, int32_t rank, int32_t world, uint64_t *comm_out) {
    if (!id || !comm_out || world < 1 || rank < 0 || rank >= world) return MI_ERR_BAD_ARGS;
    if (!rccl().ok) return MI_ERR_NO_DEVICE;
    const int32_t st = ensure_init();
    if (st != MI_OK) return st;
    std::shared_ptr<Comm> c(new Comm());
    c->world = world;
    c->members.resize(1);
    CommMember &m = c->members[0];
    m.cx = &primary();  // one process per GPU: this process's device
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof uid);
    try {
        MI_HIP(hipSetDevice(m.cx->device));
        const int32_t s = nccl_status(rccl().CommInitRank(&m.comm, world, uid, rank), "ncclCommInitRank");
        if (s != MI_OK) return s;
        MI_HIP(hipStreamCreateWithFlags(&m.st, hipStreamNonBlocking));
    } catch (const HipError &h) {
        member_free(m);
        return status_of(h);
    }
    std::lock_guard<std::mutex> lk(g_comm_mu);
    *comm_out = g_comm_next++;
    g_comms[*comm_out] = c;
    return MI_OK;
}

int32_t mi_comm_init_local(uint64_t *comm_out) {
    if (!comm_out) return MI_ERR_BAD_ARGS;
    const int32_t st = ensure_init();
    if (st != MI_OK) return st;
    std::shared_ptr<Comm> c = comm_local();
    if (!c) return MI_ERR_NO_DEVICE;
    std::lock_guard<std::mutex> lk(g_comm_mu);
    for (auto &kv : g_comms)
        if (kv.second == c) {
            *comm_out = kv.first;
            return MI_OK;
        }
    return MI_ERR_KERNEL;
}

int32_t mi_comm_world(uint64_t comm) {
    std::shared_ptr<Comm> c = comm_find(comm);
    return c ? c->world : MI_ERR_BAD_ARGS;
}

int32_t mi_comm_destroy(uint64_t comm) {
    std::shared_ptr<Comm> c;
    {
        std::lock_guard<std::mutex> lk(g_comm_mu);
        auto it = g_comms.find(comm);
        if (it == g_comms.end()) return MI_ERR_BAD_ARGS;
        c = it->second;
        g_comms.erase(it);
    }
    std::lock_guard<std::mutex> lk(c->mu);
    for (auto &m : c->members) member_free(m);
    return MI_OK;
}

// host buffers in, host buffers out; a rank-per-process communicator has one member
int32_t mi_comm_all_gather(uint64_t comm, const void *mine, size_t bytes, void *all) {
    if (!mine || !all || !bytes) return MI_ERR_BAD_ARGS;
    std::shared_ptr<Comm> c = comm_find(comm);
    if (!c || c->members.size() != 1) return MI_ERR_BAD_ARGS;
    const void *src[1] = {mine};
    void *dst[1] = {all};
    return comm_all_gather(*c, src, bytes, dst);
}

// mi_vs_exchange_fn for mi_vs_stream_open: user = (void *)(uintptr_t)communicator handle of mi_comm_init_rank
int32_t mi_vs_exchange_rccl(void *user, const mi_vs_partial *mine, size_t count, mi_vs_partial *all, size_t max_partials) {
    std::shared_ptr<Comm> c = comm_find((uint64_t)(uintptr_t)user);
    if (!c || c->members.size() != 1 || !mine || !all || !count) return MI_ERR_BAD_ARGS;
    if ((size_t)c->world * count > max_partials) return MI_ERR_BAD_ARGS;
    const void *src[1] = {mine};
    void *dst[1] = {all};
    const int32_t st = comm_all_gather(*c, src, count * sizeof(mi_vs_partial), dst);
    return st == MI_OK ? c->world : st;
}

}  // extern "C"
