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
#include <fcntl.h>
#include <rccl/rccl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
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

// ---------------------------------------------------------------- the host transport (same node, no RCCL)
// One process per GPU on ONE node is the shape of SURVEY.md 8e, so when RCCL cannot serve a communicator -- two ranks
// rehearsing on one device ("invalid usage"), no librccl, an init that fails on some rank -- the ranks still share the
// node's memory: a POSIX shared-memory segment named after the communicator's unique id carries the all-gather.  It is
// also where the ranks AGREE on the transport (every rank publishes whether RCCL came up; one failure sends all of them
// here), so the decision is the library's, not a watchdog's in the caller.
// Protocol (round s, parity p = s & 1): wait until every rank has read round s - 2 (the slot about to be overwritten),
// write the own slot, publish written = s + 1 (release); wait for every rank's written >= s + 1 (acquire), copy all
// slots, publish read = s + 1.  A fresh segment is zero-filled: that is the valid initial state, nobody initialises it.
#define SHM_MAX_RANKS 64
#define SHM_SLOT_BYTES 8192  // a group of up to 8 partials (8 x 592) per rank and round
struct ShmSeg {
    std::atomic<uint32_t> attached, rccl_votes, rccl_failed, pad;
    std::atomic<uint64_t> device_key[SHM_MAX_RANKS];  // PCI location + 1 of every rank's device (0: not yet published)
    std::atomic<uint64_t> written[SHM_MAX_RANKS], read[SHM_MAX_RANKS];
    uint8_t slot[2][SHM_MAX_RANKS][SHM_SLOT_BYTES];
};
struct Shm {
    ShmSeg *seg = nullptr;
    char name[64] = {0};
    int rank = 0, world = 1;
    uint64_t round = 0;
    ~Shm() {
        if (seg) munmap(seg, sizeof(ShmSeg));
    }
};
static bool shm_wait(const std::function<bool()> &ready, double seconds) {
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t spin = 0; !ready(); spin++) {
        if (spin < 2000) continue;  // ~ microseconds: the peers are usually there already
        sched_yield();
        if ((spin & 1023) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > seconds) return false;
    }
    return true;
}
static std::unique_ptr<Shm> shm_attach(const uint8_t *id, int rank, int world) {
    if (world > SHM_MAX_RANKS) return nullptr;
    std::unique_ptr<Shm> s(new Shm());
    uint64_t h = 0xcbf29ce484222325ull;  // FNV-1a over the 128 id bytes
    for (int i = 0; i < MI_COMM_ID_BYTES; i++) h = (h ^ id[i]) * 0x100000001b3ull;
    snprintf(s->name, sizeof s->name, "/mi355x_snark_%016llx", (unsigned long long)h);
    const int fd = shm_open(s->name, O_CREAT | O_RDWR, 0600);
    if (fd < 0) return nullptr;
    if (ftruncate(fd, sizeof(ShmSeg)) != 0) {
        close(fd);
        return nullptr;
    }
    void *p = mmap(nullptr, sizeof(ShmSeg), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return nullptr;
    s->seg = (ShmSeg *)p;
    s->rank = rank;
    s->world = world;
    s->seg->attached.fetch_add(1);
    ShmSeg *seg = s->seg;
    if (!shm_wait([=] { return seg->attached.load() >= (uint32_t)world; }, 120.0)) return nullptr;
    if (rank == 0) shm_unlink(s->name);  // everybody holds a mapping: the name has done its work
    return s;
}
static int32_t shm_all_gather(Shm &s, const void *mine, size_t bytes, void *all) {
    if (bytes > SHM_SLOT_BYTES) return MI_ERR_BAD_ARGS;
    ShmSeg *seg = s.seg;
    const uint64_t r = s.round++;
    const int par = (int)(r & 1), world = s.world;
    if (r >= 2 && !shm_wait([=] {
            for (int k = 0; k < world; k++)
                if (seg->read[k].load(std::memory_order_acquire) + 1 < r) return false;
            return true;
        }, 120.0))
        return MI_ERR_KERNEL;
    memcpy(seg->slot[par][s.rank], mine, bytes);
    seg->written[s.rank].store(r + 1, std::memory_order_release);
    if (!shm_wait([=] {
            for (int k = 0; k < world; k++)
                if (seg->written[k].load(std::memory_order_acquire) < r + 1) return false;
            return true;
        }, 120.0))
        return MI_ERR_KERNEL;
    for (int k = 0; k < world; k++) memcpy((uint8_t *)all + (size_t)k * bytes, seg->slot[par][k], bytes);
    seg->read[s.rank].store(r + 1, std::memory_order_release);
    return MI_OK;
}

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
    std::unique_ptr<Shm> shm;  // rank-per-process communicators: the agreement channel, and the transport when RCCL is not used
    bool use_rccl = true;
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
    StatExchange timer;
    if (!c.use_rccl) return c.shm ? shm_all_gather(*c.shm, mine[0], bytes, all[0]) : MI_ERR_NO_DEVICE;
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
    static_assert(sizeof(ncclUniqueId) == MI_COMM_ID_BYTES, "ncclUniqueId size");
    if (rccl().ok) {
        ncclUniqueId id;
        if (rccl().GetUniqueId(&id) == ncclSuccess) {
            memcpy(id_out, &id, sizeof id);
            return MI_OK;
        }
    }
    // no RCCL: the id only has to name the host transport's segment
    memset(id_out, 0, MI_COMM_ID_BYTES);
    const int fd = open("/dev/urandom", O_RDONLY);
    if (fd < 0 || read(fd, id_out, 32) != 32) {
        if (fd >= 0) close(fd);
        return MI_ERR_KERNEL;
    }
    close(fd);
    return MI_OK;
}

// Collective over the ranks.  RCCL when it comes up on EVERY rank (the ranks vote through the shared segment; ranks
// that share a device do not try: ncclCommInitRank refuses that), the host transport otherwise.  MI_SNARK_EXCHANGE=host
// in the environment of every rank skips RCCL.
int32_t mi_comm_init_rank(const uint8_t *id, int32_t rank, int32_t world, uint64_t *comm_out) {
    if (!id || !comm_out || world < 1 || rank < 0 || rank >= world) return MI_ERR_BAD_ARGS;
    const bool have_device = ensure_init() == MI_OK;  // without a device the communicator still serves host records (tests)
    std::shared_ptr<Comm> c(new Comm());
    c->world = world;
    c->members.resize(1);
    CommMember &m = c->members[0];
    m.cx = have_device ? &primary() : nullptr;  // one process per GPU: this process's device
    c->shm = shm_attach(id, rank, world);
    if (!c->shm) return MI_ERR_NO_DEVICE;
    ShmSeg *seg = c->shm->seg;
    // ---- may RCCL be tried at all?  every rank publishes its device; two ranks on one device rule it out for all
    char bus[64] = {0};
    uint64_t key = 1000 + (uint64_t)rank;  // no device: a key of its own
    if (have_device && hipDeviceGetPCIBusId(bus, sizeof bus, m.cx->device) == hipSuccess) {
        key = 1;
        for (const char *q = bus; *q; q++) key = key * 131 + (uint8_t)*q;
    }
    seg->device_key[rank].store(key | 1ull << 63);
    if (!shm_wait([=] {
            for (int k = 0; k < world; k++)
                if (!seg->device_key[k].load()) return false;
            return true;
        }, 120.0))
        return MI_ERR_KERNEL;
    bool shared_device = false;
    for (int a = 0; a < world; a++)
        for (int b = 0; b < a; b++)
            if (seg->device_key[a].load() == seg->device_key[b].load()) shared_device = true;
    const char *e = getenv("MI_SNARK_EXCHANGE");
    bool try_rccl = have_device && rccl().ok && !shared_device && !(e && !strcmp(e, "host"));
    bool mine_ok = false;
    if (try_rccl) {
        ncclUniqueId uid;
        memcpy(&uid, id, sizeof uid);
        try {
            MI_HIP(hipSetDevice(m.cx->device));
            mine_ok = nccl_status(rccl().CommInitRank(&m.comm, world, uid, rank), "ncclCommInitRank") == MI_OK;
            if (mine_ok) MI_HIP(hipStreamCreateWithFlags(&m.st, hipStreamNonBlocking));
        } catch (const HipError &h) {
            (void)status_of(h);
            mine_ok = false;
        }
    }
    // ---- the vote: one rank without RCCL sends every rank to the host transport
    if (!mine_ok) seg->rccl_failed.fetch_add(1);
    seg->rccl_votes.fetch_add(1);
    if (!shm_wait([=] { return seg->rccl_votes.load() >= (uint32_t)world; }, 120.0)) return MI_ERR_KERNEL;
    c->use_rccl = seg->rccl_failed.load() == 0;
    if (!c->use_rccl && m.comm) {
        (void)rccl().CommDestroy(m.comm);
        m.comm = nullptr;
    }
    std::lock_guard<std::mutex> lk(g_comm_mu);
    *comm_out = g_comm_next++;
    g_comms[*comm_out] = c;
    return MI_OK;
}
int32_t mi_comm_transport(uint64_t comm) {
    std::shared_ptr<Comm> c = comm_find(comm);
    if (!c) return MI_ERR_BAD_ARGS;
    return c->use_rccl ? MI_COMM_RCCL : MI_COMM_HOST;
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
