// Device context, workspace cache and error plumbing shared by every entry point.
#pragma once
#include <hip/hip_runtime.h>
#include <sched.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <condition_variable>
#include <deque>
#include <functional>
#include <future>
#include <atomic>
#include <chrono>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <string>
#include <vector>
#include "../../include/mi355x_snark.h"

namespace mi {

struct HipError {
    hipError_t e;
    const char *what;
    int line;
};

#define MI_HIP(expr)                                                     \
    do {                                                                 \
        hipError_t _e = (expr);                                          \
        if (_e != hipSuccess) throw ::mi::HipError{_e, #expr, __LINE__}; \
    } while (0)

inline int32_t status_of(const HipError &h) {
    fprintf(stderr, "[mi355x_snark] HIP error %d (%s) at %s:%d\n", (int)h.e, hipGetErrorString(h.e), h.what, h.line);
    if (h.e == hipErrorOutOfMemory) return MI_ERR_OOM;
    if (h.e == hipErrorNoDevice || h.e == hipErrorInvalidDevice) return MI_ERR_NO_DEVICE;
    return MI_ERR_KERNEL;
}

// A few persistent host workers for the short host-side folds (spawning a std::thread per fold costs tens of
// microseconds, several times per batch).  submit() returns a future; tasks must not throw.
class WorkerPool {
  public:
    explicit WorkerPool(int n) {
        for (int i = 0; i < n; i++)
            threads_.emplace_back([this] {
                for (;;) {
                    std::function<void()> job;
                    {
                        std::unique_lock<std::mutex> lk(mu_);
                        cv_.wait(lk, [this] { return stop_ || !q_.empty(); });
                        if (stop_ && q_.empty()) return;
                        job = std::move(q_.front());
                        q_.pop_front();
                    }
                    job();
                }
            });
    }
    ~WorkerPool() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto &t : threads_) t.join();
    }
    std::future<void> submit(std::function<void()> f) {
        auto task = std::make_shared<std::packaged_task<void()>>(std::move(f));
        std::future<void> fut = task->get_future();
        {
            std::lock_guard<std::mutex> lk(mu_);
            q_.emplace_back([task] { (*task)(); });
        }
        cv_.notify_one();
        return fut;
    }

  private:
    std::vector<std::thread> threads_;
    std::deque<std::function<void()>> q_;
    std::mutex mu_;
    std::condition_variable cv_;
    bool stop_ = false;
};
// Workers: what the process may run on (its affinity mask) less the driver's own threads (submitter, folders, finisher,
// publisher), between 4 and 16; MI_SNARK_POOL overrides.  A burst of a dozen lane passes ends with a dozen folds of 8
// tasks each at about the same time: with 8 workers that queue was the tail of a short stream.
inline int worker_pool_size() {
    if (const char *e = getenv("MI_SNARK_POOL")) {
        const int x = atoi(e);
        if (x >= 1 && x <= 64) return x;
    }
    int cores = (int)std::thread::hardware_concurrency();
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof set, &set) == 0) {
        const int c = CPU_COUNT(&set);
        if (c > 0 && (cores <= 0 || c < cores)) cores = c;
    }
    {   // a container's CPU quota counts, not the host's core count (cgroup v2: "max" or "<quota> <period>")
        FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r");
        if (f) {
            long long q = 0, per = 0;
            if (fscanf(f, "%lld %lld", &q, &per) == 2 && q > 0 && per > 0) {
                const int c = (int)(q / per);
                if (c > 0 && c < cores) cores = c;
            }
            fclose(f);
        }
    }
    int n = cores - 4;
    if (n < 4) n = 4;
    if (n > 16) n = 16;
    return n;
}
inline WorkerPool &worker_pool() {
    static WorkerPool p(worker_pool_size());
    return p;
}

// Host-side wall-clock accounting of code sections (MI_SNARK_HOSTPROF=1; report on stderr at mi_shutdown / exit).
struct HostProf {
    struct Acc { double s = 0, mn = 1e9, mx = 0; uint64_t n = 0; };
    std::map<std::string, Acc> acc;
    std::mutex mu;
    bool on = false;
    HostProf() {
        const char *e = getenv("MI_SNARK_HOSTPROF");
        on = e && e[0] == '1';
    }
    ~HostProf() { report(); }
    void report() {
        if (!on) return;
        for (auto &kv : acc)
            fprintf(stderr, "[hostprof] %-28s %8.1f us/call x %llu  (min %.1f, max %.1f)\n", kv.first.c_str(),
                    1e6 * kv.second.s / (double)kv.second.n, (unsigned long long)kv.second.n, 1e6 * kv.second.mn, 1e6 * kv.second.mx);
        acc.clear();
    }
};
inline HostProf &host_prof() {
    static HostProf h;
    return h;
}
struct HostSpan {
    const char *name;
    std::chrono::steady_clock::time_point t0;
    explicit HostSpan(const char *nm) : name(nm) {
        if (host_prof().on) t0 = std::chrono::steady_clock::now();
    }
    ~HostSpan() {
        if (!host_prof().on) return;
        std::lock_guard<std::mutex> lk(host_prof().mu);
        auto &a = host_prof().acc[name];
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        a.s += dt;
        if (dt < a.mn) a.mn = dt;
        if (dt > a.mx) a.mx = dt;
        a.n++;
    }
};

// Process-wide counters behind mi_stats_read (the build-side analogues of the reference's
// prometheus metrics).  Relaxed atomics: any thread adds, the reader takes a snapshot.
struct Stats {
    std::atomic<uint64_t> batches_verified{0}, proofs_verified{0}, proofs_rejected{0}, verify_batch_nanos{0}, msm_calls{0},
        msm_points{0}, ntt_calls{0}, ntt_elements{0}, pairing_products{0}, pairing_pairs{0}, points_decompressed{0},
        fallback_batches{0}, exchange_calls{0}, exchange_nanos{0};
};
inline Stats &stats() {
    static Stats s;
    return s;
}
struct StatTimer {  // adds its lifetime to verify_batch_nanos
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    ~StatTimer() {
        stats().verify_batch_nanos += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
    }
};

struct StatExchange {  // one all-gather of partials (any transport): call count and host wall time
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    ~StatExchange() {
        stats().exchange_calls++;
        stats().exchange_nanos += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
    }
};

// A grow-only device buffer (no hipMalloc on the steady-state path).
// Lifetime rule (DESIGN.md "Workspace lifetime"): a regrow never frees memory that enqueued work may still touch --
// it drains the device first (regrows are rare: first call, or a larger problem than ever before).  Buffers that an
// asynchronous entry point leaves in flight on a CALLER's stream must not be name-keyed at all: they come from a
// ScratchPool (below), whose entries are guarded by an event.
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    uint64_t gen = 0;  // bumped by every (re)allocation: content-carrying users compare it
    void *get(size_t bytes) {
        if (bytes > cap) {
            HostSpan hs(p ? "devbuf.regrow" : "devbuf.alloc");
            gen++;
            if (p) {
                MI_HIP(hipDeviceSynchronize());  // nothing enqueued earlier, on any stream, may still use the old block
                MI_HIP(hipFree(p));
            }
            p = nullptr;
            cap = 0;
            size_t want = bytes + bytes / 8 + 256;
            MI_HIP(hipMalloc(&p, want));
            cap = want;
        }
        return p;
    }
    void release() {
        if (p) {
            (void)hipDeviceSynchronize();
            (void)hipFree(p);
        }
        p = nullptr;
        cap = 0;
    }
};

// Scratch buffers for entry points that return while their kernels are still running on the caller's stream
// (mi_ntt_fr_device): every acquisition gets a buffer no other in-flight call is using.  An entry is reusable once
// the event recorded after its last user's final kernel has completed; a regrow of such an entry is then safe.
struct ScratchPool {
    struct Entry {
        DevBuf buf;
        hipEvent_t done = nullptr;
        hipStream_t last = nullptr;  // the stream of the last user
        bool used = false;           // `done` has been recorded at least once
    };
    std::vector<std::unique_ptr<Entry>> entries;
    // returns an entry holding >= bytes that the new user may write as soon as its kernels run on `st`: one whose last
    // user ran on the same stream (stream order does the rest), or one whose last user has finished (the context lock
    // is held by the caller)
    Entry *acquire(size_t bytes, hipStream_t st) {
        Entry *pick = nullptr;
        for (auto &e : entries)
            if (e->used && e->last == st && (!pick || (e->buf.cap >= bytes && pick->buf.cap < bytes))) pick = e.get();
        if (!pick)
            for (auto &e : entries) {
                if (e->used && hipEventQuery(e->done) != hipSuccess) continue;  // in flight on another stream
                if (!pick || (e->buf.cap >= bytes && (pick->buf.cap < bytes || e->buf.cap < pick->buf.cap))) pick = e.get();
            }
        if (!pick) {
            if (entries.size() >= 16) {  // bound the pool: wait for the oldest user instead of growing without limit
                pick = entries.front().get();
                MI_HIP(hipEventSynchronize(pick->done));
            } else {
                entries.emplace_back(new Entry());
                pick = entries.back().get();
                MI_HIP(hipEventCreateWithFlags(&pick->done, hipEventDisableTiming));
            }
        }
        pick->buf.get(bytes);
        return pick;
    }
    // call after the last kernel that touches the entry has been enqueued on `st`
    static void release(Entry *e, hipStream_t st) {
        MI_HIP(hipEventRecord(e->done, st));
        e->last = st;
        e->used = true;
    }
    void free_all() {
        for (auto &e : entries) {
            e->buf.release();
            if (e->done) (void)hipEventDestroy(e->done);
        }
        entries.clear();
    }
};

// HIP-event records of named kernels (mi_prof_enable / mi_prof_read).
struct ProfRecord {
    hipEvent_t e0, e1;
    uint64_t units;
};
struct Profiler {
    int level = 0;  // 0 off; 1 the headline kernels only (cheap enough to leave on in a timed region); 2 every section
    std::map<std::string, std::vector<ProfRecord>> recs;
    std::vector<hipEvent_t> pool;  // recycled events: creating one costs microseconds
    static bool headline(const char *name) {
        return !strcmp(name, "pairing") || !strcmp(name, "msm_accumulate") || !strcmp(name, "vs_derive") ||
               !strcmp(name, "ntt_pass") || !strcmp(name, "msm_total") || !strcmp(name, "msm_bitsums") || !strcmp(name, "vs_wire_decode");
    }
    bool active(const char *name) const { return level >= 2 || (level == 1 && headline(name)); }
    hipEvent_t get() {
        if (!pool.empty()) {
            hipEvent_t e = pool.back();
            pool.pop_back();
            return e;
        }
        hipEvent_t e;
        MI_HIP(hipEventCreate(&e));
        return e;
    }
    // begin/end bracket one launch on `st`; no-ops when profiling is off
    void begin(const char *name, hipStream_t st, uint64_t units) {
        if (!active(name)) return;
        ProfRecord r;
        r.units = units;
        r.e0 = get();
        r.e1 = get();
        MI_HIP(hipEventRecord(r.e0, st));
        recs[name].push_back(r);
    }
    void end(const char *name, hipStream_t st) {
        if (!active(name)) return;
        MI_HIP(hipEventRecord(recs[name].back().e1, st));
    }
};

// One Context per HIP device the process uses (mi_init(device_ids, n), or created on first use of a device pointer).
// Everything bound to a device lives here -- streams, workspaces, cached tables, the verifier's lanes and product
// slots (module slots below) -- so one host process can drive all GPUs of a node (the reference is one process:
// reference cli/src/commands/start.rs:616-642).
struct Context {
    std::mutex mu;  // serialises GPU work issued to THIS device through the library (callers are rayon / tokio threads)
    std::condition_variable lane_cv;  // signalled when an accumulation lane or a product slot of this device frees up
    Profiler prof;
    bool inited = false;
    int device = 0;   // HIP ordinal
    int index = 0;    // position in the library's device list (0 = primary)
    int n_cus = 256;
    ScratchPool scratch;
    // device-bound state of the other translation units (twiddle tables, pairing constants, verifier pipeline, ...),
    // created on first use, destroyed by release_modules() while the device is still alive
    std::map<std::string, std::shared_ptr<void>> modules;
    template <class T>
    T &mod(const char *name) {
        auto it = modules.find(name);
        if (it == modules.end()) it = modules.emplace(name, std::shared_ptr<void>(new T(), [](void *p) { delete (T *)p; })).first;
        return *(T *)it->second.get();
    }
    // kernels whose dynamic-LDS limit has been raised on this device (hipFuncSetAttribute is per device)
    std::map<const void *, bool> func_attr_done;
    void raise_lds_limit(const void *fn, int bytes) {
        if (func_attr_done[fn]) return;
        MI_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        func_attr_done[fn] = true;
    }
    hipStream_t stream = nullptr;  // library-owned stream for host-pointer entry points
    std::map<std::string, DevBuf> bufs;
    std::map<uint64_t, DevBuf> handles;  // registered bases
    std::map<uint64_t, size_t> handle_len;
    uint64_t next_handle = 1;
    void *pinned = nullptr;  // small pinned staging area for results
    size_t pinned_cap = 0;

    std::map<std::string, std::pair<void *, size_t>> pins;  // named pinned staging buffers
    std::map<std::string, hipEvent_t> events;  // named events (completion of an enqueued job)
    hipEvent_t event_named(const char *name) {
        auto it = events.find(name);
        if (it != events.end()) return it->second;
        hipEvent_t e;
        MI_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        events[name] = e;
        return e;
    }
    void *buf(const char *name, size_t bytes) { return bufs[name].get(bytes); }
    void *pin_named(const char *name, size_t bytes) {
        auto &e = pins[name];
        if (bytes > e.second) {
            if (e.first) MI_HIP(hipHostFree(e.first));
            e.first = nullptr;
            MI_HIP(hipHostMalloc(&e.first, bytes + 4096, hipHostMallocDefault));
            e.second = bytes + 4096;
        }
        return e.first;
    }
    void *pin(size_t bytes) {
        if (bytes > pinned_cap) {
            if (pinned) MI_HIP(hipHostFree(pinned));
            pinned = nullptr;
            MI_HIP(hipHostMalloc(&pinned, bytes + 4096, hipHostMallocDefault));
            pinned_cap = bytes + 4096;
        }
        return pinned;
    }
};

// ---- device registry (capi.hip)
Context &primary();                                  // device_ids[0] of mi_init, or the current HIP device
int32_t ensure_init();                               // the primary exists
Context *context_at(int index);                      // nullptr when out of range
int n_contexts();
Context *context_for_device(int hip_device);         // created on first use; nullptr if the device cannot be used
Context *context_of_ptr(const void *device_ptr);     // the device that owns a device pointer (primary if unknown)

// The context mutex guards the context's maps and the verifier contexts' bookkeeping.  Entry points hold it for
// their whole body, except around long host-side waits and folds (Unlocked below), so that one host thread can
// enqueue batch k+1 while another folds batch k and a third waits for a pairing product.
inline thread_local std::unique_lock<std::mutex> *tl_ctx_lock = nullptr;
struct Unlocked {
    std::unique_lock<std::mutex> *lk;
    Unlocked() : lk(tl_ctx_lock) {
        if (lk) lk->unlock();
    }
    ~Unlocked() {
        if (lk) lk->lock();
    }
};
// run f(cx) with the device's lock held and the device current
template <class F>
int32_t guarded_on(Context *cxp, F &&f) {
    struct Reset {
        std::unique_lock<std::mutex> *prev;
        ~Reset() { tl_ctx_lock = prev; }
    };
    try {
        if (!cxp) return MI_ERR_NO_DEVICE;
        Context &cx = *cxp;
        std::unique_lock<std::mutex> lk(cx.mu);
        Reset reset{tl_ctx_lock};  // destroyed before lk
        tl_ctx_lock = &lk;
        MI_HIP(hipSetDevice(cx.device));  // never rely on a foreign thread's current device
        return f(cx);
    } catch (const HipError &h) {
        return status_of(h);
    } catch (const std::bad_alloc &) {
        return MI_ERR_OOM;
    } catch (...) {
        return MI_ERR_KERNEL;
    }
}
// host-pointer entry points run on the primary device
template <class F>
int32_t guarded(F &&f) {
    const int32_t s = ensure_init();
    if (s != MI_OK) return s;
    return guarded_on(&primary(), std::forward<F>(f));
}
// device-pointer entry points run on the device that owns the pointer
template <class F>
int32_t guarded_ptr(const void *device_ptr, F &&f) {
    const int32_t s = ensure_init();
    if (s != MI_OK) return s;
    Context *cx;
    {
        HostSpan hs("route.ptr");
        cx = context_of_ptr(device_ptr);
    }
    return guarded_on(cx, std::forward<F>(f));
}

inline unsigned cdiv(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }

}  // namespace mi
