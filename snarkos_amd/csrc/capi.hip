// libmi355x_snark.so -- C ABI entry points (include/mi355x_snark.h) over the HIP kernels.
#include <string.h>
#include <atomic>
#include <new>
#include <vector>
#include "api.hpp"
#include "field28.hpp"
#include "tower.hpp"
#include "codec28.hpp"

extern "C" int mi_host_cpu_ok(void);  // cpu_probe.cpp (built without the x86-64-v3 / ADX flags)

namespace mi {

static bool host_cpu_ok() { return mi_host_cpu_ok() != 0; }

// ------------------------------------------------------------ device registry
static std::mutex g_reg_mu;
static std::vector<std::unique_ptr<Context>> g_ctxs;  // index 0 = primary; entries are never removed before mi_shutdown

static Context *create_context_locked(int dev) {  // g_reg_mu held
    for (auto &c : g_ctxs)
        if (c->device == dev) return c.get();
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || dev < 0 || dev >= count) return nullptr;
    int prev = -1;
    (void)hipGetDevice(&prev);
    if (hipSetDevice(dev) != hipSuccess) return nullptr;
    std::unique_ptr<Context> c(new Context());
    c->device = dev;
    c->index = (int)g_ctxs.size();
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess) c->n_cus = prop.multiProcessorCount;
    const bool ok = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) == hipSuccess;
    if (prev >= 0 && prev != dev) (void)hipSetDevice(prev);
    if (!ok) return nullptr;
    c->inited = true;
    g_ctxs.push_back(std::move(c));
    return g_ctxs.back().get();
}
Context &primary() { return *g_ctxs[0]; }  // callers hold ensure_init() == MI_OK
int n_contexts() {
    std::lock_guard<std::mutex> lk(g_reg_mu);
    return (int)g_ctxs.size();
}
Context *context_at(int index) {
    std::lock_guard<std::mutex> lk(g_reg_mu);
    return index >= 0 && index < (int)g_ctxs.size() ? g_ctxs[index].get() : nullptr;
}
Context *context_for_device(int hip_device) {
    std::lock_guard<std::mutex> lk(g_reg_mu);
    return create_context_locked(hip_device);
}
Context *context_of_ptr(const void *device_ptr) {
    hipPointerAttribute_t attr;
    if (device_ptr && hipPointerGetAttributes(&attr, device_ptr) == hipSuccess && attr.type == hipMemoryTypeDevice)
        return context_for_device(attr.device);
    (void)hipGetLastError();  // not a HIP allocation we can place: the primary device decides (and faults loudly if wrong)
    return &primary();
}
int32_t ensure_init() {
    std::lock_guard<std::mutex> lk(g_reg_mu);
    if (!g_ctxs.empty()) return MI_OK;
    if (!host_cpu_ok()) return MI_ERR_NO_DEVICE;  // the host side of this build needs BMI2 + ADX + AVX2
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return MI_ERR_NO_DEVICE;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return MI_ERR_NO_DEVICE;
    return create_context_locked(dev) ? MI_OK : MI_ERR_NO_DEVICE;
}

// ------------------------------------------------------------ diagnostics kernels
__global__ void dbg_fq_mul_kernel(const Fq *a, const Fq *b, Fq *out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = mul(a[i], b[i]);
}
__global__ void dbg_fr_mul_kernel(const Fr *a, const Fr *b, Fr *out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = mul(a[i], b[i]);
}
__global__ void dbg_g1_add_kernel(const G1Affine *a, const G1Affine *b, G1Affine *out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = to_affine(add_mixed(G1Xyzz::from_affine(a[i]), b[i]));
}
// the reduced-radix arithmetic (field28.hpp) through its load / store conversions
__global__ void dbg_fq_mul28_kernel(const Fq *a, const Fq *b, Fq *out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Fq28 x = load28(a[i]), y = load28(b[i]);
    // even lanes: a * b; odd lanes: the same value through squarings and lazy additions: ((a + b)^2 - (a - b)^2) / 4
    // = a b, with the division folded into the comparison by the caller (the kernel returns 4 a b there)
    if ((i & 1) == 0) out[i] = store28(mul28(x, y));
    else out[i] = store28(sub28(sqr28(normalise28(add28(x, y))), sqr28(sub28(x, y))));
}
__global__ void dbg_g1_add28_kernel(const G1Affine *a, const G1Affine *b, G1Affine *out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // mixed additions on 28-bit limbs: out = a + b + b (every branch of add_mixed28 and xyzz_add28: doubling, opposite
    // points, infinity on either side)
    G1Affine r = G1Affine::inf();
    G1Xyzz28 acc;
    bool have = false;
    const G1Affine pa = a[i], pb = b[i];
    const Fq28 one = f28_one();
    if (!pa.is_inf()) {
        acc = G1Xyzz28{mul28(load28(pa.x), one), mul28(load28(pa.y), one), one, one};
        have = true;
    }
    if (!pb.is_inf()) {
        const Fq28 qx = load28(pb.x), qy = load28(pb.y);
        for (int rep = 0; rep < 2; rep++) {
            if (!have) {
                acc = G1Xyzz28{mul28(qx, one), mul28(qy, one), one, one};
                have = true;
            } else {
                have = add_mixed28(acc, qx, qy);
            }
        }
        // and once through the full addition: + (-b)
        G1Xyzz28 nb{mul28(qx, one), neg28(mul28(qy, one)), one, one};
        nb.y = normalise28(nb.y);
        xyzz_acc28(acc, have, nb, true);
    }
    if (have) r = to_affine(xyzz28_store(acc));
    out[i] = r;  // a + b
}
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__global__ void __launch_bounds__(128) dbg_gen_points_kernel(uint64_t seed, size_t n, G1Affine g, G1Affine *out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t k = splitmix64(seed + i) | 1ull;
    uint32_t kw[2] = {(uint32_t)k, (uint32_t)(k >> 32)};
    G1Xyzz acc = G1Xyzz::inf();
    for (int b = 63; b >= 0; b--) {
        acc = dbl(acc);
        if ((kw[b >> 5] >> (b & 31)) & 1u) acc = add_mixed(acc, g);
    }
    G1Affine a = to_affine(acc);
    uint4 *o = reinterpret_cast<uint4 *>(out + i);
#pragma unroll
    for (int q = 0; q < 3; q++) {
        o[q] = make_uint4(a.x.l[4 * q], a.x.l[4 * q + 1], a.x.l[4 * q + 2], a.x.l[4 * q + 3]);
        o[3 + q] = make_uint4(a.y.l[4 * q], a.y.l[4 * q + 1], a.y.l[4 * q + 2], a.y.l[4 * q + 3]);
    }
}
__global__ void __launch_bounds__(256) dbg_fq_mul_throughput_kernel(Fq *out, uint32_t iters) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    Fq a = Fq::one(), b = Fq::r2();
    a.l[0] += (uint32_t)i;
    for (uint32_t k = 0; k < iters; k++) {
        a = mul(a, b);
        b = mul(b, a);
    }
    out[i] = add(a, b);
}

// ------------------------------------------------------------ Poseidon sponge over Fq, one lane per sponge
__device__ __forceinline__ Fq pow17(const Fq &x) {
    Fq x2 = sqr(x), x4 = sqr(x2), x8 = sqr(x4), x16 = sqr(x8);
    return mul(x16, x);
}
__device__ __attribute__((noinline)) void poseidon_permute(Fq st[3], const PoseidonConsts *__restrict__ pc) {
    for (int r = 0; r < MI_POSEIDON_ROUNDS; r++) {
#pragma unroll
        for (int i = 0; i < 3; i++) st[i] = add(st[i], pc->ark[r][i]);
        const bool full = r < MI_POSEIDON_FULL_HALF || r >= MI_POSEIDON_FULL_HALF + MI_POSEIDON_PARTIAL;
        st[0] = pow17(st[0]);
        if (full) {  // wave-uniform branch
            st[1] = pow17(st[1]);
            st[2] = pow17(st[2]);
        }
        Fq o[3];
#pragma unroll
        for (int i = 0; i < 3; i++) o[i] = add(add(mul(pc->mds[i][0], st[0]), mul(pc->mds[i][1], st[1])), mul(pc->mds[i][2], st[2]));
#pragma unroll
        for (int i = 0; i < 3; i++) st[i] = o[i];
    }
}
// two waves per 64 points (codec28.hpp): one takes the square root, the other runs the subgroup test beside it
__global__ void __launch_bounds__(128) g1_decompress_kernel(const uint8_t *__restrict__ in, size_t n, int check_subgroup,
                                                            G1Affine *__restrict__ out, uint8_t *__restrict__ ok, SqrtDev T) {
    extern __shared__ int32_t c28_lds[];  // C28_LDS_BYTES
    const size_t i = (size_t)blockIdx.x * 64 + (threadIdx.x & 63);
    G1Affine p;
    const bool good = g1_decode28(i < n ? in + 48 * i : nullptr, check_subgroup != 0, T, c28_lds, p);  // 48 i is a multiple of 16
    if (threadIdx.x < 64 && i < n) {
        out[i] = p;
        ok[i] = good ? 1 : 0;
    }
}
__global__ void __launch_bounds__(64) g1_compress_kernel(const G1Affine *__restrict__ in, size_t n, uint8_t *__restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    uint8_t b[48];
    g1_compress_one(in[i], b);
    for (int k = 0; k < 48; k++) out[48 * i + k] = b[k];
}
void g1_decompress_device(Context &cx, const uint8_t *d_in, size_t n, int check_subgroup, G1Affine *d_out, uint8_t *d_ok, hipStream_t st) {
    if (!n) return;
    stats().points_decompressed += n;
    cx.prof.begin("g1_decompress", st, n);
    hipLaunchKernelGGL(g1_decompress_kernel, dim3(cdiv(n, 64)), dim3(128), C28_LDS_BYTES, st, d_in, n, check_subgroup, d_out, d_ok,
                       sqrt_tables(cx, st));
    cx.prof.end("g1_decompress", st);
    MI_HIP(hipGetLastError());
}

__global__ void __launch_bounds__(64) poseidon_hash_many_kernel(const Fq *__restrict__ in, size_t n, size_t m, size_t k,
                                                                 const PoseidonConsts *__restrict__ pc, Fq *__restrict__ out) {
    size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    Fq st[3] = {Fq::zero(), Fq::zero(), Fq::zero()};
    const Fq *x = in + s * m;
    for (size_t i = 0; i < m; i += 2) {
        st[1] = add(st[1], x[i]);
        if (i + 1 < m) st[2] = add(st[2], x[i + 1]);
        poseidon_permute(st, pc);
    }
    if (m == 0) poseidon_permute(st, pc);
    size_t produced = 0;
    while (produced < k) {
        out[s * k + produced++] = st[1];
        if (produced < k) out[s * k + produced++] = st[2];
        if (produced < k) poseidon_permute(st, pc);
    }
}
struct PoseidonState {
    DevBuf consts;
    bool ready = false;
    ~PoseidonState() { consts.release(); }
};
const PoseidonConsts *poseidon_consts(Context &cx, hipStream_t st) {
    PoseidonState &ps = cx.mod<PoseidonState>("poseidon");
    if (!ps.ready) {
        static PoseidonConsts h;
        static std::once_flag once;
        std::call_once(once, [] {
            auto cv = [](const uint64_t *l) {
                Fq r;
                for (int i = 0; i < 6; i++) { r.l[2 * i] = (uint32_t)l[i]; r.l[2 * i + 1] = (uint32_t)(l[i] >> 32); }
                return r;
            };
            for (int r = 0; r < MI_POSEIDON_ROUNDS; r++)
                for (int i = 0; i < 3; i++) h.ark[r][i] = cv(MI_POSEIDON_ARK[r][i]);
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 3; j++) h.mds[i][j] = cv(MI_POSEIDON_MDS[i][j]);
        });
        void *d = ps.consts.get(sizeof(PoseidonConsts));
        MI_HIP(hipMemcpyAsync(d, &h, sizeof h, hipMemcpyHostToDevice, st));
        MI_HIP(hipStreamSynchronize(st));
        ps.ready = true;
    }
    return (const PoseidonConsts *)ps.consts.p;
}
static void poseidon_device(Context &cx, const Fq *d_in, size_t n, size_t m, size_t k, Fq *d_out, hipStream_t st) {
    const PoseidonConsts *pc = poseidon_consts(cx, st);
    cx.prof.begin("poseidon", st, n);
    hipLaunchKernelGGL(poseidon_hash_many_kernel, dim3(cdiv(n, 64)), dim3(64), 0, st, d_in, n, m, k, pc, d_out);
    cx.prof.end("poseidon", st);
    MI_HIP(hipGetLastError());
}

template <class T, class K>
static int32_t dbg_binary(const T *a, const T *b, T *out, size_t n, K kernel) {
    if (!a || !b || !out) return MI_ERR_BAD_ARGS;
    return guarded([&](Context &cx) -> int32_t {
        T *da = (T *)cx.buf("dbg.a", sizeof(T) * n), *db = (T *)cx.buf("dbg.b", sizeof(T) * n), *dc = (T *)cx.buf("dbg.c", sizeof(T) * n);
        MI_HIP(hipMemcpyAsync(da, a, sizeof(T) * n, hipMemcpyHostToDevice, cx.stream));
        MI_HIP(hipMemcpyAsync(db, b, sizeof(T) * n, hipMemcpyHostToDevice, cx.stream));
        hipLaunchKernelGGL(kernel, dim3(cdiv(n, 128)), dim3(128), 0, cx.stream, da, db, dc, n);
        MI_HIP(hipGetLastError());
        MI_HIP(hipMemcpyAsync(out, dc, sizeof(T) * n, hipMemcpyDeviceToHost, cx.stream));
        MI_HIP(hipStreamSynchronize(cx.stream));
        return MI_OK;
    });
}

}  // namespace mi

using namespace mi;

extern "C" {

int32_t mi_init(const int32_t *device_ids, int32_t n_devices) {
    if (n_devices < 0 || n_devices > 64 || (n_devices > 0 && !device_ids)) return MI_ERR_BAD_ARGS;
    if (!host_cpu_ok()) return MI_ERR_NO_DEVICE;
    // The streamed verifier keeps ~20 HIP streams busy; the runtime maps them onto 4 hardware queues unless
    // GPU_MAX_HW_QUEUES says otherwise, and a 20-batch stream then takes 11 ms instead of 6.  The variable is the
    // embedding process's to set, before its first HIP call and before it has threads: the library
    // never calls setenv (glibc's setenv races with any concurrent getenv of a multithreaded caller);
    // mi_runtime_hints() reports when it is missing.
    if (n_devices == 0) return ensure_init();  // the current HIP device becomes the primary
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return MI_ERR_NO_DEVICE;
    for (int i = 0; i < n_devices; i++) {
        if (device_ids[i] < 0 || device_ids[i] >= count) return MI_ERR_NO_DEVICE;
        for (int j = 0; j < i; j++)
            if (device_ids[j] == device_ids[i]) return MI_ERR_BAD_ARGS;
    }
    std::lock_guard<std::mutex> lk(g_reg_mu);
    for (int i = 0; i < n_devices; i++)  // idempotent: devices already in the list keep their place
        if (!create_context_locked(device_ids[i])) return MI_ERR_NO_DEVICE;
    return MI_OK;
}
int32_t mi_device_count(void) { return n_contexts(); }
int32_t mi_device_id(int32_t index) {
    Context *c = context_at(index);
    return c ? c->device : MI_ERR_BAD_ARGS;
}
int32_t mi_shutdown(void) {
    comm_release_all();
    stream_release_all();
    std::lock_guard<std::mutex> lk(g_reg_mu);
    verify_release_keys();  // verifier contexts: per-device key copies are freed with the devices below
    for (auto &c : g_ctxs) {
        std::lock_guard<std::mutex> lk2(c->mu);
        (void)hipSetDevice(c->device);
        (void)hipDeviceSynchronize();
        c->modules.clear();  // twiddle tables, pairing constants, the verifier pipeline's streams ...
        c->scratch.free_all();
        for (auto &kv : c->bufs) kv.second.release();
        for (auto &kv : c->handles) kv.second.release();
        c->bufs.clear();
        c->handles.clear();
        c->handle_len.clear();
        if (c->pinned) (void)hipHostFree(c->pinned);
        c->pinned = nullptr;
        c->pinned_cap = 0;
        (void)hipStreamDestroy(c->stream);
        c->stream = nullptr;
        for (auto &kv : c->events) (void)hipEventDestroy(kv.second);
        c->events.clear();
        for (auto &kv : c->pins)
            if (kv.second.first) (void)hipHostFree(kv.second.first);
        c->pins.clear();
        c->inited = false;
    }
    g_ctxs.clear();
    return MI_OK;
}
const char *mi_error_string(int32_t s) {
    switch (s) {
    case MI_OK: return "ok";
    case MI_ERR_BAD_ARGS: return "bad arguments";
    case MI_ERR_NO_DEVICE: return "no HIP device";
    case MI_ERR_OOM: return "out of device memory";
    case MI_ERR_KERNEL: return "HIP runtime or kernel failure";
    case MI_ERR_NOT_INIT: return "not initialised";
    case MI_ERR_AGAIN: return "not ready yet";
    case MI_ERR_DONE: return "stream finished";
    default: return "unknown status";
    }
}
int32_t mi_abi_version(void) { return 3; }
int32_t mi_runtime_hints(void) {
    int32_t h = 0;
    const char *e = getenv("GPU_MAX_HW_QUEUES");
    if (!e || atoi(e) < 24) h |= MI_HINT_HW_QUEUES;
    return h;
}
int32_t mi_stats_read(mi_stats *out) {
    if (!out) return MI_ERR_BAD_ARGS;
    memset(out, 0, sizeof *out);
    const Stats &s = stats();
    out->batches_verified = s.batches_verified.load();
    out->proofs_verified = s.proofs_verified.load();
    out->proofs_rejected = s.proofs_rejected.load();
    out->verify_batch_nanos = s.verify_batch_nanos.load();
    out->msm_calls = s.msm_calls.load();
    out->msm_points = s.msm_points.load();
    out->ntt_calls = s.ntt_calls.load();
    out->ntt_elements = s.ntt_elements.load();
    out->pairing_products = s.pairing_products.load();
    out->pairing_pairs = s.pairing_pairs.load();
    out->points_decompressed = s.points_decompressed.load();
    out->fallback_batches = s.fallback_batches.load();
    out->exchange_calls = s.exchange_calls.load();
    out->exchange_nanos = s.exchange_nanos.load();
    return MI_OK;
}

// ------------------------------------------------------------ MSM
// entries per MSM are indexed with 32 bits (index << 1 | sign; offsets over all windows): n <= 2^27
static const size_t MI_MSM_MAX_N = (size_t)1 << 27;

int32_t mi_msm_g1(const mi_g1_affine *bases, const mi_bigint256 *scalars, size_t n, mi_g1_jac *out) {
    if (!out || (n && (!bases || !scalars)) || n > MI_MSM_MAX_N) return MI_ERR_BAD_ARGS;
    return guarded([&](Context &cx) -> int32_t {
        G1Affine *dp = (G1Affine *)cx.buf("msm.in_points", sizeof(G1Affine) * (n ? n : 1));
        uint32_t *ds = (uint32_t *)cx.buf("msm.in_scalars", 32 * (n ? n : 1));
        if (n) {
            MI_HIP(hipMemcpyAsync(dp, bases, sizeof(G1Affine) * n, hipMemcpyHostToDevice, cx.stream));
            MI_HIP(hipMemcpyAsync(ds, scalars, 32 * n, hipMemcpyHostToDevice, cx.stream));
        }
        msm_device(cx, dp, ds, n, out, cx.stream);
        return MI_OK;
    });
}
int32_t mi_msm_g1_device(const void *d_bases, const void *d_scalars, size_t n, mi_g1_jac *out, void *stream) {
    if (!out || (n && (!d_bases || !d_scalars)) || n > MI_MSM_MAX_N) return MI_ERR_BAD_ARGS;
    return guarded_ptr(d_bases, [&](Context &cx) -> int32_t {
        msm_device(cx, (const G1Affine *)d_bases, (const uint32_t *)d_scalars, n, out, (hipStream_t)stream);
        return MI_OK;
    });
}
int32_t mi_g1_sum(const mi_g1_jac *partials, size_t n, mi_g1_jac *out) {
    if (!out || (n && !partials)) return MI_ERR_BAD_ARGS;
    return guarded([&](Context &cx) -> int32_t {
        G1Xyzz acc = G1Xyzz::inf();
        const Fq one = Fq::one();
        for (size_t i = 0; i < n; i++) {
            G1Jac p;
            memcpy(&p, &partials[i], sizeof p);
            if (p.z.is_zero()) continue;
            if (!(p.z == one)) return MI_ERR_BAD_ARGS;  // not a normalised result
            acc = add_mixed(acc, G1Affine{p.x, p.y});
        }
        G1Jac res = affine_to_jac(to_affine(acc));
        memcpy(out, &res, sizeof res);
        return MI_OK;
    });
}
static std::atomic<uint64_t> g_next_handle{1};
int32_t mi_bases_register(const mi_g1_affine *pts, size_t n, uint64_t *handle_out) {
    if (!pts || !n || !handle_out) return MI_ERR_BAD_ARGS;
    int32_t st = ensure_init();
    if (st != MI_OK) return st;
    const uint64_t h = g_next_handle++;
    // resident on every device of the list: any of them may run the MSM
    for (int i = 0; i < n_contexts(); i++) {
        st = guarded_on(context_at(i), [&](Context &cx) -> int32_t {
            void *d = cx.handles[h].get(sizeof(G1Affine) * n);
            cx.handle_len[h] = n;
            MI_HIP(hipMemcpy(d, pts, sizeof(G1Affine) * n, hipMemcpyHostToDevice));
            return MI_OK;
        });
        if (st != MI_OK) {
            (void)mi_bases_release(h);
            return st;
        }
    }
    *handle_out = h;
    return MI_OK;
}
int32_t mi_bases_release(uint64_t handle) {
    int32_t st = ensure_init();
    if (st != MI_OK) return st;
    bool found = false;
    for (int i = 0; i < n_contexts(); i++) {
        st = guarded_on(context_at(i), [&](Context &cx) -> int32_t {
            auto it = cx.handles.find(handle);
            if (it == cx.handles.end()) return MI_OK;
            found = true;
            it->second.release();  // drains the device first
            cx.handles.erase(it);
            cx.handle_len.erase(handle);
            return MI_OK;
        });
        if (st != MI_OK) return st;
    }
    return found ? MI_OK : MI_ERR_BAD_ARGS;
}
int32_t mi_msm_g1_h(uint64_t handle, size_t offset, const mi_bigint256 *scalars, size_t n, mi_g1_jac *out) {
    if (!out || (n && !scalars) || n > MI_MSM_MAX_N) return MI_ERR_BAD_ARGS;
    return guarded([&](Context &cx) -> int32_t {
        auto it = cx.handles.find(handle);
        if (it == cx.handles.end() || offset + n > cx.handle_len[handle]) return MI_ERR_BAD_ARGS;
        uint32_t *ds = (uint32_t *)cx.buf("msm.in_scalars", 32 * (n ? n : 1));
        if (n) MI_HIP(hipMemcpyAsync(ds, scalars, 32 * n, hipMemcpyHostToDevice, cx.stream));
        msm_device(cx, (const G1Affine *)it->second.p + offset, ds, n, out, cx.stream);
        return MI_OK;
    });
}

// ------------------------------------------------------------ NTT
int32_t mi_ntt_fr(mi_fr *data, uint32_t log_n, int32_t inverse, int32_t coset) {
    if (!data || log_n > 30) return MI_ERR_BAD_ARGS;  // 2^30 x 32 B = 32 GiB per buffer; the coset tables cover 30 bits
    return guarded([&](Context &cx) -> int32_t {
        size_t bytes = sizeof(Fr) << log_n;
        Fr *d = (Fr *)cx.buf("ntt.io", bytes);
        MI_HIP(hipMemcpyAsync(d, data, bytes, hipMemcpyHostToDevice, cx.stream));
        ntt_device(cx, d, log_n, inverse != 0, coset != 0, cx.stream);
        MI_HIP(hipMemcpyAsync(data, d, bytes, hipMemcpyDeviceToHost, cx.stream));
        MI_HIP(hipStreamSynchronize(cx.stream));
        return MI_OK;
    });
}
int32_t mi_ntt_fr_device(void *d_data, uint32_t log_n, int32_t inverse, int32_t coset, void *stream) {
    if (!d_data || log_n > 30) return MI_ERR_BAD_ARGS;
    return guarded_ptr(d_data, [&](Context &cx) -> int32_t {
        ntt_device(cx, (Fr *)d_data, log_n, inverse != 0, coset != 0, (hipStream_t)stream);
        return MI_OK;
    });
}

// ------------------------------------------------------------ prover-side reuse (KZG10::commit, polynomial product)
__global__ void __launch_bounds__(256) fr_to_canonical_kernel(const Fr *__restrict__ in, size_t n, uint32_t *__restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr v = from_mont(in[i]);
#pragma unroll
    for (int k = 0; k < 8; k++) out[i * 8 + k] = v.l[k];
}
// a[i] *= b[i]
__global__ void __launch_bounds__(256) fr_pointwise_mul_kernel(Fr *__restrict__ a, const Fr *__restrict__ b, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = mul(a[i], b[i]);
}
static int32_t kzg_commit_impl(Context &cx, uint64_t handle, size_t offset, const Fr *d_coeffs, size_t n, mi_g1_jac *out,
                               hipStream_t st) {
    auto it = cx.handles.find(handle);
    if (it == cx.handles.end() || offset + n > cx.handle_len[handle]) return MI_ERR_BAD_ARGS;
    uint32_t *ds = (uint32_t *)cx.buf("kzg.commit_scalars", 32 * (n ? n : 1));
    if (n) hipLaunchKernelGGL(fr_to_canonical_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, d_coeffs, n, ds);
    msm_device(cx, (const G1Affine *)it->second.p + offset, ds, n, out, st);
    return MI_OK;
}
int32_t mi_kzg_commit_h(uint64_t handle, size_t offset, const mi_fr *coeffs, size_t n, mi_g1_jac *out) {
    if (!out || (n && !coeffs) || n > MI_MSM_MAX_N) return MI_ERR_BAD_ARGS;
    return guarded([&](Context &cx) -> int32_t {
        Fr *d = (Fr *)cx.buf("kzg.commit_coeffs", 32 * (n ? n : 1));
        if (n) MI_HIP(hipMemcpyAsync(d, coeffs, 32 * n, hipMemcpyHostToDevice, cx.stream));
        return kzg_commit_impl(cx, handle, offset, d, n, out, cx.stream);
    });
}
int32_t mi_kzg_commit_h_device(uint64_t handle, size_t offset, const void *d_coeffs, size_t n, mi_g1_jac *out, void *stream) {
    if (!out || (n && !d_coeffs) || n > MI_MSM_MAX_N) return MI_ERR_BAD_ARGS;
    return guarded_ptr(d_coeffs, [&](Context &cx) -> int32_t {
        return kzg_commit_impl(cx, handle, offset, (const Fr *)d_coeffs, n, out, (hipStream_t)stream);
    });
}
// ---- witness polynomial q(X) = (p(X) - p(z)) / (X - z): q_{i-1} = S_i, p(z) = S_0 with the suffix Horner sums
// S_i = c_i + z S_{i+1}.  Three kernels: (1) every thread folds OPEN_R consecutive coefficients into (L, z^R) and the
// block scans these affine maps from the top (suffix scan, Hillis-Steele over 256 threads); (2) one workgroup carries
// the block totals across blocks; (3) every thread re-walks its coefficients from the value entering its range.
#define OPEN_R 8
#define OPEN_BLOCK 256
struct OpenMap {  // x -> l + w x
    Fr l, w;
};
__device__ __forceinline__ OpenMap open_compose(const OpenMap &lo, const OpenMap &hi) {  // lo after hi (hi is nearer the top)
    return OpenMap{add(lo.l, mul(lo.w, hi.l)), mul(lo.w, hi.w)};
}
__global__ void __launch_bounds__(OPEN_BLOCK) open_local_kernel(const Fr *__restrict__ c, size_t n, Fr z, OpenMap *__restrict__ thread_maps,
                                                                OpenMap *__restrict__ block_maps) {
    __shared__ OpenMap sh[OPEN_BLOCK];
    const int t = threadIdx.x;
    const size_t base = ((size_t)blockIdx.x * OPEN_BLOCK + t) * OPEN_R;
    OpenMap m{Fr::zero(), Fr::one()};
    for (int k = OPEN_R - 1; k >= 0; k--) {  // x -> c_base + z (c_{base+1} + z (... + z x)); past the end: identity
        if (base + k < n) {
            m.l = add(c[base + k], mul(z, m.l));
            m.w = mul(z, m.w);
        }
    }
    // suffix scan: after the loop sh[t] maps the value entering the block's top to S at this thread's first index
    sh[t] = m;
    __syncthreads();
    for (int d = 1; d < OPEN_BLOCK; d <<= 1) {
        OpenMap other;
        const bool has = t + d < OPEN_BLOCK;
        if (has) other = sh[t + d];
        __syncthreads();
        if (has) {
            m = open_compose(m, other);
            sh[t] = m;
        }
        __syncthreads();
    }
    thread_maps[(size_t)blockIdx.x * OPEN_BLOCK + t] = m;
    if (t == 0) block_maps[blockIdx.x] = m;
}
// carry[b] = S at the first index of block b + 1 (0 for the last block); single workgroup, serial over the blocks in
// chunks: thread t owns blocks [t * per, (t + 1) * per)
__global__ void __launch_bounds__(1024) open_carry_kernel(const OpenMap *__restrict__ block_maps, size_t nblocks, Fr *__restrict__ carry) {
    __shared__ OpenMap sh[1024];
    const int t = threadIdx.x;
    const size_t per = (nblocks + 1023) / 1024;
    const size_t lo = (size_t)t * per, hi = lo + per < nblocks ? lo + per : nblocks;
    OpenMap m{Fr::zero(), Fr::one()};
    for (size_t b = hi; b-- > lo;) m = open_compose(block_maps[b], m);  // map over this thread's blocks
    sh[t] = m;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        OpenMap other;
        const bool has = t + d < 1024;
        if (has) other = sh[t + d];
        __syncthreads();
        if (has) {
            m = open_compose(m, other);
            sh[t] = m;
        }
        __syncthreads();
    }
    // value entering this thread's topmost block = S just above it = (suffix map of the threads above)(0)
    Fr x = t + 1 < 1024 ? sh[t + 1].l : Fr::zero();
    for (size_t b = hi; b-- > lo;) {
        carry[b] = x;
        x = add(block_maps[b].l, mul(block_maps[b].w, x));
    }
}
__global__ void __launch_bounds__(OPEN_BLOCK) open_apply_kernel(const Fr *__restrict__ c, size_t n, Fr z,
                                                                const OpenMap *__restrict__ thread_maps, const Fr *__restrict__ carry,
                                                                Fr *__restrict__ q, Fr *__restrict__ value) {
    const int t = threadIdx.x;
    const size_t tid = (size_t)blockIdx.x * OPEN_BLOCK + t, base = tid * OPEN_R;
    if (base >= n) return;
    const Fr cb = carry[blockIdx.x];
    Fr x = cb;
    if (t + 1 < OPEN_BLOCK) {
        const OpenMap m = thread_maps[tid + 1];
        x = add(m.l, mul(m.w, cb));
    }
    for (int k = OPEN_R - 1; k >= 0; k--) {
        const size_t i = base + k;
        if (i < n) {
            x = add(c[i], mul(z, x));  // S_i
            if (i > 0) q[i - 1] = x;
            else *value = x;
        }
    }
}
static int32_t kzg_open_impl(Context &cx, uint64_t handle, const mi_fr *coeffs, size_t n, const mi_fr *point, mi_fr *value_out,
                             mi_g1_jac *witness_out) {
    {
        hipStream_t st = cx.stream;
        if (n == 0) {
            memset(value_out, 0, sizeof *value_out);
            return kzg_commit_impl(cx, handle, 0, nullptr, 0, witness_out, st);
        }
        if (cx.handles.find(handle) == cx.handles.end() || n - 1 > cx.handle_len[handle]) return MI_ERR_BAD_ARGS;
        Fr z;
        memcpy(&z, point, sizeof z);
        const size_t nthreads = (n + OPEN_R - 1) / OPEN_R, nblocks = (nthreads + OPEN_BLOCK - 1) / OPEN_BLOCK;
        Fr *dc = (Fr *)cx.buf("open.coeffs", sizeof(Fr) * n);
        Fr *dq = (Fr *)cx.buf("open.q", sizeof(Fr) * n);  // q (n - 1 coefficients) followed by the value
        OpenMap *tm = (OpenMap *)cx.buf("open.thread_maps", sizeof(OpenMap) * nblocks * OPEN_BLOCK);
        OpenMap *bm = (OpenMap *)cx.buf("open.block_maps", sizeof(OpenMap) * nblocks);
        Fr *carry = (Fr *)cx.buf("open.carry", sizeof(Fr) * nblocks);
        MI_HIP(hipMemcpyAsync(dc, coeffs, sizeof(Fr) * n, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(open_local_kernel, dim3((unsigned)nblocks), dim3(OPEN_BLOCK), 0, st, dc, n, z, tm, bm);
        hipLaunchKernelGGL(open_carry_kernel, dim3(1), dim3(1024), 0, st, bm, nblocks, carry);
        hipLaunchKernelGGL(open_apply_kernel, dim3((unsigned)nblocks), dim3(OPEN_BLOCK), 0, st, dc, n, z, tm, carry, dq, dq + (n - 1));
        MI_HIP(hipGetLastError());
        Fr *hv = (Fr *)cx.pin(sizeof(Fr));
        MI_HIP(hipMemcpyAsync(hv, dq + (n - 1), sizeof(Fr), hipMemcpyDeviceToHost, st));
        int32_t s = kzg_commit_impl(cx, handle, 0, dq, n - 1, witness_out, st);  // msm_device waits for the stream's work
        if (s != MI_OK) return s;
        MI_HIP(hipStreamSynchronize(st));
        memcpy(value_out, hv, sizeof(Fr));
        return MI_OK;
    }
}
int32_t mi_kzg_open_h(uint64_t handle, const mi_fr *coeffs, size_t n, const mi_fr *point, mi_fr *value_out, mi_g1_jac *witness_out) {
    if (!point || !value_out || !witness_out || (n && !coeffs) || n > MI_MSM_MAX_N) return MI_ERR_BAD_ARGS;
    return guarded([&](Context &cx) -> int32_t { return kzg_open_impl(cx, handle, coeffs, n, point, value_out, witness_out); });
}
// host sum of two normalised results (z = 1, or z = 0 for infinity)
static void g1_jac_add2(const mi_g1_jac &a, const mi_g1_jac &b, mi_g1_jac *out) {
    G1Xyzz acc = G1Xyzz::inf();
    for (const mi_g1_jac *j : {&a, &b}) {
        G1Jac p;
        memcpy(&p, j, sizeof p);
        if (!p.z.is_zero()) acc = add_mixed(acc, G1Affine{p.x, p.y});
    }
    G1Jac res = affine_to_jac(to_affine(acc));
    memcpy(out, &res, sizeof res);
}
// ---- hiding (KZG10::commit / open with a hiding polynomial): C = sum c_i powers_g[i] +
// sum rho_j powers_gamma_g[j]; the opening proof carries random_v = rho(z) and the witness of both quotients
int32_t mi_kzg_commit_hiding_h(uint64_t powers_handle, uint64_t gamma_powers_handle, const mi_fr *coeffs, size_t n,
                               const mi_fr *rand_coeffs, size_t m, mi_g1_jac *out) {
    if (!out || (n && !coeffs) || (m && !rand_coeffs) || n > MI_MSM_MAX_N || m > MI_MSM_MAX_N) return MI_ERR_BAD_ARGS;
    return guarded([&](Context &cx) -> int32_t {
        mi_g1_jac a, b;
        Fr *d = (Fr *)cx.buf("kzg.commit_coeffs", 32 * ((n > m ? n : m) + 1));
        if (n) MI_HIP(hipMemcpyAsync(d, coeffs, 32 * n, hipMemcpyHostToDevice, cx.stream));
        int32_t s = kzg_commit_impl(cx, powers_handle, 0, d, n, &a, cx.stream);
        if (s != MI_OK) return s;
        if (m) MI_HIP(hipMemcpyAsync(d, rand_coeffs, 32 * m, hipMemcpyHostToDevice, cx.stream));
        s = kzg_commit_impl(cx, gamma_powers_handle, 0, d, m, &b, cx.stream);
        if (s != MI_OK) return s;
        g1_jac_add2(a, b, out);
        return MI_OK;
    });
}
int32_t mi_kzg_open_hiding_h(uint64_t powers_handle, uint64_t gamma_powers_handle, const mi_fr *coeffs, size_t n,
                             const mi_fr *rand_coeffs, size_t m, const mi_fr *point, mi_fr *value_out, mi_fr *random_v_out,
                             mi_g1_jac *witness_out) {
    if (!point || !value_out || !random_v_out || !witness_out || (n && !coeffs) || (m && !rand_coeffs) || n > MI_MSM_MAX_N ||
        m > MI_MSM_MAX_N)
        return MI_ERR_BAD_ARGS;
    return guarded([&](Context &cx) -> int32_t {
        mi_g1_jac wa, wb;
        int32_t s = kzg_open_impl(cx, powers_handle, coeffs, n, point, value_out, &wa);
        if (s != MI_OK) return s;
        s = kzg_open_impl(cx, gamma_powers_handle, rand_coeffs, m, point, random_v_out, &wb);
        if (s != MI_OK) return s;
        g1_jac_add2(wa, wb, witness_out);
        return MI_OK;
    });
}
// KZG10::commit_lagrange: the polynomial is given by its evaluations over the domain and the bases are the
// Lagrange-basis SRS elements L_i(beta) G -- the same MSM, named for the reference's entry point
int32_t mi_kzg_commit_lagrange_h(uint64_t lagrange_handle, const mi_fr *evals, size_t n, mi_g1_jac *out) {
    return mi_kzg_commit_h(lagrange_handle, 0, evals, n, out);
}
// fft::EvaluationDomain::divide_by_vanishing_poly_on_coset_in_place, generalised to an evaluation domain of size
// M = 2^log_m >= N = 2^log_n: evals[i] = p(g w_M^i) is divided by Z_N(g w_M^i) = g^N (w_M^N)^i - 1, which takes M / N
// values (all non-zero: g is not in the subgroup): a table of their inverses, one product per element
__global__ void __launch_bounds__(256) fr_scale_periodic_kernel(Fr *__restrict__ a, size_t n, const Fr *__restrict__ tab, size_t period) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = mul(a[i], tab[i & (period - 1)]);
}
static int32_t divide_by_vanishing_impl(Context &cx, Fr *d, uint32_t log_m, uint32_t log_n, hipStream_t st) {
    const size_t K = (size_t)1 << (log_m - log_n);
    std::vector<Fr> tab(K);
    Fr gN = Fr::zero();  // g^N, g = 22 (the coset generator of mi_ntt_fr)
    for (int i = 0; i < 4; i++) { gN.l[2 * i] = (uint32_t)MI_FR_GENERATOR[i]; gN.l[2 * i + 1] = (uint32_t)(MI_FR_GENERATOR[i] >> 32); }
    for (uint32_t b = 0; b < log_n; b++) gN = sqr(gN);
    const Fr wK = K > 1 ? host_omega(log_m - log_n) : Fr::one();  // w_M^N has order M / N
    Fr cur = gN;
    for (size_t k = 0; k < K; k++) {
        tab[k] = inv(sub(cur, Fr::one()));
        cur = mul(cur, wK);
    }
    char nm[40];
    snprintf(nm, sizeof nm, "ntt.vanish_tab.%p", (void *)st);
    Fr *dt = (Fr *)cx.buf(nm, sizeof(Fr) * K);
    MI_HIP(hipMemcpyAsync(dt, tab.data(), sizeof(Fr) * K, hipMemcpyHostToDevice, st));
    MI_HIP(hipStreamSynchronize(st));  // tab is a local
    const size_t M = (size_t)1 << log_m;
    hipLaunchKernelGGL(fr_scale_periodic_kernel, dim3(cdiv(M, 256)), dim3(256), 0, st, d, M, (const Fr *)dt, K);
    MI_HIP(hipGetLastError());
    return MI_OK;
}
int32_t mi_ntt_divide_by_vanishing_on_coset(mi_fr *evals, uint32_t log_m, uint32_t log_n) {
    if (!evals || log_m > 30 || log_n > log_m || log_m - log_n > 12) return MI_ERR_BAD_ARGS;
    return guarded([&](Context &cx) -> int32_t {
        const size_t bytes = sizeof(Fr) << log_m;
        Fr *d = (Fr *)cx.buf("ntt.io", bytes);
        MI_HIP(hipMemcpyAsync(d, evals, bytes, hipMemcpyHostToDevice, cx.stream));
        const int32_t s = divide_by_vanishing_impl(cx, d, log_m, log_n, cx.stream);
        if (s != MI_OK) return s;
        MI_HIP(hipMemcpyAsync(evals, d, bytes, hipMemcpyDeviceToHost, cx.stream));
        MI_HIP(hipStreamSynchronize(cx.stream));
        return MI_OK;
    });
}
int32_t mi_ntt_divide_by_vanishing_on_coset_device(void *d_evals, uint32_t log_m, uint32_t log_n, void *stream) {
    if (!d_evals || log_m > 30 || log_n > log_m || log_m - log_n > 12) return MI_ERR_BAD_ARGS;
    return guarded_ptr(d_evals, [&](Context &cx) -> int32_t {
        return divide_by_vanishing_impl(cx, (Fr *)d_evals, log_m, log_n, (hipStream_t)stream);
    });
}
int32_t mi_poly_mul_fr(const mi_fr *a, size_t na, const mi_fr *b, size_t nb, mi_fr *out) {
    if (!a || !b || !out || !na || !nb || na + nb - 1 > ((size_t)1 << 30)) return MI_ERR_BAD_ARGS;
    return guarded([&](Context &cx) -> int32_t {
        const size_t nout = na + nb - 1;
        uint32_t log_n = 0;
        while (((size_t)1 << log_n) < nout) log_n++;
        const size_t N = (size_t)1 << log_n;
        Fr *da = (Fr *)cx.buf("poly.a", sizeof(Fr) * N), *db = (Fr *)cx.buf("poly.b", sizeof(Fr) * N);
        MI_HIP(hipMemsetAsync(da, 0, sizeof(Fr) * N, cx.stream));  // zero is zero in Montgomery form too
        MI_HIP(hipMemsetAsync(db, 0, sizeof(Fr) * N, cx.stream));
        MI_HIP(hipMemcpyAsync(da, a, sizeof(Fr) * na, hipMemcpyHostToDevice, cx.stream));
        MI_HIP(hipMemcpyAsync(db, b, sizeof(Fr) * nb, hipMemcpyHostToDevice, cx.stream));
        ntt_device(cx, da, log_n, false, false, cx.stream);
        ntt_device(cx, db, log_n, false, false, cx.stream);
        hipLaunchKernelGGL(fr_pointwise_mul_kernel, dim3(cdiv(N, 256)), dim3(256), 0, cx.stream, da, db, N);
        ntt_device(cx, da, log_n, true, false, cx.stream);
        MI_HIP(hipGetLastError());
        MI_HIP(hipMemcpyAsync(out, da, sizeof(Fr) * nout, hipMemcpyDeviceToHost, cx.stream));
        MI_HIP(hipStreamSynchronize(cx.stream));
        return MI_OK;
    });
}

// ------------------------------------------------------------ pairing / KZG
int32_t mi_pairing_product_is_one(const mi_g1_affine *ps, const mi_g2_affine *qs, size_t n, uint8_t *is_one) {
    if (!ps || !qs || !is_one || n == 0 || n > MI_MAX_PAIRS) return MI_ERR_BAD_ARGS;
    return guarded([&](Context &cx) -> int32_t {
        Fq12 f;
        pairing_product_host_args(cx, (const G1Affine *)ps, (const G2Affine *)qs, n, &f, cx.stream, true);
        *is_one = f.is_one() ? 1 : 0;
        return MI_OK;
    });
}
int32_t mi_pairing(const mi_g1_affine *p, const mi_g2_affine *q, mi_fq12 *out) {
    if (!p || !q || !out) return MI_ERR_BAD_ARGS;
    return guarded([&](Context &cx) -> int32_t {
        Fq12 f;
        pairing_product_host_args(cx, (const G1Affine *)p, (const G2Affine *)q, 1, &f, cx.stream, false);  // the VALUE
        memcpy(out, &f, sizeof f);
        return MI_OK;
    });
}
int32_t mi_kzg_batch_check(const mi_kzg_vk *vk, const mi_g1_affine *comms, const mi_fr *points, const mi_fr *values,
                           const mi_g1_affine *witnesses, size_t n, const uint8_t *randomizers128, uint8_t *ok_out) {
    return mi_kzg_batch_check_hiding(vk, nullptr, comms, points, values, witnesses, nullptr, n, randomizers128, ok_out);
}
int32_t mi_kzg_batch_check_hiding(const mi_kzg_vk *vk, const mi_g1_affine *gamma_g, const mi_g1_affine *comms, const mi_fr *points,
                                  const mi_fr *values, const mi_g1_affine *witnesses, const mi_fr *random_v, size_t n,
                                  const uint8_t *randomizers128, uint8_t *ok_out) {
    if (!vk || !comms || !points || !values || !witnesses || !randomizers128 || !ok_out || n == 0) return MI_ERR_BAD_ARGS;
    if ((gamma_g == nullptr) != (random_v == nullptr)) return MI_ERR_BAD_ARGS;
    return guarded([&](Context &cx) -> int32_t {
        return kzg_batch_check_host(cx, vk, gamma_g, comms, points, values, witnesses, random_v, n, randomizers128, ok_out);
    });
}

// ------------------------------------------------------------ compressed points
int32_t mi_g1_decompress(const uint8_t *compressed48, size_t n, int32_t check_subgroup, mi_g1_affine *out, uint8_t *ok) {
    if (!n || !compressed48 || !out || !ok) return MI_ERR_BAD_ARGS;
    return guarded([&](Context &cx) -> int32_t {
        uint8_t *din = (uint8_t *)cx.buf("codec.in", 48 * n + 16);
        G1Affine *dout = (G1Affine *)cx.buf("codec.out", sizeof(G1Affine) * n);
        uint8_t *dok = (uint8_t *)cx.buf("codec.ok", n);
        MI_HIP(hipMemcpyAsync(din, compressed48, 48 * n, hipMemcpyHostToDevice, cx.stream));
        g1_decompress_device(cx, din, n, check_subgroup, dout, dok, cx.stream);
        MI_HIP(hipMemcpyAsync(out, dout, sizeof(G1Affine) * n, hipMemcpyDeviceToHost, cx.stream));
        MI_HIP(hipMemcpyAsync(ok, dok, n, hipMemcpyDeviceToHost, cx.stream));
        MI_HIP(hipStreamSynchronize(cx.stream));
        return MI_OK;
    });
}
int32_t mi_g1_decompress_device(const void *d_compressed48, size_t n, int32_t check_subgroup, void *d_out, void *d_ok, void *stream) {
    if (!n || !d_compressed48 || !d_out || !d_ok || ((uintptr_t)d_compressed48 & 15)) return MI_ERR_BAD_ARGS;
    return guarded_ptr(d_out, [&](Context &cx) -> int32_t {
        g1_decompress_device(cx, (const uint8_t *)d_compressed48, n, check_subgroup, (G1Affine *)d_out, (uint8_t *)d_ok, (hipStream_t)stream);
        return MI_OK;
    });
}
int32_t mi_g1_compress(const mi_g1_affine *points, size_t n, uint8_t *compressed48_out) {
    if (!n || !points || !compressed48_out) return MI_ERR_BAD_ARGS;
    return guarded([&](Context &cx) -> int32_t {
        G1Affine *din = (G1Affine *)cx.buf("codec.out", sizeof(G1Affine) * n);
        uint8_t *dout = (uint8_t *)cx.buf("codec.in", 48 * n + 16);
        MI_HIP(hipMemcpyAsync(din, points, sizeof(G1Affine) * n, hipMemcpyHostToDevice, cx.stream));
        hipLaunchKernelGGL(g1_compress_kernel, dim3(cdiv(n, 64)), dim3(64), 0, cx.stream, din, n, dout);
        MI_HIP(hipGetLastError());
        MI_HIP(hipMemcpyAsync(compressed48_out, dout, 48 * n, hipMemcpyDeviceToHost, cx.stream));
        MI_HIP(hipStreamSynchronize(cx.stream));
        return MI_OK;
    });
}

// ------------------------------------------------------------ Poseidon
int32_t mi_poseidon_fq_hash_many(const mi_fq *inputs, size_t n, size_t m, size_t k, mi_fq *outputs) {
    if (!n || !k || !outputs || (m && !inputs)) return MI_ERR_BAD_ARGS;
    return guarded([&](Context &cx) -> int32_t {
        Fq *din = (Fq *)cx.buf("poseidon.in", sizeof(Fq) * n * (m ? m : 1));
        Fq *dout = (Fq *)cx.buf("poseidon.out", sizeof(Fq) * n * k);
        if (m) MI_HIP(hipMemcpyAsync(din, inputs, sizeof(Fq) * n * m, hipMemcpyHostToDevice, cx.stream));
        poseidon_device(cx, din, n, m, k, dout, cx.stream);
        MI_HIP(hipMemcpyAsync(outputs, dout, sizeof(Fq) * n * k, hipMemcpyDeviceToHost, cx.stream));
        MI_HIP(hipStreamSynchronize(cx.stream));
        return MI_OK;
    });
}
int32_t mi_poseidon_fq_hash_many_device(const void *d_inputs, size_t n, size_t m, size_t k, void *d_outputs, void *stream) {
    if (!n || !k || !d_outputs || (m && !d_inputs)) return MI_ERR_BAD_ARGS;
    return guarded_ptr(d_outputs, [&](Context &cx) -> int32_t {
        poseidon_device(cx, (const Fq *)d_inputs, n, m, k, (Fq *)d_outputs, (hipStream_t)stream);
        return MI_OK;
    });
}

// ------------------------------------------------------------ VarunaShaped-v1
int32_t mi_vs_ctx_create(const mi_kzg_vk *vk, const mi_vs_circuit_vk *circuits, size_t n_circuits, uint64_t *ctx_out) {
    if (!vk || !circuits || !n_circuits || !ctx_out) return MI_ERR_BAD_ARGS;
    return guarded([&](Context &cx) -> int32_t { return vs_ctx_create(cx, vk, nullptr, nullptr, circuits, n_circuits, 0, ctx_out); });
}
int32_t mi_vs_ctx_create_v2(const mi_kzg_vk *vk, const mi_vs_key_ext *ext, const mi_vs_shape *shape, const mi_vs_circuit_vk *circuits,
                            size_t n_circuits, uint32_t flags, uint64_t *ctx_out) {
    if (!vk || !shape || !circuits || !n_circuits || !ctx_out || (flags & ~(uint32_t)(MI_VS_FS_POSEIDON | MI_VS_WIRE))) return MI_ERR_BAD_ARGS;
    return guarded([&](Context &cx) -> int32_t { return vs_ctx_create(cx, vk, ext, shape, circuits, n_circuits, flags, ctx_out); });
}
int32_t mi_vs_ctx_create_ex(const mi_kzg_vk *vk, const mi_vs_circuit_vk *circuits, size_t n_circuits, uint32_t flags, uint64_t *ctx_out) {
    if (!vk || !circuits || !n_circuits || !ctx_out || (flags & ~(uint32_t)(MI_VS_FS_POSEIDON | MI_VS_WIRE))) return MI_ERR_BAD_ARGS;
    return guarded([&](Context &cx) -> int32_t { return vs_ctx_create(cx, vk, nullptr, nullptr, circuits, n_circuits, flags, ctx_out); });
}
int32_t mi_vs_ctx_destroy(uint64_t c) {
    int32_t st = ensure_init();
    if (st != MI_OK) return st;
    for (int i = n_contexts() - 1; i >= 0; i--) {  // the primary last: its call drops the host master
        const int32_t s = guarded_on(context_at(i), [&](Context &cx) -> int32_t { return vs_ctx_destroy(cx, c); });
        if (s != MI_OK) st = s;
    }
    return st;
}
int32_t mi_vs_strides(uint64_t c, size_t *proof_bytes, size_t *rand_bytes) {
    if (!proof_bytes || !rand_bytes) return MI_ERR_BAD_ARGS;
    return vs_strides(c, proof_bytes, rand_bytes) ? MI_OK : MI_ERR_BAD_ARGS;
}
int32_t mi_vs_wire_encode(uint64_t c, const void *records, size_t n, uint8_t *wire_out) {
    if (!records || !wire_out) return MI_ERR_BAD_ARGS;
    try {
        return vs_wire_encode_host(c, records, n, wire_out);
    } catch (...) {
        return MI_ERR_KERNEL;
    }
}
int32_t mi_vs_verify_batch(uint64_t c, const mi_vs_proof *proofs, size_t n, const uint8_t *randomizers128,
                           uint8_t *verdicts, uint8_t *all_ok) {
    if (!proofs || !n || !randomizers128 || !verdicts || !all_ok) return MI_ERR_BAD_ARGS;
    return guarded([&](Context &cx) -> int32_t {
        size_t pb = 0, rb = 0;
        if (!vs_strides(c, &pb, &rb)) return MI_ERR_BAD_ARGS;
        void *dp = cx.buf("vs.in_proofs", pb * n);
        void *dr = cx.buf("vs.in_rand", rb * n);
        MI_HIP(hipMemcpyAsync(dp, proofs, pb * n, hipMemcpyHostToDevice, cx.stream));
        MI_HIP(hipMemcpyAsync(dr, randomizers128, rb * n, hipMemcpyHostToDevice, cx.stream));
        return vs_verify_batch_device(cx, c, dp, n, dr, verdicts, all_ok, cx.stream);
    });
}
int32_t mi_vs_verify_batch_device(uint64_t c, const void *d_proofs, size_t n, const void *d_randomizers128,
                                  uint8_t *verdicts, uint8_t *all_ok, void *stream) {
    if (!d_proofs || !n || !d_randomizers128 || !verdicts || !all_ok) return MI_ERR_BAD_ARGS;
    return guarded_ptr(d_proofs, [&](Context &cx) -> int32_t {
        return vs_verify_batch_device(cx, c, d_proofs, n, d_randomizers128, verdicts, all_ok, (hipStream_t)stream);
    });
}
int32_t mi_vs_identify_device(uint64_t c, const void *d_proofs, size_t n, const void *d_randomizers128, uint8_t *verdicts,
                              void *stream) {
    if (!d_proofs || !n || !d_randomizers128 || !verdicts) return MI_ERR_BAD_ARGS;
    return guarded_ptr(d_proofs, [&](Context &cx) -> int32_t {
        return vs_identify_device(cx, c, d_proofs, n, d_randomizers128, verdicts, (hipStream_t)stream);
    });
}
int32_t mi_vs_verify_block(uint64_t c, const mi_vs_proof *proofs, size_t n, const uint8_t *randomizers128, uint8_t *verdicts,
                           uint8_t *all_ok) {
    if (!proofs || !n || !randomizers128 || !verdicts || !all_ok) return MI_ERR_BAD_ARGS;
    const int32_t st = ensure_init();
    if (st != MI_OK) return st;
    const int nd = n_contexts();
    size_t pb = 0, rb = 0;
    if (!vs_strides(c, &pb, &rb)) return MI_ERR_BAD_ARGS;
    std::vector<const void *> p(nd), r(nd);
    std::vector<size_t> cnt(nd);
    const size_t base = n / nd, rem = n % nd;  // contiguous shards, sizes differ by at most one (sharding.shard_bounds)
    size_t lo = 0;
    for (int d = 0; d < nd; d++) {
        cnt[d] = base + ((size_t)d < rem ? 1 : 0);
        p[d] = (const uint8_t *)proofs + pb * lo;
        r[d] = randomizers128 + rb * lo;
        lo += cnt[d];
    }
    try {
        return vs_verify_block(c, p.data(), r.data(), cnt.data(), nd, false, verdicts, all_ok);
    } catch (const HipError &h) {
        return status_of(h);
    } catch (...) {
        return MI_ERR_KERNEL;
    }
}
int32_t mi_vs_verify_block_device(uint64_t c, const void *const *d_proofs, const void *const *d_randomizers128,
                                  const size_t *n_per_shard, int32_t n_shards, uint8_t *verdicts, uint8_t *all_ok) {
    if (!d_proofs || !d_randomizers128 || !n_per_shard || n_shards < 1 || n_shards > 64 || !verdicts || !all_ok) return MI_ERR_BAD_ARGS;
    for (int s = 0; s < n_shards; s++)
        if (n_per_shard[s] && (!d_proofs[s] || !d_randomizers128[s])) return MI_ERR_BAD_ARGS;
    const int32_t st = ensure_init();
    if (st != MI_OK) return st;
    try {
        return vs_verify_block(c, d_proofs, d_randomizers128, n_per_shard, n_shards, true, verdicts, all_ok);
    } catch (const HipError &h) {
        return status_of(h);
    } catch (...) {
        return MI_ERR_KERNEL;
    }
}
int32_t mi_vs_accumulate_device(uint64_t c, const void *d_proofs, size_t n, const void *d_randomizers128,
                                mi_vs_partial *partial_out, void *stream) {
    if (!d_proofs || !n || !d_randomizers128 || !partial_out) return MI_ERR_BAD_ARGS;
    return guarded_ptr(d_proofs, [&](Context &cx) -> int32_t {
        return vs_accumulate_device(cx, c, d_proofs, n, d_randomizers128, partial_out, (hipStream_t)stream);
    });
}
int32_t mi_vs_finalize(uint64_t c, const mi_vs_partial *partials, size_t n_partials, uint8_t *all_ok) {
    if (!partials || !n_partials || !all_ok) return MI_ERR_BAD_ARGS;
    return guarded([&](Context &cx) -> int32_t { return vs_finalize(cx, c, partials, n_partials, all_ok, cx.stream); });
}

int32_t mi_vs_accumulate_begin(uint64_t c, const void *d_proofs, size_t n, const void *d_randomizers128, void *stream,
                               uint64_t *handle_out) {
    if (!d_proofs || !n || !d_randomizers128 || !handle_out) return MI_ERR_BAD_ARGS;
    return guarded_ptr(d_proofs, [&](Context &cx) -> int32_t {
        return vs_accumulate_begin(cx, c, d_proofs, n, d_randomizers128, (hipStream_t)stream, handle_out);
    });
}
int32_t mi_vs_accumulate_end(uint64_t c, uint64_t handle, mi_vs_partial *partial_out) {
    if (!partial_out) return MI_ERR_BAD_ARGS;
    if (ensure_init() != MI_OK) return MI_ERR_NO_DEVICE;
    return guarded_on(context_at((int)((handle >> 8) & 0xff)), [&](Context &cx) -> int32_t { return vs_accumulate_end(cx, c, handle, partial_out); });
}
int32_t mi_vs_accumulate_begin_multi(uint64_t c, int32_t count, const void *const *d_proofs, const size_t *n,
                                     const void *const *d_randomizers128, void *stream, uint64_t *handle_out) {
    if (count < 1 || count > MI_VS_MAX_BATCHES_PER_PASS || !d_proofs || !n || !d_randomizers128 || !handle_out) return MI_ERR_BAD_ARGS;
    for (int k = 0; k < count; k++)
        if (!d_proofs[k] || !n[k] || !d_randomizers128[k]) return MI_ERR_BAD_ARGS;
    return guarded_ptr(d_proofs[0], [&](Context &cx) -> int32_t {
        return vs_accumulate_begin_multi(cx, c, count, d_proofs, n, d_randomizers128, (hipStream_t)stream, handle_out);
    });
}
int32_t mi_vs_accumulate_begin_merged(uint64_t c, int32_t count, const void *const *d_proofs, const size_t *n,
                                      const void *const *d_randomizers128, void *stream, uint64_t *handle_out) {
    if (count < 2 || count > MI_VS_MAX_BATCHES_PER_PASS || !d_proofs || !n || !d_randomizers128 || !handle_out) return MI_ERR_BAD_ARGS;
    for (int k = 0; k < count; k++)
        if (!d_proofs[k] || !n[k] || !d_randomizers128[k]) return MI_ERR_BAD_ARGS;
    return guarded_ptr(d_proofs[0], [&](Context &cx) -> int32_t {
        return vs_accumulate_begin_merged(cx, c, count, d_proofs, n, d_randomizers128, (hipStream_t)stream, handle_out);
    });
}
int32_t mi_vs_accumulate_end_multi(uint64_t c, uint64_t handle, mi_vs_partial *partials_out, int32_t count) {
    if (!partials_out) return MI_ERR_BAD_ARGS;
    if (ensure_init() != MI_OK) return MI_ERR_NO_DEVICE;
    return guarded_on(context_at((int)((handle >> 8) & 0xff)),
                      [&](Context &cx) -> int32_t { return vs_accumulate_end_multi(cx, c, handle, partials_out, count); });
}
int32_t mi_vs_accumulate_begin_pair(uint64_t c, const void *d_proofs_a, size_t n_a, const void *d_randomizers128_a,
                                    const void *d_proofs_b, size_t n_b, const void *d_randomizers128_b, void *stream,
                                    uint64_t *handle_out) {
    const void *pr[2] = {d_proofs_a, d_proofs_b}, *rn[2] = {d_randomizers128_a, d_randomizers128_b};
    const size_t ns[2] = {n_a, n_b};
    return mi_vs_accumulate_begin_multi(c, 2, pr, ns, rn, stream, handle_out);
}
int32_t mi_vs_accumulate_end_pair(uint64_t c, uint64_t handle, mi_vs_partial *partials_out) {
    return mi_vs_accumulate_end_multi(c, handle, partials_out, 2);
}
int32_t mi_vs_batches_per_pass(uint64_t c, size_t n) {
    if (!n) return MI_ERR_BAD_ARGS;
    return guarded([&](Context &cx) -> int32_t { return vs_batches_per_pass(cx, c, n); });
}
int32_t mi_vs_batches_per_merged_pass(uint64_t c, size_t n) {
    if (!n) return MI_ERR_BAD_ARGS;
    return guarded([&](Context &cx) -> int32_t { return vs_batches_per_merged_pass(cx, c, n); });
}
int32_t mi_vs_finalize_async(uint64_t c, const mi_vs_partial *partials, size_t n_partials, uint64_t *ticket_out) {
    if (!partials || !n_partials || !ticket_out) return MI_ERR_BAD_ARGS;
    return guarded([&](Context &cx) -> int32_t { return vs_finalize_async(cx, c, partials, n_partials, ticket_out); });
}
int32_t mi_vs_finalize_async_multi(uint64_t c, const mi_vs_partial *partials, size_t n_partials_each, int32_t count, uint64_t *ticket_out) {
    if (!partials || !n_partials_each || !ticket_out || count < 1 || count > MI_VS_MAX_PRODUCTS_PER_TICKET) return MI_ERR_BAD_ARGS;
    return guarded([&](Context &cx) -> int32_t { return vs_finalize_async_multi(cx, c, partials, n_partials_each, count, ticket_out); });
}
int32_t mi_vs_finalize_wait_multi(uint64_t c, uint64_t ticket, uint8_t *all_ok, int32_t count) {
    if (!all_ok) return MI_ERR_BAD_ARGS;
    if (ensure_init() != MI_OK) return MI_ERR_NO_DEVICE;
    return guarded_on(context_at((int)((ticket >> 8) & 0xff)),
                      [&](Context &cx) -> int32_t { return vs_finalize_wait_multi(cx, c, ticket, all_ok, count); });
}
int32_t mi_vs_wait_product_slot(uint64_t c, int32_t timeout_ms) {
    (void)c;  // slots belong to the device the context's passes run on: the primary one
    return guarded([&](Context &cx) -> int32_t { return vs_wait_product_slot(cx, timeout_ms); });
}
int32_t mi_vs_finalize_wait(uint64_t c, uint64_t ticket, uint8_t *all_ok) {
    if (!all_ok) return MI_ERR_BAD_ARGS;
    if (ensure_init() != MI_OK) return MI_ERR_NO_DEVICE;
    return guarded_on(context_at((int)((ticket >> 8) & 0xff)), [&](Context &cx) -> int32_t { return vs_finalize_wait(cx, c, ticket, all_ok); });
}

// ------------------------------------------------------------ diagnostics
int32_t mi_dbg_fq_mul(const mi_fq *a, const mi_fq *b, mi_fq *out, size_t n) {
    return dbg_binary((const Fq *)a, (const Fq *)b, (Fq *)out, n, dbg_fq_mul_kernel);
}
int32_t mi_dbg_fr_mul(const mi_fr *a, const mi_fr *b, mi_fr *out, size_t n) {
    return dbg_binary((const Fr *)a, (const Fr *)b, (Fr *)out, n, dbg_fr_mul_kernel);
}
int32_t mi_dbg_g1_add(const mi_g1_affine *a, const mi_g1_affine *b, mi_g1_affine *out, size_t n) {
    return dbg_binary((const G1Affine *)a, (const G1Affine *)b, (G1Affine *)out, n, dbg_g1_add_kernel);
}
int32_t mi_dbg_fq_mul28(const mi_fq *a, const mi_fq *b, mi_fq *out, size_t n) {
    return dbg_binary((const Fq *)a, (const Fq *)b, (Fq *)out, n, dbg_fq_mul28_kernel);
}
int32_t mi_dbg_g1_add28(const mi_g1_affine *a, const mi_g1_affine *b, mi_g1_affine *out, size_t n) {
    return dbg_binary((const G1Affine *)a, (const G1Affine *)b, (G1Affine *)out, n, dbg_g1_add28_kernel);
}
int32_t mi_dbg_fq_mul_throughput(size_t n_threads, uint32_t iters, double *gmul_per_s) {
    if (!gmul_per_s || !n_threads) return MI_ERR_BAD_ARGS;
    return guarded([&](Context &cx) -> int32_t {
        n_threads = (n_threads + 255) / 256 * 256;
        Fq *d = (Fq *)cx.buf("dbg.c", sizeof(Fq) * n_threads);
        hipEvent_t e0, e1;
        MI_HIP(hipEventCreate(&e0));
        MI_HIP(hipEventCreate(&e1));
        hipLaunchKernelGGL(dbg_fq_mul_throughput_kernel, dim3(n_threads / 256), dim3(256), 0, cx.stream, d, 8u);
        MI_HIP(hipEventRecord(e0, cx.stream));
        hipLaunchKernelGGL(dbg_fq_mul_throughput_kernel, dim3(n_threads / 256), dim3(256), 0, cx.stream, d, iters);
        MI_HIP(hipEventRecord(e1, cx.stream));
        MI_HIP(hipEventSynchronize(e1));
        float ms = 0;
        MI_HIP(hipEventElapsedTime(&ms, e0, e1));
        *gmul_per_s = 2.0 * iters * (double)n_threads / (ms * 1e-3) / 1e9;
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        return MI_OK;
    });
}

int32_t mi_dbg_gen_points_device(uint64_t seed, size_t n, void *d_out, void *stream) {
    if (!d_out || !n) return MI_ERR_BAD_ARGS;
    return guarded_ptr(d_out, [&](Context &cx) -> int32_t {
        G1Affine g;
        for (int i = 0; i < 6; i++) {
            g.x.l[2 * i] = (uint32_t)MI_G1_GEN_X[i]; g.x.l[2 * i + 1] = (uint32_t)(MI_G1_GEN_X[i] >> 32);
            g.y.l[2 * i] = (uint32_t)MI_G1_GEN_Y[i]; g.y.l[2 * i + 1] = (uint32_t)(MI_G1_GEN_Y[i] >> 32);
        }
        hipLaunchKernelGGL(dbg_gen_points_kernel, dim3(cdiv(n, 128)), dim3(128), 0, (hipStream_t)stream, seed, n, g, (G1Affine *)d_out);
        MI_HIP(hipGetLastError());
        return MI_OK;
    });
}

int32_t mi_dbg_msm_plan(size_t n, int32_t scalar_bits, int32_t *c, int32_t *nwin, int32_t *chunk, int32_t *small_path) {
    if (!n || !c || !nwin || !chunk || !small_path) return MI_ERR_BAD_ARGS;
    int a, b, t, s;
    msm_plan_query(n, scalar_bits, &a, &b, &t, &s);
    *c = a;
    *nwin = b;
    *chunk = t;
    *small_path = s;
    return MI_OK;
}
int32_t mi_prof_enable(int32_t on) {
    return guarded([&](Context &cx) -> int32_t {
        cx.prof.level = on < 0 ? 0 : (on > 2 ? 2 : on);
        return MI_OK;
    });
}
int32_t mi_prof_read(const char *kernel, double *total_ms, uint64_t *launches, uint64_t *units) {
    if (!kernel || !total_ms || !launches || !units) return MI_ERR_BAD_ARGS;
    return guarded([&](Context &cx) -> int32_t {
        auto &v = cx.prof.recs[kernel];
        double ms = 0;
        uint64_t u = 0;
        for (auto &r : v) {
            MI_HIP(hipEventSynchronize(r.e1));
            float t = 0;
            MI_HIP(hipEventElapsedTime(&t, r.e0, r.e1));
            ms += t;
            u += r.units;
            cx.prof.pool.push_back(r.e0);
            cx.prof.pool.push_back(r.e1);
        }
        *total_ms = ms;
        *launches = v.size();
        *units = u;
        v.clear();
        return MI_OK;
    });
}

}  // extern "C"
