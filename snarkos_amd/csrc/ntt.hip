// Radix-2 NTT over the BLS12-377 scalar field Fr on gfx950: Stockham autosort, radix-2^K register
// passes (K <= 4), natural order in and out, ping-pong between the caller's buffer and a workspace.
//
// Replaces snarkvm_algorithms::fft::EvaluationDomain::{fft,ifft,coset_fft,coset_ifft}_in_place
// (snarkVM 1.0.0 @ dea322b, un-vendored): group_gen = ROOT^(2^(47-log_n)) with
// ROOT = 22^((r-1)/2^47); the inverse multiplies by size_inv; the coset shifts by the multiplicative
// generator 22 (forward: a[i] *= g^i before; inverse: a[i] *= g^-i after).
//
// Pass with Ns = 2^log_ns, R = 2^K, lane j of N/R:  k = j mod Ns,
//   x[r]   = in[j + r*N/R] * w_N^(k * r * N/(Ns*R))          (coalesced loads, stride N/R)
//   X      = DFT_R(x)                                        (K in-register radix-2 stages)
//   out[(j-k)*R + k + r*Ns] = X[r]
// Twiddles come from one table TW[e] = w_N^e, e < N/2, cached per log_n in HBM (w^(e+N/2) = -w^e).
#include <stdlib.h>
#include <string.h>
#include "api.hpp"

namespace mi {

__device__ __forceinline__ Fr ntt_load(const Fr *p) {
    const uint4 *q = reinterpret_cast<const uint4 *>(p);
    uint4 a = q[0], b = q[1];
    Fr r;
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
    r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
    return r;
}
__device__ __forceinline__ void ntt_store(Fr *p, const Fr &v) {
    uint4 *q = reinterpret_cast<uint4 *>(p);
    q[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    q[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
}

// w^e for e in [0, N): table holds e < N/2
__device__ __forceinline__ Fr ntt_tw(const Fr *__restrict__ tw, size_t e, size_t half) {
    if (e < half) return ntt_load(tw + e);
    return neg(ntt_load(tw + (e - half)));
}

// Two-level form for large domains: w^e = tw[e & 2047] * tw_hi[e >> 11] (tw_hi[i] = w^(2048 i), a few KB that stay in
// cache), so a pass does not gather 32-byte twiddles all over a table as large as the data itself.
struct TwRef {
    const Fr *tw;     // w^e, e < N/2
    const Fr *tw_hi;  // w^(2048 i), i < N/4096, or nullptr: single-level
};
__device__ __forceinline__ Fr ntt_tw2(const TwRef &t, size_t e, size_t half) {
    if (!t.tw_hi) return ntt_tw(t.tw, e, half);
    const bool flip = e >= half;
    if (flip) e -= half;
    const size_t lo = e & 2047, hi = e >> 11;
    Fr v = ntt_load(t.tw_hi + hi);
    if (lo) v = mul(v, ntt_load(t.tw + lo));
    return flip ? neg(v) : v;
}
__global__ void __launch_bounds__(256) ntt_table_hi_kernel(const Fr *__restrict__ tw, Fr *__restrict__ tw_hi, size_t count) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) ntt_store(tw_hi + i, ntt_load(tw + (i << 11)));
}

// tw[i] = w^i by square-and-multiply from the per-bit powers pw[b] = w^(2^b)
__global__ void __launch_bounds__(256) ntt_table_kernel(Fr *tw, size_t count, const Fr *__restrict__ pw, int bits) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    Fr acc = Fr::one();
    for (int b = 0; b < bits; b++)
        if ((i >> b) & 1) acc = mul(acc, ntt_load(pw + b));
    ntt_store(tw + i, acc);
}

template <int K>
__global__ void __launch_bounds__(256) ntt_pass_kernel(const Fr *__restrict__ in, Fr *__restrict__ out,
                                                       const Fr *__restrict__ tw, uint32_t log_n, uint32_t log_ns,
                                                       int inverse, int scale_out, Fr scale) {
    constexpr int R = 1 << K;
    const size_t N = (size_t)1 << log_n, half = N >> 1, T = N >> K;
    const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= T) return;
    const size_t Ns = (size_t)1 << log_ns;
    const size_t k = j & (Ns - 1);
    Fr x[R];
#pragma unroll
    for (int r = 0; r < R; r++) x[r] = ntt_load(in + j + (size_t)r * T);
    if (k != 0) {
        const size_t stride = N >> (log_ns + K);
#pragma unroll
        for (int r = 1; r < R; r++) {
            size_t e = k * r * stride;  // < N
            if (inverse) e = (N - e) & (N - 1);
            x[r] = mul(x[r], ntt_tw(tw, e, half));
        }
    }
    // in-register DIF: after the stages x[i] holds X[bitrev_K(i)]
#pragma unroll
    for (int s = K - 1; s >= 0; s--) {
        const int h = 1 << s;
#pragma unroll
        for (int blk = 0; blk < R; blk += 2 * h) {
#pragma unroll
            for (int t = 0; t < h; t++) {
                Fr u = x[blk + t], v = x[blk + t + h];
                x[blk + t] = add(u, v);
                Fr d = sub(u, v);
                if (t != 0) {
                    // w_{2h}^t = w_N^(t * N/(2h));  inverse: w^-e = -w^(N/2 - e)
                    size_t e = (size_t)t * (N >> (s + 1));
                    Fr w = inverse ? neg(ntt_load(tw + (half - e))) : ntt_load(tw + e);
                    d = mul(d, w);
                }
                x[blk + t + h] = d;
            }
        }
    }
    const size_t base = (j - k) * R + k;
#pragma unroll
    for (int i = 0; i < R; i++) {
        int rev = 0;
#pragma unroll
        for (int b = 0; b < K; b++) rev |= ((i >> b) & 1) << (K - 1 - b);
        Fr v = x[i];
        if (scale_out) v = mul(v, scale);
        ntt_store(out + base + (size_t)rev * Ns, v);
    }
}

// ---------------------------------------------------------------- LDS-tiled Stockham pass (radix 2^P, 4 <= P <= 9)
// One workgroup (128 lanes, 8 elements each) owns a tile of C = 1024 / 2^P consecutive columns j of the global
// Stockham step of radix R = 2^P: it loads the R x C tile with C contiguous elements per row (coalesced), applies the
// inter-pass twiddles, runs the R-point transform of every column entirely in LDS as 2-3 radix-8/4 Stockham rounds
// (each lane does one radix-8 or two radix-4 butterflies per round in registers, rounds exchange through LDS), and
// stores the tile.  A 2^22 transform is 4 such trips through HBM (6+6+5+5 bits) instead of 6 register-radix-16 trips.
// LDS image: column c at c * (R + 1) elements (one element of padding: conflict-free for both access directions).
template <int K>
__device__ __forceinline__ void ntt_dft_regs(Fr *x, const Fr *__restrict__ tw, size_t N, size_t half, int inverse) {
    // in-register DIF of 2^K values: afterwards x[i] holds X[bitrev_K(i)]
    constexpr int R = 1 << K;
#pragma unroll
    for (int s = K - 1; s >= 0; s--) {
        const int h = 1 << s;
#pragma unroll
        for (int blk = 0; blk < R; blk += 2 * h) {
#pragma unroll
            for (int t = 0; t < h; t++) {
                Fr u = x[blk + t], v = x[blk + t + h];
                x[blk + t] = add(u, v);
                Fr d = sub(u, v);
                if (t != 0) {
                    size_t e = (size_t)t * (N >> (s + 1));
                    Fr w = inverse ? neg(ntt_load(tw + (half - e))) : ntt_load(tw + e);
                    d = mul(d, w);
                }
                x[blk + t + h] = d;
            }
        }
    }
}
template <int K>
__device__ __forceinline__ int ntt_bitrev(int i) {
    int rev = 0;
#pragma unroll
    for (int b = 0; b < K; b++) rev |= ((i >> b) & 1) << (K - 1 - b);
    return rev;
}
__device__ __forceinline__ Fr lds_ld(const Fr *p) {
    const uint4 *q = reinterpret_cast<const uint4 *>(p);
    uint4 a = q[0], b = q[1];
    Fr r;
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
    r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
    return r;
}
__device__ __forceinline__ void lds_st(Fr *p, const Fr &v) {
    uint4 *q = reinterpret_cast<uint4 *>(p);
    q[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    q[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
}
// one in-LDS Stockham round of radix 2^PR on a column of R = 2^P points (sub-transform size done so far: 2^lns)
template <int P, int PR>
__device__ __forceinline__ void ntt_lds_round(Fr *colp, int l, int lns, const TwRef &tr, size_t N, size_t half,
                                              uint32_t log_n, int inverse) {
    const Fr *tw = tr.tw;
    constexpr int R = 1 << P, RP = 1 << PR, LPC = R / 8, G = 8 / RP, NG = R / RP;
    Fr x[8];
    const int ns = 1 << lns;
#pragma unroll
    for (int gi = 0; gi < G; gi++) {
        const int jj = gi * LPC + l, kk = jj & (ns - 1);
#pragma unroll
        for (int r = 0; r < RP; r++) {
            Fr v = lds_ld(colp + jj + r * NG);
            if (r != 0 && kk != 0) {
                // w_R^(kk * r * R / (ns * RP)) = w_N^(... * N / R)
                size_t e = ((size_t)(kk * r) << (P - lns - PR)) << (log_n - P);
                if (inverse) e = (N - e) & (N - 1);
                v = mul(v, ntt_tw2(tr, e, half));
            }
            x[gi * RP + r] = v;
        }
        ntt_dft_regs<PR>(x + gi * RP, tw, N, half, inverse);
    }
    __syncthreads();
#pragma unroll
    for (int gi = 0; gi < G; gi++) {
        const int jj = gi * LPC + l, kk = jj & (ns - 1);
#pragma unroll
        for (int i = 0; i < RP; i++) lds_st(colp + (jj - kk) * RP + kk + ntt_bitrev<PR>(i) * ns, x[gi * RP + i]);
    }
    __syncthreads();
}
template <int P>
__global__ void __launch_bounds__(128) ntt_lds_pass_kernel(const Fr *__restrict__ in, Fr *__restrict__ out,
                                                           const Fr *__restrict__ tw, const Fr *__restrict__ tw_hi,
                                                           uint32_t log_n, uint32_t log_ns, int inverse, int scale_out,
                                                           Fr scale) {
    const TwRef tr{tw, tw_hi};
    constexpr int R = 1 << P, C = 1024 / R, LPC = R / 8;
    extern __shared__ uint4 ntt_lds4[];
    Fr *tile = reinterpret_cast<Fr *>(ntt_lds4);
    const size_t N = (size_t)1 << log_n, half = N >> 1, T = N >> P, Ns = (size_t)1 << log_ns;
    const size_t j0 = (size_t)blockIdx.x * C;
    const int t = threadIdx.x;
    // ---- load: consecutive lanes take consecutive columns (C contiguous elements per row)
    const size_t stride = N >> (log_ns + P);
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int e = i * 128 + t, r = e / C, c = e % C;
        const size_t j = j0 + c, k = j & (Ns - 1);
        Fr v = ntt_load(in + j + (size_t)r * T);
        if (r != 0 && k != 0) {
            size_t ex = k * r * stride;
            if (inverse) ex = (N - ex) & (N - 1);
            v = mul(v, ntt_tw2(tr, ex, half));
        }
        lds_st(tile + c * (R + 1) + r, v);
    }
    __syncthreads();
    // ---- the R-point transform of each column, in LDS
    Fr *colp = tile + (t / LPC) * (R + 1);
    const int l = t % LPC;
    if (P == 4) {
        ntt_lds_round<P, 2>(colp, l, 0, tr, N, half, log_n, inverse);
        ntt_lds_round<P, 2>(colp, l, 2, tr, N, half, log_n, inverse);
    } else if (P == 5) {
        ntt_lds_round<P, 3>(colp, l, 0, tr, N, half, log_n, inverse);
        ntt_lds_round<P, 2>(colp, l, 3, tr, N, half, log_n, inverse);
    } else if (P == 6) {
        ntt_lds_round<P, 3>(colp, l, 0, tr, N, half, log_n, inverse);
        ntt_lds_round<P, 3>(colp, l, 3, tr, N, half, log_n, inverse);
    } else if (P == 7) {
        ntt_lds_round<P, 3>(colp, l, 0, tr, N, half, log_n, inverse);
        ntt_lds_round<P, 2>(colp, l, 3, tr, N, half, log_n, inverse);
        ntt_lds_round<P, 2>(colp, l, 5, tr, N, half, log_n, inverse);
    } else if (P == 8) {
        ntt_lds_round<P, 3>(colp, l, 0, tr, N, half, log_n, inverse);
        ntt_lds_round<P, 3>(colp, l, 3, tr, N, half, log_n, inverse);
        ntt_lds_round<P, 2>(colp, l, 6, tr, N, half, log_n, inverse);
    } else {
        ntt_lds_round<P, 3>(colp, l, 0, tr, N, half, log_n, inverse);
        ntt_lds_round<P, 3>(colp, l, 3, tr, N, half, log_n, inverse);
        ntt_lds_round<P, 3>(colp, l, 6, tr, N, half, log_n, inverse);
    }
    // ---- store: out[(j - k) * R + k + r * Ns]
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int e = i * 128 + t;
        int r, c;
        if (log_ns == 0) { c = e / R; r = e % R; }  // first pass: a column's outputs are contiguous
        else { r = e / C; c = e % C; }              // later passes: consecutive columns are contiguous
        const size_t j = j0 + c, k = j & (Ns - 1);
        Fr v = lds_ld(tile + c * (R + 1) + r);
        if (scale_out) v = mul(v, scale);
        ntt_store(out + (j - k) * R + k + (size_t)r * Ns, v);
    }
}

// a[i] *= g^(+-i) * scale, with g^i = glo[i & 1023] * ghi[i >> 10 & 1023] * ghh[i >> 20 ...]
__global__ void __launch_bounds__(256) ntt_coset_kernel(Fr *a, size_t n, const Fr *__restrict__ gtab, Fr scale) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr g = mul(ntt_load(gtab + (i & 1023)), ntt_load(gtab + 1024 + ((i >> 10) & 1023)));
    if (i >> 20) g = mul(g, ntt_load(gtab + 2048 + ((i >> 20) & 1023)));
    ntt_store(a + i, mul(mul(ntt_load(a + i), g), scale));
}

struct NttTables {  // per device (Context::mod)
    DevBuf tw[48];
    DevBuf tw_hi[48];
    DevBuf coset_fwd, coset_inv;  // 3 x 1024 entries each: g^i, g^(1024 i), g^(2^20 i)
    bool coset_ready = false;
    ~NttTables() {
        for (auto &b : tw) b.release();
        for (auto &b : tw_hi) b.release();
        coset_fwd.release();
        coset_inv.release();
    }
};
static NttTables &ntt_tables(Context &cx) { return cx.mod<NttTables>("ntt.tables"); }

static Fr host_fr_from_limbs(const uint64_t *l) {
    Fr r;
    for (int i = 0; i < 4; i++) { r.l[2 * i] = (uint32_t)l[i]; r.l[2 * i + 1] = (uint32_t)(l[i] >> 32); }
    return r;
}
Fr host_omega(uint32_t log_n) {
    Fr w = host_fr_from_limbs(MI_FR_ROOT_OF_UNITY);
    for (uint32_t i = log_n; i < MI_FR_TWO_ADICITY; i++) w = sqr(w);
    return w;
}

static const Fr *ntt_get_table(Context &cx, uint32_t log_n, hipStream_t st) {
    NttTables &t = ntt_tables(cx);
    if (log_n == 0) return nullptr;
    size_t count = (size_t)1 << (log_n - 1);
    if (t.tw[log_n].p) return (const Fr *)t.tw[log_n].p;
    Fr *d = (Fr *)t.tw[log_n].get(sizeof(Fr) * count);
    Fr pw[48];
    pw[0] = host_omega(log_n);
    for (int b = 1; b < 48; b++) pw[b] = sqr(pw[b - 1]);
    Fr *dpw = (Fr *)cx.buf("ntt.pw", sizeof(pw));
    MI_HIP(hipMemcpyAsync(dpw, pw, sizeof(pw), hipMemcpyHostToDevice, st));
    MI_HIP(hipStreamSynchronize(st));  // pw is a stack array
    hipLaunchKernelGGL(ntt_table_kernel, dim3(cdiv(count, 256)), dim3(256), 0, st, d, count, dpw, (int)log_n - 1);
    if (log_n >= 25) {  // measured: at 2^22 the extra product costs more than the gathers (0.90 vs 0.80 ms); only for
                        // domains whose single-level table (>= 0.5 GB) would thrash: compact table of w^(2048 i)
        const size_t chi = count >> 11;
        Fr *dh = (Fr *)t.tw_hi[log_n].get(sizeof(Fr) * chi);
        hipLaunchKernelGGL(ntt_table_hi_kernel, dim3(cdiv(chi, 256)), dim3(256), 0, st, d, dh, chi);
    }
    MI_HIP(hipGetLastError());
    MI_HIP(hipStreamSynchronize(st));  // one-time: later calls may come in on another stream
    return d;
}
static void ntt_get_coset_tables(Context &cx, hipStream_t st) {
    NttTables &t = ntt_tables(cx);
    if (t.coset_ready) return;
    std::vector<Fr> h(3 * 1024);
    for (int dir = 0; dir < 2; dir++) {
        Fr g = host_fr_from_limbs(dir == 0 ? MI_FR_GENERATOR : MI_FR_GENERATOR_INV);
        for (int lvl = 0; lvl < 3; lvl++) {
            Fr acc = Fr::one();
            for (int i = 0; i < 1024; i++) { h[lvl * 1024 + i] = acc; acc = mul(acc, g); }
            g = acc;  // g^1024
        }
        DevBuf &b = dir == 0 ? t.coset_fwd : t.coset_inv;
        void *d = b.get(sizeof(Fr) * h.size());
        MI_HIP(hipMemcpyAsync(d, h.data(), sizeof(Fr) * h.size(), hipMemcpyHostToDevice, st));
        MI_HIP(hipStreamSynchronize(st));
    }
    t.coset_ready = true;
}

template <int K>
static void ntt_launch_pass(const Fr *in, Fr *out, const Fr *tw, uint32_t log_n, uint32_t log_ns, bool inverse,
                            bool scale_out, const Fr &scale, hipStream_t st) {
    size_t T = (size_t)1 << (log_n - K);
    hipLaunchKernelGGL(ntt_pass_kernel<K>, dim3(cdiv(T, 256)), dim3(256), 0, st, in, out, tw, log_n, log_ns,
                       inverse ? 1 : 0, scale_out ? 1 : 0, scale);
}

template <int P>
static void ntt_launch_lds_pass(Context &cx, const Fr *in, Fr *out, const Fr *tw, const Fr *tw_hi, uint32_t log_n, uint32_t log_ns,
                                bool inverse, bool scale_out, const Fr &scale, hipStream_t st) {
    constexpr int R = 1 << P, C = 1024 / R;
    const size_t lds_bytes = sizeof(Fr) * (size_t)C * (R + 1);
    cx.raise_lds_limit(reinterpret_cast<const void *>(&ntt_lds_pass_kernel<P>), (int)lds_bytes);
    const size_t T = (size_t)1 << (log_n - P);
    hipLaunchKernelGGL(ntt_lds_pass_kernel<P>, dim3((unsigned)(T / C)), dim3(128), lds_bytes, st, in, out, tw, tw_hi, log_n,
                       log_ns, inverse ? 1 : 0, scale_out ? 1 : 0, scale);
}

// in place on d_data (device pointer, 2^log_n elements)
void ntt_device(Context &cx, Fr *d_data, uint32_t log_n, bool inverse, bool coset, hipStream_t st) {
    stats().ntt_calls++;
    stats().ntt_elements += (uint64_t)1 << log_n;
    const size_t n = (size_t)1 << log_n;
    if (coset) ntt_get_coset_tables(cx, st);
    if (coset && !inverse)
        hipLaunchKernelGGL(ntt_coset_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, d_data, n, (const Fr *)ntt_tables(cx).coset_fwd.p, Fr::one());
    if (log_n > 0) {
        const Fr *tw = ntt_get_table(cx, log_n, st);
        // The Stockham twin.  This call may return with its passes still running on the caller's stream (the
        // `_device` entry point), and the next call may come on another stream: every call takes a buffer that no
        // in-flight transform is using (event-guarded pool), never a name-keyed workspace.
        ScratchPool::Entry *scr = cx.scratch.acquire(sizeof(Fr) * n, st);
        Fr *tmp = (Fr *)scr->buf.p;
        // split log_n into passes of at most 4 bits, an even number of them when log_n >= 2
        int npass = (int)(log_n + 3) / 4;
        if ((npass & 1) && log_n >= 2) npass++;
        Fr n_inv = Fr::one();
        if (inverse) {
            Fr nn = Fr::zero();
            nn.l[0] = (uint32_t)n; nn.l[1] = (uint32_t)((uint64_t)n >> 32);
            n_inv = inv(to_mont(nn));
        }
        const bool fold_scale = inverse && !coset;
        Fr *src = d_data, *dst = tmp;
        uint32_t log_ns = 0;
        // measured on MI355X: the LDS-tiled passes win up to 2^20 (2^16: 0.079 vs 0.143 ms, 2^20: 0.184 vs 0.224 ms); at 2^22
        // both are bound by the same integer work and the six register-radix-16 passes are marginally faster (0.78 vs 0.80)
        const bool force_lds = getenv("MI_SNARK_NTT_LDS") != nullptr;
        const bool use_lds = log_n >= 11 && (log_n <= 20 || force_lds) && !getenv("MI_SNARK_NTT_REGS");
        const Fr *tw_hi = (const Fr *)ntt_tables(cx).tw_hi[log_n].p;  // null below 2^25
        if (use_lds) {
            // an even number of LDS-tiled passes of 4..9 bits each
            int np = 2 * (int)((log_n + 15) / 16);  // <= 8 bits per pass
            if (const char *e = getenv("MI_SNARK_NTT_NP")) {  // experiment: any number of passes of <= 9 bits (an odd number ends with a copy)
                const int x = atoi(e);
                if (x >= 1 && (int)(log_n + x - 1) / x <= 9 && (int)log_n / x >= 4) np = x;
            }
            for (int p = 0; p < np; p++) {
                int P = (int)(log_n - log_ns + (np - p) - 1) / (np - p);
                bool last = p == np - 1;
                cx.prof.begin("ntt_pass", st, n);
                switch (P) {
                case 4: ntt_launch_lds_pass<4>(cx, src, dst, tw, tw_hi, log_n, log_ns, inverse, last && fold_scale, n_inv, st); break;
                case 5: ntt_launch_lds_pass<5>(cx, src, dst, tw, tw_hi, log_n, log_ns, inverse, last && fold_scale, n_inv, st); break;
                case 6: ntt_launch_lds_pass<6>(cx, src, dst, tw, tw_hi, log_n, log_ns, inverse, last && fold_scale, n_inv, st); break;
                case 7: ntt_launch_lds_pass<7>(cx, src, dst, tw, tw_hi, log_n, log_ns, inverse, last && fold_scale, n_inv, st); break;
                case 8: ntt_launch_lds_pass<8>(cx, src, dst, tw, tw_hi, log_n, log_ns, inverse, last && fold_scale, n_inv, st); break;
                default: ntt_launch_lds_pass<9>(cx, src, dst, tw, tw_hi, log_n, log_ns, inverse, last && fold_scale, n_inv, st); break;
                }
                cx.prof.end("ntt_pass", st);
                log_ns += P;
                Fr *t = src; src = dst; dst = t;
            }
            npass = 0;
        }
        for (int p = 0; p < npass; p++) {
            int K = (int)(log_n - log_ns + (npass - p) - 1) / (npass - p);
            bool last = p == npass - 1;
            cx.prof.begin("ntt_pass", st, n);
            switch (K) {
            case 1: ntt_launch_pass<1>(src, dst, tw, log_n, log_ns, inverse, last && fold_scale, n_inv, st); break;
            case 2: ntt_launch_pass<2>(src, dst, tw, log_n, log_ns, inverse, last && fold_scale, n_inv, st); break;
            case 3: ntt_launch_pass<3>(src, dst, tw, log_n, log_ns, inverse, last && fold_scale, n_inv, st); break;
            default: ntt_launch_pass<4>(src, dst, tw, log_n, log_ns, inverse, last && fold_scale, n_inv, st); break;
            }
            cx.prof.end("ntt_pass", st);
            log_ns += K;
            Fr *t = src; src = dst; dst = t;
        }
        if (src != d_data) MI_HIP(hipMemcpyAsync(d_data, src, sizeof(Fr) * n, hipMemcpyDeviceToDevice, st));
        if (inverse && coset)
            hipLaunchKernelGGL(ntt_coset_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, d_data, n, (const Fr *)ntt_tables(cx).coset_inv.p, n_inv);
        ScratchPool::release(scr, st);
    } else if (inverse && coset) {
        // n = 1: g^0 * 1 = identity
    }
    MI_HIP(hipGetLastError());
}

}  // namespace mi
