// Variable-base MSM over BLS12-377 G1 on gfx950: signed-digit Pippenger, sort-by-bucket, load-balanced bucket
// accumulation in register-resident XYZZ accumulators, two reductions (by size), host Horner fold.
//
// Replaces snarkvm_algorithms::msm::VariableBase::msm (snarkVM 1.0.0 @ dea322b, un-vendored; SURVEY.md
// 8a row A3): out = sum_i scalars[i] * bases[i], scalars canonical BigInteger256, bases Montgomery
// affine.  The CPU reference walks c-bit windows with batched-affine bucket adds on rayon threads; this
// is a different algorithm with the same (unique) group result.
//
// Common front (one stream, no host synchronisation until the result):
//   msm_digits     lane per scalar: signed c-bit digits -> digits[w][i]; bucket histogram whose atomic also returns
//                  the entry's rank in its bucket (lanes of a wave sharing a bucket take one atomic together)
//   scan / plan    bucket offsets; buckets over T entries are cut into chunks (offsets of those too)
//   msm_scatter    lane per (w, i): sorted[offset[bucket] + rank] = (i << 1) | negate        (no atomics)
//   msm_accumulate lane per CHUNK: gather its points (96 B each, random), mixed-add into XYZZ
// Large path (n > 2^17): chunk map + chunks ordered by length; multi-chunk buckets summed (msm_combine_*);
//   msm_segments (running sums per L-bucket segment + (gL) fix-up); point_tree_sum per window; the host folds the
//   nwin window sums (<= 256 doublings) and normalises to z = 1.
// Small path (n <= 2^17, latency bound; takes optional split points: up to four MSMs as one launch sequence, each
//   further problem as another set of windows): ONE planning workgroup instead of scans / memsets / chunk kernels, accumulate finds its
//   bucket by bisection, msm_bit_sums_* forms per-(window, bit) subset sums straight from the chunk partials into
//   pinned host memory, and the host folds over bit positions (halves of each problem on pool threads).
// HBM traffic per (point, scalar) pair: 32 B scalar read once, 96 B point read once per window from
// HBM/Infinity Cache (the 100 MB base table of a 2^20 MSM stays cache resident), 4 B digit + 4 B rank + 4 B index
// written and read once per window.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <map>
#include <thread>
#include "api.hpp"
#include "hostfold.hpp"
#include "field28.hpp"

namespace mi {

struct MsmPlan {
    int c, nwin;
    uint32_t B;  // buckets per window = 2^(c-1)
};

// largest n that takes the small (latency-bound) path: 2^17 since round 3 (with the gather stage ahead of the per-bit sums
// the small path wins there: 1.51 against 1.68 ms; at 2^18 it loses, 2.77 against 1.76: tools/msm_small_limit.py);
// MI_SNARK_MSM_SMALL_LOG overrides for experiments
// chunk length on the small path: longer chunks = fewer partials for the bit-sum reduction (less GPU work per MSM),
// shorter = shorter accumulation chains (latency); MI_SNARK_MSM_SMALL_T overrides for experiments
static uint32_t msm_small_chunk() {
    static uint32_t t = 0;
    if (!t) {
        t = 16;
        if (const char *e = getenv("MI_SNARK_MSM_SMALL_T")) {
            const int x = atoi(e);
            if (x >= 4 && x <= 256) t = (uint32_t)x;
        }
    }
    return t;
}
size_t msm_small_limit() {
    static size_t lim = 0;
    if (!lim) {
        int lg = 17;
        if (const char *e = getenv("MI_SNARK_MSM_SMALL_LOG")) {
            const int x = atoi(e);
            if (x >= 6 && x <= 18) lg = x;
        }
        lim = (size_t)1 << lg;
    }
    return lim;
}
// Window width of the large path relative to floor(log2 n) - 4.  Measured (252-bit scalars): one bit wider pays up to
// 2^18 (2^17: 3.16 -> 2.70 ms, 2^18: 3.50 -> 2.87 ms: fewer windows to accumulate), not at 2^19 (4.16 vs 4.35 ms: the
// bucket reduction grows).  MI_SNARK_MSM_C_BIAS overrides for experiments.
static int msm_large_c_bias(int lg) {
    static int b = 99;
    if (b == 99) {
        b = 98;
        if (const char *e = getenv("MI_SNARK_MSM_C_BIAS")) {
            const int x = atoi(e);
            if (x >= -4 && x <= 2) b = x;
        }
    }
    return b == 98 ? (lg <= 18 ? 1 : 0) : b;
}
static MsmPlan msm_plan(size_t n, int scalar_bits, int force_c) {
    int lg = 0;
    while (((size_t)1 << (lg + 1)) <= n) lg++;
    int c = lg - 4;
    if (c < 6) c = 6;
    if (c > 16) c = 16;
    if (force_c >= 6 && force_c <= 16) c = force_c;
    else if (n <= msm_small_limit()) {
        // one bit narrower than floor(log2 n) - 4 (measured: 2^15 1.49 -> 1.39 ms; the verifier's 13 409-point pair:
        // accumulate stage 1.30 -> 1.21 ms): the per-bit reduction costs c / 2 additions per chunk
        static const int sb = getenv("MI_SNARK_MSM_SMALL_C_BIAS") ? atoi(getenv("MI_SNARK_MSM_SMALL_C_BIAS")) : -1;
        c += sb;
        if (c < 6) c = 6;
        if (c > 16) c = 16;
    } else {
        c += msm_large_c_bias(lg);
        if (c < 6) c = 6;
        if (c > 16) c = 16;
    }
    MsmPlan p;
    p.c = c;
    const int bits = scalar_bits > 128 ? 254 : 129;  // one spare bit so the top signed digit never overflows
    p.nwin = (bits + c - 1) / c;
    p.B = 1u << (c - 1);
    return p;
}

// the plan an MSM of n points would run on (diagnostics: bench.py's work counts)
void msm_plan_query(size_t n, int scalar_bits, int *c, int *nwin, int *chunk, int *small) {
    const MsmPlan p = msm_plan(n, scalar_bits, 0);
    *c = p.c;
    *nwin = p.nwin;
    *small = n <= msm_small_limit() ? 1 : 0;
    uint32_t T = (uint32_t)(2 * ((n + p.B - 1) / p.B));
    if (T < 16) T = 16;
    *chunk = *small ? (int)msm_small_chunk() : (int)T;
}

// ---------------------------------------------------------------- signed digit of one window, without the windows below
// carry_w = [ k mod 2^(c w)  >  H_w ],  H_w = sum_{j < w} 2^(c j + c - 1)   (the recoding d = raw + carry > 2^(c-1) ? raw +
// carry - 2^c : raw + carry, by induction on w): one comparison instead of the carry chain through the lower windows.
__device__ __forceinline__ int32_t msm_digit_at(const uint32_t (&s)[8], int c, int w, const uint32_t *__restrict__ hw) {
    // carry into window w (uniform w: every branch below is wave-uniform)
    uint32_t carry = 0;
    const int nbits = c * w;
    if (w > 0) {
        bool gt = false, decided = false;
#pragma unroll
        for (int j = 7; j >= 0; j--) {
            uint32_t limb = 0;
            if (32 * j < nbits) limb = 32 * (j + 1) <= nbits ? s[j] : (s[j] & ((1u << (nbits - 32 * j)) - 1u));
            const uint32_t h = hw[j];
            if (!decided && limb != h) {
                gt = limb > h;
                decided = true;
            }
        }
        carry = gt ? 1u : 0u;
    }
    const int li = nbits >> 5, off = nbits & 31;
    uint32_t lo = 0, hi = 0;
    switch (li) {
    case 0: lo = s[0]; hi = s[1]; break;
    case 1: lo = s[1]; hi = s[2]; break;
    case 2: lo = s[2]; hi = s[3]; break;
    case 3: lo = s[3]; hi = s[4]; break;
    case 4: lo = s[4]; hi = s[5]; break;
    case 5: lo = s[5]; hi = s[6]; break;
    case 6: lo = s[6]; hi = s[7]; break;
    case 7: lo = s[7]; hi = 0; break;
    default: break;
    }
    const uint32_t mask = (1u << c) - 1u, B = 1u << (c - 1);
    const uint32_t raw = ((uint32_t)((((uint64_t)hi << 32) | lo) >> off) & mask) + carry;
    return raw > B ? (int32_t)raw - (int32_t)(1u << c) : (int32_t)raw;
}
// the limbs of H_w for a compile-time width: the bits at positions = C - 1 mod C, below C w (w uniform: scalar code)
template <int C>
__device__ __forceinline__ void msm_half_limbs(int w, uint32_t (&hw)[8]) {
    const int top = C * w;  // bits below this position
#pragma unroll
    for (int j = 0; j < 8; j++) {
        uint32_t pat = 0;  // compile-time constant per (C, j)
#pragma unroll
        for (int b = 0; b < 32; b++)
            if ((32 * j + b) % C == C - 1) pat |= 1u << b;
        const int room = top - 32 * j;
        hw[j] = room >= 32 ? pat : (room <= 0 ? 0u : (pat & ((1u << room) - 1u)));
    }
}

// ---------------------------------------------------------------- K1: signed digits + histogram
// One thread per (scalar, group of up to four windows): blockIdx.y is the group, a workgroup takes MSM_DIGITS_THREADS
// consecutive scalars.  (One thread per
// scalar walking its 32 windows was a chain of 32 returning global atomics: 56 us of the 1024-proof batch's latency for
// microseconds of work, and n x windows atomics at the memory-side rate.)  A thread takes its window's carry from one
// comparison (msm_digit_at), counts its digit in the workgroup's LDS histogram, which
// also numbers it among the workgroup's entries of that bucket; one global atomic per (workgroup, live bucket) then
// reserves the workgroup's range of ranks.  Buckets of several problems (split points inside the workgroup's range of
// scalars) sit side by side in LDS.  Window widths whose histogram does not fit (c > 11: never the verifier's passes) count
// in global memory directly, hot buckets of a wave through one atomic.
// windows per thread: a thread reads its scalar once for G windows (one thread per window re-read 32 bytes per window
// through L2: 1.4 % of the streamed rate) and walks them with a chain of G LDS atomics
template <int C>
constexpr int msm_digits_group() { return C <= 9 ? 4 : (C == 10 ? 2 : 1); }  // G x 2^(C-1) x MSM_MAX_PROBLEMS counters: 32 KB at most
template <int C, int BITS, int MSM_DIGITS_THREADS>
__global__ void __launch_bounds__(MSM_DIGITS_THREADS) msm_digits_kernel(const uint32_t *__restrict__ scalars, size_t n,
                                                                        int32_t *__restrict__ digits, uint32_t *__restrict__ ranks,
                                                                        uint32_t *__restrict__ counts_all, MsmSplits splits) {
    constexpr int NWIN = (BITS + C - 1) / C;
    constexpr uint32_t B = 1u << (C - 1);
    constexpr bool IN_LDS = C <= 11;
    constexpr int G = msm_digits_group<C>();
    __shared__ uint32_t hist[IN_LDS ? G * B * MSM_MAX_PROBLEMS : 1];
    const size_t first = (size_t)blockIdx.x * MSM_DIGITS_THREADS, i = first + threadIdx.x;
    const bool live_thread = i < n;
    const int w0 = (int)blockIdx.y * G;  // uniform: this workgroup's windows w0 .. w0 + G - 1 (below NWIN)
    const uint32_t p_first = splits.problem_of(first);
    const size_t last = first + MSM_DIGITS_THREADS - 1 < n ? first + MSM_DIGITS_THREADS - 1 : n - 1;
    const uint32_t n_prob = splits.problem_of(last) - p_first + 1;  // problems this workgroup's scalars belong to
    const uint32_t per_window = B * n_prob;
    if (IN_LDS) {
        for (uint32_t e = threadIdx.x; e < G * per_window; e += MSM_DIGITS_THREADS) hist[e] = 0;
        __syncthreads();
    }
    int32_t d[G];
    uint32_t local[G];
#pragma unroll
    for (int g = 0; g < G; g++) { d[g] = 0; local[g] = 0; }
    uint32_t prob = p_first;
    if (live_thread) {
        prob = splits.problem_of(i);
        const uint4 *sp = reinterpret_cast<const uint4 *>(scalars + i * 8);
        uint4 lo = sp[0], hi = sp[1];
        const uint32_t s[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
        for (int g = 0; g < G; g++) {
            const int w = w0 + g;
            if (w < NWIN) {  // uniform
                uint32_t hw[8];
                msm_half_limbs<C>(w, hw);
                d[g] = msm_digit_at(s, C, w, hw);
                digits[(size_t)w * n + i] = d[g];
            }
        }
    }
    // the histogram count also hands this entry its rank inside the bucket, so the scatter needs no atomics
    if (IN_LDS) {
        const uint32_t pslot = (prob - p_first) * B;
#pragma unroll
        for (int g = 0; g < G; g++) {
            const uint32_t mag = d[g] < 0 ? (uint32_t)(-d[g]) : (uint32_t)d[g];
            if (d[g] != 0) local[g] = atomicAdd(&hist[g * per_window + pslot + (mag - 1)], 1u);
        }
        __syncthreads();
        for (uint32_t e = threadIdx.x; e < G * per_window; e += MSM_DIGITS_THREADS) {
            const uint32_t cnt = hist[e];
            const uint32_t g = e / per_window, r = e - g * per_window;
            if (cnt) hist[e] = atomicAdd(&counts_all[(p_first + r / B) * (uint32_t)NWIN * B + (uint32_t)(w0 + (int)g) * B + r % B], cnt);
        }
        __syncthreads();
#pragma unroll
        for (int g = 0; g < G; g++) {
            const uint32_t mag = d[g] < 0 ? (uint32_t)(-d[g]) : (uint32_t)d[g];
            if (d[g] != 0) ranks[(size_t)(w0 + g) * n + i] = hist[g * per_window + pslot + (mag - 1)] + local[g];
        }
    } else {
        // (G = 1) Windows with few live buckets (the top window of a 253-bit scalar holds one or two digits; skewed
        // scalars anywhere) would serialise thousands of atomics on one address (measured: 1.7 ms of a 4.9 ms MSM at
        // 2^18): while the first live lane's bucket is shared by 8+ lanes of the wave, those lanes take ONE atomic and
        // number themselves by their position in the match mask.
        const int lane = threadIdx.x & 63;
        const int w = w0;
        const uint32_t mag = d[0] < 0 ? (uint32_t)(-d[0]) : (uint32_t)d[0];
        bool live = d[0] != 0;
        const uint32_t bkt = prob * (uint32_t)NWIN * B + (uint32_t)w * B + (mag - 1);
        uint32_t rank = 0;
#pragma unroll 1
        for (int it = 0; it < 4; it++) {
            const uint64_t am = __ballot(live);
            if (!am) break;
            const int leader = __ffsll((long long)am) - 1;
            const uint32_t fb = (uint32_t)__shfl((int)bkt, leader, 64);
            const bool mine = live && bkt == fb;
            const uint64_t mm = __ballot(mine);
            if (__popcll(mm) < 8) break;
            uint32_t base = 0;
            if (lane == leader) base = atomicAdd(&counts_all[fb], (uint32_t)__popcll(mm));
            base = (uint32_t)__shfl((int)base, leader, 64);
            if (mine) {
                rank = base + (uint32_t)__popcll(mm & ((1ull << lane) - 1ull));
                live = false;
            }
        }
        if (live) rank = atomicAdd(&counts_all[bkt], 1u);
        if (d[0] != 0) ranks[(size_t)w * n + i] = rank;
    }
}

// ---------------------------------------------------------------- K2: exclusive scan (3 small kernels)
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}
// each block scans a tile of 256*8 elements; out = exclusive prefix inside the tile; tile_sums[block] = total
__global__ void __launch_bounds__(256) scan_tiles_kernel(const uint32_t *__restrict__ in, uint32_t *__restrict__ out,
                                                         uint32_t *__restrict__ tile_sums, size_t n) {
    __shared__ uint32_t wsum[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    size_t base = ((size_t)blockIdx.x * 256 + threadIdx.x) * 8;
    uint32_t v[8], s = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        v[k] = base + k < n ? in[base + k] : 0u;
        s += v[k];
    }
    uint32_t incl = wave_incl_scan(s, lane);
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t woff = 0;
    for (int k = 0; k < wave; k++) woff += wsum[k];
    uint32_t run = woff + incl - s;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        if (base + k < n) out[base + k] = run;
        run += v[k];
    }
    if (threadIdx.x == 255) tile_sums[blockIdx.x] = woff + incl;
}
// single block: exclusive scan of up to 1024*K tile sums, in place
__global__ void __launch_bounds__(1024) scan_sums_kernel(uint32_t *sums, size_t m) {
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t carry_s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (size_t base = 0; base < m; base += 1024) {
        size_t i = base + threadIdx.x;
        uint32_t v = i < m ? sums[i] : 0u;
        uint32_t incl = wave_incl_scan(v, lane);
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        uint32_t woff = carry_s;
        for (int k = 0; k < wave; k++) woff += wsum[k];
        if (i < m) sums[i] = woff + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = woff + incl;
        __syncthreads();
    }
}
// offsets[i] += tile_sums[tile]; also write the grand total at offsets[n] and copy to cursor
__global__ void __launch_bounds__(256) scan_finish_kernel(uint32_t *__restrict__ offsets, uint32_t *__restrict__ cursor,
                                                          const uint32_t *__restrict__ tile_sums,
                                                          const uint32_t *__restrict__ counts, size_t n) {
    size_t base = ((size_t)blockIdx.x * 256 + threadIdx.x) * 8;
    uint32_t add = tile_sums[blockIdx.x];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        size_t i = base + k;
        if (i < n) {
            uint32_t o = offsets[i] + add;
            offsets[i] = o;
            if (cursor) cursor[i] = o;
            if (i == n - 1) offsets[n] = o + counts[i];
        }
    }
}

// ---------------------------------------------------------------- K3: scatter indices into bucket order
__global__ void __launch_bounds__(256) msm_scatter_kernel(const int32_t *__restrict__ digits,
                                                          const uint32_t *__restrict__ ranks, size_t n, int nwin, uint32_t B,
                                                          const uint32_t *__restrict__ offsets, uint32_t *__restrict__ sorted,
                                                          MsmSplits sp) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * (size_t)nwin) return;
    int32_t d = digits[t];
    if (d == 0) return;
    size_t w = t / n, i = t - w * n;
    w += (size_t)nwin * sp.problem_of(i);  // every problem has its own windows
    uint32_t mag = d < 0 ? (uint32_t)(-d) : (uint32_t)d;
    uint32_t pos = offsets[w * B + (mag - 1)] + ranks[t];
    sorted[pos] = ((uint32_t)i << 1) | (d < 0 ? 1u : 0u);
}

// out-of-line group addition for the cold reduction kernels
// (group law inlined, field products out of line and register-passed: curve.hpp fm<false>)
__device__ __forceinline__ G1Xyzz xyzz_add_ni(const G1Xyzz &a, const G1Xyzz &b) { return add<false>(a, b); }
__device__ __forceinline__ G1Xyzz xyzz_shfl_down(const G1Xyzz &a, int delta) {
    G1Xyzz r;
    const Fq *src[4] = {&a.x, &a.y, &a.zz, &a.zzz};
    Fq *dst[4] = {&r.x, &r.y, &r.zz, &r.zzz};
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
        for (int k = 0; k < 12; k++) dst[q]->l[k] = __shfl_down(src[q]->l[k], delta, 64);
    return r;
}

// ---------------------------------------------------------------- K4: bucket accumulation, load-balanced
// A bucket with more than T entries is cut into ceil(count / T) chunks, one lane each; the top window of a
// 253-bit scalar has few distinct digits, and a skewed scalar distribution can put most points into a handful
// of buckets, so lane-per-bucket would leave one lane walking thousands of points.
// Small MSMs: ONE workgroup turns the bucket histogram into everything the later kernels index by -- bucket offsets
// (exclusive scan of counts), chunk offsets (exclusive scan of ceil(count / T)) -- and clears the histogram for the
// next MSM in this workspace.  Replaces 2 memsets + 8 launches of the general path: a 10 k-point MSM is bound by
// launch latency, not by work.
// sc_off (may be null): offsets of the SHORTENED chunk list, ceil(chunks / gather) entries per bucket (msm_gather_chunks_kernel)
__global__ void __launch_bounds__(1024) msm_plan_small_kernel(uint32_t *__restrict__ counts, size_t nb, uint32_t T,
                                                              uint32_t *__restrict__ offsets, uint32_t *__restrict__ chunk_off,
                                                              uint32_t gather, uint32_t *__restrict__ sc_off) {
    __shared__ uint32_t w1[16], w2[16], w3[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t per = (nb + 1023) / 1024;
    size_t lo = (size_t)threadIdx.x * per, hi = lo + per;
    if (lo > nb) lo = nb;
    if (hi > nb) hi = nb;
    uint32_t s1 = 0, s2 = 0, s3 = 0;
    for (size_t b = lo; b < hi; b++) {
        const uint32_t c = counts[b], ch = (c + T - 1) / T;
        s1 += c;
        s2 += ch;
        s3 += (ch + gather - 1) / gather;
    }
    const uint32_t i1 = wave_incl_scan(s1, lane), i2 = wave_incl_scan(s2, lane), i3 = wave_incl_scan(s3, lane);
    if (lane == 63) {
        w1[wave] = i1;
        w2[wave] = i2;
        w3[wave] = i3;
    }
    __syncthreads();
    uint32_t o1 = i1 - s1, o2 = i2 - s2, o3 = i3 - s3;
    for (int k = 0; k < wave; k++) {
        o1 += w1[k];
        o2 += w2[k];
        o3 += w3[k];
    }
    for (size_t b = lo; b < hi; b++) {
        const uint32_t c = counts[b], ch = (c + T - 1) / T;
        offsets[b] = o1;
        chunk_off[b] = o2;
        if (sc_off) sc_off[b] = o3;
        o1 += c;
        o2 += ch;
        o3 += (ch + gather - 1) / gather;
        counts[b] = 0;
    }
    if (threadIdx.x == 1023) {  // its range is the last (possibly empty) one: o1 / o2 / o3 are the grand totals
        offsets[nb] = o1;
        chunk_off[nb] = o2;
        if (sc_off) sc_off[nb] = o3;
    }
}
// The same plan over MANY buckets (merged passes: c = 11, 49 k buckets -- the one workgroup above took 150 us of every
// such pass's chain): a workgroup per 1024 buckets, coalesced.  First launch: the workgroups' totals; second: every
// workgroup adds up the totals before its own (at most a few dozen), scans its buckets and writes the offsets.
__global__ void __launch_bounds__(1024) msm_plan_sums_kernel(const uint32_t *__restrict__ counts, size_t nb, uint32_t T, uint32_t gather,
                                                             uint32_t *__restrict__ block_sums) {
    __shared__ uint32_t w[3][16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t b = (size_t)blockIdx.x * 1024 + threadIdx.x;
    const uint32_t c = b < nb ? counts[b] : 0u, ch = (c + T - 1) / T, sc = (ch + gather - 1) / gather;
    const uint32_t i1 = wave_incl_scan(c, lane), i2 = wave_incl_scan(ch, lane), i3 = wave_incl_scan(sc, lane);
    if (lane == 63) {
        w[0][wave] = i1;
        w[1][wave] = i2;
        w[2][wave] = i3;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        uint32_t t = 0;
        for (int k = 0; k < 16; k++) t += w[threadIdx.x][k];
        block_sums[3 * blockIdx.x + threadIdx.x] = t;
    }
}
__global__ void __launch_bounds__(1024) msm_plan_offsets_kernel(uint32_t *__restrict__ counts, size_t nb, uint32_t T, uint32_t gather,
                                                                const uint32_t *__restrict__ block_sums, uint32_t *__restrict__ offsets,
                                                                uint32_t *__restrict__ chunk_off, uint32_t *__restrict__ sc_off) {
    __shared__ uint32_t w[3][16], base[3];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave == 0) {  // the totals of the workgroups before this one
        uint32_t t0 = 0, t1 = 0, t2 = 0;
        for (uint32_t j = (uint32_t)lane; j < blockIdx.x; j += 64) {
            t0 += block_sums[3 * j];
            t1 += block_sums[3 * j + 1];
            t2 += block_sums[3 * j + 2];
        }
        t0 = wave_incl_scan(t0, lane);
        t1 = wave_incl_scan(t1, lane);
        t2 = wave_incl_scan(t2, lane);
        if (lane == 63) {
            base[0] = t0;
            base[1] = t1;
            base[2] = t2;
        }
    }
    const size_t b = (size_t)blockIdx.x * 1024 + threadIdx.x;
    const uint32_t c = b < nb ? counts[b] : 0u, ch = (c + T - 1) / T, sc = (ch + gather - 1) / gather;
    const uint32_t i1 = wave_incl_scan(c, lane), i2 = wave_incl_scan(ch, lane), i3 = wave_incl_scan(sc, lane);
    if (lane == 63) {
        w[0][wave] = i1;
        w[1][wave] = i2;
        w[2][wave] = i3;
    }
    __syncthreads();
    uint32_t o1 = base[0] + i1 - c, o2 = base[1] + i2 - ch, o3 = base[2] + i3 - sc;
    for (int k = 0; k < wave; k++) {
        o1 += w[0][k];
        o2 += w[1][k];
        o3 += w[2][k];
    }
    if (b < nb) {
        offsets[b] = o1;
        chunk_off[b] = o2;
        if (sc_off) sc_off[b] = o3;
        counts[b] = 0;
    }
    if (b == nb - 1) {  // the grand totals
        offsets[nb] = o1 + c;
        chunk_off[nb] = o2 + ch;
        if (sc_off) sc_off[nb] = o3 + sc;
    }
}
__global__ void __launch_bounds__(256) msm_chunk_count_kernel(const uint32_t *__restrict__ counts, size_t nb, uint32_t T,
                                                              uint32_t *__restrict__ nchunks) {
    size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b < nb) nchunks[b] = (counts[b] + T - 1) / T;
}
// lane per bucket: chunk_bucket[chunk_off[b] + j] = b; buckets with more than `heavy` chunks go on the heavy list
__global__ void __launch_bounds__(256) msm_chunk_map_kernel(const uint32_t *__restrict__ chunk_off, size_t nb,
                                                            uint32_t *__restrict__ chunk_bucket, uint32_t heavy,
                                                            uint32_t *__restrict__ heavy_list,
                                                            uint32_t *__restrict__ heavy_count) {
    size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb) return;
    uint32_t lo = chunk_off[b], hi = chunk_off[b + 1];
    for (uint32_t t = lo; t < hi; t++) chunk_bucket[t] = (uint32_t)b;
    if (hi - lo > heavy) heavy_list[atomicAdd(heavy_count, 1u)] = (uint32_t)b;
}
// ---- order the chunks by length (longest first) so the 64 lanes of a wave walk equally long chunks ----
__device__ __forceinline__ uint32_t chunk_len(uint32_t t, const uint32_t *__restrict__ offsets,
                                              const uint32_t *__restrict__ chunk_off, const uint32_t *__restrict__ chunk_bucket,
                                              uint32_t T) {
    const uint32_t b = chunk_bucket[t];
    const uint32_t k = offsets[b] + (t - chunk_off[b]) * T, end = offsets[b + 1];
    return end - k > T ? T : end - k;
}
// hist[T - len]++ (bin 0 = full chunks); per-block LDS histogram, one global atomic per used bin
__global__ void __launch_bounds__(256) msm_len_hist_kernel(const uint32_t *__restrict__ offsets,
                                                           const uint32_t *__restrict__ chunk_off,
                                                           const uint32_t *__restrict__ chunk_bucket, size_t nb, uint32_t T,
                                                           uint32_t *__restrict__ hist) {
    extern __shared__ uint32_t lh[];
    for (uint32_t i = threadIdx.x; i <= T; i += 256) lh[i] = 0;
    __syncthreads();
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < chunk_off[nb]) atomicAdd(&lh[T - chunk_len((uint32_t)t, offsets, chunk_off, chunk_bucket, T)], 1u);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i <= T; i += 256)
        if (lh[i]) atomicAdd(&hist[i], lh[i]);
}
// single block: exclusive scan of the T+1 bins into cursor[]
__global__ void __launch_bounds__(64) msm_len_scan_kernel(const uint32_t *__restrict__ hist, uint32_t T, uint32_t *__restrict__ cursor) {
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (uint32_t i = 0; i <= T; i++) { cursor[i] = run; run += hist[i]; }
    }
}
// order[cursor[bin]++] = chunk id, with block-local ranks so only one global atomic per (block, bin)
__global__ void __launch_bounds__(256) msm_len_scatter_kernel(const uint32_t *__restrict__ offsets,
                                                              const uint32_t *__restrict__ chunk_off,
                                                              const uint32_t *__restrict__ chunk_bucket, size_t nb, uint32_t T,
                                                              uint32_t *__restrict__ cursor, uint32_t *__restrict__ order) {
    extern __shared__ uint32_t lh[];  // [0, T]: local counts, then [T+1, 2T+1]: global base of this block's run
    for (uint32_t i = threadIdx.x; i < 2 * (T + 1); i += 256) lh[i] = 0;
    __syncthreads();
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = t < chunk_off[nb];
    uint32_t bin = 0, rank = 0;
    if (live) {
        bin = T - chunk_len((uint32_t)t, offsets, chunk_off, chunk_bucket, T);
        rank = atomicAdd(&lh[bin], 1u);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i <= T; i += 256)
        if (lh[i]) lh[T + 1 + i] = atomicAdd(&cursor[i], lh[i]);
    __syncthreads();
    if (live) order[lh[T + 1 + bin] + rank] = (uint32_t)t;
}

__global__ void __launch_bounds__(128, 2) msm_accumulate_kernel(const G1Affine *__restrict__ points,
                                                             const uint32_t *__restrict__ sorted,
                                                             const uint32_t *__restrict__ offsets,
                                                             const uint32_t *__restrict__ chunk_off,
                                                             uint32_t *__restrict__ chunk_bucket,
                                                             const uint32_t *__restrict__ order, size_t nb,
                                                             uint32_t T, G1Xyzz *__restrict__ chunk_partial, int find_bucket) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= chunk_off[nb]) return;
    if (order) t = order[t];
    uint32_t b;
    if (find_bucket) {
        // small path: no chunk map was built; bisect chunk_off (chunk_off[b] <= t < chunk_off[b + 1]) and record the
        // bucket for the reduction kernel
        uint32_t lo = 0, hi = (uint32_t)nb;
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (chunk_off[mid] <= (uint32_t)t) lo = mid; else hi = mid;
        }
        b = lo;
        chunk_bucket[t] = b;
    } else {
        b = chunk_bucket[t];
    }
    uint32_t k = offsets[b] + ((uint32_t)t - chunk_off[b]) * T;
    uint32_t end = offsets[b + 1];
    if (end - k > T) end = k + T;
    // The chunk's sum in the reduced-radix arithmetic of field28.hpp (one v_mad_i64_i32 per limb product, carry-free
    // additions): the accumulator lives in 28-bit limbs from the first point to the last, the gathered points are
    // re-read 8 bits higher (no arithmetic), the result goes back to the library's form once per chunk.
    G1Xyzz28 acc;
    bool have = false;
    uint32_t e_nxt = sorted[k];
    G1Affine nxt = load_point(points, e_nxt & ~1u);
    while (k < end) {
        const G1Affine cur = nxt;
        const uint32_t e_cur = e_nxt;
        k++;
        if (k < end) {  // issue the next gather before the add
            e_nxt = sorted[k];
            nxt = load_point(points, e_nxt & ~1u);
        }
        if (cur.is_inf()) continue;
        const Fq28 qx = load28(cur.x);
        Fq28 qy = load28(cur.y);
        if (e_cur & 1u) qy = neg28(qy);
        if (!have) {
            const Fq28 one = f28_one();
            acc = G1Xyzz28{mul28(qx, one), mul28(qy, one), one, one};
            have = true;
            continue;
        }
        have = add_mixed28(acc, qx, qy);
    }
    G1Xyzz out = G1Xyzz::inf();
    if (have) out = xyzz28_store(acc);
    store_xyzz(&chunk_partial[t], out);
}
// Buckets cut into more than one chunk ("heavy": the top window, skewed scalars): sum their chunk partials into
// buckets[b].  Up to SERIAL_MAX chunks one LANE walks them (a realistic top window is cut into 4-8 chunks per bucket);
// beyond that a whole block does (strided partial sums, LDS tree).
constexpr uint32_t MSM_HEAVY_SERIAL_MAX = 16;
__global__ void __launch_bounds__(64) msm_combine_serial_kernel(const G1Xyzz *__restrict__ chunk_partial,
                                                                const uint32_t *__restrict__ chunk_off,
                                                                const uint32_t *__restrict__ heavy_list,
                                                                const uint32_t *__restrict__ heavy_count,
                                                                G1Xyzz *__restrict__ buckets) {
    const uint32_t nheavy = *heavy_count;
    for (uint32_t h = blockIdx.x * 64 + threadIdx.x; h < nheavy; h += gridDim.x * 64) {
        const uint32_t b = heavy_list[h];
        const uint32_t lo = chunk_off[b], hi = chunk_off[b + 1];
        if (hi - lo > MSM_HEAVY_SERIAL_MAX) continue;
        G1Xyzz28 acc;
        bool have = false;
        for (uint32_t e = lo; e < hi; e++) {
            const G1Xyzz v = load_xyzz(&chunk_partial[e]);
            xyzz_acc28(acc, have, xyzz28_from(v), !v.is_inf());
        }
        store_xyzz(&buckets[b], have ? xyzz28_store(acc) : G1Xyzz::inf());
    }
}
__global__ void __launch_bounds__(256) msm_combine_heavy_kernel(const G1Xyzz *__restrict__ chunk_partial,
                                                                const uint32_t *__restrict__ chunk_off,
                                                                const uint32_t *__restrict__ heavy_list,
                                                                const uint32_t *__restrict__ heavy_count,
                                                                G1Xyzz *__restrict__ buckets) {
    __shared__ G1Xyzz28 lds[128];
    __shared__ uint32_t ldsf[128];
    const int t = threadIdx.x;
    const uint32_t nheavy = *heavy_count;
    for (uint32_t h = blockIdx.x; h < nheavy; h += gridDim.x) {
        const uint32_t b = heavy_list[h];
        const uint32_t lo = chunk_off[b], hi = chunk_off[b + 1];
        if (hi - lo <= MSM_HEAVY_SERIAL_MAX) continue;  // block-uniform
        G1Xyzz28 acc;
        bool have = false;
        for (uint32_t i = lo + t; i < hi; i += 256) {
            const G1Xyzz v = load_xyzz(&chunk_partial[i]);
            xyzz_acc28(acc, have, xyzz28_from(v), !v.is_inf());
        }
        for (int stride = 128; stride >= 1; stride >>= 1) {
            if (t >= stride && t < 2 * stride) {
                lds[t - stride] = acc;
                ldsf[t - stride] = have ? 1u : 0u;
            }
            __syncthreads();
            if (t < stride) xyzz_acc28(acc, have, lds[t], ldsf[t] != 0);
            __syncthreads();
        }
        if (t == 0) store_xyzz(&buckets[b], have ? xyzz28_store(acc) : G1Xyzz::inf());
    }
}

// ---------------------------------------------------------------- K5: segment partials
// thread (w, g): part = sum_{j=0}^{L-1} (g*L + j + 1) * bucket[w][g*L + j]
__global__ void __launch_bounds__(128) msm_segments_kernel(const G1Xyzz *__restrict__ chunk_partial,
                                                           const uint32_t *__restrict__ chunk_off, uint32_t heavy,
                                                           const G1Xyzz *__restrict__ heavy_buckets, uint32_t B, uint32_t L,
                                                           size_t n_segments_total, G1Xyzz *__restrict__ partials) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_segments_total) return;
    uint32_t G = B / L;
    size_t w = t / G;
    uint32_t g = (uint32_t)(t - w * G);
    const size_t b0 = w * B + (size_t)g * L;
    // running sums and the (g L) * run fix-up in the reduced-radix arithmetic of field28.hpp; one conversion back per segment
    G1Xyzz28 run, wsum;
    bool have_run = false, have_w = false;
    for (int j = (int)L - 1; j >= 0; j--) {
        // the bucket's value is its single chunk partial (no separate combine pass); buckets cut into more chunks
        // (heavy = 1) were summed by the heavy kernel
        const uint32_t lo = chunk_off[b0 + j], hi = chunk_off[b0 + j + 1];
        G1Xyzz v = G1Xyzz::inf();
        if (hi - lo > heavy) v = load_xyzz(&heavy_buckets[b0 + j]);
        else if (lo < hi) v = load_xyzz(&chunk_partial[lo]);
        xyzz_acc28(run, have_run, xyzz28_from(v), !v.is_inf());
        xyzz_acc28(wsum, have_w, run, have_run);
    }
    // + (g*L) * run  (double-and-add)
    uint32_t k = g * L;
    if (k != 0 && have_run) {
        G1Xyzz28 acc;
        bool have_acc = false;
        for (int b = 31 - __clz(k); b >= 0; b--) {
            if (have_acc) acc = xyzz_dbl28(acc);
            if ((k >> b) & 1u) xyzz_acc28(acc, have_acc, run, true);
        }
        xyzz_acc28(wsum, have_w, acc, have_acc);
    }
    store_xyzz(&partials[t], have_w ? xyzz28_store(wsum) : G1Xyzz::inf());
}

// ---------------------------------------------------------------- K6: per-group tree sum through LDS
// in[group][m] -> out[group][gridDim.x]; block (x, group) sums a strided slice, then an LDS tree.
__global__ void __launch_bounds__(256) point_tree_sum_kernel(const G1Xyzz *__restrict__ in, size_t m,
                                                             G1Xyzz *__restrict__ out) {
    __shared__ G1Xyzz28 lds[128];  // 28 KiB: the upper half of the live lanes parks its sum here each level
    __shared__ uint32_t ldsf[128];
    const size_t group = blockIdx.y;
    const G1Xyzz *src = in + group * m;
    G1Xyzz28 acc;
    bool have = false;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < m; i += (size_t)gridDim.x * 256) {
        const G1Xyzz v = load_xyzz(&src[i]);
        xyzz_acc28(acc, have, xyzz28_from(v), !v.is_inf());
    }
    const int t = threadIdx.x;
    for (int stride = 128; stride >= 1; stride >>= 1) {
        if (t >= stride && t < 2 * stride) {
            lds[t - stride] = acc;
            ldsf[t - stride] = have ? 1u : 0u;
        }
        __syncthreads();
        if (t < stride) xyzz_acc28(acc, have, lds[t], ldsf[t] != 0);
        __syncthreads();
    }
    if (t == 0) store_xyzz(&out[group * gridDim.x + blockIdx.x], have ? xyzz28_store(acc) : G1Xyzz::inf());
}

// ---------------------------------------------------------------- small-MSM reduction: shorter chunk lists first
// A bucket of the verifier's passes is 6-7 chunks long, and the per-bit sums below add every chunk partial into the
// c / 2 groups whose bit its bucket carries: ~3 400 partials per window of a merged four-batch pass, 27 + 6 additions in
// sequence per group.  Here one lane adds up to MSM_GATHER consecutive chunk partials of ONE bucket (a bucket of
// MSM_GATHER chunks or fewer becomes one entry; the few heavy buckets -- the top window's, skewed scalars' -- become
// several, so no lane walks more than MSM_GATHER): at most MSM_GATHER additions in sequence, once, and the groups then
// sum a list that is 6-7 times shorter: 4-5 + 6 additions in sequence for c = 10, and a third of the products.  The new
// list has the old one's shape (offsets per bucket from the plan kernel, the bucket of every entry), so the per-bit
// kernel does not know the difference.
constexpr uint32_t MSM_GATHER = 8;
// Two lanes per entry: each adds half of the entry's partials (at most gather / 2 in sequence), the even lane then adds
// its neighbour's sum -- five additions in sequence instead of eight.
__global__ void __launch_bounds__(64) msm_gather_chunks_kernel(const G1Xyzz *__restrict__ chunk_partial, const uint32_t *__restrict__ chunk_off,
                                                              const uint32_t *__restrict__ sc_off, size_t nb, uint32_t gather,
                                                              G1Xyzz *__restrict__ out, uint32_t *__restrict__ sc_bucket) {
    const uint32_t s = blockIdx.x * 32 + (threadIdx.x >> 1), half = threadIdx.x & 1;
    const bool live = s < sc_off[nb];  // (no early return: the neighbours exchange their sums below)
    uint32_t lo = 0, hi = 0;
    if (live) {
        uint32_t lo_b = 0, hi_b = (uint32_t)nb;  // sc_off[b] <= s < sc_off[b + 1]
        while (hi_b - lo_b > 1) {
            const uint32_t mid = (lo_b + hi_b) >> 1;
            if (sc_off[mid] <= s) lo_b = mid; else hi_b = mid;
        }
        const uint32_t b = lo_b;
        if (!half) sc_bucket[s] = b;
        const uint32_t first = chunk_off[b] + (s - sc_off[b]) * gather, end = chunk_off[b + 1];
        const uint32_t last = end - first > gather ? first + gather : end;
        const uint32_t mid = first + (gather + 1) / 2 < last ? first + (gather + 1) / 2 : last;
        lo = half ? mid : first;
        hi = half ? last : mid;
    }
    G1Xyzz28 acc;
    bool have = false;
    for (uint32_t e = lo; e < hi; e++) {
        const G1Xyzz v = load_xyzz(&chunk_partial[e]);
        xyzz_acc28(acc, have, xyzz28_from(v), !v.is_inf());
    }
    const G1Xyzz28 other = xyzz28_shfl_down(acc, 1);
    const bool have_other = __shfl_down((int)have, 1, 64) != 0;
    if (!half) {
        xyzz_acc28(acc, have, other, have_other);
        if (live) store_xyzz(&out[s], have ? xyzz28_store(acc) : G1Xyzz::inf());
    }
}

// ---------------------------------------------------------------- small-MSM reduction: per-bit subset sums
// sum_b mag(b) * bucket_b = sum_bit 2^bit * (sum of the buckets whose magnitude has that bit set).  Group
// g = w * c + bit (which is also the bit position of 2^(c w + bit) in the scalar) sums, straight from the chunk
// partials of window w, those whose bucket magnitude has `bit` set: log depth instead of a running sum, at
// c/2 times the additions -- the right trade when the whole MSM is latency bound (n <= 2^17).
// One WAVE per group (window, bit): every lane walks its stride class of the window's chunk list, skipping entries
// whose bucket magnitude lacks the bit without holding the other lanes up (the wave only synchronises on additions),
// then 6 shuffle levels.  A quarter of the waves of the 256-lane version and no barrier: the whole verifier is bound
// by wave-slots x time of this reduction, not by its latency.  `out` may be pinned host memory (results go straight
// to the folding thread, no copy kernel).
// One WAVE per group, balanced: the wave reads the window's chunk list 64 entries at a time, the lanes whose entry has
// the bit set append it to a small LDS queue (ballot + prefix count), and whenever 64 entries are queued every lane
// takes one and adds it.  ceil(matches / 64) additions per lane instead of the longest lane's share of a strided walk
// (~7 instead of ~10 for the verifier's windows), then the 6 shuffle levels.
__global__ void __launch_bounds__(64) msm_bit_sums_queue_kernel(const G1Xyzz *__restrict__ chunk_partial,
                                                                const uint32_t *__restrict__ chunk_off,
                                                                const uint32_t *__restrict__ chunk_bucket, uint32_t B, int c,
                                                                G1Xyzz *__restrict__ out) {
    __shared__ uint32_t queue[128];
    const int lane = threadIdx.x;
    const uint32_t g = blockIdx.x, w = g / (uint32_t)c, bit = g - w * (uint32_t)c;
    const uint32_t lo = chunk_off[(size_t)w * B], hi = chunk_off[(size_t)(w + 1) * B];
    // sums in the reduced-radix arithmetic of field28.hpp from the first partial to the last shuffle level; one
    // conversion back per group
    G1Xyzz28 acc;
    bool have = false;
    uint32_t head = 0, tail = 0;  // wave-uniform
    for (uint32_t base = lo; base < hi; base += 64) {
        const uint32_t e = base + lane;
        const bool match = e < hi && (((chunk_bucket[e] - w * B + 1) >> bit) & 1u);
        const uint64_t mm = __ballot(match);
        if (match) queue[(tail + (uint32_t)__popcll(mm & ((1ull << lane) - 1ull))) & 127u] = e;
        tail += (uint32_t)__popcll(mm);
        __syncthreads();  // one wave: orders the queue writes before the reads below
        if (tail - head >= 64) {
            const uint32_t q = queue[(head + lane) & 127u];
            head += 64;
            const G1Xyzz v = load_xyzz(&chunk_partial[q]);
            xyzz_acc28(acc, have, xyzz28_from(v), !v.is_inf());
        }
        __syncthreads();
    }
    if ((uint32_t)lane < tail - head) {
        const G1Xyzz v = load_xyzz(&chunk_partial[queue[(head + lane) & 127u]]);
        xyzz_acc28(acc, have, xyzz28_from(v), !v.is_inf());
    }
#pragma unroll 1
    for (int stride = 32; stride >= 1; stride >>= 1) {
        const G1Xyzz28 other = xyzz28_shfl_down(acc, stride);
        const bool have_other = __shfl_down((int)have, stride, 64) != 0;
        if (lane < stride) xyzz_acc28(acc, have, other, have_other);
    }
    if (lane == 0) store_xyzz(&out[g], have ? xyzz28_store(acc) : G1Xyzz::inf());
}

// (Variants of this reduction were built, measured and REMOVED in round 3: several groups per wave (less work, a longer
// chain: slower), a strided walk without the queue (slower), and four groups per 256-lane block with one tree through
// LDS (+2.5 % streamed throughput, but one 400-batch run in five to seventeen ended in a memory fault at a 2 MB-aligned
// address or an illegal-instruction error whose cause was not found by reading its code, its descriptor or its ISA:
// DESIGN.md section 3 item 5.  A verifier library does not ship a path with a known fault behind a switch.)

// ---------------------------------------------------------------- large MSMs: tiled histogram in LDS
// The global histogram atomics run at the memory-side atomic rate (21 G/s: 0.7 ms of a 2^20 MSM, 10.6 ms at 2^24).
// Here a workgroup owns one (window, tile of points): the window's whole histogram (2^(c-1) counters, <= 128 KB) lives
// in its LDS, the atomic that counts an entry and returns its rank is an LDS atomic, and the per-tile counts go out in
// one coalesced store.  A column scan over the tiles turns them into per-tile bases (and the window's total counts);
// a second pass over the tile adds the base to each entry's rank.  Scalars are re-read once per window (L2 / HBM
// streaming) and the signed-digit carry into window w comes from one comparison instead of the chain through the
// lower windows:  carry_w = [ k mod 2^(c w)  >  sum_{j < w} 2^(c j + c - 1) ]   (the recoding d = raw + carry > 2^(c-1)
// ? raw + carry - 2^c : raw + carry, by induction on w).
struct MsmTileConsts {
    uint32_t half[24][8];  // half[w] = sum_{j < w} 2^(c j + c - 1), little-endian limbs (w >= 1)
};
__global__ void __launch_bounds__(1024) msm_digits_tiled_kernel(const uint32_t *__restrict__ scalars, size_t n, size_t tile_size,
                                                                int c, const MsmTileConsts *__restrict__ tc,
                                                                int32_t *__restrict__ digits, uint32_t *__restrict__ ranks,
                                                                uint32_t *__restrict__ tile_counts) {
    extern __shared__ uint32_t hist[];
    const uint32_t B = 1u << (c - 1);
    const int w = blockIdx.y;
    const size_t lo = (size_t)blockIdx.x * tile_size, hi = lo + tile_size < n ? lo + tile_size : n;
    for (uint32_t b = threadIdx.x; b < B; b += 1024) hist[b] = 0;
    __syncthreads();
    const uint32_t *hw = tc->half[w];
    for (size_t i = lo + threadIdx.x; i < hi; i += 1024) {
        const uint4 *sp = reinterpret_cast<const uint4 *>(scalars + i * 8);
        const uint4 x = sp[0], y = sp[1];
        const uint32_t s[8] = {x.x, x.y, x.z, x.w, y.x, y.y, y.z, y.w};
        const int32_t d = msm_digit_at(s, c, w, hw);
        digits[(size_t)w * n + i] = d;
        if (d != 0) {
            const uint32_t mag = d < 0 ? (uint32_t)(-d) : (uint32_t)d;
            ranks[(size_t)w * n + i] = atomicAdd(&hist[mag - 1], 1u);  // rank inside (tile, bucket)
        }
    }
    __syncthreads();
    uint32_t *out = tile_counts + ((size_t)w * gridDim.x + blockIdx.x) * B;
    for (uint32_t b = threadIdx.x; b < B; b += 1024) out[b] = hist[b];
}
// lane per (window, bucket): exclusive scan of the tiles' counts in place; the total is the bucket's count
__global__ void __launch_bounds__(256) msm_tile_scan_kernel(uint32_t *__restrict__ tile_counts, size_t nb, uint32_t B, uint32_t n_tiles,
                                                            uint32_t *__restrict__ counts) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nb) return;
    const size_t w = t / B, b = t - w * B;
    uint32_t run = 0;
    for (uint32_t k = 0; k < n_tiles; k++) {
        uint32_t *p = tile_counts + ((size_t)w * n_tiles + k) * B + b;
        const uint32_t v = *p;
        *p = run;
        run += v;
    }
    counts[t] = run;
}
// ranks[w][i] += base of (tile of i, bucket of i)
__global__ void __launch_bounds__(1024) msm_rank_fixup_kernel(const int32_t *__restrict__ digits, size_t n, size_t tile_size, int c,
                                                              const uint32_t *__restrict__ tile_base, uint32_t *__restrict__ ranks) {
    extern __shared__ uint32_t base[];
    const uint32_t B = 1u << (c - 1);
    const int w = blockIdx.y;
    const size_t lo = (size_t)blockIdx.x * tile_size, hi = lo + tile_size < n ? lo + tile_size : n;
    const uint32_t *src = tile_base + ((size_t)w * gridDim.x + blockIdx.x) * B;
    for (uint32_t b = threadIdx.x; b < B; b += 1024) base[b] = src[b];
    __syncthreads();
    for (size_t i = lo + threadIdx.x; i < hi; i += 1024) {
        const int32_t d = digits[(size_t)w * n + i];
        if (d != 0) ranks[(size_t)w * n + i] += base[(d < 0 ? (uint32_t)(-d) : (uint32_t)d) - 1];
    }
}

// ---------------------------------------------------------------- host driver
template <int C, int T>
static void launch_digits_t(const uint32_t *scalars, size_t n, int32_t *digits, uint32_t *ranks, uint32_t *counts,
                            int scalar_bits, MsmSplits split, hipStream_t st) {
    if (scalar_bits > 128)
        hipLaunchKernelGGL((msm_digits_kernel<C, 254, T>), dim3(cdiv(n, T), ((254 + C - 1) / C + msm_digits_group<C>() - 1) / msm_digits_group<C>()), dim3(T), 0, st, scalars, n, digits, ranks, counts, split);
    else
        hipLaunchKernelGGL((msm_digits_kernel<C, 129, T>), dim3(cdiv(n, T), ((129 + C - 1) / C + msm_digits_group<C>() - 1) / msm_digits_group<C>()), dim3(T), 0, st, scalars, n, digits, ranks, counts, split);
}
// workgroups of 512: measured 7.78e6 proofs/s streamed against 7.6e6 with 1024 (a workgroup
// that large waits for a whole CU's worth of wave slots) and 7.3-7.7e6 with 256 (four times the global atomics of 1024)
template <int C>
static void launch_digits(const uint32_t *scalars, size_t n, int32_t *digits, uint32_t *ranks, uint32_t *counts,
                          int scalar_bits, MsmSplits split, hipStream_t st) {
    launch_digits_t<C, 512>(scalars, n, digits, ranks, counts, scalar_bits, split, st);
}

// Sum groups of XYZZ points on the device: in[groups][m] -> result[groups] (device pointer into workspace)
static G1Xyzz *device_group_sum(Context &cx, G1Xyzz *in, size_t groups, size_t m, hipStream_t st, const char *tag_a,
                                const char *tag_b) {
    G1Xyzz *cur = in;
    bool flip = false;
    while (m > 1) {
        unsigned bx = (unsigned)((m + 2047) / 2048);  // >= 8 elements per lane before the tree
        if (bx < 1) bx = 1;
        G1Xyzz *dst = (G1Xyzz *)cx.buf(flip ? tag_b : tag_a, sizeof(G1Xyzz) * groups * bx);
        hipLaunchKernelGGL(point_tree_sum_kernel, dim3(bx, (unsigned)groups), dim3(256), 0, st, cur, m, dst);
        cur = dst;
        m = bx;
        flip = !flip;
    }
    return cur;
}

struct CleanMark {
    uint64_t gen = 0;
    size_t nb = 0;
    const void *p = nullptr;
};
struct MsmState {  // per device (Context::mod)
    std::map<int, CleanMark> clean;  // per workspace slot: the small path's planning kernel left the histogram cleared
    uint64_t tc_gen[64] = {};        // per slot: generation of the tile-constant buffer that holds an uploaded table
    bool tc_ready[64] = {};
};

// Enqueue the whole device pipeline of one MSM on `st` (no host synchronisation); `slot` names the workspace set,
// so two MSMs with different slots may run on two streams at once.  msm_finish() waits and folds on the host.
void msm_enqueue(Context &cx, const G1Affine *d_points, const uint32_t *d_scalars, size_t n, hipStream_t st, int scalar_bits,
                 int slot, MsmJob *job, int force_c, int pin_parity, MsmSplits split) {
    job->n = n;
    job->groups = 0;
    job->problems = 1;
    if (n == 0) return;
    char ws[32];
    auto W = [&](const char *name) {
        snprintf(ws, sizeof ws, "msm%d.%s", slot, name);
        return ws;
    };
    MsmPlan pl = msm_plan(n, scalar_bits, force_c);
    stats().msm_calls++;
    stats().msm_points += n;
    job->c = pl.c;
    job->nwin = pl.nwin;
    // A small MSM is latency bound (one mixed addition is ~10k instructions; its ~17 dependent launches cost more
    // than their work): it takes short chunks, one fused planning kernel and the log-depth reduction below.
    const bool small = n <= msm_small_limit();
    int problems = 1;
    for (int k = 0; k < MSM_MAX_PROBLEMS - 1; k++) {
        if (split.at[k] == 0 || split.at[k] >= n) split.at[k] = n;  // unused: no point lies beyond it
        else problems = k + 2;
    }
    for (int k = 0; k + 1 < problems; k++)
        if (split.at[k] == n || (k && split.at[k] < split.at[k - 1])) throw HipError{hipErrorInvalidValue, "msm split points", __LINE__};
    if (problems > 1 && !small) throw HipError{hipErrorInvalidValue, "msm split needs n <= 2^17", __LINE__};
    job->problems = problems;
    cx.prof.begin("msm_total", st, n);
    HostSpan hs_bufs("msm.enqueue_all");
    const size_t nb = (size_t)problems * pl.nwin * pl.B;  // all buckets (both problems' windows)
    const int vwin = problems * pl.nwin;
    int32_t *digits = (int32_t *)cx.buf(W("digits"), sizeof(int32_t) * n * pl.nwin);
    uint32_t *sorted = (uint32_t *)cx.buf(W("sorted"), sizeof(uint32_t) * n * pl.nwin);
    DevBuf &counts_buf = cx.bufs[W("counts")];
    uint32_t *counts = (uint32_t *)counts_buf.get(sizeof(uint32_t) * nb);
    uint32_t *offsets = (uint32_t *)cx.buf(W("offsets"), sizeof(uint32_t) * (nb + 1));
    uint32_t *ranks = (uint32_t *)cx.buf(W("ranks"), sizeof(uint32_t) * n * pl.nwin);
    const unsigned tiles = cdiv(nb, 2048);
    uint32_t *tile_sums = (uint32_t *)cx.buf(W("tiles"), sizeof(uint32_t) * tiles);

    // the small path's planning kernel leaves the histogram cleared: skip the memset when this workspace's last
    // user was that kernel, on the same allocation, over at least as many buckets
    MsmState &ms = cx.mod<MsmState>("msm");
    CleanMark &mark = ms.clean[slot];
    const bool clean = small && mark.gen == counts_buf.gen && mark.nb >= nb && mark.p == counts;
    mark = CleanMark();
    static const bool tiled_front = !(getenv("MI_SNARK_MSM_TILED") && getenv("MI_SNARK_MSM_TILED")[0] == '0');
    const bool tiled = !small && tiled_front && pl.nwin <= 24;
    if (!clean && !tiled) MI_HIP(hipMemsetAsync(counts, 0, sizeof(uint32_t) * nb, st));
    HostSpan hs_launch("msm.launches");
    cx.prof.begin("msm_digits", st, n);
    if (tiled) {
        // tiles of >= 32 Ki points, at most 64 per window (the per-tile count table is nwin x tiles x B words)
        unsigned n_tiles = (unsigned)((n + 32767) / 32768);
        if (n_tiles > 64) n_tiles = 64;
        const size_t tile_size = (n + n_tiles - 1) / n_tiles;
        uint32_t *tile_counts = (uint32_t *)cx.buf(W("tile_counts"), sizeof(uint32_t) * (size_t)pl.nwin * n_tiles * pl.B);
        DevBuf &tcb = cx.bufs[W("tile_consts")];
        MsmTileConsts *d_tc = (MsmTileConsts *)tcb.get(sizeof(MsmTileConsts) * 17);  // one table per window width c
        const bool have_tc = slot >= 0 && slot < 64 && ms.tc_ready[slot] && ms.tc_gen[slot] == tcb.gen;
        if (!have_tc) {
            static MsmTileConsts h[17];
            static std::once_flag once;
            std::call_once(once, [] {
                memset(h, 0, sizeof h);
                for (int cc = 6; cc <= 16; cc++)
                    for (int w = 1; w < 24; w++)
                        for (int j = 0; j < w; j++) {
                            const int bit = cc * j + cc - 1;
                            if (bit < 256) h[cc].half[w][bit >> 5] |= 1u << (bit & 31);
                        }
            });
            MI_HIP(hipMemcpyAsync(d_tc, h, sizeof h, hipMemcpyHostToDevice, st));
            MI_HIP(hipStreamSynchronize(st));
            if (slot >= 0 && slot < 64) {
                ms.tc_ready[slot] = true;
                ms.tc_gen[slot] = tcb.gen;
            }
        }
        const size_t lds = sizeof(uint32_t) * pl.B;
        cx.raise_lds_limit((const void *)msm_digits_tiled_kernel, 131072);
        cx.raise_lds_limit((const void *)msm_rank_fixup_kernel, 131072);
        hipLaunchKernelGGL(msm_digits_tiled_kernel, dim3(n_tiles, (unsigned)pl.nwin), dim3(1024), lds, st, d_scalars, n, tile_size, pl.c,
                           d_tc + pl.c, digits, ranks, tile_counts);
        hipLaunchKernelGGL(msm_tile_scan_kernel, dim3(cdiv(nb, 256)), dim3(256), 0, st, tile_counts, nb, pl.B, n_tiles, counts);
        hipLaunchKernelGGL(msm_rank_fixup_kernel, dim3(n_tiles, (unsigned)pl.nwin), dim3(1024), lds, st, digits, n, tile_size, pl.c,
                           tile_counts, ranks);
    } else {
        switch (pl.c) {
#define MI_CASE(C) case C: launch_digits<C>(d_scalars, n, digits, ranks, counts, scalar_bits, split, st); break;
            MI_CASE(6) MI_CASE(7) MI_CASE(8) MI_CASE(9) MI_CASE(10) MI_CASE(11) MI_CASE(12) MI_CASE(13) MI_CASE(14) MI_CASE(15) MI_CASE(16)
#undef MI_CASE
        }
    }
    cx.prof.end("msm_digits", st);
    // chunking: T = twice the mean bucket load of a full window, so only over-full buckets are split
    uint32_t T = (uint32_t)(2 * ((n + pl.B - 1) / pl.B));
    if (T < 16) T = 16;
    if (small) {
        T = msm_small_chunk();
    }
    const uint32_t HEAVY = 1;  // every bucket with more than one chunk is summed by the heavy kernel (a wave each)
    const size_t max_chunks = nb + (n * (size_t)pl.nwin) / T + 1;
    uint32_t *chunk_off = (uint32_t *)cx.buf(W("chunk_off"), sizeof(uint32_t) * (nb + 1));
    uint32_t *chunk_bucket = (uint32_t *)cx.buf(W("chunk_bucket"), sizeof(uint32_t) * max_chunks);
    G1Xyzz *chunk_partial = (G1Xyzz *)cx.buf(W("chunk_partial"), sizeof(G1Xyzz) * max_chunks);
    uint32_t *heavy_list = nullptr, *heavy_count = nullptr, *order = nullptr, *sc_off = nullptr;
    bool gather = false;
    uint32_t gather_len = MSM_GATHER;
    if (small) {
        cx.prof.begin("msm_scan", st, n);
        // long chunk lists (merged passes) are shortened before the per-bit sums: msm_gather_chunks_kernel
        static const int gather_knob = getenv("MI_SNARK_MSM_GATHER") ? atoi(getenv("MI_SNARK_MSM_GATHER")) : -1;
        gather = gather_knob >= 0 ? gather_knob != 0 : (double)n * pl.nwin / (double)T / (double)vwin >= 1000.0;
        gather_len = gather_knob > 1 ? (uint32_t)gather_knob : MSM_GATHER;
        if (gather) sc_off = (uint32_t *)cx.buf(W("sc_off"), sizeof(uint32_t) * (nb + 1));
        if (nb > 16384) {  // many buckets: a workgroup per 1024 of them, two launches
            const unsigned nblk = (unsigned)cdiv(nb, 1024);
            uint32_t *plan_sums = (uint32_t *)cx.buf(W("plan_sums"), sizeof(uint32_t) * 3 * nblk);
            hipLaunchKernelGGL(msm_plan_sums_kernel, dim3(nblk), dim3(1024), 0, st, counts, nb, T, gather_len, plan_sums);
            hipLaunchKernelGGL(msm_plan_offsets_kernel, dim3(nblk), dim3(1024), 0, st, counts, nb, T, gather_len, plan_sums, offsets, chunk_off, sc_off);
        } else {
            hipLaunchKernelGGL(msm_plan_small_kernel, dim3(1), dim3(1024), 0, st, counts, nb, T, offsets, chunk_off, gather_len, sc_off);
        }
        cx.prof.end("msm_scan", st);
        mark.gen = counts_buf.gen;
        mark.nb = nb;
        mark.p = counts;
        cx.prof.begin("msm_scatter", st, n);
        hipLaunchKernelGGL(msm_scatter_kernel, dim3(cdiv(n * pl.nwin, 256)), dim3(256), 0, st, digits, ranks, n, pl.nwin, pl.B, offsets, sorted, split);
        cx.prof.end("msm_scatter", st);
    } else {
        cx.prof.begin("msm_scan", st, n);
        hipLaunchKernelGGL(scan_tiles_kernel, dim3(tiles), dim3(256), 0, st, counts, offsets, tile_sums, nb);
        hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(1024), 0, st, tile_sums, (size_t)tiles);
        hipLaunchKernelGGL(scan_finish_kernel, dim3(tiles), dim3(256), 0, st, offsets, (uint32_t *)nullptr, tile_sums, counts, nb);
        cx.prof.end("msm_scan", st);
        cx.prof.begin("msm_scatter", st, n);
        hipLaunchKernelGGL(msm_scatter_kernel, dim3(cdiv(n * pl.nwin, 256)), dim3(256), 0, st, digits, ranks, n, pl.nwin, pl.B, offsets, sorted, split);
        cx.prof.end("msm_scatter", st);
        uint32_t *nchunks = (uint32_t *)cx.buf(W("nchunks"), sizeof(uint32_t) * nb);
        heavy_list = (uint32_t *)cx.buf(W("heavy"), sizeof(uint32_t) * (nb + 1));
        heavy_count = heavy_list + nb;
        cx.prof.begin("msm_chunks", st, n);
        MI_HIP(hipMemsetAsync(heavy_count, 0, sizeof(uint32_t), st));
        hipLaunchKernelGGL(msm_chunk_count_kernel, dim3(cdiv(nb, 256)), dim3(256), 0, st, counts, nb, T, nchunks);
        hipLaunchKernelGGL(scan_tiles_kernel, dim3(tiles), dim3(256), 0, st, nchunks, chunk_off, tile_sums, nb);
        hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(1024), 0, st, tile_sums, (size_t)tiles);
        hipLaunchKernelGGL(scan_finish_kernel, dim3(tiles), dim3(256), 0, st, chunk_off, (uint32_t *)nullptr, tile_sums, nchunks, nb);
        hipLaunchKernelGGL(msm_chunk_map_kernel, dim3(cdiv(nb, 256)), dim3(256), 0, st, chunk_off, nb, chunk_bucket, HEAVY, heavy_list, heavy_count);
        order = (uint32_t *)cx.buf(W("order"), sizeof(uint32_t) * max_chunks);
        uint32_t *lhist = (uint32_t *)cx.buf(W("lenhist"), sizeof(uint32_t) * 2 * (T + 1));
        uint32_t *lcursor = lhist + (T + 1);
        MI_HIP(hipMemsetAsync(lhist, 0, sizeof(uint32_t) * (T + 1), st));
        hipLaunchKernelGGL(msm_len_hist_kernel, dim3(cdiv(max_chunks, 256)), dim3(256), sizeof(uint32_t) * (T + 1), st, offsets,
                           chunk_off, chunk_bucket, nb, T, lhist);
        hipLaunchKernelGGL(msm_len_scan_kernel, dim3(1), dim3(64), 0, st, lhist, T, lcursor);
        hipLaunchKernelGGL(msm_len_scatter_kernel, dim3(cdiv(max_chunks, 256)), dim3(256), sizeof(uint32_t) * 2 * (T + 1), st,
                           offsets, chunk_off, chunk_bucket, nb, T, lcursor, order);
        cx.prof.end("msm_chunks", st);
    }
    cx.prof.begin("msm_accumulate", st, n);
    hipLaunchKernelGGL(msm_accumulate_kernel, dim3(cdiv(max_chunks, 128)), dim3(128), 0, st, d_points, sorted, offsets, chunk_off,
                       chunk_bucket, order, nb, T, chunk_partial, small ? 1 : 0);
    cx.prof.end("msm_accumulate", st);
    G1Xyzz *result;
    char ta[32], tb[32];
    snprintf(ta, sizeof ta, "msm%d.tree_a", slot);
    snprintf(tb, sizeof tb, "msm%d.tree_b", slot);
    if (small) {
        // per-bit subset sums straight from the chunk partials; the host walks the bit positions (Horner)
        job->groups = (size_t)pl.nwin * pl.c;  // per problem
        job->per_bit = true;
        const size_t all_groups = (size_t)vwin * pl.c;
        char pn[32];
        snprintf(pn, sizeof pn, "msm%d.pin%d", slot, pin_parity);
        job->h = (G1Xyzz *)cx.pin_named(pn, sizeof(G1Xyzz) * all_groups);
        cx.prof.begin("msm_bitsums", st, n);
        // one wave per group, its own six tree levels, results straight into pinned host memory
        if (gather) {
            const size_t max_sc = nb + max_chunks / gather_len + 1;
            G1Xyzz *sc_partial = (G1Xyzz *)cx.buf(W("sc_partial"), sizeof(G1Xyzz) * max_sc);
            uint32_t *sc_bucket = (uint32_t *)cx.buf(W("sc_bucket"), sizeof(uint32_t) * max_sc);
            hipLaunchKernelGGL(msm_gather_chunks_kernel, dim3((unsigned)cdiv(max_sc, 32)), dim3(64), 0, st, chunk_partial, chunk_off, sc_off, nb,
                               gather_len, sc_partial, sc_bucket);
            hipLaunchKernelGGL(msm_bit_sums_queue_kernel, dim3((unsigned)all_groups), dim3(64), 0, st, sc_partial, sc_off, sc_bucket, pl.B, pl.c,
                               job->h);
        } else {
            hipLaunchKernelGGL(msm_bit_sums_queue_kernel, dim3((unsigned)all_groups), dim3(64), 0, st, chunk_partial, chunk_off,
                               chunk_bucket, pl.B, pl.c, job->h);
        }
        cx.prof.end("msm_bitsums", st);
        result = nullptr;
    } else {
        G1Xyzz *buckets = (G1Xyzz *)cx.buf(W("buckets"), sizeof(G1Xyzz) * nb);
        cx.prof.begin("msm_combine", st, n);
        hipLaunchKernelGGL(msm_combine_serial_kernel, dim3(256), dim3(64), 0, st, chunk_partial, chunk_off, heavy_list, heavy_count, buckets);
        hipLaunchKernelGGL(msm_combine_heavy_kernel, dim3(1024), dim3(256), 0, st, chunk_partial, chunk_off, heavy_list, heavy_count, buckets);
        cx.prof.end("msm_combine", st);
        uint32_t L = pl.B >= 4096 ? 8 : 4;
        const size_t nseg = nb / L;
        G1Xyzz *partials = (G1Xyzz *)cx.buf(W("partials"), sizeof(G1Xyzz) * nseg);
        cx.prof.begin("msm_segments", st, n);
        hipLaunchKernelGGL(msm_segments_kernel, dim3(cdiv(nseg, 128)), dim3(128), 0, st, chunk_partial, chunk_off, HEAVY, buckets, pl.B,
                           L, nseg, partials);
        cx.prof.end("msm_segments", st);
        cx.prof.begin("msm_tree", st, n);
        result = device_group_sum(cx, partials, pl.nwin, pl.B / L, st, ta, tb);
        cx.prof.end("msm_tree", st);
        job->groups = pl.nwin;
        job->per_bit = false;
    }
    MI_HIP(hipGetLastError());
    cx.prof.end("msm_total", st);
    char pn[32];
    snprintf(pn, sizeof pn, "msm%d.pin%d", slot, pin_parity);
    if (result) {  // the window sums of the large path: copy out
        job->h = (G1Xyzz *)cx.pin_named(pn, sizeof(G1Xyzz) * job->groups * problems);
        MI_HIP(hipMemcpyAsync(job->h, result, sizeof(G1Xyzz) * job->groups * problems, hipMemcpyDeviceToHost, st));
    }
    job->done = cx.event_named(pn);
    MI_HIP(hipEventRecord(job->done, st));
}

// (A hipGraph replay of the small path's launch sequence existed in round 1 and was removed: an instantiated graph bakes
// in the addresses of name-keyed workspaces and pinned buffers and the "histogram already clear" decision of the moment
// of capture, and nothing invalidated it when an eager call on the same slot regrew a workspace or left the histogram
// dirty -- stale pointers and an unzeroed histogram turn into out-of-range scatter writes.  DESIGN.md "Workspace lifetime".)

// wait for the job's stream, fold on the host (Horner over windows or bit positions), normalise to z = 1
// One problem's per-bit sums -> its point, as K independent parts: part k folds the bit positions [at[k], at[k+1])
// (a doubling and an addition each) and is then shifted up by at[k] doublings; the boundaries balance the parts with the
// measured costs of hostfold.hpp (a folded bit ~2.7 shift doublings: the top part is the shortest).  The parts of all
// problems of a job run side by side on the pool (no task waits for another).  K = 3 when several folds share the pool
// (the streamed verifier), K = 4 when this fold has it to itself (one batch on its own: the critical path drops by a
// quarter; 6 buys nothing more).
struct FoldSplit {
    static constexpr int KMAX = 4;
    const G1Xyzz *h;
    size_t G;
    int K;
    size_t at[KMAX + 1];
    G1Xyzz part[KMAX];
    FoldSplit(const G1Xyzz *h_, size_t G_, int K_) : h(h_), G(G_), K(K_) {
        while (G > 0 && h[G - 1].is_inf()) G--;  // short scalars leave the top positions empty
        if (K < 1) K = 1;
        if (K > KMAX) K = KMAX;
        // equal cost C per part: len_k 27 + at[k] 10 = C; bisect C so that the lengths add up to G
        double lo = 0, hi = 27.0 * (double)G + 1;
        for (int it = 0; it < 50; it++) {
            const double C = 0.5 * (lo + hi);
            double a = 0, tot = 0;
            for (int k = 0; k < K; k++) {
                double len = (C - 10.0 * a) / 27.0;
                if (len < 0) len = 0;
                tot += len;
                a += len;
            }
            if (tot >= (double)G) hi = C; else lo = C;
        }
        double a = 0;
        at[0] = 0;
        for (int k = 0; k < K; k++) {
            double len = (hi - 10.0 * a) / 27.0;
            if (len < 0) len = 0;
            a += len;
            size_t e = (size_t)(a + 0.5);
            if (e > G || k == K - 1) e = G;
            if (e < at[k]) e = at[k];
            at[k + 1] = e;
        }
        for (int k = 0; k < K; k++) part[k] = G1Xyzz::inf();
    }
    void run(int k) { part[k] = hostfold::xyzz(hostfold::shift(hostfold::fold_bits(h, at[k], at[k + 1]), at[k])); }
    G1Jac result() const {
        G1Xyzz acc = part[0];
        for (int k = 1; k < K; k++) acc = add(acc, part[k]);
        return affine_to_jac(to_affine(acc));
    }
};
static std::atomic<int> g_folds_in_flight{0};
void msm_finish(Context &cx, MsmJob *job, hipStream_t st, mi_g1_jac *out, mi_g1_jac *out_b, mi_g1_jac *out_c, mi_g1_jac *out_d) {
    mi_g1_jac *outs[4] = {out, out_b, out_c, out_d};
    msm_finish_all(cx, job, st, outs, 4);
}
void msm_finish_all(Context &cx, MsmJob *job, hipStream_t st, mi_g1_jac *const *outs_in, int n_outs) {
    (void)cx;
    (void)st;
    mi_g1_jac *outs[MSM_MAX_PROBLEMS] = {};
    for (int p = 0; p < n_outs && p < MSM_MAX_PROBLEMS; p++) outs[p] = outs_in[p];
    mi_g1_jac *out = outs[0];
    G1Jac res = G1Jac{Fq::one(), Fq::one(), Fq::zero()};
    if (job->n == 0) {
        for (mi_g1_jac *o : outs)
            if (o) memcpy(o, &res, sizeof res);
        return;
    }
    {
        HostSpan hs("finish.wait");
        MI_HIP(hipEventSynchronize(job->done));  // this job only: later work on the same stream is not waited for
    }
    HostSpan hs_fold("finish.fold");
    const G1Xyzz *h = job->h;
    if (job->per_bit) {
        // every problem's bit sums fold as two independent halves; all halves of the job run side by side on the pool
        // (this thread takes the first), then the normalisations (one Fq inversion each) do the same
        const int np = job->problems;
        struct InFlight {
            int others;
            InFlight() : others(g_folds_in_flight.fetch_add(1)) {}
            ~InFlight() { g_folds_in_flight.fetch_sub(1); }
        } in_flight;
        const int K = in_flight.others == 0 ? 4 : 3;  // measured (a sweep of this split, 20 and 400 batches): busy 1 / 2 / 3 within noise of each other, 3 ahead on both rates
        std::vector<FoldSplit> f;
        f.reserve(MSM_MAX_PROBLEMS);
        for (int p = 0; p < np; p++) f.emplace_back(h + (size_t)p * job->groups, job->groups, job->groups > 160 ? K : (K < 2 ? K : 2));
        std::vector<std::future<void>> tasks;
        for (int p = 0; p < np; p++) {
            FoldSplit *fp = &f[p];
            for (int k = fp->K - 1; k >= 0; k--) {
                if (p == 0 && k == 0) continue;  // this thread's share
                tasks.push_back(worker_pool().submit([fp, k]() { fp->run(k); }));
            }
        }
        std::exception_ptr err;
        {
            HostSpan hs_parts("finish.fold.parts");
            try {
                HostSpan hs_own("finish.fold.own_part");
                f[0].run(0);
            } catch (...) {
                err = std::current_exception();
            }
            for (auto &t : tasks) {
                try {
                    t.get();
                } catch (...) {
                    if (!err) err = std::current_exception();
                }
            }
        }
        if (err) std::rethrow_exception(err);
        HostSpan hs_norm("finish.fold.normalise");
        G1Jac r[MSM_MAX_PROBLEMS];
        tasks.clear();
        for (int p = 1; p < np; p++) {
            FoldSplit *fp = &f[p];
            G1Jac *rp = &r[p];
            if (outs[p]) tasks.push_back(worker_pool().submit([fp, rp]() { *rp = fp->result(); }));
        }
        r[0] = f[0].result();
        for (auto &t : tasks) t.get();
        for (int p = 0; p < np; p++)
            if (outs[p]) memcpy(outs[p], &r[p], sizeof r[p]);
        return;
    }
    hostfold::PX acc = hostfold::px(h[job->nwin - 1]);
    for (int w = job->nwin - 2; w >= 0; w--) acc = hostfold::add(hostfold::shift(acc, (size_t)job->c), hostfold::px(h[w]));
    res = affine_to_jac(to_affine(hostfold::xyzz(acc)));
    memcpy(out, &res, sizeof res);
}

// d_points / d_scalars: device pointers.  Result: normalised Jacobian (z = 1) or infinity (z = 0).
void msm_device(Context &cx, const G1Affine *d_points, const uint32_t *d_scalars, size_t n, mi_g1_jac *out,
                hipStream_t st, int scalar_bits, int force_c) {
    MsmJob job;
    msm_enqueue(cx, d_points, d_scalars, n, st, scalar_bits, 0, &job, force_c);
    msm_finish(cx, &job, st, out);
}

}  // namespace mi
