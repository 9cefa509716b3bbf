// BLS12-377 prime fields for gfx950 (and the host side of the same library).
//
// Replaces snarkvm_fields::{Fp256<FrParameters>, Fp384<FqParameters>} (snarkVM 1.0.0 @ dea322b, the
// un-vendored dependency pinned at reference Cargo.toml:47-52) on the device.
// Memory layout is the ABI's: little-endian 64-bit limbs, Montgomery form with R = 2^256 (Fr) /
// 2^384 (Fq).  CDNA4 has no 64-bit integer multiplier, so the arithmetic runs on 32-bit limbs
// (v_mad_u64_u32: 32x32+64 -> 64) over the very same bytes; R and therefore every stored value is
// unchanged.  Both moduli are 1 mod 2^32, so the Montgomery factor is m = -t0 (no multiply).
//
// The same header compiles for the host (plain C++), which the library uses for the short serial
// tails (window fold of an MSM, G2 line precomputation at key registration).
#pragma once
#include <string.h>
#include <stdint.h>
#include "constants.hpp"

#ifndef MI_COMBA_LANES
#define MI_COMBA_LANES 1
#endif
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define MI_HD __host__ __device__ __forceinline__
#define MI_D __device__ __forceinline__
#else
#define MI_HD inline
#define MI_D inline
#endif

namespace mi {

struct FqP {
    static constexpr int N = 12;
    static constexpr uint64_t INV64 = MI_FQ_INV;
    static constexpr uint32_t MOD[12] = {0x00000001u, 0x8508c000u, 0x30000000u, 0x170b5d44u, 0xba094800u, 0x1ef3622fu,
                                         0x00f5138fu, 0x1a22d9f3u, 0x6ca1493bu, 0xc63b05c0u, 0x17c510eau, 0x01ae3a46u};
    // R mod q
    static constexpr uint32_t ONE[12] = {0xffffff68u, 0x02cdffffu, 0x7fffffb1u, 0x51409f83u, 0x8a7d3ff2u, 0x9f7db3a9u,
                                         0x6e7c6305u, 0x7b4e97b7u, 0x803c84e8u, 0x4cf495bfu, 0xe2fdf49au, 0x008d6661u};
    static constexpr uint32_t R2[12] = {0x9400cd22u, 0xb786686cu, 0xb00431b1u, 0x0329fcaau, 0x62d6b46du, 0x22a5f111u,
                                        0x827dc3acu, 0xbfdf7d03u, 0x41790bf9u, 0x837e92f0u, 0x1e914b88u, 0x006dfccbu};
};
struct FrP {
    static constexpr int N = 8;
    static constexpr uint64_t INV64 = MI_FR_INV;
    static constexpr uint32_t MOD[8] = {0x00000001u, 0x0a118000u, 0xd0000001u, 0x59aa76feu,
                                        0x5c37b001u, 0x60b44d1eu, 0x9a2ca556u, 0x12ab655eu};
    static constexpr uint32_t ONE[8] = {0xfffffff3u, 0x7d1c7fffu, 0x6ffffff2u, 0x7257f50fu,
                                        0x512c0feeu, 0x16d81575u, 0x2bbb9a9du, 0x0d4bda32u};
    static constexpr uint32_t R2[8] = {0xb861857bu, 0x25d577bau, 0x8860591fu, 0xcc2c27b5u,
                                       0xe5dc8593u, 0xa7cc008fu, 0xeff1c939u, 0x011fdae7u};
};

template <class P>
struct Fp {
    using Params = P;
    static constexpr int N = P::N;
    uint32_t l[N];

    MI_HD static Fp zero() {
        Fp r;
#pragma unroll
        for (int i = 0; i < N; i++) r.l[i] = 0;
        return r;
    }
    MI_HD static Fp one() {
        Fp r;
#pragma unroll
        for (int i = 0; i < N; i++) r.l[i] = P::ONE[i];
        return r;
    }
    MI_HD static Fp r2() {
        Fp r;
#pragma unroll
        for (int i = 0; i < N; i++) r.l[i] = P::R2[i];
        return r;
    }
    MI_HD bool is_zero() const {
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < N; i++) o |= l[i];
        return o == 0;
    }
    MI_HD bool operator==(const Fp &b) const {
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < N; i++) o |= l[i] ^ b.l[i];
        return o == 0;
    }
    MI_HD bool operator!=(const Fp &b) const { return !(*this == b); }
};

// add / subtract with carry, one instruction per limb on the device (v_addc_co_u32 / v_subb_co_u32 chains)
#if defined(__clang__)
MI_HD uint32_t addc32(uint32_t a, uint32_t b, uint32_t cin, uint32_t *cout) {
    unsigned c;
    uint32_t r = __builtin_addc(a, b, cin, &c);
    *cout = c;
    return r;
}
MI_HD uint32_t subc32(uint32_t a, uint32_t b, uint32_t bin, uint32_t *bout) {
    unsigned c;
    uint32_t r = __builtin_subc(a, b, bin, &c);
    *bout = c;
    return r;
}
#else
MI_HD uint32_t addc32(uint32_t a, uint32_t b, uint32_t cin, uint32_t *cout) {
    uint64_t s = (uint64_t)a + b + cin;
    *cout = (uint32_t)(s >> 32);
    return (uint32_t)s;
}
MI_HD uint32_t subc32(uint32_t a, uint32_t b, uint32_t bin, uint32_t *bout) {
    uint64_t d = (uint64_t)a - b - bin;
    *bout = (uint32_t)(d >> 63);
    return (uint32_t)d;
}
#endif

// r = a - MOD if a >= MOD (a < 2*MOD on entry)
template <class P>
MI_HD void reduce_once(Fp<P> &a) {
    constexpr int N = P::N;
    uint32_t t[N];
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < N; i++) t[i] = subc32(a.l[i], P::MOD[i], borrow, &borrow);
#pragma unroll
    for (int i = 0; i < N; i++) a.l[i] = borrow ? a.l[i] : t[i];
}

template <class P>
MI_HD Fp<P> add(const Fp<P> &a, const Fp<P> &b) {
    constexpr int N = P::N;
    Fp<P> r;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < N; i++) r.l[i] = addc32(a.l[i], b.l[i], c, &c);
    reduce_once(r);  // both moduli leave spare top bits: no carry out of the top limb
    return r;
}

template <class P>
MI_HD Fp<P> sub(const Fp<P> &a, const Fp<P> &b) {
    constexpr int N = P::N;
    Fp<P> r;
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < N; i++) r.l[i] = subc32(a.l[i], b.l[i], borrow, &borrow);
    // add back MOD masked by the borrow (branch-free: lanes diverge on data)
    const uint32_t mask = 0u - borrow;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < N; i++) r.l[i] = addc32(r.l[i], P::MOD[i] & mask, c, &c);
    return r;
}

template <class P>
MI_HD Fp<P> neg(const Fp<P> &a) {
    return sub(Fp<P>::zero(), a);
}
template <class P>
MI_HD Fp<P> dbl(const Fp<P> &a) {
    return add(a, a);
}

// Montgomery product a*b*R^-1 mod p, fully reduced.  Operand-scanning CIOS on 32-bit limbs; the
// modulus is 1 mod 2^32, so m = -t0 and m*MOD[0] + t0 contributes only its carry (1 when t0 != 0).
#if !defined(__HIP_DEVICE_COMPILE__)
// host build of the same product: 64-bit limbs and a 128-bit accumulator (the short serial tails that run on
// the host -- window fold, G2 line preparation -- are ~3x faster this way than with the 32-bit limb code)
template <class P>
inline Fp<P> mul_host64(const Fp<P> &a, const Fp<P> &b) {
    constexpr int M = P::N / 2;
    uint64_t A[M], B[M], Q[M], t[M + 1];
    memcpy(A, a.l, sizeof A);  // the same little-endian bytes, as 64-bit limbs
    memcpy(B, b.l, sizeof B);
    for (int i = 0; i < M; i++) {
        Q[i] = ((uint64_t)P::MOD[2 * i + 1] << 32) | P::MOD[2 * i];
        t[i] = 0;
    }
    t[M] = 0;
    for (int i = 0; i < M; i++) {
        unsigned __int128 acc;
        uint64_t carry = 0;
        for (int j = 0; j < M; j++) {
            acc = (unsigned __int128)A[j] * B[i] + t[j] + carry;
            t[j] = (uint64_t)acc;
            carry = (uint64_t)(acc >> 64);
        }
        t[M] = carry;
        uint64_t m = t[0] * P::INV64;
        acc = (unsigned __int128)m * Q[0] + t[0];
        carry = (uint64_t)(acc >> 64);
        for (int j = 1; j < M; j++) {
            acc = (unsigned __int128)m * Q[j] + t[j] + carry;
            t[j - 1] = (uint64_t)acc;
            carry = (uint64_t)(acc >> 64);
        }
        t[M - 1] = t[M] + carry;
    }
    // one conditional subtraction, in 64-bit limbs
    uint64_t d[M];
    unsigned __int128 bw = 0;
    for (int i = 0; i < M; i++) {
        unsigned __int128 x = (unsigned __int128)t[i] - Q[i] - (uint64_t)bw;
        d[i] = (uint64_t)x;
        bw = (x >> 64) & 1;
    }
    Fp<P> r;
    memcpy(r.l, bw ? t : d, sizeof A);
    return r;
}
#endif

template <class P>
MI_HD Fp<P> mul32(const Fp<P> &a, const Fp<P> &b) {
    constexpr int N = P::N;
    uint32_t t[N + 1];
#pragma unroll
    for (int i = 0; i <= N; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        uint32_t bi = b.l[i];
        uint64_t acc = 0;
#pragma unroll
        for (int j = 0; j < N; j++) {
            acc = (uint64_t)a.l[j] * bi + t[j] + (acc >> 32);
            t[j] = (uint32_t)acc;
        }
        t[N] = (uint32_t)(acc >> 32);
        uint32_t m = 0u - t[0];
        acc = (uint64_t)(t[0] != 0);  // carry of m*1 + t0
#pragma unroll
        for (int j = 1; j < N; j++) {
            acc = (uint64_t)m * P::MOD[j] + t[j] + (uint32_t)acc;
            t[j - 1] = (uint32_t)acc;
            acc >>= 32;
        }
        t[N - 1] = t[N] + (uint32_t)acc;
    }
    Fp<P> r;
#pragma unroll
    for (int i = 0; i < N; i++) r.l[i] = t[i];
    reduce_once(r);
    return r;
}

}  // namespace mi
#if defined(__HIP_DEVICE_COMPILE__)
// Device product: finely integrated product scanning (Comba), generated fully unrolled by tools/gen_comba.py.
// Column k sums a_i b_(k-i) and m_i p_(k-i) into a 96-bit accumulator {ovf, acc}: v_mad_u64_u32 adds the 64-bit
// product into acc and leaves the carry in VCC, v_addc_co_u32 folds it into ovf: 2 VALU instructions per limb
// product and no register shuffling, against ~4.4 for the compiler's code from mul32 (v_mad + 64-bit add + two
// v_mov to build the zero-extended addend).  LANES = 2 splits each column over two accumulators.
#include "comba_gen.hpp"
#endif
namespace mi {

// Fq product WITHOUT the final conditional subtraction (device: result < q (1 + q/R) for canonical inputs, to be
// summed lazily and reduced once); the host build returns the reduced product.
MI_HD Fp<FqP> mul_nr(const Fp<FqP> &a, const Fp<FqP> &b);

template <class P>
MI_HD Fp<P> mul(const Fp<P> &a, const Fp<P> &b) {
#if !defined(__HIP_DEVICE_COMPILE__)
    return mul_host64(a, b);
#elif defined(MI_MUL32)
    return mul32(a, b);
#else
    return mul_comba_gen<P, MI_COMBA_LANES>(a, b);
#endif
}
MI_HD Fp<FqP> mul_nr(const Fp<FqP> &a, const Fp<FqP> &b) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MI_MUL32)
    return mul_comba_nr_gen<FqP>(a, b);
#else
    return mul(a, b);
#endif
}

// a^2: the device has a dedicated squaring (n (n + 1) / 2 limb products for the square, the same reduction: ~0.78 of a
// product's instructions); the value is the product's, bit for bit
template <class P>
MI_HD Fp<P> sqr(const Fp<P> &a) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MI_MUL32)
    return sqr_comba_gen<P>(a);
#else
    return mul(a, a);
#endif
}

template <class P>
MI_HD Fp<P> operator+(const Fp<P> &a, const Fp<P> &b) { return add(a, b); }
template <class P>
MI_HD Fp<P> operator-(const Fp<P> &a, const Fp<P> &b) { return sub(a, b); }
template <class P>
MI_HD Fp<P> operator*(const Fp<P> &a, const Fp<P> &b) { return mul(a, b); }

// a^e for a little-endian 32-bit-limb exponent (square-and-multiply, MSB first)
template <class P>
MI_HD Fp<P> pow_limbs(const Fp<P> &a, const uint32_t *e, int n_limbs) {
    Fp<P> acc = Fp<P>::one();
    bool started = false;
    for (int i = n_limbs * 32 - 1; i >= 0; i--) {
        if (started) acc = sqr(acc);
        if ((e[i >> 5] >> (i & 31)) & 1u) {
            if (started) acc = mul(acc, a);
            else { acc = a; started = true; }
        }
    }
    return acc;
}

// Fermat inverse a^(p-2); inv(0) = 0.  The exponent is MOD - 2 and MOD[0] = 1, so the low limb
// borrows: (MOD - 2) = {0xffffffff, MOD[1]-1, MOD[2], ...}.
template <class P>
MI_HD Fp<P> inv(const Fp<P> &a) {
    uint32_t e[P::N];
    e[0] = 0xffffffffu;
    e[1] = P::MOD[1] - 1u;
    for (int i = 2; i < P::N; i++) e[i] = P::MOD[i];
    return pow_limbs(a, e, P::N);
}

// canonical little-endian integer (< p) -> Montgomery
template <class P>
MI_HD Fp<P> to_mont(const Fp<P> &canonical) {
    return mul(canonical, Fp<P>::r2());
}
template <class P>
MI_HD Fp<P> from_mont(const Fp<P> &m) {
    Fp<P> one = Fp<P>::zero();
    one.l[0] = 1;
    return mul(m, one);
}
template <class P>
MI_HD Fp<P> from_u32(uint32_t v) {
    Fp<P> t = Fp<P>::zero();
    t.l[0] = v;
    return to_mont(t);
}

using Fq = Fp<FqP>;
using Fr = Fp<FrP>;

// Out-of-line copies: long serial chains (pairing tower, reductions, scalar multiplications) call these
// instead of inlining a ~600-instruction product at every site.
#if defined(__HIPCC__)
#define MI_NI __host__ __device__ __attribute__((noinline))
#else
#define MI_NI __attribute__((noinline))
#endif
// operands by value: 24 + 12 VGPRs travel in registers (by reference they would go through scratch memory)
inline MI_NI Fq mulq(Fq a, Fq b) { return mul(a, b); }
inline MI_NI Fq sqrq(Fq a) { return sqr(a); }
// Inverse by the binary extended Euclid algorithm (shifts and subtractions only: ~10x fewer instructions
// than the Fermat power on one lane).  Input and output in Montgomery form; inv(0) = 0.
template <class P>
MI_HD Fp<P> inv_egcd(const Fp<P> &a) {
    constexpr int N = P::N;
    if (a.is_zero()) return a;
    uint32_t u[N], v[N], x1[N], x2[N];
#pragma unroll
    for (int i = 0; i < N; i++) { u[i] = a.l[i]; v[i] = P::MOD[i]; x1[i] = 0; x2[i] = 0; }
    x1[0] = 1;
    auto is_one = [](const uint32_t *t) {
        uint32_t o = t[0] ^ 1u;
#pragma unroll
        for (int i = 1; i < N; i++) o |= t[i];
        return o == 0;
    };
    auto shr1 = [](uint32_t *t, uint32_t top) {
#pragma unroll
        for (int i = 0; i < N - 1; i++) t[i] = (t[i] >> 1) | (t[i + 1] << 31);
        t[N - 1] = (t[N - 1] >> 1) | (top << 31);
    };
    auto halve_mod = [&](uint32_t *x) {  // x <- x/2 mod p
        uint32_t carry = 0;
        if (x[0] & 1u) {
#pragma unroll
            for (int i = 0; i < N; i++) {
                uint64_t t = (uint64_t)x[i] + P::MOD[i] + carry;
                x[i] = (uint32_t)t;
                carry = (uint32_t)(t >> 32);
            }
        }
        shr1(x, carry);
    };
    auto geq = [](const uint32_t *p_, const uint32_t *q_) {  // p >= q
        uint32_t borrow = 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
            uint64_t d = (uint64_t)p_[i] - q_[i] - borrow;
            borrow = (uint32_t)(d >> 63);
        }
        return borrow == 0;
    };
    auto sub_to = [](uint32_t *p_, const uint32_t *q_) {  // p -= q (p >= q)
        uint32_t borrow = 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
            uint64_t d = (uint64_t)p_[i] - q_[i] - borrow;
            p_[i] = (uint32_t)d;
            borrow = (uint32_t)(d >> 63);
        }
    };
    auto sub_mod = [&](uint32_t *p_, const uint32_t *q_) {  // p <- p - q mod MOD, both < MOD
        uint32_t borrow = 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
            uint64_t d = (uint64_t)p_[i] - q_[i] - borrow;
            p_[i] = (uint32_t)d;
            borrow = (uint32_t)(d >> 63);
        }
        uint32_t mask = 0u - borrow, c = 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
            uint64_t t = (uint64_t)p_[i] + (P::MOD[i] & mask) + c;
            p_[i] = (uint32_t)t;
            c = (uint32_t)(t >> 32);
        }
    };
    for (int guard = 0; guard < 4 * 32 * N; guard++) {
        if (is_one(u) || is_one(v)) break;
        while (!(u[0] & 1u)) { shr1(u, 0); halve_mod(x1); }
        while (!(v[0] & 1u)) { shr1(v, 0); halve_mod(x2); }
        if (geq(u, v)) { sub_to(u, v); sub_mod(x1, x2); }
        else { sub_to(v, u); sub_mod(x2, x1); }
    }
    Fp<P> r;
    const bool use1 = is_one(u);
#pragma unroll
    for (int i = 0; i < N; i++) r.l[i] = use1 ? x1[i] : x2[i];
    // r = (aR)^-1 = a^-1 R^-1; two Montgomery products by R^2 give a^-1 R
    return mul(mul(r, Fp<P>::r2()), Fp<P>::r2());
}
// r = | sf ? -(x fa) : (x fa)  +  sg ? -(y ga) : (y ga) |, N + 1 limbs; returns the sign of the combination
template <int N>
MI_HD bool bingcd_combine(const uint32_t *xx, uint32_t fa, bool sf, const uint32_t *yy, uint32_t ga, bool sg, uint32_t *r) {
    uint32_t tx[N + 1], ty[N + 1];
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < N; i++) { c = (uint64_t)xx[i] * fa + (uint32_t)c; tx[i] = (uint32_t)c; c >>= 32; }
    tx[N] = (uint32_t)c;
    c = 0;
#pragma unroll
    for (int i = 0; i < N; i++) { c = (uint64_t)yy[i] * ga + (uint32_t)c; ty[i] = (uint32_t)c; c >>= 32; }
    ty[N] = (uint32_t)c;
    if (sf == sg) {
        uint32_t cy = 0;
#pragma unroll
        for (int i = 0; i <= N; i++) r[i] = addc32(tx[i], ty[i], cy, &cy);
        return sf;
    }
    uint32_t bw = 0;
#pragma unroll
    for (int i = 0; i <= N; i++) r[i] = subc32(tx[i], ty[i], bw, &bw);
    if (!bw) return sf;  // |x fa| >= |y ga|: the sign of the first term
    uint32_t cy = 1;     // negate
#pragma unroll
    for (int i = 0; i <= N; i++) r[i] = addc32(~r[i], 0u, cy, &cy);
    return sg;
}
// out = (neg ? -w : w) / 2^K mod p for w < 2^K p (N + 1 limbs), K = 30
template <class P>
MI_HD void bingcd_mont_shift(uint32_t *w, bool neg, uint32_t *out) {
    constexpr int N = P::N, K = 30;
    constexpr uint32_t MASK = (1u << K) - 1u;
        const uint32_t tq = (0u - w[0]) & MASK;  // w + tq p = 0 mod 2^K (p = 1 mod 2^32)
        uint64_t c = 0;
        uint32_t cy = 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
            c = (uint64_t)tq * P::MOD[i] + (uint32_t)c;
            w[i] = addc32(w[i], (uint32_t)c, cy, &cy);
            c >>= 32;
        }
        w[N] = w[N] + (uint32_t)c + cy;  // < 2^(K+1) p / 2^(32 N): no overflow
        uint32_t r[N], d[N], bw = 0;
#pragma unroll
        for (int i = 0; i < N; i++) r[i] = (w[i] >> K) | (w[i + 1] << (32 - K));  // < 2 p
#pragma unroll
        for (int i = 0; i < N; i++) d[i] = subc32(r[i], P::MOD[i], bw, &bw);
#pragma unroll
        for (int i = 0; i < N; i++) r[i] = bw ? r[i] : d[i];  // < p
        uint32_t nz = 0;
#pragma unroll
        for (int i = 0; i < N; i++) nz |= r[i];
        if (neg && nz) {
            bw = 0;
#pragma unroll
            for (int i = 0; i < N; i++) r[i] = subc32(P::MOD[i], r[i], bw, &bw);
        }
#pragma unroll
        for (int i = 0; i < N; i++) out[i] = r[i];
}
// Inverse by the approximated binary GCD of Pornin ("Optimized Binary GCD for Modular Inversion", 2020, Alg. 2) on
// 32-bit limbs: each outer iteration runs K = 30 plain binary-GCD steps on 62-bit approximations of (a, b) -- their low
// 30 bits and their top 32 bits -- which yields update factors |f|, |g| <= 2^30; the factors are then applied once to
// the full-size a, b (exact division by 2^30) and to the Bezout coefficients u, v (division by 2^30 modulo p, the
// Montgomery way: both moduli are 1 mod 2^32).  len(a) + len(b) drops by >= 30 bits per iteration, so
// ceil((2 len(p) - 1) / 30) iterations always suffice (26 for Fq, 17 for Fr; the loop leaves as soon as a = 0:
// 18-20 in practice).  One lane of a pairing workgroup runs this for the norm of the Miller value: the plain binary
// EGCD above took 300-400 us there (a quarter of the whole pairing product); this one takes ~1.5 k instructions per
// iteration.  Input and output in Montgomery form; inv(0) = 0.  Pinned against the oracle in tests/test_host_arith.py.
template <class P>
MI_HD Fp<P> inv_bingcd(const Fp<P> &x) {
    constexpr int N = P::N, K = 30;
    constexpr uint32_t MASK = (1u << K) - 1u;
    constexpr int BITS = N == 12 ? 377 : 253;
    constexpr int ITERS = (2 * BITS - 1 + K - 1) / K;
    if (x.is_zero()) return x;
    uint32_t a[N], b[N], u[N], v[N];
#pragma unroll
    for (int i = 0; i < N; i++) { a[i] = x.l[i]; b[i] = P::MOD[i]; u[i] = 0; v[i] = 0; }
    u[0] = 1;
    for (int it = 0; it < ITERS; it++) {
        // ---- the loop ends with a = 0, b = gcd = 1
        uint32_t anz = 0;
#pragma unroll
        for (int i = 0; i < N; i++) anz |= a[i];
        if (anz == 0) break;
        // ---- approximations: three-word window at the top non-zero limb of a | b, aligned to its top bit
        uint32_t ah = 0, am = 0, al = 0, bh = 0, bm = 0, bl = 0;
        bool found = false, small = true;  // small: a and b fit in two limbs
#pragma unroll
        for (int i = N - 1; i >= 2; i--) {
            const bool here = !found && ((a[i] | b[i]) != 0);
            if (here) {
                ah = a[i]; am = a[i - 1]; al = a[i - 2];
                bh = b[i]; bm = b[i - 1]; bl = b[i - 2];
                found = true;
                small = false;
            }
        }
        uint64_t abar, bbar;
        if (small) {
            const uint64_t a64 = ((uint64_t)a[1] << 32) | a[0], b64 = ((uint64_t)b[1] << 32) | b[0];
            const uint64_t top = a64 | b64;
            if ((top >> (2 * K + 2)) == 0) {  // at most 62 bits: exact
                abar = a64;
                bbar = b64;
            } else {                          // 63 or 64 bits
                const int sh = (top >> 63) ? 32 : 31;  // n - 32
                abar = (a64 & MASK) | ((a64 >> sh) << K);
                bbar = (b64 & MASK) | ((b64 >> sh) << K);
            }
        } else {
            const uint32_t topw = ah | bh;
            const int s = __builtin_clz(topw);  // topw != 0 here
            const uint32_t ta = s ? ((ah << s) | (am >> (32 - s))) : ah, tb = s ? ((bh << s) | (bm >> (32 - s))) : bh;
            (void)al; (void)bl;  // the top 32 bits never reach the third word
            abar = (uint64_t)(a[0] & MASK) | ((uint64_t)ta << K);
            bbar = (uint64_t)(b[0] & MASK) | ((uint64_t)tb << K);
        }
        // ---- K binary-GCD steps on the approximations
        int32_t f0 = 1, g0 = 0, f1 = 0, g1 = 1;
        for (int j = 0; j < K; j++) {
            if (abar & 1u) {
                if (abar < bbar) {
                    const uint64_t tt = abar; abar = bbar; bbar = tt;
                    int32_t ti = f0; f0 = f1; f1 = ti;
                    ti = g0; g0 = g1; g1 = ti;
                }
                abar -= bbar;
                f0 -= f1;
                g0 -= g1;
            }
            abar >>= 1;
            f1 <<= 1;
            g1 <<= 1;
        }
        // ---- (a, b) <- (|a f0 + b g0|, |a f1 + b g1|) / 2^K, exact; a flipped sign flips the row of factors
        const uint32_t f0a = (uint32_t)(f0 < 0 ? -f0 : f0), g0a = (uint32_t)(g0 < 0 ? -g0 : g0);
        const uint32_t f1a = (uint32_t)(f1 < 0 ? -f1 : f1), g1a = (uint32_t)(g1 < 0 ? -g1 : g1);
        uint32_t ra[N + 1], rb[N + 1];
        const bool na = bingcd_combine<N>(a, f0a, f0 < 0, b, g0a, g0 < 0, ra);
        const bool nb = bingcd_combine<N>(a, f1a, f1 < 0, b, g1a, g1 < 0, rb);
#pragma unroll
        for (int i = 0; i < N; i++) {
            a[i] = (ra[i] >> K) | (ra[i + 1] << (32 - K));
            b[i] = (rb[i] >> K) | (rb[i + 1] << (32 - K));
        }
        // ---- (u, v) <- ((u f0 + v g0), (u f1 + v g1)) / 2^K mod p, rows negated where a / b were
        uint32_t ru[N + 1], rv[N + 1];
        const bool su = bingcd_combine<N>(u, f0a, (f0 < 0) != na, v, g0a, (g0 < 0) != na, ru);
        const bool sv = bingcd_combine<N>(u, f1a, (f1 < 0) != nb, v, g1a, (g1 < 0) != nb, rv);
        bingcd_mont_shift<P>(ru, su, u);
        bingcd_mont_shift<P>(rv, sv, v);
    }
    // b = 1 and v = (x R)^-1 = x^-1 R^-1 as an integer mod p; two Montgomery products by R^2 give x^-1 R
    Fp<P> r;
#pragma unroll
    for (int i = 0; i < N; i++) r.l[i] = v[i];
    return mul(mul(r, Fp<P>::r2()), Fp<P>::r2());
}
inline MI_NI Fq invq(const Fq &a) { return inv_bingcd(a); }
inline MI_NI Fr invr(const Fr &a) { return inv_bingcd(a); }
inline MI_NI Fr mulr(Fr a, Fr b) { return mul(a, b); }

static_assert(sizeof(Fq) == 48 && sizeof(Fr) == 32, "ABI layout");

}  // namespace mi
