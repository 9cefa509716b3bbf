// Host side of the small-MSM reduction: Horner over per-bit sums on the CPU (msm.hip msm_finish_all).
//
// The chain is serial (one doubling per bit position of the scalars, ~250 for the verifier) and sits between the
// last MSM kernel and the pairing launch of every batch, so it is written for the host it runs on: Fq on six 64-bit
// limbs with a BMI2/ADX Montgomery product (two carry chains: mulx + adcx/adox; the compilers' code from the 128-bit
// accumulator form of field.hpp runs ~1.4x longer), 64-bit additions, and Jacobian doublings (2M + 5S) for the pure
// shifts where the XYZZ doubling costs 6M + 3S.  Values are the library's: same Montgomery form, same bytes as Fq /
// G1Xyzz; the result leaves through to_affine, so the choice of formulas cannot change an output bit.
// Host only: the device pass of a .hip file sees the types and declarations (it parses the host functions that use
// them), the bodies -- x86 inline assembly among them -- exist in the host pass alone.
#pragma once
#include <stdlib.h>
#include "curve.hpp"

namespace mi {
namespace hostfold {
struct H6 {
    uint64_t l[6];
};
static_assert(sizeof(H6) == sizeof(Fq), "same bytes");
struct PX {  // extended Jacobian, as G1Xyzz
    H6 x, y, zz, zzz;
    inline bool is_inf() const { return (zz.l[0] | zz.l[1] | zz.l[2] | zz.l[3] | zz.l[4] | zz.l[5]) == 0; }
};
#if defined(__HIP_DEVICE_COMPILE__)
#define MI_HF_DECL
#else
#define MI_HF_DECL inline
#endif
MI_HF_DECL PX px(const G1Xyzz &p);
MI_HF_DECL G1Xyzz xyzz(const PX &p);
MI_HF_DECL PX add(const PX &a, const PX &b);
MI_HF_DECL PX shift(const PX &p, size_t k);
MI_HF_DECL PX fold_bits(const G1Xyzz *h, size_t lo, size_t hi);
#undef MI_HF_DECL
}  // namespace hostfold
}  // namespace mi

#if !defined(__HIP_DEVICE_COMPILE__)
namespace mi {
namespace hostfold {

inline constexpr uint64_t q64(int i) { return ((uint64_t)FqP::MOD[2 * i + 1] << 32) | FqP::MOD[2 * i]; }
alignas(64) static const uint64_t HQ[6] = {q64(0), q64(1), q64(2), q64(3), q64(4), q64(5)};

inline H6 h6(const Fq &a) {
    H6 r;
    memcpy(r.l, a.l, sizeof r.l);
    return r;
}
inline Fq fq(const H6 &a) {
    Fq r;
    memcpy(r.l, a.l, sizeof a.l);
    return r;
}

#if defined(__x86_64__)
#define MI_HOSTFOLD_ADX 1
// t (seven words, the top one zero on entry) += x[0..5] * m: low halves on the CF chain, high halves on the OF chain
#define MI_HF_MULADD6(x)                                                                                               \
    asm("xorl %%eax, %%eax\n\t"                                                                                       \
        "mulxq 0(%[p]), %%rax, %%rbx\n\t"                                                                             \
        "adcxq %%rax, %[t0]\n\t"                                                                                      \
        "adoxq %%rbx, %[t1]\n\t"                                                                                      \
        "mulxq 8(%[p]), %%rax, %%rbx\n\t"                                                                             \
        "adcxq %%rax, %[t1]\n\t"                                                                                      \
        "adoxq %%rbx, %[t2]\n\t"                                                                                      \
        "mulxq 16(%[p]), %%rax, %%rbx\n\t"                                                                            \
        "adcxq %%rax, %[t2]\n\t"                                                                                      \
        "adoxq %%rbx, %[t3]\n\t"                                                                                      \
        "mulxq 24(%[p]), %%rax, %%rbx\n\t"                                                                            \
        "adcxq %%rax, %[t3]\n\t"                                                                                      \
        "adoxq %%rbx, %[t4]\n\t"                                                                                      \
        "mulxq 32(%[p]), %%rax, %%rbx\n\t"                                                                            \
        "adcxq %%rax, %[t4]\n\t"                                                                                      \
        "adoxq %%rbx, %[t5]\n\t"                                                                                      \
        "mulxq 40(%[p]), %%rax, %%rbx\n\t"                                                                            \
        "adcxq %%rax, %[t5]\n\t"                                                                                      \
        "adoxq %%rbx, %[t6]\n\t"                                                                                      \
        "movl $0, %%eax\n\t"                                                                                          \
        "adcxq %%rax, %[t6]\n\t"                                                                                      \
        : [t0] "+r"(t0), [t1] "+r"(t1), [t2] "+r"(t2), [t3] "+r"(t3), [t4] "+r"(t4), [t5] "+r"(t5), [t6] "+r"(t6)     \
        : [p] "r"(x), "d"(m), "m"(*(const uint64_t(*)[6])(x))                                                         \
        : "rax", "rbx", "cc")

// Montgomery product, CIOS: t += a b_i; t += (-t0 / q mod 2^64) q; t >>= 64.  t < 2q throughout (q < 2^377).
__attribute__((target("bmi2,adx"))) inline H6 mul_adx(const H6 &a, const H6 &b) {
    uint64_t t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0, t6 = 0, m;
    const uint64_t *ap = a.l, *qp = HQ;
    for (int i = 0; i < 6; i++) {
        m = b.l[i];
        MI_HF_MULADD6(ap);
        const uint64_t top = t6;
        t6 = 0;
        m = t0 * FqP::INV64;
        MI_HF_MULADD6(qp);  // t0 becomes zero
        t0 = t1; t1 = t2; t2 = t3; t3 = t4; t4 = t5; t5 = t6 + top; t6 = 0;
    }
    uint64_t d[6], t[6] = {t0, t1, t2, t3, t4, t5}, bw = 0;
    for (int i = 0; i < 6; i++) {
        const unsigned __int128 x = (unsigned __int128)t[i] - HQ[i] - bw;
        d[i] = (uint64_t)x;
        bw = (uint64_t)(x >> 64) & 1;
    }
    H6 r;
    for (int i = 0; i < 6; i++) r.l[i] = bw ? t[i] : d[i];
    return r;
}
#undef MI_HF_MULADD6
inline bool have_adx() {  // MI_SNARK_HOSTFOLD_NO_ADX=1: the portable product (tests run both)
    static const bool ok = __builtin_cpu_supports("bmi2") && __builtin_cpu_supports("adx") && !getenv("MI_SNARK_HOSTFOLD_NO_ADX");
    return ok;
}
#endif

inline H6 mul(const H6 &a, const H6 &b) {
#if defined(MI_HOSTFOLD_ADX)
    if (have_adx()) return mul_adx(a, b);
#endif
    return h6(mi::mul(fq(a), fq(b)));
}
inline H6 sqr(const H6 &a) { return mul(a, a); }
inline H6 add(const H6 &a, const H6 &b) {
    uint64_t s[6], d[6], c = 0, bw = 0;
    for (int i = 0; i < 6; i++) {
        const unsigned __int128 x = (unsigned __int128)a.l[i] + b.l[i] + c;
        s[i] = (uint64_t)x;
        c = (uint64_t)(x >> 64);
    }
    for (int i = 0; i < 6; i++) {  // no carry out of the top limb: 2q < 2^384
        const unsigned __int128 x = (unsigned __int128)s[i] - HQ[i] - bw;
        d[i] = (uint64_t)x;
        bw = (uint64_t)(x >> 64) & 1;
    }
    H6 r;
    for (int i = 0; i < 6; i++) r.l[i] = bw ? s[i] : d[i];
    return r;
}
inline H6 sub(const H6 &a, const H6 &b) {
    uint64_t d[6], bw = 0, c = 0;
    for (int i = 0; i < 6; i++) {
        const unsigned __int128 x = (unsigned __int128)a.l[i] - b.l[i] - bw;
        d[i] = (uint64_t)x;
        bw = (uint64_t)(x >> 64) & 1;
    }
    const uint64_t mask = 0 - bw;
    H6 r;
    for (int i = 0; i < 6; i++) {
        const unsigned __int128 x = (unsigned __int128)d[i] + (HQ[i] & mask) + c;
        r.l[i] = (uint64_t)x;
        c = (uint64_t)(x >> 64);
    }
    return r;
}
inline H6 dbl(const H6 &a) { return add(a, a); }
inline bool is_zero(const H6 &a) { return (a.l[0] | a.l[1] | a.l[2] | a.l[3] | a.l[4] | a.l[5]) == 0; }

inline PX px(const G1Xyzz &p) { return PX{h6(p.x), h6(p.y), h6(p.zz), h6(p.zzz)}; }
inline G1Xyzz xyzz(const PX &p) { return G1Xyzz{fq(p.x), fq(p.y), fq(p.zz), fq(p.zzz)}; }
inline PX px_inf() { return px(G1Xyzz::inf()); }

// dbl-2008-s-1, a = 0
inline PX dbl(const PX &p) {
    if (p.is_inf()) return p;
    const H6 u = dbl(p.y), v = sqr(u), w = mul(u, v), s = mul(p.x, v), xx = sqr(p.x), m = add(dbl(xx), xx);
    const H6 x3 = sub(sqr(m), dbl(s)), y3 = sub(mul(m, sub(s, x3)), mul(w, p.y));
    return PX{x3, y3, mul(v, p.zz), mul(w, p.zzz)};
}
// add-2008-s, every special case
inline PX add(const PX &a, const PX &b) {
    if (a.is_inf()) return b;
    if (b.is_inf()) return a;
    const H6 u1 = mul(a.x, b.zz), u2 = mul(b.x, a.zz), s1 = mul(a.y, b.zzz), s2 = mul(b.y, a.zzz);
    const H6 p = sub(u2, u1), r = sub(s2, s1);
    if (is_zero(p)) return is_zero(r) ? dbl(a) : px_inf();
    const H6 pp = sqr(p), ppp = mul(p, pp), qq = mul(u1, pp);
    const H6 x3 = sub(sub(sqr(r), ppp), dbl(qq)), y3 = sub(mul(r, sub(qq, x3)), mul(s1, ppp));
    return PX{x3, y3, mul(mul(a.zz, b.zz), pp), mul(mul(a.zzz, b.zzz), ppp)};
}
// p * 2^k.  The same point as Jacobian (X ZZ, Y ZZZ, ZZ) -- x = X / ZZ = X ZZ / ZZ^2, y = Y / ZZZ = Y ZZZ / ZZ^3 since
// ZZ^3 = ZZZ^2 -- takes dbl-2009-l doublings (2M + 5S against 6M + 3S), and (X, Y, Z) is (X, Y, Z^2, Z^3) again.
inline PX shift(const PX &p, size_t k) {
    if (p.is_inf() || k == 0) return p;
    H6 X = mul(p.x, p.zz), Y = mul(p.y, p.zzz), Z = p.zz;
    for (size_t j = 0; j < k; j++) {
        const H6 A = sqr(X), B = sqr(Y), C = sqr(B);
        H6 D = sub(sub(sqr(add(X, B)), A), C);
        D = dbl(D);
        const H6 E = add(dbl(A), A), F = sqr(E);
        const H6 X3 = sub(F, dbl(D));
        const H6 C8 = dbl(dbl(dbl(C)));
        const H6 Z3 = dbl(mul(Y, Z));
        Y = sub(mul(E, sub(D, X3)), C8);
        X = X3;
        Z = Z3;
    }
    const H6 zz = sqr(Z);
    return PX{X, Y, zz, mul(zz, Z)};
}
// Horner over the per-bit sums h[lo..hi): sum_g 2^(g - lo) h[g]
inline PX fold_bits(const G1Xyzz *h, size_t lo, size_t hi) {
    PX acc = px_inf();
    for (size_t g = hi; g-- > lo;) {
        acc = dbl(acc);
        if (!h[g].is_inf()) acc = add(acc, px(h[g]));
    }
    return acc;
}

}  // namespace hostfold
}  // namespace mi
#endif
