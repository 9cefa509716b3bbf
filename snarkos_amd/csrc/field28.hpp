// Reduced-radix Fq for the bucket accumulation: 14 SIGNED limbs of 28 bits (value = sum l_i 2^(28 i)).
//
// The 32-bit-limb Montgomery product (field.hpp / comba_gen.hpp) spends two VALU instructions per limb product:
// v_mad_u64_u32 and the v_addc_co_u32 that catches the carry out of the 64-bit column accumulator.  With 28-bit limbs a
// whole product-scanning column (<= 14 + 13 products below 2^59) fits a signed 64-bit accumulator: ONE v_mad_i64_i32 per
// limb product, and the carry into the next column is an arithmetic shift.  Signed limbs make additions and
// subtractions limb-wise with neither a carry chain nor a modular correction: values are lazy (a few q either side of
// zero), limbs a few bits over 28; only what feeds the next group operation is carried back (normalise28).
// 392 = 14 * 28 products against 288 + 288 instructions: msm_accumulate, the kernel this is for, is ALU-bound on
// exactly this product.
//
// Montgomery radix here: R' = 2^392.  The library's data is Montgomery for R = 2^384: a loaded x R is read 8 bits
// higher (x R 2^8 = x R' as a lazy value below 2^385, no reduction needed), a result goes back through one product
// with 2^384 and a canonical fix-up (store28).  tools/model_f28.py models every operation limb for limb and asserts
// the accumulator and limb bounds used here; tests/test_f28_model.py runs it, tests/test_gpu_msm.py pins the kernel.
// Device only.
#pragma once
#include "curve.hpp"

#if defined(__HIPCC__)
namespace mi {

struct Fq28 {
    int32_t l[14];
};
constexpr uint32_t F28_MASK = (1u << 28) - 1u;

struct F28Consts {
    uint32_t mod[14], one[14];
};
constexpr uint32_t f28_limb_of(const uint32_t *w12, int i, int shift_up) {
    // bits [28 i - shift_up, 28 i - shift_up + 28) of the 384-bit value w12
    uint32_t out = 0;
    for (int b = 0; b < 28; b++) {
        const int src = 28 * i - shift_up + b;
        if (src >= 0 && src < 384 && ((w12[src >> 5] >> (src & 31)) & 1u)) out |= 1u << b;
    }
    return out;
}
constexpr F28Consts f28_make_consts() {
    F28Consts c{};
    for (int i = 0; i < 14; i++) c.mod[i] = f28_limb_of(FqP::MOD, i, 0);
    // 2^392 mod q = (2^384 mod q) * 2^8 mod q: the library's ONE shifted up, then reduced by repeated subtraction of q
    // in limbs (at most 255 times)
    uint32_t v[14] = {};
    for (int i = 0; i < 14; i++) v[i] = f28_limb_of(FqP::ONE, i, 8);
    for (int round = 0; round < 256; round++) {
        // v >= q ?
        bool ge = true;
        for (int i = 13; i >= 0; i--) {
            if (v[i] != c.mod[i]) {
                ge = v[i] > c.mod[i];
                break;
            }
        }
        if (!ge) break;
        int64_t borrow = 0;
        for (int i = 0; i < 14; i++) {
            int64_t d = (int64_t)v[i] - (int64_t)c.mod[i] - borrow;
            borrow = d < 0 ? 1 : 0;
            if (d < 0) d += (int64_t)1 << 28;
            v[i] = (uint32_t)d;
        }
    }
    for (int i = 0; i < 14; i++) c.one[i] = v[i];
    return c;
}
constexpr F28Consts F28C = f28_make_consts();
static_assert(F28C.mod[0] == 1, "q = 1 mod 2^28: the Montgomery factor of a column is minus its low limb");

}  // namespace mi
#include "f28_gen.hpp"
namespace mi {

// (always inlined: an out-of-line copy shared by a kernel's ten call sites was measured on the streamed verifier --
// 4.9 M against 5.9 M proofs/s: the operands' 28 registers each way cost more than the smaller loop saves)
__device__ __forceinline__ Fq28 mul28(const Fq28 &a, const Fq28 &b) { return mul28_gen(a, b); }
__device__ __forceinline__ Fq28 sqr28(const Fq28 &a) { return sqr28_gen(a); }

__device__ __forceinline__ Fq28 f28_one() {
    Fq28 r;
#pragma unroll
    for (int i = 0; i < 14; i++) r.l[i] = (int32_t)F28C.one[i];
    return r;
}
// x R (canonical, 12 x 32 bits) -> x R' as a lazy value below 2^385: limb i = bits [28 i - 8, 28 i + 20)
__device__ __forceinline__ Fq28 load28(const Fq &x) {
    Fq28 r;
#pragma unroll
    for (int i = 0; i < 14; i++) {
        const int o = 28 * i - 8;
        uint32_t v;
        if (o < 0) {
            v = x.l[0] << 8;
        } else {
            const int w = o >> 5, sh = o & 31;
            v = w < 12 ? x.l[w] >> sh : 0u;
            if (sh > 4 && w + 1 < 12) v |= x.l[w + 1] << (32 - sh);
        }
        r.l[i] = (int32_t)(v & F28_MASK);
    }
    return r;
}
__device__ __forceinline__ Fq28 sub28(const Fq28 &a, const Fq28 &b) {
    Fq28 r;
#pragma unroll
    for (int i = 0; i < 14; i++) r.l[i] = a.l[i] - b.l[i];
    return r;
}
__device__ __forceinline__ Fq28 dbl28(const Fq28 &a) {
    Fq28 r;
#pragma unroll
    for (int i = 0; i < 14; i++) r.l[i] = a.l[i] * 2;
    return r;
}
__device__ __forceinline__ Fq28 neg28(const Fq28 &a) {
    Fq28 r;
#pragma unroll
    for (int i = 0; i < 14; i++) r.l[i] = -a.l[i];
    return r;
}
// carry pass: limbs 0..12 into [0, 2^28), the top limb keeps the sign
__device__ __forceinline__ Fq28 normalise28(const Fq28 &a) {
    Fq28 r = a;
#pragma unroll
    for (int i = 0; i < 13; i++) {
        const int32_t c = r.l[i] >> 28;
        r.l[i] &= (int32_t)F28_MASK;
        r.l[i + 1] += c;
    }
    return r;
}
// may a lazy value in (-k q, k q) be 0 mod q?  k q has low limb k (q = 1 mod 2^28): never misses, hits a non-multiple
// with probability (2k + 1) 2^-28
__device__ __forceinline__ bool maybe_zero28(const Fq28 &a, uint32_t k) {
    return (((uint32_t)a.l[0] + k) & F28_MASK) <= 2 * k;
}
// |a| < 256 q  ->  the library's canonical Montgomery form (x R, 12 x 32 bits)
__device__ __forceinline__ Fq store28(const Fq28 &a) {
    Fq28 to32;
#pragma unroll
    for (int i = 0; i < 14; i++) to32.l[i] = 0;
    to32.l[13] = 1 << 20;  // 2^384
    Fq28 v = normalise28(mul28(a, to32));  // in (-q, 2q)
    const int32_t negm = v.l[13] >> 31;  // all ones if negative
    Fq28 t;
#pragma unroll
    for (int i = 0; i < 14; i++) t.l[i] = v.l[i] + ((int32_t)F28C.mod[i] & negm);
    v = normalise28(t);  // in [0, 2q)
#pragma unroll
    for (int i = 0; i < 14; i++) t.l[i] = v.l[i] - (int32_t)F28C.mod[i];
    t = normalise28(t);
    const int32_t keep = t.l[13] >> 31;  // negative: v < q, keep v
#pragma unroll
    for (int i = 0; i < 14; i++) v.l[i] = (v.l[i] & keep) | (t.l[i] & ~keep);
    // pack 14 x 28 -> 12 x 32
    Fq r;
#pragma unroll
    for (int j = 0; j < 12; j++) {
        const int o = 32 * j, i = o / 28, sh = o % 28;
        uint32_t w = (uint32_t)v.l[i] >> sh;
        w |= (uint32_t)v.l[i + 1] << (28 - sh);
        if (28 - sh + 28 < 32 && i + 2 < 14) w |= (uint32_t)v.l[i + 2] << (56 - sh);
        r.l[j] = w;
    }
    return r;
}

// ---- XYZZ accumulator in this arithmetic (madd-2008-s, the formulas of curve.hpp add_mixed)
struct G1Xyzz28 {
    Fq28 x, y, zz, zzz;  // x, y: normalised lazy values (|x| < 6q, |y| < 3q); zz, zzz: product outputs
};
__device__ __forceinline__ G1Xyzz xyzz28_store(const G1Xyzz28 &a) { return G1Xyzz{store28(a.x), store28(a.y), store28(a.zz), store28(a.zzz)}; }
__device__ __forceinline__ Fq28 add28(const Fq28 &a, const Fq28 &b) {
    Fq28 r;
#pragma unroll
    for (int i = 0; i < 14; i++) r.l[i] = a.l[i] + b.l[i];
    return r;
}
// acc (not infinity) += q (affine, not infinity), coordinates of q as loaded (load28).  Returns false when the sum is
// the point at infinity (acc untouched).  The degenerate branches of the group law (same or opposite point) sit behind
// the low-limb filter, decided exactly there, and stay in this arithmetic: everything lives in registers.
__device__ __forceinline__ bool add_mixed28(G1Xyzz28 &acc, const Fq28 &qx, const Fq28 &qy) {
    // ordered so that few values are live at a time
    const Fq28 p = sub28(mul28(qx, acc.zz), acc.x);
    if (maybe_zero28(p, 8)) {  // one addition in ~2^24 comes here without being degenerate
        if (store28(p).is_zero()) {
            const Fq28 rr = sub28(mul28(qy, acc.zzz), acc.y);
            if (!store28(rr).is_zero()) return false;  // opposite points
            // the same point: 2 (qx, qy)  (mdbl-2008-s-1, a = 0)
            const Fq28 one = f28_one();
            const Fq28 x = mul28(qx, one), y = mul28(qy, one);  // loaded values are lazy up to 256 q
            const Fq28 u = dbl28(y), v = sqr28(u), w = mul28(u, v), sx = mul28(x, v), xx = sqr28(x);
            const Fq28 m = normalise28(add28(dbl28(xx), xx));
            const Fq28 x3 = sub28(sqr28(m), dbl28(sx));
            const Fq28 y3 = sub28(mul28(m, sub28(sx, x3)), mul28(w, y));
            acc.x = normalise28(x3);
            acc.y = normalise28(y3);
            acc.zz = v;
            acc.zzz = w;
            return true;
        }
    }
    const Fq28 r = sub28(mul28(qy, acc.zzz), acc.y);
    const Fq28 pp = sqr28(p);
    acc.zz = mul28(acc.zz, pp);
    const Fq28 ppp = mul28(p, pp);
    const Fq28 qq = mul28(acc.x, pp);
    acc.zzz = mul28(acc.zzz, ppp);
    const Fq28 x3 = sub28(sub28(sqr28(r), ppp), dbl28(qq));
    const Fq28 yp = mul28(acc.y, ppp);
    acc.y = normalise28(sub28(mul28(r, sub28(qq, x3)), yp));
    acc.x = normalise28(x3);
    return true;
}

// ---- full XYZZ addition for reductions of partial sums (tools/model_f28.py xyzz_add / xyzz_dbl): operands with limbs
// below 2^28 -- normalised x, y and product outputs zz, zzz, or coordinates as loaded
__device__ __forceinline__ G1Xyzz28 xyzz28_from(const G1Xyzz &a) { return G1Xyzz28{load28(a.x), load28(a.y), load28(a.zz), load28(a.zzz)}; }
// 2 a (dbl-2008-s-1, a = 0); cold
__device__ __forceinline__ G1Xyzz28 xyzz_dbl28(const G1Xyzz28 &a) {
    const Fq28 u = dbl28(a.y), v = sqr28(u), w = mul28(u, v), s = mul28(a.x, v), xx = sqr28(a.x);
    const Fq28 m = normalise28(add28(dbl28(xx), xx));
    const Fq28 x3 = sub28(sqr28(m), dbl28(s));
    const Fq28 y3 = sub28(mul28(m, sub28(s, x3)), mul28(w, a.y));
    return G1Xyzz28{normalise28(x3), normalise28(y3), mul28(v, a.zz), mul28(w, a.zzz)};
}
// a += b, neither infinity (add-2008-s: 12M + 2S).  Returns false when the sum is the point at infinity (a untouched).
__device__ __forceinline__ bool xyzz_add28(G1Xyzz28 &a, const G1Xyzz28 &b) {
    const Fq28 u1 = mul28(a.x, b.zz), s1 = mul28(a.y, b.zzz);
    const Fq28 p = sub28(mul28(b.x, a.zz), u1);
    if (maybe_zero28(p, 8)) {
        if (store28(p).is_zero()) {
            if (!store28(sub28(mul28(b.y, a.zzz), s1)).is_zero()) return false;
            a = xyzz_dbl28(a);
            return true;
        }
    }
    const Fq28 r = sub28(mul28(b.y, a.zzz), s1);
    const Fq28 pp = sqr28(p), ppp = mul28(p, pp), qq = mul28(u1, pp);
    a.zz = mul28(mul28(a.zz, b.zz), pp);
    a.zzz = mul28(mul28(a.zzz, b.zzz), ppp);
    const Fq28 x3 = sub28(sub28(sqr28(r), ppp), dbl28(qq));
    a.y = normalise28(sub28(mul28(r, sub28(qq, x3)), mul28(s1, ppp)));
    a.x = normalise28(x3);
    return true;
}
// the same with either side possibly infinity (a flag each)
__device__ __forceinline__ void xyzz_acc28(G1Xyzz28 &a, bool &have_a, const G1Xyzz28 &b, bool have_b) {
    if (!have_b) return;
    if (!have_a) {
        a = b;
        have_a = true;
        return;
    }
    have_a = xyzz_add28(a, b);
}

__device__ __forceinline__ Fq28 shfl_down28(const Fq28 &a, int delta) {
    Fq28 r;
#pragma unroll
    for (int k = 0; k < 14; k++) r.l[k] = __shfl_down(a.l[k], delta, 64);
    return r;
}
__device__ __forceinline__ G1Xyzz28 xyzz28_shfl_down(const G1Xyzz28 &a, int delta) {
    return G1Xyzz28{shfl_down28(a.x, delta), shfl_down28(a.y, delta), shfl_down28(a.zz, delta), shfl_down28(a.zzz, delta)};
}

// ---- Jacobian doubling and mixed addition for the fallback's scalar-multiplication chains (glv.hpp), limb bounds as in
// tools/model_f28.py jac_dbl / jac_madd: X, Y normalised, Z with limbs below 2^29
struct G1Jac28 {
    Fq28 x, y, z;
};
__device__ __forceinline__ Fq28 scale28(const Fq28 &a, int32_t k) {
    Fq28 r;
#pragma unroll
    for (int i = 0; i < 14; i++) r.l[i] = a.l[i] * k;
    return r;
}
// 2 (X, Y, Z), a = 0 (EFD dbl-2009-l: 2M + 5S)
__device__ __forceinline__ G1Jac28 jac_dbl28(const G1Jac28 &p) {
    const Fq28 a = sqr28(p.x), b = sqr28(p.y), c = sqr28(b);
    const Fq28 d = normalise28(dbl28(sub28(sub28(sqr28(add28(p.x, b)), a), c)));
    const Fq28 e = normalise28(add28(dbl28(a), a));
    G1Jac28 r;
    r.x = normalise28(sub28(sqr28(e), dbl28(d)));
    r.y = normalise28(sub28(mul28(e, sub28(d, r.x)), scale28(c, 8)));
    r.z = dbl28(mul28(p.y, p.z));
    return r;
}
// (X, Y, Z) + affine (qx, qy) as loaded (EFD madd-2007-bl: 7M + 4S); the accumulator is not infinity.
// Returns 0: done; 1: the sum is the point at infinity; 2: same point (the caller doubles the affine point instead)
__device__ __forceinline__ int jac_add_mixed28(G1Jac28 &p, const Fq28 &qx, const Fq28 &qy) {
    const Fq28 z1z1 = sqr28(p.z);
    const Fq28 h = sub28(mul28(qx, z1z1), p.x);
    const Fq28 rr = sub28(mul28(mul28(qy, p.z), z1z1), p.y);  // r / 2
    if (maybe_zero28(h, 8)) {
        if (store28(h).is_zero()) return store28(rr).is_zero() ? 2 : 1;
    }
    const Fq28 hh = sqr28(h), i4 = scale28(hh, 4);
    const Fq28 j = mul28(h, i4), v = mul28(p.x, i4);
    const Fq28 x3 = normalise28(sub28(sub28(scale28(sqr28(rr), 4), j), dbl28(v)));
    p.y = normalise28(dbl28(sub28(mul28(rr, sub28(v, x3)), mul28(p.y, j))));
    p.z = normalise28(sub28(sub28(sqr28(normalise28(add28(p.z, h))), z1z1), hh));
    p.x = x3;
    return 0;
}

}  // namespace mi
#endif
