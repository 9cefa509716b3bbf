// BLS12-377 G1 group law for gfx950 (host + device).
//
// Replaces snarkvm_curves::bls12_377::{G1Affine, G1Projective} (snarkVM 1.0.0 @ dea322b, un-vendored;
// SURVEY.md 8a row A2) on the device.  E/Fq: y^2 = x^3 + 1 (a = 0).
//   * G1Affine  : ABI type, 96 B {x, y} Montgomery; the point at infinity is x = y = 0
//                 ((0,0) is not on the curve, so the encoding is unambiguous).
//   * G1Xyzz    : extended Jacobian (X, Y, ZZ, ZZZ), x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2; infinity: ZZ = 0.
//                 Bucket accumulators use it because the mixed addition is 8M + 2S and needs no
//                 field inversion (formulas: Sutherland 2008, EFD "xyzz").
//   * G1Jac     : ABI output type (X, Y, Z), x = X/Z^2, y = Y/Z^3; infinity: Z = 0.
#pragma once
#include "field.hpp"

namespace mi {

struct G1Affine {
    Fq x, y;
    MI_HD bool is_inf() const { return x.is_zero() && y.is_zero(); }
    MI_HD static G1Affine inf() { return G1Affine{Fq::zero(), Fq::zero()}; }
};
struct G1Jac {
    Fq x, y, z;
};
struct G1Xyzz {
    Fq x, y, zz, zzz;
    MI_HD bool is_inf() const { return zz.is_zero(); }
    MI_HD static G1Xyzz inf() { return G1Xyzz{Fq::one(), Fq::one(), Fq::zero(), Fq::zero()}; }
    MI_HD static G1Xyzz from_affine(const G1Affine &p) {
        if (p.is_inf()) return inf();
        return G1Xyzz{p.x, p.y, Fq::one(), Fq::one()};
    }
};
static_assert(sizeof(G1Affine) == 96 && sizeof(G1Jac) == 144 && sizeof(G1Xyzz) == 192, "ABI layout");

// Field products in the group law go through fm<I>/fs<I>: I = true inlines the 12-limb Montgomery
// product (hot bucket-accumulation loop), I = false calls the shared out-of-line copy (everything
// else: keeps code size and compile time bounded).
template <bool I>
MI_HD Fq fm(const Fq &a, const Fq &b) {
    if constexpr (I) return mul(a, b);
    else return mulq(a, b);
}
template <bool I>
MI_HD Fq fs(const Fq &a) {
    if constexpr (I) return sqr(a);
    else return sqrq(a);
}

MI_HD G1Affine neg(const G1Affine &p) {
    if (p.is_inf()) return p;
    return G1Affine{p.x, neg(p.y)};
}

// 2 * (affine p), p not infinity  (mdbl-2008-s-1, a = 0)
template <bool I = false>
MI_HD G1Xyzz dbl_affine(const G1Affine &p) {
    Fq u = dbl(p.y);
    Fq v = fs<I>(u);
    Fq w = fm<I>(u, v);
    Fq s = fm<I>(p.x, v);
    Fq xx = fs<I>(p.x);
    Fq m = add(dbl(xx), xx);
    Fq x3 = sub(fs<I>(m), dbl(s));
    Fq y3 = sub(fm<I>(m, sub(s, x3)), fm<I>(w, p.y));
    return G1Xyzz{x3, y3, v, w};
}

// 2 * p  (dbl-2008-s-1, a = 0)
template <bool I = false>
MI_HD G1Xyzz dbl(const G1Xyzz &p) {
    if (p.is_inf()) return p;
    Fq u = dbl(p.y);
    Fq v = fs<I>(u);
    Fq w = fm<I>(u, v);
    Fq s = fm<I>(p.x, v);
    Fq xx = fs<I>(p.x);
    Fq m = add(dbl(xx), xx);
    Fq x3 = sub(fs<I>(m), dbl(s));
    Fq y3 = sub(fm<I>(m, sub(s, x3)), fm<I>(w, p.y));
    return G1Xyzz{x3, y3, fm<I>(v, p.zz), fm<I>(w, p.zzz)};
}

// acc + (affine q)  (madd-2008-s), all special cases handled
template <bool I = false>
MI_HD G1Xyzz add_mixed(const G1Xyzz &a, const G1Affine &q) {
    if (q.is_inf()) return a;
    if (a.is_inf()) return G1Xyzz{q.x, q.y, Fq::one(), Fq::one()};
    Fq u2 = fm<I>(q.x, a.zz);
    Fq s2 = fm<I>(q.y, a.zzz);
    Fq p = sub(u2, a.x);
    Fq r = sub(s2, a.y);
    if (p.is_zero()) {
        if (r.is_zero()) return dbl_affine<I>(q);
        return G1Xyzz::inf();
    }
    Fq pp = fs<I>(p);
    Fq ppp = fm<I>(p, pp);
    Fq qq = fm<I>(a.x, pp);
    Fq x3 = sub(sub(fs<I>(r), ppp), dbl(qq));
    Fq y3 = sub(fm<I>(r, sub(qq, x3)), fm<I>(a.y, ppp));
    return G1Xyzz{x3, y3, fm<I>(a.zz, pp), fm<I>(a.zzz, ppp)};
}

// a + b  (add-2008-s)
template <bool I = false>
MI_HD G1Xyzz add(const G1Xyzz &a, const G1Xyzz &b) {
    if (a.is_inf()) return b;
    if (b.is_inf()) return a;
    Fq u1 = fm<I>(a.x, b.zz);
    Fq u2 = fm<I>(b.x, a.zz);
    Fq s1 = fm<I>(a.y, b.zzz);
    Fq s2 = fm<I>(b.y, a.zzz);
    Fq p = sub(u2, u1);
    Fq r = sub(s2, s1);
    if (p.is_zero()) {
        if (r.is_zero()) return dbl<I>(a);
        return G1Xyzz::inf();
    }
    Fq pp = fs<I>(p);
    Fq ppp = fm<I>(p, pp);
    Fq qq = fm<I>(u1, pp);
    Fq x3 = sub(sub(fs<I>(r), ppp), dbl(qq));
    Fq y3 = sub(fm<I>(r, sub(qq, x3)), fm<I>(s1, ppp));
    return G1Xyzz{x3, y3, fm<I>(fm<I>(a.zz, b.zz), pp), fm<I>(fm<I>(a.zzz, b.zzz), ppp)};
}

template <bool I = false>
MI_HD G1Affine to_affine(const G1Xyzz &p) {
    if (p.is_inf()) return G1Affine::inf();
    // 1/ZZ and 1/ZZZ from one inversion: i = 1/(ZZ*ZZZ)
    Fq i = invq(fm<I>(p.zz, p.zzz));
    return G1Affine{fm<I>(p.x, fm<I>(i, p.zzz)), fm<I>(p.y, fm<I>(i, p.zz))};
}

MI_HD G1Jac affine_to_jac(const G1Affine &p) {
    if (p.is_inf()) return G1Jac{Fq::one(), Fq::one(), Fq::zero()};
    return G1Jac{p.x, p.y, Fq::one()};
}

// k * p by double-and-add over `bits` bits of a little-endian 32-bit-limb scalar (canonical integer)
template <bool I = false>
MI_HD G1Xyzz scalar_mul(const G1Affine &p, const uint32_t *k, int bits) {
    G1Xyzz acc = G1Xyzz::inf();
    for (int i = bits - 1; i >= 0; i--) {
        acc = dbl<I>(acc);
        if ((k[i >> 5] >> (i & 31)) & 1u) acc = add_mixed<I>(acc, p);
    }
    return acc;
}

#if defined(__HIPCC__)
// ---------------------------------------------------------------- 16-byte vector loads/stores of points
__device__ __forceinline__ G1Affine load_point(const G1Affine *__restrict__ pts, uint32_t entry) {
    const uint4 *p = reinterpret_cast<const uint4 *>(pts + (entry >> 1));
    uint4 v[6];
#pragma unroll
    for (int k = 0; k < 6; k++) v[k] = p[k];
    G1Affine r;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        r.x.l[4 * k + 0] = v[k].x; r.x.l[4 * k + 1] = v[k].y; r.x.l[4 * k + 2] = v[k].z; r.x.l[4 * k + 3] = v[k].w;
        r.y.l[4 * k + 0] = v[3 + k].x; r.y.l[4 * k + 1] = v[3 + k].y; r.y.l[4 * k + 2] = v[3 + k].z; r.y.l[4 * k + 3] = v[3 + k].w;
    }
    if (entry & 1u) r = neg(r);
    return r;
}
__device__ __forceinline__ void store_xyzz(G1Xyzz *dst, const G1Xyzz &a) {
    uint4 *o = reinterpret_cast<uint4 *>(dst);
    const Fq *f[4] = {&a.x, &a.y, &a.zz, &a.zzz};
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
        for (int k = 0; k < 3; k++) o[q * 3 + k] = make_uint4(f[q]->l[4 * k], f[q]->l[4 * k + 1], f[q]->l[4 * k + 2], f[q]->l[4 * k + 3]);
}
__device__ __forceinline__ G1Xyzz load_xyzz(const G1Xyzz *src) {
    const uint4 *p = reinterpret_cast<const uint4 *>(src);
    G1Xyzz a;
    Fq *f[4] = {&a.x, &a.y, &a.zz, &a.zzz};
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
        for (int k = 0; k < 3; k++) {
            uint4 v = p[q * 3 + k];
            f[q]->l[4 * k] = v.x; f[q]->l[4 * k + 1] = v.y; f[q]->l[4 * k + 2] = v.z; f[q]->l[4 * k + 3] = v.w;
        }
    return a;
}

#endif

}  // namespace mi
