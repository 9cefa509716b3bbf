// Scalar multiplication of ONE point by ONE scalar on a lane (the per-proof fallback of the batch verifier and
// mi_g1_mul_many_device): the MSM machinery shares its doublings over thousands of points, a lone product cannot, and
// its latency is its doubling chain.  Two levers cut that chain and the work beside it:
//   * GLV: phi(x, y) = (beta x, y) is multiplication by lambda = x^2 - 1 on G1 (lambda^2 + lambda + 1 = r), so
//     s P = s1 P + s2 phi(P) with s1, s2 < 2^127: two half-length chains that run on two lanes;
//   * signed fixed windows of W bits: lanes of a wave run in lockstep, so a data-dependent addition (double-and-add,
//     NAF) costs every lane an addition per bit whatever its own digit; with fixed windows every lane adds once per
//     W doublings.  The table {P, 2P, .., 2^(W-1) P} is made affine with one shared inversion (every addition of the
//     chain is then a mixed one) and lives in global memory (96 B per entry and lane: too large for registers or
//     LDS at the occupancy wanted), read once per window;
//   * the accumulator is Jacobian: a chain is mostly doublings, 2M + 5S there against 6M + 3S in XYZZ, with the
//     dedicated squaring of field.hpp.
// Reference: snarkvm_curves' `ProjectiveCurve::mul` / `mul_bits` double-and-add (snarkVM 1.0.0 @ dea322b, curves/src/
// templates/short_weierstrass_jacobian/projective.rs, un-vendored) and bls12_377::g1's GLV
// parameters; constants here are derived in tools/gen_constants.py.
#pragma once
#include "curve.hpp"

namespace mi {

MI_HD Fq glv_beta() {
    Fq b;
#pragma unroll
    for (int i = 0; i < 6; i++) {
        b.l[2 * i] = (uint32_t)MI_GLV_BETA[i];
        b.l[2 * i + 1] = (uint32_t)(MI_GLV_BETA[i] >> 32);
    }
    return b;
}

// s (canonical, < r) = s1 + lambda s2 with 0 <= s1 < lambda and 0 <= s2 <= lambda + 1 (both below 2^127)
MI_HD void glv_split(const uint32_t (&s)[8], uint32_t (&s1)[4], uint32_t (&s2)[4]) {
    const uint32_t LAM[4] = {(uint32_t)MI_GLV_LAMBDA[0], (uint32_t)(MI_GLV_LAMBDA[0] >> 32), (uint32_t)MI_GLV_LAMBDA[1],
                             (uint32_t)(MI_GLV_LAMBDA[1] >> 32)};
    const uint32_t MU[5] = {(uint32_t)MI_GLV_MU[0], (uint32_t)(MI_GLV_MU[0] >> 32), (uint32_t)MI_GLV_MU[1],
                            (uint32_t)(MI_GLV_MU[1] >> 32), (uint32_t)MI_GLV_MU[2]};
    // qhat = floor(s mu / 2^256), mu = floor(2^256 / lambda): the quotient or one less
    uint32_t prod[13];
#pragma unroll
    for (int i = 0; i < 13; i++) prod[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint32_t carry = 0;
#pragma unroll
        for (int j = 0; j < 5; j++) {
            const uint64_t t = (uint64_t)s[i] * MU[j] + prod[i + j] + carry;
            prod[i + j] = (uint32_t)t;
            carry = (uint32_t)(t >> 32);
        }
        prod[i + 5] = carry;
    }
    uint32_t q[4] = {prod[8], prod[9], prod[10], prod[11]};
    // rem = s - qhat lambda, below 2 lambda < 2^128: five limbs carry it
    uint32_t t5[5] = {0, 0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 4; i++) {
        uint32_t carry = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (i + j < 5) {
                const uint64_t t = (uint64_t)q[i] * LAM[j] + t5[i + j] + carry;
                t5[i + j] = (uint32_t)t;
                carry = (uint32_t)(t >> 32);
            }
        }
        if (i + 4 < 5) t5[i + 4] = carry;
    }
    uint32_t rem[5];
    {
        uint32_t borrow = 0;
#pragma unroll
        for (int i = 0; i < 5; i++) {
            const uint64_t d = (uint64_t)s[i] - t5[i] - borrow;
            rem[i] = (uint32_t)d;
            borrow = (uint32_t)(d >> 63);
        }
    }
#pragma unroll
    for (int round = 0; round < 2; round++) {
        // rem >= lambda ?
        uint32_t d[5], borrow = 0;
#pragma unroll
        for (int i = 0; i < 5; i++) {
            const uint64_t t = (uint64_t)rem[i] - (i < 4 ? LAM[i] : 0u) - borrow;
            d[i] = (uint32_t)t;
            borrow = (uint32_t)(t >> 63);
        }
        const uint32_t ge = 1u - borrow;
        uint32_t c = ge;
#pragma unroll
        for (int i = 0; i < 5; i++) rem[i] = ge ? d[i] : rem[i];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint64_t t = (uint64_t)q[i] + c;
            q[i] = (uint32_t)t;
            c = (uint32_t)(t >> 32);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; i++) { s1[i] = rem[i]; s2[i] = q[i]; }
}

// Signed digits of a value below 2^128 in windows of W bits, packed as nibbles (digit + 8), low window first.
// W = 3: digits -3..4, 43 windows (the top one holds bits 126..128: at most 3 + carry = 4, no carry out);
// W = 4: digits -8..7, 33 windows (the top one holds the last carry only).
template <int W>
struct GlvDigits {
    static constexpr int NW = (128 + W) / W;          // 43 (W = 3), 33 (W = 4)
    static constexpr int WORDS = (NW + 7) / 8;
    uint32_t w[WORDS];
    MI_HD void recode(const uint32_t (&v)[4]) {
#pragma unroll
        for (int i = 0; i < WORDS; i++) w[i] = 0;
        uint32_t carry = 0;
#pragma unroll
        for (int i = 0; i < NW; i++) {
            const int bit = W * i, word = bit >> 5, off = bit & 31;
            uint32_t raw = word < 4 ? v[word] >> off : 0u;
            if (off + W > 32 && word + 1 < 4) raw |= v[word + 1] << (32 - off);
            raw = (raw & ((1u << W) - 1)) + carry;
            carry = raw > (W == 4 ? 7u : (1u << (W - 1))) ? 1u : 0u;
            const uint32_t nib = raw + 8u - (carry << W);  // digit + 8
            w[i >> 3] |= nib << (4 * (i & 7));
        }
    }
    // digit of window i (dynamic i, no register indexing)
    MI_HD int digit(int i) const {
        uint32_t word = 0;
#pragma unroll
        for (int j = 0; j < WORDS; j++) word = (i >> 3) == j ? w[j] : word;
        return (int)((word >> (4 * (i & 7))) & 15u) - 8;
    }
};

// ---- the chain's own group law: Jacobian doubling is 2M + 5S against 6M + 3S in XYZZ, and a chain is mostly doublings
// 2 (X, Y, Z), a = 0 (EFD dbl-2009-l); Z = 0 stays Z = 0
MI_HD G1Jac glv_dbl(const G1Jac &p) {
    const Fq a = sqrq(p.x), b = sqrq(p.y), c = sqrq(b);
    Fq d = sub(sub(sqrq(add(p.x, b)), a), c);
    d = dbl(d);
    const Fq e = add(dbl(a), a), f = sqrq(e);
    G1Jac r;
    r.x = sub(f, dbl(d));
    r.y = sub(mulq(e, sub(d, r.x)), dbl(dbl(dbl(c))));
    r.z = dbl(mulq(p.y, p.z));
    return r;
}
// (X, Y, Z) + affine q, q not infinity (EFD madd-2007-bl: 7M + 4S), all special cases handled
MI_HD G1Jac glv_add_mixed(const G1Jac &p, const G1Affine &q) {
    if (p.z.is_zero()) return G1Jac{q.x, q.y, Fq::one()};
    const Fq z1z1 = sqrq(p.z), u2 = mulq(q.x, z1z1), s2 = mulq(mulq(q.y, p.z), z1z1);
    const Fq h = sub(u2, p.x), rr = sub(s2, p.y);
    if (h.is_zero()) {
        if (rr.is_zero()) return glv_dbl(p);
        return G1Jac{Fq::one(), Fq::one(), Fq::zero()};
    }
    const Fq hh = sqrq(h), i = dbl(dbl(hh)), j = mulq(h, i), r2 = dbl(rr), v = mulq(p.x, i);
    G1Jac r;
    r.x = sub(sub(sqrq(r2), j), dbl(v));
    r.y = sub(mulq(r2, sub(v, r.x)), dbl(mulq(p.y, j)));
    r.z = sub(sub(sqrq(add(p.z, h)), z1z1), hh);
    return r;
}
MI_HD G1Xyzz glv_to_xyzz(const G1Jac &p) {
    if (p.z.is_zero()) return G1Xyzz::inf();
    const Fq zz = sqrq(p.z);
    return G1Xyzz{p.x, p.y, zz, mulq(zz, p.z)};
}
// v P from a table of P's multiples (T::get(e, negate) = +-(e + 1) P, affine), v < 2^128: NW windows, W doublings and
// one mixed addition each
template <int W, class T>
MI_HD G1Xyzz glv_chain(const T &tab, const uint32_t (&v)[4]) {
    GlvDigits<W> dg;
    dg.recode(v);
    G1Jac acc = G1Jac{Fq::one(), Fq::one(), Fq::zero()};
    for (int i = GlvDigits<W>::NW - 1; i >= 0; i--) {
        if (i != GlvDigits<W>::NW - 1) {
#pragma unroll 1
            for (int k = 0; k < W; k++) acc = glv_dbl(acc);
        }
        const int d = dg.digit(i);
        if (d != 0) {
            const G1Affine t = tab.get((uint32_t)((d < 0 ? -d : d) - 1), d < 0);
            if (!t.is_inf()) acc = glv_add_mixed(acc, t);
        }
    }
    return glv_to_xyzz(acc);
}

#if defined(__HIPCC__)
}  // namespace mi
#include "field28.hpp"
namespace mi {
__device__ __forceinline__ void glv_store_affine(G1Affine *dst, const G1Affine &a) {
    uint4 *o = reinterpret_cast<uint4 *>(dst);
#pragma unroll
    for (int k = 0; k < 3; k++) {
        o[k] = make_uint4(a.x.l[4 * k], a.x.l[4 * k + 1], a.x.l[4 * k + 2], a.x.l[4 * k + 3]);
        o[3 + k] = make_uint4(a.y.l[4 * k], a.y.l[4 * k + 1], a.y.l[4 * k + 2], a.y.l[4 * k + 3]);
    }
}

// table[e] = (e + 1) P for e < 2^(W-1), AFFINE (one shared inversion: every addition of the chain is then a mixed one),
// in this lane's slice of global memory; P not infinity.  An entry that is the point at infinity (P of small order)
// is stored as (0, 0).
template <int W>
__device__ __forceinline__ void glv_build_table(G1Affine *tab, const G1Affine &P) {
    constexpr int E = 1 << (W - 1);
    glv_store_affine(&tab[0], P);
    G1Xyzz m[E];  // m[e] = (e + 1) P; m[0] unused
    m[1] = dbl_affine(P);
    m[2] = add_mixed(m[1], P);
    m[3] = dbl(m[1]);
    if constexpr (W == 4) {
        m[4] = add_mixed(m[3], P);
        m[5] = dbl(m[2]);
        m[6] = add_mixed(m[5], P);
        m[7] = dbl(m[3]);
    }
    // 1 / (zz zzz) of every entry from one inversion (Montgomery's trick); an infinite entry enters as 1
    Fq w[E], pre[E];
    Fq run = Fq::one();
#pragma unroll
    for (int e = 1; e < E; e++) {
        w[e] = m[e].is_inf() ? Fq::one() : mulq(m[e].zz, m[e].zzz);
        pre[e] = run;
        run = mulq(run, w[e]);
    }
    Fq inv = invq(run);
#pragma unroll
    for (int e = E - 1; e >= 1; e--) {
        const Fq iw = mulq(inv, pre[e]);  // 1 / w[e]
        inv = mulq(inv, w[e]);
        G1Affine a = G1Affine::inf();
        if (!m[e].is_inf()) a = G1Affine{mulq(m[e].x, mulq(iw, m[e].zzz)), mulq(m[e].y, mulq(iw, m[e].zz))};
        glv_store_affine(&tab[e], a);
    }
}

// The device's chain: the same windows over the same affine table, with the accumulator in the reduced-radix arithmetic
// of field28.hpp (one instruction per limb product, carry-free additions); table entries are read as they are stored
// (canonical, 32-bit limbs) and enter 8 bits higher.
template <int W>
__device__ __forceinline__ G1Xyzz glv_chain28(const G1Affine *tab, const uint32_t (&v)[4]) {
    GlvDigits<W> dg;
    dg.recode(v);
    G1Jac28 acc;
    bool have = false;
    for (int i = GlvDigits<W>::NW - 1; i >= 0; i--) {
        if (have) {
#pragma unroll 1
            for (int k = 0; k < W; k++) acc = jac_dbl28(acc);
        }
        const int d = dg.digit(i);
        if (d != 0) {
            const G1Affine t = load_point(tab, (uint32_t)(((d < 0 ? -d : d) - 1) << 1));
            if (!t.is_inf()) {
                const Fq28 qx = load28(t.x);
                Fq28 qy = load28(t.y);
                if (d < 0) qy = neg28(qy);
                const Fq28 one = f28_one();
                int st = 2;  // an empty accumulator takes the point itself (as a doubling of nothing: same code path)
                if (have) st = jac_add_mixed28(acc, qx, qy);
                if (st == 1) have = false;
                if (st == 2) {
                    acc = G1Jac28{mul28(qx, one), mul28(qy, one), one};
                    if (have) acc = jac_dbl28(acc);  // the accumulator was this very point
                    have = true;
                }
            }
        }
    }
    if (!have) return G1Xyzz::inf();
    const Fq28 zz = sqr28(acc.z);
    return G1Xyzz{store28(acc.x), store28(acc.y), store28(zz), store28(mul28(zz, acc.z))};
}

struct GlvGlobalTable {
    const G1Affine *tab;
    __device__ __forceinline__ G1Affine get(uint32_t e, bool negate) const { return load_point(tab, (e << 1) | (negate ? 1u : 0u)); }
};
#endif

}  // namespace mi
