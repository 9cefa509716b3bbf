// Compressed G1 points: snarkVM's CanonicalSerialize of a short-Weierstrass affine point with compression.
//
// The reference node never touches this encoding itself (it lives in the un-vendored snarkvm-curves crate: templates/
// short_weierstrass_jacobian + SWFlags, [RECALL]): the commitments of a Varuna `Proof` and of a `CircuitVerifyingKey`
// arrive in it inside `Transaction` bytes (reference node/router/messages/src/unconfirmed_transaction.rs:25-28,
// node/bft/events/src/block_response.rs:64-70) and snarkVM decodes them one point at a time on the CPU -- a square
// root in Fq (~1 200 products) and a subgroup check per point, ~11 points per proof.  Here a batch decodes in one
// launch, a lane per point.  The convention (parity unpinned; the tests' CPU restatements follow the same text):
//   48 bytes, x little-endian and below q; bit 7 of byte 47: y is the larger root (y > (q - 1) / 2);
//   bit 6: the point at infinity (x = 0, bit 7 clear).
// Host and device (MI_HD).
#pragma once
#include "constants.hpp"
#include "curve.hpp"

namespace mi {

MI_HD Fq fq_from_u64x6(const uint64_t *c) {
    Fq r;
#pragma unroll
    for (int i = 0; i < 6; i++) {
        r.l[2 * i] = (uint32_t)c[i];
        r.l[2 * i + 1] = (uint32_t)(c[i] >> 32);
    }
    return r;
}
// a < b as 384-bit integers (limbs, not field elements)
MI_HD bool limbs_less(const Fq &a, const Fq &b) {
    for (int i = 11; i >= 0; i--) {
        if (a.l[i] < b.l[i]) return true;
        if (a.l[i] > b.l[i]) return false;
    }
    return false;
}
// a^e for the 330-bit exponent (t - 1) / 2 with 4-bit windows: 14 products for a^2 .. a^15, then 4 squarings and at
// most one product per window (~330 squarings + ~90 products instead of + ~165)
MI_HD Fq fq_pow_t_minus1_half(const Fq &a) {
    Fq tab[16];
    tab[0] = Fq::one();
    tab[1] = a;
#pragma unroll 1
    for (int i = 2; i < 16; i++) tab[i] = (i & 1) ? mul(tab[i - 1], a) : sqr(tab[i >> 1]);
    Fq acc = Fq::one();
    bool started = false;
#pragma unroll 1
    for (int w = 95; w >= 0; w--) {  // 96 windows of 4 bits cover 384 bits
        const unsigned nib = (unsigned)(MI_FQ_T_MINUS1_HALF[w >> 4] >> (4 * (w & 15))) & 15u;
        if (started) {
            acc = sqr(acc);
            acc = sqr(acc);
            acc = sqr(acc);
            acc = sqr(acc);
        }
        if (nib) {
            acc = started ? mul(acc, tab[nib]) : tab[nib];
            started = true;
        }
    }
    return acc;
}
// Tonelli-Shanks over the 2^46-th roots of unity (q - 1 = 2^46 t): root of a (Montgomery) into `root`; false when a
// is not a square.  With c_0 = 15^t the loop's c is always c_0^(2^(46 - m)) and its g = c_0^(2^(45 - i)): table
// entries (MI_FQ_2SYLOW_POW), so a round costs the i squarings that find the order of b and two products.
MI_HD bool fq_sqrt(const Fq &a, Fq &root) {
    if (a.is_zero()) {
        root = a;
        return true;
    }
    const Fq one = Fq::one();
    const Fq w = fq_pow_t_minus1_half(a);  // a^((t-1)/2)
    Fq x = mul(a, w);                      // a^((t+1)/2)
    Fq b = mul(x, w);                      // a^t
    int m = MI_FQ_TWO_ADICITY;
    while (!(b == one)) {
        int i = 0;
        Fq b2 = b;
        while (!(b2 == one)) {
            b2 = sqr(b2);
            if (++i == m) return false;  // b has order 2^m: not a square
        }
        x = mul(x, fq_from_u64x6(MI_FQ_2SYLOW_POW[MI_FQ_TWO_ADICITY - 1 - i]));  // g
        b = mul(b, fq_from_u64x6(MI_FQ_2SYLOW_POW[MI_FQ_TWO_ADICITY - i]));      // g^2 (i >= 1: index <= 45)
        m = i;
    }
    root = x;
    return true;
}
MI_HD bool fq_is_larger_root(const Fq &y_mont) {
    return limbs_less(fq_from_u64x6(MI_FQ_HALF), from_mont(y_mont));
}
// u Q for the curve's seed u = 0x8508c00000000001 (64 bits, seven of them set), Q projective
MI_HD G1Xyzz mul_by_seed(const G1Xyzz &q) {
    G1Xyzz acc = q;
    for (int i = 62; i >= 0; i--) {
        acc = dbl(acc);
        if ((MI_BLS_X >> i) & 1ull) acc = add(acc, q);
    }
    return acc;
}
// Is the curve point p in the subgroup of order r?  The endomorphism phi(x, y) = (beta x, y) satisfies phi^2 + phi + 1 = 0
// on the whole curve and acts on the subgroup as lambda = u^2 - 1 (glv.hpp), so phi^2 acts there as -u^2.  Conversely
// phi^2(p) = -u^2 p forces (u^4 - u^2 + 1) p = r p = O: the test  u (u p) == -phi^2(p)  is exact (no false accepts),
// for 126 doublings and 12 additions instead of 253 and ~125 (M. Scott, "A note on group membership tests for G1,
// G2 and GT on BLS pairing-friendly curves", section 6: the form arkworks uses for this curve).
MI_HD bool g1_in_subgroup(const G1Affine &p) {
    if (p.is_inf()) return true;
    const G1Xyzz u2p = mul_by_seed(mul_by_seed(G1Xyzz::from_affine(p)));
    if (u2p.is_inf()) return false;  // -phi^2(p) is a finite point
    const Fq beta = fq_from_u64x6(MI_GLV_BETA);
    const Fq x2 = mul(mul(beta, beta), p.x);  // phi^2(p) = (beta^2 x, y); its negative has -y
    return u2p.x == mul(x2, u2p.zz) && u2p.y == mul(neg(p.y), u2p.zzz);
}
// 48 bytes -> affine point (Montgomery); false: not an encoding of a point (of the order-r subgroup when checked)
MI_HD bool g1_decompress_one(const uint8_t *in, bool check_subgroup, G1Affine &out) {
    out = G1Affine::inf();
    const uint8_t flags = in[47] & 0xC0;
    Fq xc;
    bool zero = true;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        uint32_t v = 0;
        for (int k = 0; k < 4; k++) {
            const int idx = 4 * i + k;
            v |= (uint32_t)(idx == 47 ? (in[idx] & 0x3F) : in[idx]) << (8 * k);
        }
        xc.l[i] = v;
        zero = zero && v == 0;
    }
    if (flags & 0x40) return zero && !(flags & 0x80);
    if (!limbs_less(xc, fq_from_u64x6(MI_FQ_MODULUS))) return false;
    const Fq x = to_mont(xc);
    const Fq rhs = add(mul(sqr(x), x), Fq::one());
    Fq y;
    if (!fq_sqrt(rhs, y)) return false;
    if (!(sqr(y) == rhs)) return false;
    if (y.is_zero() && (flags & 0x80)) return false;  // y = 0 is its own negative: the sign flag selects nothing (non-canonical)
    if (fq_is_larger_root(y) != ((flags & 0x80) != 0)) y = neg(y);
    const G1Affine p{x, y};
    if (check_subgroup && !g1_in_subgroup(p)) return false;
    out = p;
    return true;
}
MI_HD void g1_compress_one(const G1Affine &p, uint8_t *out) {
    for (int i = 0; i < 48; i++) out[i] = 0;
    if (p.is_inf()) {
        out[47] = 0x40;
        return;
    }
    const Fq xc = from_mont(p.x);
    for (int i = 0; i < 48; i++) out[i] = (uint8_t)(xc.l[i >> 2] >> (8 * (i & 3)));
    if (fq_is_larger_root(p.y)) out[47] |= 0x80;
}

}  // namespace mi
