// BLS12-377 optimal-ate pairing on gfx950 (host + device): tower Fq2/Fq6/Fq12, prepared G2 lines,
// Miller loop over x = 0x8508c00000000001, final exponentiation.
//
// Replaces snarkvm_curves::bls12_377::Bls12_377::{miller_loop, final_exponentiation,
// product_of_pairings} and G2Prepared (snarkVM 1.0.0 @ dea322b, un-vendored).
//   Fq2 = Fq[u]/(u^2+5), Fq6 = Fq2[v]/(v^3-u), Fq12 = Fq6[w]/(w^2-v); G2 on the D-type twist
//   y^2 = x^3 + 1/u; untwist (x',y') -> (x' w^2, y' w^3); the line through T with slope L evaluated at
//   P = (xP, yP) is  yP + (-L xP) w + (L xT - yT) w^3  = c0:(yP,0,0), c1:(-L xP, L xT - yT, 0).
// G2 points are fixed per verifying key (h, beta_h), so their 69 line coefficient pairs are computed
// once on the host at key registration (the reference does the same: prepared_h / prepared_beta_h).
// final_exponentiation returns f^(3 (q^12-1)/r): the x-chain computes the cube of the reduced pairing,
//   3 (q^4-q^2+1)/r = (x-1)^2 (x+q) (x^2+q^2-1) + 3   (identity checked in tools/gen_constants.py).
#pragma once
#include <string.h>
#include "curve.hpp"

#define MI_MAX_PAIRS 4
#define MI_MILLER_LINES 69 /* 63 doublings + 6 additions */

namespace mi {

// ---------------------------------------------------------------- Fq2
struct Fq2 {
    Fq c0, c1;
    MI_HD static Fq2 zero() { return Fq2{Fq::zero(), Fq::zero()}; }
    MI_HD static Fq2 one() { return Fq2{Fq::one(), Fq::zero()}; }
    MI_HD bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
    MI_HD bool operator==(const Fq2 &o) const { return c0 == o.c0 && c1 == o.c1; }
};
MI_HD Fq mul5(const Fq &a) { Fq t = dbl(dbl(a)); return add(t, a); }
MI_HD Fq2 add(const Fq2 &a, const Fq2 &b) { return Fq2{add(a.c0, b.c0), add(a.c1, b.c1)}; }
MI_HD Fq2 sub(const Fq2 &a, const Fq2 &b) { return Fq2{sub(a.c0, b.c0), sub(a.c1, b.c1)}; }
MI_HD Fq2 neg(const Fq2 &a) { return Fq2{neg(a.c0), neg(a.c1)}; }
MI_HD Fq2 dbl(const Fq2 &a) { return Fq2{dbl(a.c0), dbl(a.c1)}; }
inline MI_NI Fq2 mul(const Fq2 &a, const Fq2 &b) {
    Fq t0 = mulq(a.c0, b.c0), t1 = mulq(a.c1, b.c1);
    Fq m = mulq(add(a.c0, a.c1), add(b.c0, b.c1));
    return Fq2{sub(t0, mul5(t1)), sub(sub(m, t0), t1)};
}
inline MI_NI Fq2 sqr(const Fq2 &a) {
    // (a0 + a1 u)^2 = a0^2 - 5 a1^2 + 2 a0 a1 u
    Fq t0 = sqrq(a.c0), t1 = sqrq(a.c1), m = mulq(a.c0, a.c1);
    return Fq2{sub(t0, mul5(t1)), dbl(m)};
}
inline MI_NI Fq2 mul_fq(const Fq2 &a, const Fq &s) { return Fq2{mulq(a.c0, s), mulq(a.c1, s)}; }
MI_HD Fq2 mul_nonres(const Fq2 &a) { return Fq2{neg(mul5(a.c1)), a.c0}; }  // times u
inline MI_NI Fq2 inv(const Fq2 &a) {
    Fq n = invq(add(sqrq(a.c0), mul5(sqrq(a.c1))));
    return Fq2{mulq(a.c0, n), neg(mulq(a.c1, n))};
}

// ---------------------------------------------------------------- Fq6
struct Fq6 {
    Fq2 c0, c1, c2;
    MI_HD static Fq6 zero() { return Fq6{Fq2::zero(), Fq2::zero(), Fq2::zero()}; }
    MI_HD static Fq6 one() { return Fq6{Fq2::one(), Fq2::zero(), Fq2::zero()}; }
};
MI_HD Fq6 add(const Fq6 &a, const Fq6 &b) { return Fq6{add(a.c0, b.c0), add(a.c1, b.c1), add(a.c2, b.c2)}; }
MI_HD Fq6 sub(const Fq6 &a, const Fq6 &b) { return Fq6{sub(a.c0, b.c0), sub(a.c1, b.c1), sub(a.c2, b.c2)}; }
MI_HD Fq6 neg(const Fq6 &a) { return Fq6{neg(a.c0), neg(a.c1), neg(a.c2)}; }
MI_HD Fq6 mul_nonres(const Fq6 &a) { return Fq6{mul_nonres(a.c2), a.c0, a.c1}; }  // times v
inline MI_NI Fq6 mul(const Fq6 &a, const Fq6 &b) {
    Fq2 v0 = mul(a.c0, b.c0), v1 = mul(a.c1, b.c1), v2 = mul(a.c2, b.c2);
    Fq2 c0 = add(v0, mul_nonres(sub(sub(mul(add(a.c1, a.c2), add(b.c1, b.c2)), v1), v2)));
    Fq2 c1 = add(sub(sub(mul(add(a.c0, a.c1), add(b.c0, b.c1)), v0), v1), mul_nonres(v2));
    Fq2 c2 = add(sub(sub(mul(add(a.c0, a.c2), add(b.c0, b.c2)), v0), v2), v1);
    return Fq6{c0, c1, c2};
}
inline MI_NI Fq6 mul_by_01(const Fq6 &a, const Fq2 &b0, const Fq2 &b1) {
    Fq2 c0 = add(mul(a.c0, b0), mul_nonres(mul(a.c2, b1)));
    Fq2 c1 = add(mul(a.c0, b1), mul(a.c1, b0));
    Fq2 c2 = add(mul(a.c1, b1), mul(a.c2, b0));
    return Fq6{c0, c1, c2};
}
inline MI_NI Fq6 inv(const Fq6 &a) {
    Fq2 t0 = sub(sqr(a.c0), mul_nonres(mul(a.c1, a.c2)));
    Fq2 t1 = sub(mul_nonres(sqr(a.c2)), mul(a.c0, a.c1));
    Fq2 t2 = sub(sqr(a.c1), mul(a.c0, a.c2));
    Fq2 d = add(mul(a.c0, t0), mul_nonres(add(mul(a.c2, t1), mul(a.c1, t2))));
    d = inv(d);
    return Fq6{mul(t0, d), mul(t1, d), mul(t2, d)};
}

// ---------------------------------------------------------------- Fq12
struct Fq12 {
    Fq6 c0, c1;
    MI_HD static Fq12 one() { return Fq12{Fq6::one(), Fq6::zero()}; }
    MI_HD bool is_one() const {
        const Fq *p = &c0.c0.c0;
        if (!(p[0] == Fq::one())) return false;
        for (int i = 1; i < 12; i++)
            if (!p[i].is_zero()) return false;
        return true;
    }
};
static_assert(sizeof(Fq12) == 576, "ABI layout");
inline MI_NI Fq12 mul(const Fq12 &a, const Fq12 &b) {
    Fq6 v0 = mul(a.c0, b.c0), v1 = mul(a.c1, b.c1);
    Fq6 c1 = sub(sub(mul(add(a.c0, a.c1), add(b.c0, b.c1)), v0), v1);
    return Fq12{add(v0, mul_nonres(v1)), c1};
}
inline MI_NI Fq12 sqr(const Fq12 &a) {
    Fq6 ab = mul(a.c0, a.c1);
    Fq6 s = mul(add(a.c0, a.c1), add(a.c0, mul_nonres(a.c1)));
    return Fq12{sub(sub(s, ab), mul_nonres(ab)), add(ab, ab)};
}
MI_HD Fq12 conj(const Fq12 &a) { return Fq12{a.c0, neg(a.c1)}; }
inline MI_NI Fq12 inv(const Fq12 &a) {
    Fq6 t = inv(sub(mul(a.c0, a.c0), mul_nonres(mul(a.c1, a.c1))));
    return Fq12{mul(a.c0, t), neg(mul(a.c1, t))};
}
// Frobenius^e: the Fq coefficient of w^k (k = i + 2j + 6b for c_i.c_j.c_b) is multiplied by gamma_e^k
inline MI_NI Fq12 frobenius(const Fq12 &a, const Fq *frob_e /* 12 constants gamma_e^k */) {
    Fq12 r;
    const Fq *src = &a.c0.c0.c0;
    Fq *dst = &r.c0.c0.c0;
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < 3; j++)
            for (int b = 0; b < 2; b++) {
                int idx = (i * 3 + j) * 2 + b;
                dst[idx] = mulq(src[idx], frob_e[i + 2 * j + 6 * b]);
            }
    return r;
}
// f * (yP + b w + c w^3)
inline MI_NI Fq12 mul_by_line(const Fq12 &f, const Fq &yp, const Fq2 &b, const Fq2 &c) {
    Fq6 A = Fq6{mul_fq(f.c0.c0, yp), mul_fq(f.c0.c1, yp), mul_fq(f.c0.c2, yp)};
    Fq6 B = mul_by_01(f.c1, b, c);
    Fq2 b2 = Fq2{add(b.c0, yp), b.c1};
    Fq6 C = mul_by_01(add(f.c0, f.c1), b2, c);
    return Fq12{add(A, mul_nonres(B)), sub(sub(C, A), B)};
}

// ---------------------------------------------------------------- G2 and prepared lines
struct G2Affine {
    Fq2 x, y;
    MI_HD bool is_inf() const { return x.is_zero() && y.is_zero(); }
};
static_assert(sizeof(G2Affine) == 192, "ABI layout");
struct Line {
    Fq2 c1, c3;  // l(P) = yP + (c1 * xP) w + c3 w^3
};
struct G2Prepared {
    int32_t infinity;
    int32_t _pad[3];
    Line lines[MI_MILLER_LINES];
};

// host only: affine chord-and-tangent walk of the Miller loop (one Fq2 inversion per line, once per key)
static void g2_prepare_host(G2Prepared &out, const G2Affine &q) {
    out.infinity = q.is_inf() ? 1 : 0;
    memset(out._pad, 0, sizeof out._pad);
    if (out.infinity) {
        memset(out.lines, 0, sizeof out.lines);
        return;
    }
    G2Affine T = q;
    int n = 0;
    auto step = [&](const G2Affine &a, const G2Affine &b, bool tangent) {
        Fq2 lam;
        if (tangent) {
            Fq2 xx = sqr(a.x);
            lam = mul(add(dbl(xx), xx), inv(dbl(a.y)));
        } else {
            lam = mul(sub(b.y, a.y), inv(sub(b.x, a.x)));
        }
        Fq2 x3 = sub(sub(sqr(lam), a.x), b.x);
        Fq2 y3 = sub(mul(lam, sub(a.x, x3)), a.y);
        out.lines[n].c1 = neg(lam);
        out.lines[n].c3 = sub(mul(lam, a.x), a.y);
        n++;
        return G2Affine{x3, y3};
    };
    for (int i = 62; i >= 0; i--) {
        T = step(T, T, true);
        if ((MI_BLS_X >> i) & 1) T = step(T, q, false);
    }
}

// ---------------------------------------------------------------- Miller loop + final exponentiation
struct PairingConsts {
    Fq frob[3][12];
    // fold constants of the workgroup-cooperative rounds (pairing_wg.hpp): plain integers, D[m] = 2^(32 m - 384) mod q
    // for m < 12, 2^(32 (m - 12)) mod q for m >= 12
    Fq dtab[25];
};

inline MI_NI Fq12 exp_by_x(const Fq12 &a) {
    Fq12 acc = a;
    for (int i = 62; i >= 0; i--) {
        acc = sqr(acc);
        if ((MI_BLS_X >> i) & 1) acc = mul(acc, a);
    }
    return acc;
}
inline MI_NI Fq12 final_exponentiation(const Fq12 &f, const PairingConsts &pc) {
    Fq12 t = mul(conj(f), inv(f));
    t = mul(frobenius(t, pc.frob[1]), t);
    Fq12 a = mul(exp_by_x(t), conj(t));
    Fq12 b = mul(exp_by_x(a), conj(a));
    Fq12 c = mul(exp_by_x(b), frobenius(b, pc.frob[0]));
    Fq12 d = exp_by_x(exp_by_x(c));
    d = mul(mul(d, frobenius(c, pc.frob[1])), conj(c));
    return mul(d, mul(sqr(t), t));
}
inline MI_NI Fq12 miller_product(const G1Affine *ps, const G2Prepared *const *qs, int n) {
    Fq12 f = Fq12::one();
    int idx = 0;
    for (int i = 62; i >= 0; i--) {
        f = sqr(f);
        for (int k = 0; k < n; k++) {
            if (qs[k]->infinity || ps[k].is_inf()) continue;
            const Line &ln = qs[k]->lines[idx];
            f = mul_by_line(f, ps[k].y, mul_fq(ln.c1, ps[k].x), ln.c3);
        }
        idx++;
        if ((MI_BLS_X >> i) & 1) {
            for (int k = 0; k < n; k++) {
                if (qs[k]->infinity || ps[k].is_inf()) continue;
                const Line &ln = qs[k]->lines[idx];
                f = mul_by_line(f, ps[k].y, mul_fq(ln.c1, ps[k].x), ln.c3);
            }
            idx++;
        }
    }
    return f;
}


}  // namespace mi
