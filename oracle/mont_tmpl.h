/* Montgomery prime-field template (CPU oracle; TEST INFRASTRUCTURE ONLY).
 *
 * Include with:  FE (type name), FN(x) (function prefix macro), NL (limb count),
 *                FMOD (modulus limbs), FINV (-p^-1 mod 2^64), FONE (R mod p), FR2 (R^2 mod p).
 * Restates snarkvm_fields::{Fp256,Fp384} Montgomery arithmetic (snarkVM 1.0.0 @ dea322b,
 * not vendored): 64-bit little-endian limbs, R = 2^(64*NL),
 * CIOS multiplication, values kept fully reduced in [0, p).
 */
#ifndef FE_EXTERNAL
typedef struct { uint64_t l[NL]; } FE;
#endif

static inline int FN(is_zero)(const FE *a) {
    uint64_t o = 0;
    for (int i = 0; i < NL; i++) o |= a->l[i];
    return o == 0;
}
static inline int FN(eq)(const FE *a, const FE *b) {
    uint64_t o = 0;
    for (int i = 0; i < NL; i++) o |= a->l[i] ^ b->l[i];
    return o == 0;
}
static inline void FN(zero)(FE *a) { for (int i = 0; i < NL; i++) a->l[i] = 0; }
static inline void FN(one)(FE *a) { for (int i = 0; i < NL; i++) a->l[i] = FONE[i]; }

/* r = a - p if a >= p */
static inline void FN(reduce_once)(FE *a, uint64_t carry) {
    uint64_t t[NL];
    unsigned __int128 b = 0;
    for (int i = 0; i < NL; i++) {
        b = (unsigned __int128)a->l[i] - FMOD[i] - (uint64_t)(b >> 127);
        t[i] = (uint64_t)b;
    }
    uint64_t borrow = (uint64_t)(b >> 127);
    if (carry || !borrow)
        for (int i = 0; i < NL; i++) a->l[i] = t[i];
}
static inline void FN(add)(FE *r, const FE *a, const FE *b) {
    unsigned __int128 c = 0;
    for (int i = 0; i < NL; i++) {
        c = (unsigned __int128)a->l[i] + b->l[i] + (uint64_t)(c >> 64);
        r->l[i] = (uint64_t)c;
    }
    FN(reduce_once)(r, (uint64_t)(c >> 64));
}
static inline void FN(sub)(FE *r, const FE *a, const FE *b) {
    unsigned __int128 d = 0;
    uint64_t t[NL];
    for (int i = 0; i < NL; i++) {
        d = (unsigned __int128)a->l[i] - b->l[i] - (uint64_t)(d >> 127);
        t[i] = (uint64_t)d;
    }
    if (d >> 127) {
        unsigned __int128 c = 0;
        for (int i = 0; i < NL; i++) {
            c = (unsigned __int128)t[i] + FMOD[i] + (uint64_t)(c >> 64);
            t[i] = (uint64_t)c;
        }
    }
    for (int i = 0; i < NL; i++) r->l[i] = t[i];
}
static inline void FN(neg)(FE *r, const FE *a) {
    FE z;
    FN(zero)(&z);
    FN(sub)(r, &z, a);
}
static inline void FN(dbl)(FE *r, const FE *a) { FN(add)(r, a, a); }

/* CIOS; the modulus leaves >= 1 spare bit in the top limb, so the running value fits NL+1 words */
static inline void FN(mul)(FE *r, const FE *a, const FE *b) {
    uint64_t t[NL + 1];
#pragma GCC unroll 8
    for (int i = 0; i < NL + 1; i++) t[i] = 0;
#pragma GCC unroll 8
    for (int i = 0; i < NL; i++) {
        unsigned __int128 acc;
        uint64_t carry = 0, bi = b->l[i];
#pragma GCC unroll 8
        for (int j = 0; j < NL; j++) {
            acc = (unsigned __int128)a->l[j] * bi + t[j] + carry;
            t[j] = (uint64_t)acc;
            carry = (uint64_t)(acc >> 64);
        }
        t[NL] = carry;
        uint64_t m = t[0] * FINV;
        acc = (unsigned __int128)m * FMOD[0] + t[0];
        carry = (uint64_t)(acc >> 64);
#pragma GCC unroll 8
        for (int j = 1; j < NL; j++) {
            acc = (unsigned __int128)m * FMOD[j] + t[j] + carry;
            t[j - 1] = (uint64_t)acc;
            carry = (uint64_t)(acc >> 64);
        }
        t[NL - 1] = t[NL] + carry;
    }
#pragma GCC unroll 8
    for (int i = 0; i < NL; i++) r->l[i] = t[i];
    FN(reduce_once)(r, 0);
}
static inline void FN(sqr)(FE *r, const FE *a) { FN(mul)(r, a, a); }

/* canonical integer (little-endian limbs) -> Montgomery form; input must be < p */
static inline void FN(from_canonical)(FE *r, const uint64_t *limbs) {
    FE t, r2;
    for (int i = 0; i < NL; i++) { t.l[i] = limbs[i]; r2.l[i] = FR2[i]; }
    FN(mul)(r, &t, &r2);
}
static inline void FN(to_canonical)(uint64_t *limbs, const FE *a) {
    FE one, t;
    FN(zero)(&one);
    one.l[0] = 1;
    FN(mul)(&t, a, &one);
    for (int i = 0; i < NL; i++) limbs[i] = t.l[i];
}
static inline void FN(from_u64)(FE *r, uint64_t v) {
    uint64_t l[NL];
    for (int i = 0; i < NL; i++) l[i] = 0;
    l[0] = v;
    FN(from_canonical)(r, l);
}
/* r = a^e, e given as `n` little-endian limbs */
static inline void FN(pow)(FE *r, const FE *a, const uint64_t *e, int n) {
    FE acc, base = *a;
    FN(one)(&acc);
    int started = 0;
    for (int i = n * 64 - 1; i >= 0; i--) {
        if (started) FN(sqr)(&acc, &acc);
        if ((e[i / 64] >> (i % 64)) & 1) {
            if (started) FN(mul)(&acc, &acc, &base);
            else { acc = base; started = 1; }
        }
    }
    *r = acc;
}
/* Fermat inversion a^(p-2); inv(0) = 0 */
static inline void FN(inv)(FE *r, const FE *a) {
    uint64_t e[NL];
    unsigned __int128 b = 0;
    for (int i = 0; i < NL; i++) {
        b = (unsigned __int128)FMOD[i] - (i == 0 ? 2 : 0) - (uint64_t)(b >> 127);
        e[i] = (uint64_t)b;
    }
    FN(pow)(r, a, e, NL);
}
