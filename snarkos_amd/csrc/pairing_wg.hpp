// Workgroup-cooperative pairing product for gfx950: 192 lanes (3 waves) share one Fq12 value in LDS.
//
// A pairing product is one long dependency chain (63 squarings + 69 line multiplications per pair in the
// Miller loop, ~350 more Fq12 operations in the final exponentiation), so one lane per pairing is purely
// latency bound (~90 ms).  Here every Fq12 operation is one "round": Fq12 is held in the power basis
// Fq[w]/(w^12 + 5) (the tower basis element u^b v^j w^i IS w^(6b+2j+i), so this is a relabelling), the
// 144 (dense) or 60 (sparse line) coefficient products a_i * b_j go one per lane, and 12 lanes then fold
// them: c_k = sum_{i+j=k} - 5 sum_{i+j=k+12}, with lazy 384-bit sums and a single reduction.
// Frobenius is diagonal in this basis (c_k *= gamma_e^k), conjugation negates the odd coefficients.
// Replaces Bls12_377::{miller_loop, final_exponentiation, product_of_pairings}.
#pragma once
#include "tower.hpp"

namespace mi {

constexpr int PW_THREADS = 192;

struct PB {
    Fq c[12];  // coefficient of w^k
};

struct PwShared {
    PB f, t, a, b, c, d, x, y;
    Fq prod[144];
    Fq line[MI_MAX_PAIRS][5];  // coefficients at w^0, w^1, w^3, w^7, w^9
    Fq px[MI_MAX_PAIRS], py[MI_MAX_PAIRS];
};

// ---- LDS <-> register copies through 32-bit words (LDS objects are only 4-byte aligned as far as we assume)
__device__ __forceinline__ Fq lds_load(const Fq *p) {
    Fq r;
#pragma unroll
    for (int i = 0; i < 12; i++) r.l[i] = p->l[i];
    return r;
}
__device__ __forceinline__ void lds_store(Fq *p, const Fq &v) {
#pragma unroll
    for (int i = 0; i < 12; i++) p->l[i] = v.l[i];
}

// acc (384-bit, no reduction) += v
__device__ __forceinline__ void wide_add(uint32_t acc[12], const Fq &v) {
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        uint64_t s = (uint64_t)acc[i] + v.l[i] + c;
        acc[i] = (uint32_t)s;
        c = (uint32_t)(s >> 32);
    }
}
// (lo - 5 hi) mod q for lo < 12q, hi < 12q:  lo + 64q - 5hi in (4q, 76q), then binary conditional subtraction
__device__ __forceinline__ Fq pb_reduce(const uint32_t lo[12], const uint32_t hi[12]) {
    uint32_t t[12], v[12];
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) {  // t = 5 * hi = (hi << 2) + hi
        uint32_t sh = (hi[i] << 2) | (i ? (hi[i - 1] >> 30) : 0u);
        uint64_t s = (uint64_t)sh + hi[i] + c;
        t[i] = (uint32_t)s;
        c = (uint32_t)(s >> 32);
    }
    c = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) {  // v = lo + (q << 6)
        uint32_t q6 = (FqP::MOD[i] << 6) | (i ? (FqP::MOD[i - 1] >> 26) : 0u);
        uint64_t s = (uint64_t)lo[i] + q6 + c;
        v[i] = (uint32_t)s;
        c = (uint32_t)(s >> 32);
    }
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) {  // v -= t
        uint64_t d = (uint64_t)v[i] - t[i] - borrow;
        v[i] = (uint32_t)d;
        borrow = (uint32_t)(d >> 63);
    }
#pragma unroll
    for (int s = 6; s >= 0; s--) {  // if v >= (q << s): v -= q << s
        uint32_t d[12];
        borrow = 0;
#pragma unroll
        for (int i = 0; i < 12; i++) {
            uint32_t qs = s ? ((FqP::MOD[i] << s) | (i ? (FqP::MOD[i - 1] >> (32 - s)) : 0u)) : FqP::MOD[i];
            uint64_t dd = (uint64_t)v[i] - qs - borrow;
            d[i] = (uint32_t)dd;
            borrow = (uint32_t)(dd >> 63);
        }
#pragma unroll
        for (int i = 0; i < 12; i++) v[i] = borrow ? v[i] : d[i];
    }
    Fq r;
#pragma unroll
    for (int i = 0; i < 12; i++) r.l[i] = v[i];
    return r;
}

// out = A * B where B has m non-zero coefficients at positions pos[0..m) held compactly in bc[0..m).
// All PW_THREADS lanes must call.  out may alias A (products are read before the barrier).
__device__ __attribute__((noinline)) void pb_mul_generic(PwShared &S, PB *out, const PB *A, const Fq *bc, const int *pos, int m) {
    const int t = threadIdx.x;
    if (t < 12 * m) {
        const int i = t / m, jj = t - i * m;
        lds_store(&S.prod[t], mul(lds_load(&A->c[i]), lds_load(&bc[jj])));
    }
    __syncthreads();
    if (t < 12) {
        uint32_t lo[12], hi[12];
#pragma unroll
        for (int i = 0; i < 12; i++) { lo[i] = 0; hi[i] = 0; }
        for (int jj = 0; jj < m; jj++) {
            int i = t - pos[jj];
            if (i >= 0) wide_add(lo, lds_load(&S.prod[i * m + jj]));
            else wide_add(hi, lds_load(&S.prod[(i + 12) * m + jj]));
        }
        lds_store(&out->c[t], pb_reduce(lo, hi));
    }
    __syncthreads();
}
__device__ __forceinline__ void pb_mul(PwShared &S, PB *out, const PB *A, const PB *B) {
    const int pos[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
    pb_mul_generic(S, out, A, B->c, pos, 12);
}
__device__ __forceinline__ void pb_mul_line(PwShared &S, PB *out, const PB *A, const Fq *line5) {
    const int pos[5] = {0, 1, 3, 7, 9};
    pb_mul_generic(S, out, A, line5, pos, 5);
}
// out = frobenius^e(A): c_k *= gamma_e^k
__device__ __forceinline__ void pb_frob(PwShared &S, PB *out, const PB *A, const Fq *gamma /* global, 12 */) {
    const int t = threadIdx.x;
    Fq v;
    if (t < 12) v = mul(lds_load(&A->c[t]), gamma[t]);
    __syncthreads();
    if (t < 12) lds_store(&out->c[t], v);
    __syncthreads();
}
// out = conj(A): negate the odd coefficients
__device__ __forceinline__ void pb_conj(PwShared &S, PB *out, const PB *A) {
    const int t = threadIdx.x;
    Fq v;
    if (t < 12) {
        v = lds_load(&A->c[t]);
        if (t & 1) v = neg(v);
    }
    __syncthreads();
    if (t < 12) lds_store(&out->c[t], v);
    __syncthreads();
}
__device__ __forceinline__ void pb_copy(PwShared &S, PB *out, const PB *A) {
    const int t = threadIdx.x;
    Fq v;
    if (t < 12) v = lds_load(&A->c[t]);
    __syncthreads();
    if (t < 12) lds_store(&out->c[t], v);
    __syncthreads();
}
// out = A^x (x = BLS seed); uses S.y as the running value
__device__ __attribute__((noinline)) void pb_exp_by_x(PwShared &S, PB *out, const PB *A) {
    pb_copy(S, &S.y, A);
    for (int i = 62; i >= 0; i--) {
        pb_mul(S, &S.y, &S.y, &S.y);
        if ((MI_BLS_X >> i) & 1) pb_mul(S, &S.y, &S.y, A);
    }
    pb_copy(S, out, &S.y);
}
// out = A^-1: one lane, tower formulas (the only serial piece: ~180 products + one binary-EGCD inversion)
__device__ __attribute__((noinline)) void pb_inv(PwShared &S, PB *out, const PB *A) {
    if (threadIdx.x == 0) {
        Fq12 f;
        Fq *fc = &f.c0.c0.c0;
        for (int i = 0; i < 2; i++)
            for (int j = 0; j < 3; j++)
                for (int b = 0; b < 2; b++) fc[(i * 3 + j) * 2 + b] = lds_load(&A->c[i + 2 * j + 6 * b]);
        Fq12 g = inv(f);
        const Fq *gc = &g.c0.c0.c0;
        for (int i = 0; i < 2; i++)
            for (int j = 0; j < 3; j++)
                for (int b = 0; b < 2; b++) lds_store(&out->c[i + 2 * j + 6 * b], gc[(i * 3 + j) * 2 + b]);
    }
    __syncthreads();
}

// Miller loop over n pairs (none at infinity); P coordinates in S.px / S.py; result in S.f
__device__ __attribute__((noinline)) void pw_miller(PwShared &S, const G2Prepared *const *qs, int n) {
    const int t = threadIdx.x;
    if (t < 12) lds_store(&S.f.c[t], t == 0 ? Fq::one() : Fq::zero());
    __syncthreads();
    int idx = 0;
    for (int i = 62; i >= 0; i--) {
        const int steps = 1 + (int)((MI_BLS_X >> i) & 1);
        pb_mul(S, &S.f, &S.f, &S.f);
        for (int s = 0; s < steps; s++, idx++) {
            // line coefficients for all pairs: lane (p, role)
            if (t < 5 * n) {
                const int p = t / 5, role = t - p * 5;
                const Line *ln = &qs[p]->lines[idx];
                Fq v;
                if (role == 0) v = lds_load(&S.py[p]);
                else if (role == 1) v = mul(ln->c1.c0, lds_load(&S.px[p]));
                else if (role == 2) v = ln->c3.c0;
                else if (role == 3) v = mul(ln->c1.c1, lds_load(&S.px[p]));
                else v = ln->c3.c1;
                lds_store(&S.line[p][role], v);
            }
            __syncthreads();
            for (int p = 0; p < n; p++) pb_mul_line(S, &S.f, &S.f, S.line[p]);
        }
    }
}

// S.f <- S.f ^ (3 (q^12 - 1) / r)
__device__ __attribute__((noinline)) void pw_final_exp(PwShared &S, const PairingConsts *pc) {
    // easy part
    pb_inv(S, &S.a, &S.f);
    pb_conj(S, &S.t, &S.f);
    pb_mul(S, &S.t, &S.t, &S.a);
    pb_frob(S, &S.a, &S.t, pc->frob[1]);
    pb_mul(S, &S.t, &S.a, &S.t);
    // hard part (cubed): (x-1)^2 (x+q) (x^2+q^2-1) + 3
    pb_exp_by_x(S, &S.a, &S.t);
    pb_conj(S, &S.x, &S.t);
    pb_mul(S, &S.a, &S.a, &S.x);
    pb_exp_by_x(S, &S.b, &S.a);
    pb_conj(S, &S.x, &S.a);
    pb_mul(S, &S.b, &S.b, &S.x);
    pb_exp_by_x(S, &S.c, &S.b);
    pb_frob(S, &S.x, &S.b, pc->frob[0]);
    pb_mul(S, &S.c, &S.c, &S.x);
    pb_exp_by_x(S, &S.d, &S.c);
    pb_exp_by_x(S, &S.d, &S.d);
    pb_frob(S, &S.x, &S.c, pc->frob[1]);
    pb_mul(S, &S.d, &S.d, &S.x);
    pb_conj(S, &S.x, &S.c);
    pb_mul(S, &S.d, &S.d, &S.x);
    pb_mul(S, &S.x, &S.t, &S.t);
    pb_mul(S, &S.x, &S.x, &S.t);
    pb_mul(S, &S.f, &S.d, &S.x);
}

// full product over the pairs whose P is not infinity and whose Q is not infinity; result (tower order) to out
__device__ __forceinline__ void pw_pairing_product(PwShared &S, const G1Affine *ps, const G2Prepared *const *qs, int n,
                                                   const PairingConsts *pc, Fq12 *out) {
    __shared__ const G2Prepared *live_q[MI_MAX_PAIRS];
    __shared__ int live_n;
    const int t = threadIdx.x;
    if (t == 0) {
        int m = 0;
        for (int p = 0; p < n; p++) {
            if (qs[p]->infinity || ps[p].is_inf()) continue;
            live_q[m] = qs[p];
            S.px[m] = ps[p].x;
            S.py[m] = ps[p].y;
            m++;
        }
        live_n = m;
    }
    __syncthreads();
    pw_miller(S, live_q, live_n);
    pw_final_exp(S, pc);
    if (t < 12) {
        // tower index (i, j, b) holds the coefficient of w^(i + 2j + 6b)
        const int b = t / 6, rem = t - 6 * b, j = rem >> 1, i = rem & 1;
        Fq *o = &out->c0.c0.c0;
        o[(i * 3 + j) * 2 + b] = lds_load(&S.f.c[t]);
    }
    __syncthreads();
}

}  // namespace mi
