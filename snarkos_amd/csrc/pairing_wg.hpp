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

struct alignas(16) PB {
    Fq c[12];  // coefficient of w^k
};

struct alignas(16) PwShared {
    PB f, t, a, b, c, d, x, y;
    alignas(16) Fq line[MI_MAX_PAIRS][8];  // per pair: coefficients at w^0, w^1, w^3, w^7, w^9 (+ padding)
    alignas(16) Fq px[MI_MAX_PAIRS], py[MI_MAX_PAIRS];
};

// ---- LDS <-> register copies, 16 bytes at a time (PwShared members are 16-byte aligned)
__device__ __forceinline__ Fq lds_load(const Fq *p) {
    const uint4 *q = reinterpret_cast<const uint4 *>(p);
    uint4 v0 = q[0], v1 = q[1], v2 = q[2];
    Fq r;
    r.l[0] = v0.x; r.l[1] = v0.y; r.l[2] = v0.z; r.l[3] = v0.w;
    r.l[4] = v1.x; r.l[5] = v1.y; r.l[6] = v1.z; r.l[7] = v1.w;
    r.l[8] = v2.x; r.l[9] = v2.y; r.l[10] = v2.z; r.l[11] = v2.w;
    return r;
}
__device__ __forceinline__ void lds_store(Fq *p, const Fq &v) {
    uint4 *q = reinterpret_cast<uint4 *>(p);
    q[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    q[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
    q[2] = make_uint4(v.l[8], v.l[9], v.l[10], v.l[11]);
}

// position (power of w) of the s-th stored coefficient of the right operand: dense 0..11, line {0,1,3,7,9}
template <int M>
__device__ __forceinline__ int pb_pos(int s) {
    if (M == 12) return s;
    return s == 0 ? 0 : s == 1 ? 1 : s == 2 ? 3 : s == 3 ? 7 : 9;
}

// out = A * B, B given compactly as its M stored coefficients bc[0..M).  All PW_THREADS lanes must call.
// Lane (k = t / 16, s = t % 16): the 16-lane row k produces coefficient k.  Lane s < M multiplies
// a_((k - pos(s)) mod 12) by bc[s], turns a wrapped product p (degree >= 12) into 8q - 5p, and the row sums its
// <= 12 values with a 4-step xor-shuffle tree of lazy 384-bit additions (< 100 q < 2^384); one binary
// reduction (conditional subtraction of 64q .. q) per coefficient.  out may alias A or bc (reads finish before the barrier).
template <int M>
__device__ __attribute__((noinline)) void pb_mul_t(PwShared &S, PB *out, const PB *A, const Fq *bc) {
    const int t = threadIdx.x, k = t >> 4, s = t & 15;
    uint32_t acc[12];
#pragma unroll
    for (int i = 0; i < 12; i++) acc[i] = 0;
    if (s < M) {
        int i = k - pb_pos<M>(s);
        const bool wrap = i < 0;
        if (wrap) i += 12;
        Fq p = mul(lds_load(&A->c[i]), lds_load(&bc[s]));
#pragma unroll
        for (int j = 0; j < 12; j++) acc[j] = p.l[j];
        if (wrap) {  // w^12 = -5: contribute 8q - 5p (in (3q, 8q]) instead of a reduced -5p; the row sum stays < 2^384
            uint32_t t5[12], c = 0, bw = 0;
#pragma unroll
            for (int j = 0; j < 12; j++) t5[j] = addc32((p.l[j] << 2) | (j ? (p.l[j - 1] >> 30) : 0u), p.l[j], c, &c);
#pragma unroll
            for (int j = 0; j < 12; j++) {
                const uint32_t q8 = (FqP::MOD[j] << 3) | (j ? (FqP::MOD[j - 1] >> 29) : 0u);
                acc[j] = subc32(q8, t5[j], bw, &bw);
            }
        }
    }
#pragma unroll
    for (int d = 8; d >= 1; d >>= 1) {
        uint32_t c = 0;
#pragma unroll
        for (int j = 0; j < 12; j++) acc[j] = addc32(acc[j], (uint32_t)__shfl_xor((int)acc[j], d, 16), c, &c);
    }
    // acc < 100 q < 2^384: subtract 64q, 32q, ..., q where possible
#pragma unroll
    for (int sh = 6; sh >= 0; sh--) {
        uint32_t d[12], borrow = 0;
#pragma unroll
        for (int j = 0; j < 12; j++) {
            const uint32_t qs = sh ? ((FqP::MOD[j] << sh) | (j ? (FqP::MOD[j - 1] >> (32 - sh)) : 0u)) : FqP::MOD[j];
            d[j] = subc32(acc[j], qs, borrow, &borrow);
        }
#pragma unroll
        for (int j = 0; j < 12; j++) acc[j] = borrow ? acc[j] : d[j];
    }
    __syncthreads();
    if (s == 0) {
        Fq r;
#pragma unroll
        for (int j = 0; j < 12; j++) r.l[j] = acc[j];
        lds_store(&out->c[k], r);
    }
    __syncthreads();
}
__device__ __forceinline__ void pb_mul(PwShared &S, PB *out, const PB *A, const PB *B) { pb_mul_t<12>(S, out, A, B->c); }
__device__ __forceinline__ void pb_mul_line(PwShared &S, PB *out, const PB *A, const Fq *line5) { pb_mul_t<5>(S, out, A, line5); }
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
// out = A^-1 by the norm to Fq: A^-1 = Abar / N, Abar = prod_{e=1..11} frob_e(A), N = A * Abar in Fq.
// Frobenius is diagonal here, so this stays in the workgroup-wide rounds; the one serial step is a binary-EGCD
// inversion in Fq on a single lane.  With U = A frob1(A) frob2(A) (e = 0,1,2) and P2 = A frob1(A):
//   W = U * frob3(U) * frob6(U) * frob9(P2)   covers e = 0..10,   Abar = frob1(W),   frob6 = conj, frob9 = conj o frob3.
// Uses S.x, S.y, S.c, S.d as scratch (callers invert before they use those).
__device__ __attribute__((noinline)) void pb_inv(PwShared &S, PB *out, const PB *A, const PairingConsts *pc) {
    pb_frob(S, &S.x, A, pc->frob[0]);         // x = frob1(A)
    pb_mul(S, &S.c, A, &S.x);                 // c = P2 = A frob1(A)
    pb_frob(S, &S.x, A, pc->frob[1]);         // x = frob2(A)
    pb_mul(S, &S.d, &S.c, &S.x);              // d = U
    pb_frob(S, &S.x, &S.d, pc->frob[2]);      // x = frob3(U)
    pb_mul(S, &S.y, &S.d, &S.x);              // y = U frob3(U)
    pb_conj(S, &S.x, &S.d);                   // x = frob6(U)
    pb_mul(S, &S.y, &S.y, &S.x);              // y = U frob3(U) frob6(U)
    pb_frob(S, &S.x, &S.c, pc->frob[2]);      // x = frob3(P2)
    pb_conj(S, &S.x, &S.x);                   // x = frob9(P2)
    pb_mul(S, &S.y, &S.y, &S.x);              // y = W
    pb_frob(S, &S.x, &S.y, pc->frob[0]);      // x = Abar
    pb_mul(S, &S.y, A, &S.x);                 // y = N (coefficient 0; the others vanish)
    const int t = threadIdx.x;
    if (t == 0) lds_store(&S.y.c[1], invq(lds_load(&S.y.c[0])));  // 1/N parked in y.c[1]
    __syncthreads();
    Fq v;
    if (t < 12) v = mul(lds_load(&S.x.c[t]), lds_load(&S.y.c[1]));
    __syncthreads();
    if (t < 12) lds_store(&out->c[t], v);
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
    pb_inv(S, &S.a, &S.f, pc);
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
