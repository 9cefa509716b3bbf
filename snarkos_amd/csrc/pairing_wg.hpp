// Workgroup-cooperative pairing product for gfx950: 192 lanes (3 waves) share one Fq12 value in LDS.
//
// A pairing product is one long dependency chain (63 squarings + 69 line multiplications per pair in the
// Miller loop, ~350 more Fq12 operations in the final exponentiation), so one lane per pairing is purely
// latency bound (~90 ms).  Here every Fq12 operation is one "round": Fq12 is held in the power basis
// Fq[w]/(w^12 + 5) (the tower basis element u^b v^j w^i IS w^(6b+2j+i), so this is a relabelling), the
// 144 coefficient products a_i * b_j go one per lane, and each 16-lane row folds its twelve:
// c_k = sum_{i+j=k} - 5 sum_{i+j=k+12}, with lazy 384-bit sums and a single reduction; the rows' four spare lanes
// can carry the product of two sparse line functions in the same round (pb_round).
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
    PB f, t, a, b, c, d, x, y, z, g;
    alignas(16) Fq one;  // the constant line of a couple's missing second pair
    alignas(16) Fq px[MI_MAX_PAIRS], py[MI_MAX_PAIRS];
    // every line of the current couple's Miller loop evaluated at its P up front, [step][pair][w^1, w^3, w^7, w^9]
    // (the w^0 coefficient is py[pair]); 26 KB
    alignas(16) Fq lines2[MI_MILLER_LINES][2][4];
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

// ---------------------------------------------------------------------------------------------------------------
// The round (every Fq12 operation of the Miller loop and of the final exponentiation).  out = A * B: lane (k = t / 16,
// s = t % 16 < 12) of the 16-lane row k multiplies a_((k - s) mod 12) by b_s; the row's sum is coefficient k.
//   * products are taken without their final conditional subtraction (< 1.007 q); a wrapped product p contributes
//     6q - 5p; a row's lazy sum stays < 67 q < 2^384;
//   * the row sum is an inclusive scan over the 16-lane row with DPP row_shr:1/2/4/8 -- no LDS traffic; lane 11 ends
//     with the sum of lanes 0..11;
//   * one reduction: quotient estimate from the top limb (k <= floor(acc / q) <= k + 1), acc - k q, one conditional
//     subtraction;
//   * the four spare lanes of every row (s = 12..15) can carry a second, independent product in the same round: the
//     product of two sparse line functions (5 x 5 coefficients -> 10 non-zero rows of at most 4 products each, table
//     below), summed after the first two scan steps in lane 15.
struct Q6 {
    uint32_t l[12];
};
constexpr Q6 pw_make_q6() {
    Q6 r{};
    uint64_t c = 0;
    for (int j = 0; j < 12; j++) {
        const uint64_t v = (uint64_t)FqP::MOD[j] * 6u + c;
        r.l[j] = (uint32_t)v;
        c = v >> 32;
    }
    return r;
}
// line x line: for output row k, up to four products l1[i] * l2[j] (stored-coefficient indices; positions 0,1,3,7,9);
// code = i | j << 3 | wrapped << 6 | valid << 7
#define PW_LL(i, j, w) (uint8_t)((i) | ((j) << 3) | ((w) << 6) | 0x80)
__device__ const uint8_t PW_LL_TABLE[12][4] = {
    {PW_LL(0, 0, 0), PW_LL(2, 4, 1), PW_LL(4, 2, 1), 0},               // w^0: 0+0, 3+9, 9+3
    {PW_LL(0, 1, 0), PW_LL(1, 0, 0), 0, 0},                            // w^1
    {PW_LL(1, 1, 0), PW_LL(3, 3, 1), 0, 0},                            // w^2: 1+1, 7+7
    {PW_LL(0, 2, 0), PW_LL(2, 0, 0), 0, 0},                            // w^3
    {PW_LL(1, 2, 0), PW_LL(2, 1, 0), PW_LL(3, 4, 1), PW_LL(4, 3, 1)},  // w^4: 1+3, 3+1, 7+9, 9+7
    {0, 0, 0, 0},                                                      // w^5
    {PW_LL(2, 2, 0), PW_LL(4, 4, 1), 0, 0},                            // w^6: 3+3, 9+9
    {PW_LL(0, 3, 0), PW_LL(3, 0, 0), 0, 0},                            // w^7
    {PW_LL(1, 3, 0), PW_LL(3, 1, 0), 0, 0},                            // w^8
    {PW_LL(0, 4, 0), PW_LL(4, 0, 0), 0, 0},                            // w^9
    {PW_LL(1, 4, 0), PW_LL(4, 1, 0), PW_LL(2, 3, 0), PW_LL(3, 2, 0)},  // w^10: 1+9, 9+1, 3+7, 7+3
    {0, 0, 0, 0},                                                      // w^11
};
#undef PW_LL

// acc += (acc of the lane N places lower in the 16-lane row), 384 bits: the DPP shift is fused into the add / add-with-
// carry instructions themselves (row_shr:N, lanes without a source read 0), 12 instructions per step instead of a
// v_mov_b32_dpp + v_addc pair per limb (measured: the four steps are 0.38 us of a 2.3 us round).
template <int N>
__device__ __forceinline__ void pw_row_scan_step(uint32_t (&acc)[12]) {
    // s_nop 1: a DPP source written by the preceding VALU instruction needs two wait states, and the compiler does not
    // look inside the asm statement
    asm volatile(
        "s_nop 1\n\t"
        "v_add_co_u32_dpp %0, vcc, %0, %0 row_shr:%12 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
        "v_addc_co_u32_dpp %1, vcc, %1, %1, vcc row_shr:%12 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
        "v_addc_co_u32_dpp %2, vcc, %2, %2, vcc row_shr:%12 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
        "v_addc_co_u32_dpp %3, vcc, %3, %3, vcc row_shr:%12 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
        "v_addc_co_u32_dpp %4, vcc, %4, %4, vcc row_shr:%12 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
        "v_addc_co_u32_dpp %5, vcc, %5, %5, vcc row_shr:%12 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
        "v_addc_co_u32_dpp %6, vcc, %6, %6, vcc row_shr:%12 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
        "v_addc_co_u32_dpp %7, vcc, %7, %7, vcc row_shr:%12 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
        "v_addc_co_u32_dpp %8, vcc, %8, %8, vcc row_shr:%12 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
        "v_addc_co_u32_dpp %9, vcc, %9, %9, vcc row_shr:%12 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
        "v_addc_co_u32_dpp %10, vcc, %10, %10, vcc row_shr:%12 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
        "v_addc_co_u32_dpp %11, vcc, %11, %11, vcc row_shr:%12 row_mask:0xf bank_mask:0xf bound_ctrl:0"
        : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]),
          "+v"(acc[8]), "+v"(acc[9]), "+v"(acc[10]), "+v"(acc[11])
        : "n"(N)
        : "vcc");
}

// out = A * B (dense).  WITH_L: also outL = l1 * l2, l1 / l2 the five stored coefficients of two line functions
// (pointer to the w^0 coefficient and to the four others).  All PW_THREADS lanes must call.
// NOALIAS: out / outL are not read by this round (the callers ping-pong between buffers), so the barrier between the
// operand reads and the result stores is dropped: one barrier per round instead of two (measured 0.12 us of 2.3).
template <bool WITH_L, bool NOALIAS = false>
__device__ __attribute__((noinline)) void pb_round(PwShared &S, PB *out, const PB *A, const PB *B, PB *outL, const Fq *l1_0,
                                                   const Fq *l1_r, const Fq *l2_0, const Fq *l2_r) {
    constexpr Q6 q6 = pw_make_q6();
    const int t = threadIdx.x, k = t >> 4, s = t & 15;
    uint32_t acc[12];
#pragma unroll
    for (int j = 0; j < 12; j++) acc[j] = 0;
    Fq a, b;
    bool active = false, wrap = false;
    if (s < 12) {
        int i = k - s;
        wrap = i < 0;
        if (wrap) i += 12;
        a = lds_load(&A->c[i]);
        b = lds_load(&B->c[s]);
        active = true;
    } else if (WITH_L) {
        const uint32_t code = PW_LL_TABLE[k][s - 12];
        if (code & 0x80u) {
            const uint32_t i = code & 7u, j = (code >> 3) & 7u;
            a = lds_load(i == 0 ? l1_0 : &l1_r[i - 1]);
            b = lds_load(j == 0 ? l2_0 : &l2_r[j - 1]);
            wrap = (code & 0x40u) != 0;
            active = true;
        }
    }
    if (active) {
        const Fq p = mul_nr(a, b);
#pragma unroll
        for (int j = 0; j < 12; j++) acc[j] = p.l[j];
        if (wrap) {  // w^12 = -5: 6q - 5p, in (0.9 q, 6 q]
            uint32_t t5[12], c = 0, bw = 0;
#pragma unroll
            for (int j = 0; j < 12; j++) t5[j] = addc32((p.l[j] << 2) | (j ? (p.l[j - 1] >> 30) : 0u), p.l[j], c, &c);
#pragma unroll
            for (int j = 0; j < 12; j++) acc[j] = subc32(q6.l[j], t5[j], bw, &bw);
        }
    }
    pw_row_scan_step<1>(acc);
    pw_row_scan_step<2>(acc);
    uint32_t v[12];
    if (WITH_L) {
#pragma unroll
        for (int j = 0; j < 12; j++) v[j] = acc[j];  // lane 15: lanes 12..15 summed = the line product's row
    }
    pw_row_scan_step<4>(acc);
    pw_row_scan_step<8>(acc);
    if (WITH_L) {
#pragma unroll
        for (int j = 0; j < 12; j++) v[j] = s == 15 ? v[j] : acc[j];
    } else {
#pragma unroll
        for (int j = 0; j < 12; j++) v[j] = acc[j];
    }
    // v < 67 q: k = floor(v11 * floor(2^40 / (q11 + 1)) / 2^40) <= v / q < k + 2
    {
        // (independent 32 x 32 products for k q were tried: v_mul_lo / v_mul_hi are slower than this chain of
        // v_mad_u64_u32: 2.70 -> 2.89 us per round)
        const uint32_t kq = __umulhi(v[11], 38996u) >> 8;
        uint32_t carry = 0, borrow = 0;
#pragma unroll
        for (int j = 0; j < 12; j++) {
            const uint64_t pr = (uint64_t)kq * FqP::MOD[j] + carry;
            carry = (uint32_t)(pr >> 32);
            v[j] = subc32(v[j], (uint32_t)pr, borrow, &borrow);
        }
        // no final conditional subtraction: v is left in [0, 2q).  The next round's unreduced products of operands
        // below 2q stay below 1.03 q (4 q^2 / R + q), so every bound above holds; the few consumers that need the
        // canonical representative (negation, inversion, the result) reduce once themselves.
    }
    if (!NOALIAS) __syncthreads();
    if (s == 11 || (WITH_L && s == 15)) {
        Fq r;
#pragma unroll
        for (int j = 0; j < 12; j++) r.l[j] = v[j];
        lds_store(s == 11 ? &out->c[k] : &outL->c[k], r);
    }
    __syncthreads();
}
__device__ __forceinline__ void pb_mul(PwShared &S, PB *out, const PB *A, const PB *B) {
    pb_round<false>(S, out, A, B, nullptr, nullptr, nullptr, nullptr, nullptr);
}
// out must not be A or B
__device__ __forceinline__ void pb_mul_na(PwShared &S, PB *out, const PB *A, const PB *B) {
    pb_round<false, true>(S, out, A, B, nullptr, nullptr, nullptr, nullptr, nullptr);
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
        reduce_once(v);  // rounds leave values in [0, 2q)
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
    // ping-pong between S.y and S.z so that no round writes an operand (one barrier per round); A is neither
    PB *cur = &S.y, *alt = &S.z;
    pb_copy(S, cur, A);
    for (int i = 62; i >= 0; i--) {
        pb_mul_na(S, alt, cur, cur);
        PB *sw = cur; cur = alt; alt = sw;
        if ((MI_BLS_X >> i) & 1) {
            pb_mul_na(S, alt, cur, A);
            sw = cur; cur = alt; alt = sw;
        }
    }
    pb_copy(S, out, cur);
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
    if (t == 0) {
        Fq nrm = lds_load(&S.y.c[0]);
        reduce_once(nrm);  // rounds leave values in [0, 2q)
        lds_store(&S.y.c[1], invq(nrm));  // 1/N parked in y.c[1]
    }
    __syncthreads();
    Fq v;
    if (t < 12) v = mul(lds_load(&S.x.c[t]), lds_load(&S.y.c[1]));
    __syncthreads();
    if (t < 12) lds_store(&out->c[t], v);
    __syncthreads();
}

// Miller loop of one COUPLE of live pairs (p0, p0 + 1) -- every KZG / verifier check is one couple -- into S.f.  All
// line coefficients first (552 values, three per lane), then per doubling step TWO rounds -- [f^2 beside l_1 l_2] and
// [f^2 * (l_1 l_2)] -- instead of three rounds plus a line evaluation; an addition step costs one more round (its line
// product rides in the second round's spare lanes).  A single pair (np = 1) runs as a couple whose second line is the
// constant 1.  Products over more pairs multiply the couples' values (pw_pairing_product).
__device__ __attribute__((noinline)) void pw_miller2(PwShared &S, const G2Prepared *const *qs, int p0, int np) {
    const int t = threadIdx.x;
    for (int item = t; item < MI_MILLER_LINES * 8; item += PW_THREADS) {
        const int idx = item >> 3, p = (item >> 2) & 1, role = item & 3;
        Fq v = Fq::zero();
        if (p < np) {
            const Line *ln = &qs[p0 + p]->lines[idx];
            if (role == 0) v = mul(ln->c1.c0, lds_load(&S.px[p0 + p]));       // w^1
            else if (role == 1) v = ln->c3.c0;                                 // w^3
            else if (role == 2) v = mul(ln->c1.c1, lds_load(&S.px[p0 + p]));  // w^7
            else v = ln->c3.c1;                                                // w^9
        }
        lds_store(&S.lines2[idx][p][role], v);
    }
    if (t < 12) lds_store(&S.f.c[t], t == 0 ? Fq::one() : Fq::zero());
    if (t == 12) lds_store(&S.one, Fq::one());
    __syncthreads();
    const Fq *y1 = &S.py[p0], *y2 = np == 2 ? &S.py[p0 + 1] : &S.one;
    int idx = 0;
    PB *cur = &S.f, *alt = &S.g;  // ping-pong: no round writes one of its operands
    for (int i = 62; i >= 0; i--) {
        pb_round<true, true>(S, alt, cur, cur, &S.t, y1, S.lines2[idx][0], y2, S.lines2[idx][1]);  // alt = cur^2, t = l l'
        idx++;
        if ((MI_BLS_X >> i) & 1) {
            pb_round<true, true>(S, cur, alt, &S.t, &S.a, y1, S.lines2[idx][0], y2, S.lines2[idx][1]);  // cur = alt t
            idx++;
            pb_mul_na(S, alt, cur, &S.a);
            PB *sw = cur; cur = alt; alt = sw;
        } else {
            pb_mul_na(S, cur, alt, &S.t);
        }
    }
    if (cur != &S.f) pb_copy(S, &S.f, cur);
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
    if (live_n == 0) {
        if (t < 12) lds_store(&S.f.c[t], t == 0 ? Fq::one() : Fq::zero());
        __syncthreads();
    }
    for (int c = 0; c < live_n; c += 2) {
        if (c > 0) pb_copy(S, &S.b, &S.f);  // the couples before this one
        pw_miller2(S, live_q, c, live_n - c >= 2 ? 2 : 1);
        if (c > 0) pb_mul(S, &S.f, &S.f, &S.b);
    }
    pw_final_exp(S, pc);
    if (t < 12) {
        // tower index (i, j, b) holds the coefficient of w^(i + 2j + 6b)
        const int b = t / 6, rem = t - 6 * b, j = rem >> 1, i = rem & 1;
        Fq *o = &out->c0.c0.c0;
        Fq res = lds_load(&S.f.c[t]);
        reduce_once(res);  // rounds leave values in [0, 2q)
        o[(i * 3 + j) * 2 + b] = res;
    }
    __syncthreads();
}

}  // namespace mi
