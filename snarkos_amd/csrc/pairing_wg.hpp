// Workgroup-cooperative pairing product for gfx950: 256 lanes share one Fq12 value in LDS (three waves carry its twelve
// coefficients, the fourth the value's scale: struct PB).
//
// A pairing product is one long dependency chain (63 squarings + 69 line multiplications per couple of pairs in the
// Miller loop, ~350 more Fq12 operations in the final exponentiation), so one lane per pairing is purely latency
// bound (~90 ms).  Here every Fq12 operation is one "round": Fq12 is held in the power basis Fq[w]/(w^12 + 5) (the
// tower basis element u^b v^j w^i IS w^(6b+2j+i), so this is a relabelling) and a round is bounded by the instruction
// count of its longest lane, which is what this file is organised around (round 2; arithmetic and bounds are modelled
// and checked in tools/model_pround.py):
//   * one 16-lane DPP row per output coefficient c_k = sum_s a_(k-s) b'_s, all 16 lanes busy: lane (j, g) forms
//     sum_{s = 3g..3g+2} a_(k-s) * (limbs 3j..3j+2 of b'_s) -- a 12 x 3-limb product-scanning pass over three products,
//     108 limb products, NO Montgomery reduction per product;
//   * the wrap w^12 = -5 costs nothing inside the round: every value in LDS carries its twin 10q - 5c (written by the
//     round that produced it), and a wrapped term simply reads the twin;
//   * the four lanes of a quad (same limb offset) add up with two DPP quad_perm steps; the Montgomery reduction of the
//     row's 768-bit sum is DISTRIBUTED: every lane multiplies four words of its quad's sum by constants
//     D[m] = 2^(32m - 384) mod q (36..48 limb products), four DPP row-rotate steps add the 16 lanes' 13-limb results
//     (every lane ends with the total), and one Barrett step with a two-word quotient leaves c_k in [0, 2q);
//   * one barrier per round (results ping-pong between buffers).
// Round 1's layout (one lane per full Montgomery product, 12 of 16 lanes active, 763-instruction product + 4 row-scan
// steps + reduction ~ 1 000 instructions per round) measured 1.95 us per round; this one is ~600 instructions.
// The line functions of a couple of pairs are multiplied together up front (69 products l_1 l_2), so a doubling step of
// the Miller loop is two rounds (f^2, then (l_1 l_2) f) and an addition step one more.
// Frobenius is diagonal in this basis (c_k *= gamma_e^k), conjugation negates the odd coefficients.
// Replaces Bls12_377::{miller_loop, final_exponentiation, product_of_pairings}.
#pragma once
#include "tower.hpp"
#include "pround_gen.hpp"

namespace mi {

constexpr int PW_THREADS = 256;  // rows 0..11: the coefficients (three waves); row 12, in the fourth wave: the scale (below)

// A value also carries a SCALE s in Fq* (c[12]; c5[12] = 2q - s, its negative): the value stands for s times a unitary
// element.  Products multiply the scales (a thirteenth row of the round, on the fourth wave: no time), Frobenius and
// conjugation keep them.  The cyclotomic squaring round needs it -- the Granger-Scott formulas hold for unitary
// elements, and for c = s u they read (s u)^2 = 3 A'^2 - 2 s conj(A') ... with the scale in the linear terms -- because the
// verdict form of the final exponentiation (pw_final_exp) runs on scaled values.  Exact products carry s = 1.
struct alignas(16) PB {
    Fq c[13];   // coefficient of w^k, in [0, 2q); c[12]: the scale
    Fq c5[13];  // its twin 10q - 5 c_k (= -5 c_k mod q, in (0, 10q]): what a wrapped term of the next round multiplies by
};

struct alignas(16) PwShared {
    PB f, g, t, a, b, c, d, x, y, z;
    alignas(16) Fq dtab[25];  // fold constants D[m]
    alignas(16) Fq one;       // the constant line of a couple's missing second pair
    alignas(16) Fq px[MI_MAX_PAIRS], py[MI_MAX_PAIRS];
    // every line of the current couple's Miller loop evaluated at its P, [step][pair][w^1, w^3, w^7, w^9]
    // (the w^0 coefficient is py[pair]); 26 KB
    alignas(16) Fq lines2[MI_MILLER_LINES][2][4];
    // the couple's line products l_1 l_2, one dense Fq12 per Miller step (operand A of a round: no twin needed); 39 KB
    alignas(16) Fq lprod[MI_MILLER_LINES][12];
};
constexpr size_t PW_LDS_BYTES = sizeof(PwShared);  // dynamic LDS of every kernel that runs a pairing product
// Everything below addresses the workgroup's PwShared through LDS-qualified pointers (32-bit, ds_* instructions, no
// generic-pointer null checks), also across the out-of-line rounds.
#define PW_LDS __attribute__((address_space(3)))
typedef PW_LDS PwShared LShared;
typedef PW_LDS PB LPB;
typedef PW_LDS Fq LFq;

// ---- LDS <-> register copies, 16 bytes at a time (PwShared members are 16-byte aligned)
typedef uint32_t pw_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ Fq lds_load(const LFq *p) {
    const PW_LDS pw_u32x4 *q = (const PW_LDS pw_u32x4 *)p;
    const pw_u32x4 v0 = q[0], v1 = q[1], v2 = q[2];
    Fq r;
    r.l[0] = v0.x; r.l[1] = v0.y; r.l[2] = v0.z; r.l[3] = v0.w;
    r.l[4] = v1.x; r.l[5] = v1.y; r.l[6] = v1.z; r.l[7] = v1.w;
    r.l[8] = v2.x; r.l[9] = v2.y; r.l[10] = v2.z; r.l[11] = v2.w;
    return r;
}
__device__ __forceinline__ void lds_store(LFq *p, const Fq &v) {
    PW_LDS pw_u32x4 *q = (PW_LDS pw_u32x4 *)p;
    q[0] = pw_u32x4{v.l[0], v.l[1], v.l[2], v.l[3]};
    q[1] = pw_u32x4{v.l[4], v.l[5], v.l[6], v.l[7]};
    q[2] = pw_u32x4{v.l[8], v.l[9], v.l[10], v.l[11]};
}

struct Q12 {
    uint32_t l[12];
};
constexpr Q12 pw_make_q10() {
    Q12 r{};
    uint64_t c = 0;
    for (int j = 0; j < 12; j++) {
        const uint64_t v = (uint64_t)FqP::MOD[j] * 10u + c;
        r.l[j] = (uint32_t)v;
        c = v >> 32;
    }
    return r;
}
// twin of c in [0, 2q]: 10q - 5c in [0, 10q]
__device__ __forceinline__ Fq pw_twin(const Fq &v) {
    constexpr Q12 q10 = pw_make_q10();
    uint32_t t5[12], c = 0, bw = 0;
    Fq r;
#pragma unroll
    for (int j = 0; j < 12; j++) t5[j] = addc32((v.l[j] << 2) | (j ? (v.l[j - 1] >> 30) : 0u), v.l[j], c, &c);
#pragma unroll
    for (int j = 0; j < 12; j++) r.l[j] = subc32(q10.l[j], t5[j], bw, &bw);
    return r;
}

// the scale's partner: 2q - s in (0, 2q] for s in [0, 2q)
__device__ __forceinline__ Fq pw_neg2q(const Fq &v) {
    constexpr Q12 q10 = pw_make_q10();  // unused limbs fold away; 2q built from the modulus
    uint32_t bw = 0, c = 0;
    Fq r;
    (void)q10;
#pragma unroll
    for (int j = 0; j < 12; j++) {
        const uint32_t q2 = addc32(FqP::MOD[j], FqP::MOD[j], c, &c);
        r.l[j] = subc32(q2, v.l[j], bw, &bw);
    }
    return r;
}

// x += (x of the quad lane named by the permutation), N limbs with carry: DPP fused into the add instructions
#define PW_DPP_ADD16(ctrl)                                                                                              \
    asm volatile("s_nop 1\n\t"                                                                                          \
                 "v_add_co_u32_dpp %0, vcc, %0, %0 " ctrl " row_mask:0xf bank_mask:0xf\n\t"                             \
                 "v_addc_co_u32_dpp %1, vcc, %1, %1, vcc " ctrl " row_mask:0xf bank_mask:0xf\n\t"                       \
                 "v_addc_co_u32_dpp %2, vcc, %2, %2, vcc " ctrl " row_mask:0xf bank_mask:0xf\n\t"                       \
                 "v_addc_co_u32_dpp %3, vcc, %3, %3, vcc " ctrl " row_mask:0xf bank_mask:0xf\n\t"                       \
                 "v_addc_co_u32_dpp %4, vcc, %4, %4, vcc " ctrl " row_mask:0xf bank_mask:0xf\n\t"                       \
                 "v_addc_co_u32_dpp %5, vcc, %5, %5, vcc " ctrl " row_mask:0xf bank_mask:0xf\n\t"                       \
                 "v_addc_co_u32_dpp %6, vcc, %6, %6, vcc " ctrl " row_mask:0xf bank_mask:0xf\n\t"                       \
                 "v_addc_co_u32_dpp %7, vcc, %7, %7, vcc " ctrl " row_mask:0xf bank_mask:0xf\n\t"                       \
                 "v_addc_co_u32_dpp %8, vcc, %8, %8, vcc " ctrl " row_mask:0xf bank_mask:0xf\n\t"                       \
                 "v_addc_co_u32_dpp %9, vcc, %9, %9, vcc " ctrl " row_mask:0xf bank_mask:0xf\n\t"                       \
                 "v_addc_co_u32_dpp %10, vcc, %10, %10, vcc " ctrl " row_mask:0xf bank_mask:0xf\n\t"                    \
                 "v_addc_co_u32_dpp %11, vcc, %11, %11, vcc " ctrl " row_mask:0xf bank_mask:0xf\n\t"                    \
                 "v_addc_co_u32_dpp %12, vcc, %12, %12, vcc " ctrl " row_mask:0xf bank_mask:0xf\n\t"                    \
                 "v_addc_co_u32_dpp %13, vcc, %13, %13, vcc " ctrl " row_mask:0xf bank_mask:0xf\n\t"                    \
                 "v_addc_co_u32_dpp %14, vcc, %14, %14, vcc " ctrl " row_mask:0xf bank_mask:0xf\n\t"                    \
                 "v_addc_co_u32_dpp %15, vcc, %15, %15, vcc " ctrl " row_mask:0xf bank_mask:0xf"                        \
                 : "+v"(P[0]), "+v"(P[1]), "+v"(P[2]), "+v"(P[3]), "+v"(P[4]), "+v"(P[5]), "+v"(P[6]), "+v"(P[7]),      \
                   "+v"(P[8]), "+v"(P[9]), "+v"(P[10]), "+v"(P[11]), "+v"(P[12]), "+v"(P[13]), "+v"(P[14]), "+v"(P[15]) \
                 :                                                                                                      \
                 : "vcc")
#define PW_DPP_ADD13(ctrl)                                                                                              \
    asm volatile("s_nop 1\n\t"                                                                                          \
                 "v_add_co_u32_dpp %0, vcc, %0, %0 " ctrl " row_mask:0xf bank_mask:0xf\n\t"                             \
                 "v_addc_co_u32_dpp %1, vcc, %1, %1, vcc " ctrl " row_mask:0xf bank_mask:0xf\n\t"                       \
                 "v_addc_co_u32_dpp %2, vcc, %2, %2, vcc " ctrl " row_mask:0xf bank_mask:0xf\n\t"                       \
                 "v_addc_co_u32_dpp %3, vcc, %3, %3, vcc " ctrl " row_mask:0xf bank_mask:0xf\n\t"                       \
                 "v_addc_co_u32_dpp %4, vcc, %4, %4, vcc " ctrl " row_mask:0xf bank_mask:0xf\n\t"                       \
                 "v_addc_co_u32_dpp %5, vcc, %5, %5, vcc " ctrl " row_mask:0xf bank_mask:0xf\n\t"                       \
                 "v_addc_co_u32_dpp %6, vcc, %6, %6, vcc " ctrl " row_mask:0xf bank_mask:0xf\n\t"                       \
                 "v_addc_co_u32_dpp %7, vcc, %7, %7, vcc " ctrl " row_mask:0xf bank_mask:0xf\n\t"                       \
                 "v_addc_co_u32_dpp %8, vcc, %8, %8, vcc " ctrl " row_mask:0xf bank_mask:0xf\n\t"                       \
                 "v_addc_co_u32_dpp %9, vcc, %9, %9, vcc " ctrl " row_mask:0xf bank_mask:0xf\n\t"                       \
                 "v_addc_co_u32_dpp %10, vcc, %10, %10, vcc " ctrl " row_mask:0xf bank_mask:0xf\n\t"                    \
                 "v_addc_co_u32_dpp %11, vcc, %11, %11, vcc " ctrl " row_mask:0xf bank_mask:0xf\n\t"                    \
                 "v_addc_co_u32_dpp %12, vcc, %12, %12, vcc " ctrl " row_mask:0xf bank_mask:0xf"                        \
                 : "+v"(F[0]), "+v"(F[1]), "+v"(F[2]), "+v"(F[3]), "+v"(F[4]), "+v"(F[5]), "+v"(F[6]), "+v"(F[7]),      \
                   "+v"(F[8]), "+v"(F[9]), "+v"(F[10]), "+v"(F[11]), "+v"(F[12])                                        \
                 :                                                                                                      \
                 : "vcc")

// ---------------------------------------------------------------------------------------------------------------
// The round: out = A * B.  A: 12 coefficients (only plain values are read: a line product or a PB's c[]); B: a PB (its
// twin supplies the wrapped terms).  All PW_THREADS lanes must call.  NOALIAS: out is neither A nor B (the callers
// ping-pong between buffers), so no barrier separates the operand reads from the result stores: one barrier per round.
// PROF (tools/round_bench.hip only): wave 0 adds the shader-clock ticks of each stage of the round to prof[0..7]
// (loads, product, quad reduction, fold, row reduction, Barrett + twin, stores + barrier)
#define PW_STAMP(i)                                                   \
    if (PROF) {                                                       \
        const uint64_t now_ = __builtin_amdgcn_s_memtime();          \
        if (threadIdx.x == 0) prof[i] += now_ - stamp_;               \
        stamp_ = now_;                                                \
    }
// everything after the lane's partial product P: quad sum, distributed Montgomery fold, row sum, Barrett, twin, store
template <bool NOALIAS, bool PROF>
__device__ __forceinline__ void pb_round_finish(uint32_t (&P)[16], LPB *out, const Fq &d0, const Fq &d1, const Fq &d2, const Fq &d3,
                                                uint64_t *prof, uint64_t &stamp_) {
    constexpr uint32_t MU0 = 0x54234331u, MU1 = 0x98u;  // floor(2^416 / q)
    const int t = threadIdx.x, k = t >> 4, L = t & 15, g = L & 3;
    const bool scale_wave = __builtin_amdgcn_readfirstlane(t >> 6) == 3;  // scalar: the compiler branches, no exec masks
    // the quad's sum Q_j (limb offset 3j), in all four lanes
    PW_DPP_ADD16("quad_perm:[1,0,3,2]");
    PW_DPP_ADD16("quad_perm:[2,3,0,1]");
    PW_STAMP(2)
    // distributed Montgomery reduction: this lane's four words of Q_j (limbs g, g+4, g+8, g+12) times D[position]
    // (lane-dependent choice by masks, branch-free: the compiler turns the nested conditional into exec-mask blocks)
    const uint32_t m0 = g == 0 ? ~0u : 0u, m1 = g == 1 ? ~0u : 0u, m2 = g == 2 ? ~0u : 0u, m3 = g == 3 ? ~0u : 0u;
    uint32_t w[4];
#pragma unroll
    for (int tt = 0; tt < 4; tt++)
        w[tt] = (P[4 * tt] & m0) | (P[4 * tt + 1] & m1) | (P[4 * tt + 2] & m2) | (P[4 * tt + 3] & m3);
    uint32_t F[13];
    pr_fold(F, w[0], w[1], w[2], w[3], d0, d1, d2, d3);
    PW_STAMP(3)
    // the row's total, in every lane (rotations within the 16-lane row)
    PW_DPP_ADD13("row_ror:8");
    PW_DPP_ADD13("row_ror:4");
    PW_DPP_ADD13("row_ror:2");
    PW_DPP_ADD13("row_ror:1");
    PW_STAMP(4)
    // Barrett: F < 2^416, quotient estimate from the bits above 2^352, c = F - qhat q in [0, 2q)
    uint64_t tq = ((uint64_t)F[11] * MU0) >> 32;
    tq += (uint64_t)F[12] * MU0;
    tq += (uint64_t)F[11] * MU1;
    const uint64_t qh = (uint64_t)F[12] * MU1 + (tq >> 32);
    uint32_t r[12];
    pr_qhat_q(r, (uint32_t)qh, (uint32_t)(qh >> 32));
    Fq c;
    uint32_t bw = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) c.l[i] = subc32(F[i], r[i], bw, &bw);
    Fq c5;
    if (!scale_wave) c5 = pw_twin(c);
    else c5 = pw_neg2q(c);  // the scale's partner is its negative
    PW_STAMP(5)
    if (!NOALIAS) __syncthreads();
    if (k <= 12) {  // rows 13..15 of the scale's wave are idle
        if (L == 0) lds_store(&out->c[k], c);
        if (L == 1) lds_store(&out->c5[k], c5);
    }
    __syncthreads();
    PW_STAMP(6)
}
// As: the scale of operand A (&A->c[12]) or nullptr (A is a line product / merged factor: scale one)
template <bool NOALIAS, bool PROF = false>
__device__ __forceinline__ void pb_round_impl(LShared &S, LPB *out, const LFq *A, const LPB *B, const Fq &d0, const Fq &d1,
                                              const Fq &d2, const Fq &d3, uint64_t *prof = nullptr, const LFq *As = nullptr) {
    const int t = threadIdx.x, k = t >> 4, L = t & 15, j = L >> 2, g = L & 3;
    uint64_t stamp_ = PROF ? __builtin_amdgcn_s_memtime() : 0;
    Fq a[3];
    uint32_t b[3][3];
    if (__builtin_amdgcn_readfirstlane(t >> 6) != 3) {  // scalar
#pragma unroll
        for (int tt = 0; tt < 3; tt++) {
            const int s = 3 * g + tt;
            int i = k - s;
            const bool wrap = i < 0;
            if (wrap) i += 12;
            a[tt] = lds_load(&A[i]);
            const PW_LDS uint32_t *bl = (wrap ? B->c5[s].l : B->c[s].l) + 3 * j;
            b[tt][0] = bl[0];
            b[tt][1] = bl[1];
            b[tt][2] = bl[2];
        }
    } else {  // the scale's row (and the three idle rows of its wave, which store nothing): s_A s_B on the lanes g = 0
        const PW_LDS uint32_t *bl = B->c[12].l + 3 * j;
        a[0] = As ? lds_load(As) : Fq::one();
        a[1] = a[2] = Fq::zero();
#pragma unroll
        for (int tt = 0; tt < 3; tt++) b[tt][0] = b[tt][1] = b[tt][2] = 0;
        if (g == 0) {
            b[0][0] = bl[0];
            b[0][1] = bl[1];
            b[0][2] = bl[2];
        }
    }
    if (PROF) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    PW_STAMP(0)
    uint32_t P[16];
    pr_qprod(P, a[0], a[1], a[2], b[0], b[1], b[2]);
    PW_STAMP(1)
    pb_round_finish<NOALIAS, PROF>(P, out, d0, d1, d2, d3, prof, stamp_);
}

// ---------------------------------------------------------------------------------------------------------------
// The cyclotomic squaring round: out = B^2 for B = s u, u in the cyclotomic subgroup (u^(q^4 - q^2 + 1) = 1: everything
// after the easy part of the final exponentiation), s = B's scale.  Granger-Scott, in this basis: with W = w^3 (W^4 = -5),
//   B = A + B w + C w^2,   A = sum_i c_(3i) W^i,  B = sum_i c_(3i+1) W^i,  C = sum_i c_(3i+2) W^i,   X~ = sum (-1)^i x_i W^i:
//   B^2 = (3 A^2 - 2 s A~) + (3 W C^2 + 2 s B~) w + (3 B^2 - 2 s C~) w^2                 (tools/model_pround.py checks it)
// A square in Fq[W]/(W^4 + 5) is ten products, so an output coefficient is two or three products of coefficients (times
// 3 or 6, times -5 through the twin where W^4 wraps) and one product with the scale: one term per lane g of the quad
// instead of the general round's three -- 12 limb products to scale the operand and 36 for the product against 108.
// Table entry: a | b << 4 | twin << 8 | multiplier << 9 (0: no term); b = 12: the scale (twin: its negative).
#define PW_CS(a, b, tw, m) (uint16_t)((a) | ((b) << 4) | ((tw) << 8) | ((m) << 9))
__device__ const uint16_t PW_CSQR_TABLE[13][4] = {
    // A-rows k = 3i: 3 [A^2]_i - 2 s (-1)^i c_k; x_m = c_(3m)
    /* 0 */ {PW_CS(0, 0, 0, 3), PW_CS(6, 6, 1, 3), PW_CS(3, 9, 1, 6), PW_CS(0, 12, 1, 2)},
    // C-rows k = 3i + 1: 3 [W C^2]_i + 2 s (-1)^i c_k; x_m = c_(3m+2); [W C^2]_0 = -5 [C^2]_3, [W C^2]_i = [C^2]_(i-1)
    /* 1 */ {PW_CS(2, 11, 1, 6), PW_CS(5, 8, 1, 6), 0, PW_CS(1, 12, 0, 2)},
    // B-rows k = 3i + 2: 3 [B^2]_i - 2 s (-1)^i c_k; x_m = c_(3m+1)
    /* 2 */ {PW_CS(1, 1, 0, 3), PW_CS(7, 7, 1, 3), PW_CS(4, 10, 1, 6), PW_CS(2, 12, 1, 2)},
    /* 3 */ {PW_CS(0, 3, 0, 6), PW_CS(6, 9, 1, 6), 0, PW_CS(3, 12, 0, 2)},
    /* 4 */ {PW_CS(2, 2, 0, 3), PW_CS(8, 8, 1, 3), PW_CS(5, 11, 1, 6), PW_CS(4, 12, 1, 2)},
    /* 5 */ {PW_CS(1, 4, 0, 6), PW_CS(7, 10, 1, 6), 0, PW_CS(5, 12, 0, 2)},
    /* 6 */ {PW_CS(0, 6, 0, 6), PW_CS(3, 3, 0, 3), PW_CS(9, 9, 1, 3), PW_CS(6, 12, 1, 2)},
    /* 7 */ {PW_CS(2, 5, 0, 6), PW_CS(8, 11, 1, 6), 0, PW_CS(7, 12, 0, 2)},
    /* 8 */ {PW_CS(1, 7, 0, 6), PW_CS(4, 4, 0, 3), PW_CS(10, 10, 1, 3), PW_CS(8, 12, 1, 2)},
    /* 9 */ {PW_CS(0, 9, 0, 6), PW_CS(3, 6, 0, 6), 0, PW_CS(9, 12, 0, 2)},
    /* 10 */ {PW_CS(2, 8, 0, 6), PW_CS(5, 5, 0, 3), PW_CS(11, 11, 1, 3), PW_CS(10, 12, 1, 2)},
    /* 11 */ {PW_CS(1, 10, 0, 6), PW_CS(4, 7, 0, 6), 0, PW_CS(11, 12, 0, 2)},
    /* 12: the scale */ {PW_CS(12, 12, 0, 1), 0, 0, 0},
};
#undef PW_CS
// this lane's term (it depends on the lane only: loops keep it in a register)
__device__ __forceinline__ uint32_t pw_csqr_entry() {
    const int k = threadIdx.x >> 4, g = threadIdx.x & 3;
    return k <= 12 ? PW_CSQR_TABLE[k][g] : 0u;  // rows 13..15: idle (zero product, nothing stored)
}
template <bool PROF = false>
__device__ __forceinline__ void pb_csqr_impl(LShared &S, LPB *out, const LPB *B, uint32_t e, const Fq &d0, const Fq &d1, const Fq &d2,
                                             const Fq &d3, uint64_t *prof = nullptr) {
    const int L = threadIdx.x & 15, j = L >> 2;
    uint64_t stamp_ = PROF ? __builtin_amdgcn_s_memtime() : 0;
    const uint32_t mult = e >> 9;
    const Fq av = lds_load(&B->c[e & 15u]);
    const PW_LDS uint32_t *bl = ((e & 0x100u) ? B->c5[(e >> 4) & 15u].l : B->c[(e >> 4) & 15u].l) + 3 * j;
    const uint32_t b[3] = {bl[0], bl[1], bl[2]};
    // the operand times its multiplier: < 6 * 2q < 2^381, twelve limbs
    Fq a;
    uint64_t carry = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        carry += (uint64_t)av.l[i] * mult;
        a.l[i] = (uint32_t)carry;
        carry >>= 32;
    }
    if (PROF) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    PW_STAMP(0)
    uint32_t P[16];
    pr_qprod1(P, a, b);
    PW_STAMP(1)
    pb_round_finish<true, PROF>(P, out, d0, d1, d2, d3, prof, stamp_);
}
#undef PW_STAMP
#undef PW_DPP_ADD16
#undef PW_DPP_ADD13
// the fold constants of this lane (they depend on the lane only: loops keep them in registers across rounds)
struct PwFold {
    Fq d0, d1, d2, d3;
};
__device__ __forceinline__ PwFold pw_fold_consts(LShared &S) {
    const int L = threadIdx.x & 15;
    const LFq *dt = &S.dtab[3 * (L >> 2) + (L & 3)];
    return PwFold{lds_load(dt), lds_load(dt + 4), lds_load(dt + 8), lds_load(dt + 12)};
}
template <bool NOALIAS>
__device__ __attribute__((noinline)) void pb_round(LShared &S, LPB *out, const LFq *A, const LPB *B, const LFq *As) {
    const PwFold fc = pw_fold_consts(S);
    pb_round_impl<NOALIAS>(S, out, A, B, fc.d0, fc.d1, fc.d2, fc.d3, nullptr, As);
}

__device__ __forceinline__ void pb_mul(LShared &S, LPB *out, const LPB *A, const LPB *B) { pb_round<false>(S, out, A->c, B, &A->c[12]); }
// out must not be A or B
__device__ __forceinline__ void pb_mul_na(LShared &S, LPB *out, const LPB *A, const LPB *B) { pb_round<true>(S, out, A->c, B, &A->c[12]); }

// store coefficient t of a value computed by lanes t < 12, with its twin
__device__ __forceinline__ void pb_put(LPB *out, int t, const Fq &v) {
    lds_store(&out->c[t], v);
    lds_store(&out->c5[t], pw_twin(v));
}
// the value's scale (one lane), with its negative
__device__ __forceinline__ void pb_put_scale(LPB *out, const Fq &s) {
    lds_store(&out->c[12], s);
    lds_store(&out->c5[12], pw_neg2q(s));
}
// out = frobenius^e(A): c_k *= gamma_e^k
__device__ __forceinline__ void pb_frob(LShared &S, LPB *out, const LPB *A, const Fq *gamma /* global, 12 */) {
    const int t = threadIdx.x;
    Fq v;
    if (t < 12) v = mul(lds_load(&A->c[t]), gamma[t]);
    if (t == 12) v = lds_load(&A->c[12]);  // Fq is fixed
    __syncthreads();
    if (t < 12) pb_put(out, t, v);
    if (t == 12) pb_put_scale(out, v);
    __syncthreads();
}
// out = conj(A): negate the odd coefficients
__device__ __forceinline__ void pb_conj(LShared &S, LPB *out, const LPB *A) {
    const int t = threadIdx.x;
    Fq v;
    if (t < 12) {
        v = lds_load(&A->c[t]);
        reduce_once(v);  // rounds leave values in [0, 2q)
        if (t & 1) v = neg(v);
    }
    if (t == 12) v = lds_load(&A->c[12]);
    __syncthreads();
    if (t < 12) pb_put(out, t, v);
    if (t == 12) pb_put_scale(out, v);
    __syncthreads();
}
__device__ __forceinline__ void pb_copy(LShared &S, LPB *out, const LPB *A) {
    const int t = threadIdx.x;
    Fq v;
    if (t < 26) v = lds_load(t < 13 ? &A->c[t] : &A->c5[t - 13]);
    __syncthreads();
    if (t < 26) lds_store(t < 13 ? &out->c[t] : &out->c5[t - 13], v);
    __syncthreads();
}
__device__ __forceinline__ void pb_set_one(LPB *out) {  // followed by the caller's barrier
    const int t = threadIdx.x;
    if (t < 12) pb_put(out, t, t == 0 ? Fq::one() : Fq::zero());
    if (t == 12) pb_put_scale(out, Fq::one());
}
// out = A^x (x = BLS seed), A = s u with u in the cyclotomic subgroup: 63 cyclotomic squarings, 6 products
__device__ __attribute__((noinline)) void pb_exp_by_x(LShared &S, LPB *out, const LPB *A) {
    // ping-pong between S.y and S.z so that no round writes an operand (one barrier per round); A is neither
    LPB *cur = &S.y, *alt = &S.z;
    pb_copy(S, cur, A);
    const PwFold fc = pw_fold_consts(S);  // in registers across the 69 rounds (a round that reloads them: 1.74 vs 1.44 us)
    const uint32_t entry = pw_csqr_entry();
    for (int i = 62; i >= 0; i--) {
        pb_csqr_impl(S, alt, cur, entry, fc.d0, fc.d1, fc.d2, fc.d3);
        LPB *sw = cur; cur = alt; alt = sw;
        if ((MI_BLS_X >> i) & 1) {
            pb_round_impl<true>(S, alt, cur->c, A, fc.d0, fc.d1, fc.d2, fc.d3, nullptr, &cur->c[12]);
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
// unscaled: out = Abar = N A^-1 with scale N -- no inversion in Fq (pw_final_exp's verdict form).
__device__ __attribute__((noinline)) void pb_inv(LShared &S, LPB *out, const LPB *A, const PairingConsts *pc, bool unscaled = false) {
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
    if (unscaled) {  // uniform: out = Abar = N A^-1, and says so: scale N (A's own scale is one: a Miller value)
        Fq v;
        if (t < 12) v = lds_load(&S.x.c[t]);
        if (t == 12) v = lds_load(&S.y.c[0]);
        __syncthreads();
        if (t < 12) pb_put(out, t, v);
        if (t == 12) pb_put_scale(out, v);
        __syncthreads();
        return;
    }
    if (t == 0) {
        Fq nrm = lds_load(&S.y.c[0]);
        reduce_once(nrm);  // rounds leave values in [0, 2q)
        lds_store(&S.y.c[1], invq(nrm));  // 1/N parked in y.c[1]
    }
    __syncthreads();
    Fq v;
    if (t < 12) v = mul(lds_load(&S.x.c[t]), lds_load(&S.y.c[1]));
    __syncthreads();
    if (t < 12) pb_put(out, t, v);
    if (t == 12) pb_put_scale(out, Fq::one());
    __syncthreads();
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

// The lines of one COUPLE of live pairs (p0, p0 + 1) for the line indices [lo, lo + count): every line's coefficients at
// its P (four per line and pair, a lane each), then the line products l_1 l_2 (twelve coefficients per line: up to four
// Montgomery products each, a lane per coefficient) into S.lprod[0 .. count).  A single pair (np = 1) runs as a couple
// whose second line is the constant 1.
// at: where the products go, S.lprod[at .. at + count) (and the line coefficients S.lines2[at ..]).
__device__ __forceinline__ void pw_line_products(LShared &S, const G2Prepared *const *qs, int p0, int np, int lo, int count, int at = 0) {
    const int t = threadIdx.x;
    for (int item = t; item < count * 8; item += PW_THREADS) {
        const int idx = item >> 3, p = (item >> 2) & 1, role = item & 3;
        Fq v = Fq::zero();
        if (p < np) {
            const Line *ln = &qs[p0 + p]->lines[lo + idx];
            if (role == 0) v = mul(ln->c1.c0, lds_load(&S.px[p0 + p]));       // w^1
            else if (role == 1) v = ln->c3.c0;                                 // w^3
            else if (role == 2) v = mul(ln->c1.c1, lds_load(&S.px[p0 + p]));  // w^7
            else v = ln->c3.c1;                                                // w^9
        }
        lds_store(&S.lines2[at + idx][p][role], v);
    }
    if (t == 12) lds_store(&S.one, Fq::one());
    __syncthreads();
    const LFq *y1 = &S.py[p0], *y2 = np == 2 ? &S.py[p0 + 1] : &S.one;
    for (int item = t; item < count * 12; item += PW_THREADS) {
        const int idx = item / 12, k = item - 12 * idx;
        Fq acc = Fq::zero();
#pragma unroll 1
        for (int e = 0; e < 4; e++) {
            const uint32_t code = PW_LL_TABLE[k][e];
            if (!(code & 0x80u)) break;
            const uint32_t i = code & 7u, j = (code >> 3) & 7u;
            const Fq p = mul(lds_load(i == 0 ? y1 : &S.lines2[at + idx][0][i - 1]), lds_load(j == 0 ? y2 : &S.lines2[at + idx][1][j - 1]));
            acc = (code & 0x40u) ? sub(acc, mul5(p)) : add(acc, p);  // w^12 = -5
        }
        lds_store(&S.lprod[at + idx][k], acc);
    }
    __syncthreads();
}

// Miller loop of one couple into S.f: all line products up front, then two rounds per doubling step -- f^2, then
// (l_1 l_2) f -- and one more per addition step.  Products over more pairs multiply the couples' values.
__device__ __attribute__((noinline)) void pw_miller2(LShared &S, const G2Prepared *const *qs, int p0, int np) {
    pw_line_products(S, qs, p0, np, 0, MI_MILLER_LINES);
    pb_set_one(&S.f);
    __syncthreads();
    int idx = 0;
    LPB *cur = &S.f, *alt = &S.g;  // ping-pong: no round writes one of its operands
    const PwFold fc = pw_fold_consts(S);
    for (int i = 62; i >= 0; i--) {
        pb_round_impl<true>(S, alt, cur->c, cur, fc.d0, fc.d1, fc.d2, fc.d3);           // alt = cur^2
        pb_round_impl<true>(S, cur, S.lprod[idx], alt, fc.d0, fc.d1, fc.d2, fc.d3);     // cur = (l l') alt
        idx++;
        if ((MI_BLS_X >> i) & 1) {
            pb_round_impl<true>(S, alt, S.lprod[idx], cur, fc.d0, fc.d1, fc.d2, fc.d3);  // the addition step's lines
            idx++;
            LPB *sw = cur; cur = alt; alt = sw;
        }
    }
    if (cur != &S.f) pb_copy(S, &S.f, cur);
}

// ---- the Miller loop in two launches (pairing.hip): f_(i+1) = f_i^2 X_i with X_i the step's line product (times the
// addition step's where x has a bit), so three steps are f^8 (X_a^4 X_b^2 X_c).  The merged factors M_g = X_a^4 X_b^2 X_c
// do not depend on f: a first launch computes the 21 factors on 21 workgroups side by side (over all the couples of
// pairs at once: pw_miller_group), the product kernel then runs 3 squarings + one multiplication per group: 80 rounds
// instead of 132 for one couple and of 264 for two.
#define PW_MILLER_GROUPS 21
static_assert(3 * PW_MILLER_GROUPS == 63, "63 doubling steps in triples");
// workgroup g: M_g of ALL the live pairs (one or two couples) -> out[12] (plain coefficients in [0, 2q)).  With two
// couples every step's line products are multiplied together first, X_i = (l_1 l_2)(l_3 l_4) -- at most five more
// rounds here, spread over the 21 workgroups -- so that the product kernel multiplies ONE factor per group.
__device__ __forceinline__ void pw_miller_group(LShared &S, const G2Prepared *const *qs, int live_n, int g, Fq *out) {
    // line indices of the group's steps: step s (bit 62 - s of x) has its doubling line at s + (addition steps before s)
    int lo = 0;
    for (int s = 0; s < 3 * g; s++) lo += 1 + (int)((MI_BLS_X >> (62 - s)) & 1);
    int count = 0, nadd[3];
    for (int k = 0; k < 3; k++) {
        nadd[k] = (int)((MI_BLS_X >> (62 - (3 * g + k))) & 1);
        count += 1 + nadd[k];
    }
    constexpr int SECOND = 8;  // count <= 6: the second couple's products sit at S.lprod[8 ..]
    pw_line_products(S, qs, 0, live_n >= 2 ? 2 : 1, lo, count, 0);
    if (live_n > 2) pw_line_products(S, qs, 2, live_n - 2 >= 2 ? 2 : 1, lo, count, SECOND);
    const int t = threadIdx.x;
    const PwFold fc = pw_fold_consts(S);
    // X[i]: the step's factor as operand A of a round (plain coefficients)
    LPB *const merged[6] = {&S.t, &S.a, &S.b, &S.c, &S.d, &S.z};
    const LFq *X[6];
    for (int i = 0; i < count; i++) X[i] = S.lprod[i];
    if (live_n > 2) {
        for (int i = 0; i < count; i++) {
            if (t < 12) pb_put(&S.x, t, lds_load(&S.lprod[SECOND + i][t]));
            if (t == 12) pb_put_scale(&S.x, Fq::one());
            __syncthreads();
            pb_round_impl<true>(S, merged[i], S.lprod[i], &S.x, fc.d0, fc.d1, fc.d2, fc.d3);
            X[i] = merged[i]->c;
        }
    }
    // acc = X_a; then twice: acc = acc^2 X_next.  X_k = X[dbl] (* X[add]); products keep the running value as operand B
    // (a PB with its twin) and the factor as operand A (plain)
    LPB *cur = &S.f, *alt = &S.g;
    int idx = 0;
    if (t < 12) pb_put(cur, t, lds_load(&X[0][t]));
    if (t == 12) pb_put_scale(cur, Fq::one());
    __syncthreads();
    idx = 1;
    if (nadd[0]) {
        pb_round_impl<true>(S, alt, X[idx], cur, fc.d0, fc.d1, fc.d2, fc.d3);
        idx++;
        LPB *sw = cur; cur = alt; alt = sw;
    }
    for (int k = 1; k < 3; k++) {
        pb_round_impl<true>(S, alt, cur->c, cur, fc.d0, fc.d1, fc.d2, fc.d3);  // alt = cur^2
        pb_round_impl<true>(S, cur, X[idx], alt, fc.d0, fc.d1, fc.d2, fc.d3);  // cur = X_k alt
        idx++;
        if (nadd[k]) {
            pb_round_impl<true>(S, alt, X[idx], cur, fc.d0, fc.d1, fc.d2, fc.d3);
            idx++;
            LPB *sw = cur; cur = alt; alt = sw;
        }
    }
    if (t < 12) out[t] = lds_load(&cur->c[t]);
}
// S.f = the Miller value of all the pairs from their merged factors m[group][12] (global memory)
__device__ __attribute__((noinline)) void pw_miller_merged(LShared &S, const Fq *m) {
    const int t = threadIdx.x;
    for (int item = t; item < PW_MILLER_GROUPS * 12; item += PW_THREADS) lds_store(&S.lprod[item / 12][item % 12], m[item]);
    __syncthreads();
    // f = M_0, then per group three squarings and the group's factor
    LPB *cur = &S.f, *alt = &S.g;
    if (t < 12) pb_put(cur, t, lds_load(&S.lprod[0][t]));
    if (t == 12) pb_put_scale(cur, Fq::one());
    __syncthreads();
    const PwFold fc = pw_fold_consts(S);
    for (int g = 1; g < PW_MILLER_GROUPS; g++) {
        for (int k = 0; k < 3; k++) {
            pb_round_impl<true>(S, alt, cur->c, cur, fc.d0, fc.d1, fc.d2, fc.d3);
            LPB *sw = cur; cur = alt; alt = sw;
        }
        pb_round_impl<true>(S, alt, S.lprod[g], cur, fc.d0, fc.d1, fc.d2, fc.d3);
        LPB *sw = cur; cur = alt; alt = sw;
    }
    if (cur != &S.f) pb_copy(S, &S.f, cur);
}

// S.f <- S.f ^ (3 (q^12 - 1) / r)
// verdict_only: S.f <- s * that value for some s in Fq* -- the same chain WITHOUT the inversion in Fq.  The easy part
// takes f^-1 as Abar / N with N = Norm(f) in Fq; leaving 1/N out scales the value by N, an element of Fq*, and every
// later step keeps "unitary value times an Fq* scalar" in that shape: Frobenius and conjugation fix Fq, products and
// powers multiply the scalars, and where the chain uses conj(t) for t^-1 the scaled conj(c t) = c t^-1 is c^2 times the
// inverse.  With z = f^(3 (q^12 - 1) / r) in mu_r and gcd(r, q - 1) = 1:  s z lies in Fq  <=>  z = 1, so the caller's
// test "is one" becomes "coefficients 1..11 vanish and coefficient 0 does not" (pw_pairing_product writes one / not one).
__device__ __attribute__((noinline)) void pw_final_exp(LShared &S, const PairingConsts *pc, bool verdict_only) {
    // easy part
    pb_inv(S, &S.a, &S.f, pc, verdict_only);
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

// the pairs whose P is not infinity and whose Q is not infinity, packed to the front: S.px / S.py and live_q; returns
// their number in every lane
__device__ __forceinline__ int pw_live_pairs(LShared &S, const G1Affine *ps, const G2Prepared *const *qs, int n,
                                              const G2Prepared **live_q, int *live_n, const PairingConsts *pc) {
    const int t = threadIdx.x;
    if (t == 0) {
        int m = 0;
        for (int p = 0; p < n; p++) {
            if (qs[p]->infinity || ps[p].is_inf()) continue;
            live_q[m] = qs[p];
            lds_store(&S.px[m], ps[p].x);
            lds_store(&S.py[m], ps[p].y);
            m++;
        }
        *live_n = m;
    }
    if (t >= 64 && t < 64 + 25) lds_store(&S.dtab[t - 64], pc->dtab[t - 64]);
    __syncthreads();
    return *live_n;
}
// full product over the live pairs; result (tower order) to out.  merged: the couples' merged Miller factors from
// pw_miller_group (the two-launch form), or null: everything in this workgroup.
// verdict_only: *out is one exactly when the product is one, and otherwise SOME value that is not one (pw_final_exp).
__device__ __forceinline__ void pw_pairing_product(LShared &S, const G1Affine *ps, const G2Prepared *const *qs, int n,
                                                   const PairingConsts *pc, Fq12 *out, const Fq *merged = nullptr,
                                                   bool verdict_only = false) {
    __shared__ const G2Prepared *live_q[MI_MAX_PAIRS];
    __shared__ int live_n_s;
    const int t = threadIdx.x;
    const int live_n = pw_live_pairs(S, ps, qs, n, live_q, &live_n_s, pc);
    if (live_n == 0) {
        pb_set_one(&S.f);
        __syncthreads();
    } else if (merged) {
        pw_miller_merged(S, merged);
    } else {
        for (int c = 0; c < live_n; c += 2) {
            if (c > 0) pb_copy(S, &S.b, &S.f);  // the couples before this one
            pw_miller2(S, live_q, c, live_n - c >= 2 ? 2 : 1);
            if (c > 0) pb_mul(S, &S.f, &S.f, &S.b);
        }
    }
    pw_final_exp(S, pc, verdict_only);
    Fq res = Fq::zero();
    if (t < 12) {
        res = lds_load(&S.f.c[t]);
        reduce_once(res);  // rounds leave values in [0, 2q)
    }
    if (verdict_only && t < 64) {  // wave 0 holds all twelve coefficients: s in Fq* (times one) becomes one
        const uint64_t nz = __ballot(t < 12 && !res.is_zero());
        if ((nz & 0xfffull) == 1ull) res = t == 0 ? Fq::one() : Fq::zero();
    }
    if (t < 12) {
        // tower index (i, j, b) holds the coefficient of w^(i + 2j + 6b)
        const int b = t / 6, rem = t - 6 * b, j = rem >> 1, i = rem & 1;
        Fq *o = &out->c0.c0.c0;
        o[(i * 3 + j) * 2 + b] = res;
    }
    __syncthreads();
}

}  // namespace mi
