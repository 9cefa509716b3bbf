// Host build (plain g++, no HIP) of the product's shared host/device arithmetic headers, so the
// CPU test tier can pin snarkos_amd/csrc/{field,curve,tower}.hpp against the oracle without a GPU.
// Test infrastructure: compiled by tests/test_host_arith.py into tests/hostlib/libhostcheck.so.
#include <vector>
#include <string.h>
#include "../../snarkos_amd/csrc/tower.hpp"
#include "../../snarkos_amd/csrc/glv.hpp"
#include "../../snarkos_amd/csrc/codec.hpp"
#include "../../snarkos_amd/csrc/hostfold.hpp"

using namespace mi;

static Fq fq_from64(const uint64_t *l) {
    Fq r;
    for (int i = 0; i < 6; i++) { r.l[2 * i] = (uint32_t)l[i]; r.l[2 * i + 1] = (uint32_t)(l[i] >> 32); }
    return r;
}

// the fallback's scalar multiplication (glv.hpp) on the host: split, signed windows, Jacobian chain; the table is built
// one entry at a time here (the device shares one inversion over it)
template <int W>
static G1Affine glv_mul_host(const G1Affine &p, const uint32_t *k8) {
    struct Tab {
        G1Affine t[8];
        G1Affine get(uint32_t e, bool negate) const { return negate ? neg(t[e]) : t[e]; }
    };
    uint32_t s[8], s1[4], s2[4];
    for (int i = 0; i < 8; i++) s[i] = k8[i];
    glv_split(s, s1, s2);
    G1Xyzz sum = G1Xyzz::inf();
    for (int half = 0; half < 2; half++) {
        G1Affine q = p;
        if (half) q.x = mul(q.x, glv_beta());
        if (q.is_inf()) continue;
        Tab tab;
        G1Xyzz m = G1Xyzz::from_affine(q);
        for (int e = 0; e < (1 << (W - 1)); e++) {
            tab.t[e] = to_affine(m);
            m = add_mixed(m, q);
        }
        sum = add(sum, glv_chain<W>(tab, half ? s2 : s1));
    }
    return to_affine(sum);
}
extern "C" {
void hc_fq_mul(const Fq *a, const Fq *b, Fq *o) { *o = mul(*a, *b); }
void hc_fq_mul32(const Fq *a, const Fq *b, Fq *o) { *o = mul32(*a, *b); }
void hc_fr_mul32(const Fr *a, const Fr *b, Fr *o) { *o = mul32(*a, *b); }
void hc_fq_add(const Fq *a, const Fq *b, Fq *o) { *o = add(*a, *b); }
void hc_fq_sub(const Fq *a, const Fq *b, Fq *o) { *o = sub(*a, *b); }
void hc_fq_inv(const Fq *a, Fq *o) { *o = inv(*a); }
void hc_fq_inv_egcd(const Fq *a, Fq *o) { *o = inv_egcd(*a); }
void hc_fr_inv_egcd(const Fr *a, Fr *o) { *o = inv_egcd(*a); }
void hc_fq_inv_bingcd(const Fq *a, Fq *o) { *o = invq(*a); }  // what the library uses (field.hpp: inv_bingcd)
void hc_fr_inv_bingcd(const Fr *a, Fr *o) { *o = invr(*a); }
void hc_fr_mul(const Fr *a, const Fr *b, Fr *o) { *o = mul(*a, *b); }
void hc_fr_inv(const Fr *a, Fr *o) { *o = inv(*a); }
void hc_fr_to_mont(const Fr *a, Fr *o) { *o = to_mont(*a); }
void hc_fr_from_mont(const Fr *a, Fr *o) { *o = from_mont(*a); }
void hc_g1_add_mixed(const G1Affine *a, const G1Affine *b, G1Affine *o) { *o = to_affine(add_mixed(G1Xyzz::from_affine(*a), *b)); }
void hc_g1_add(const G1Affine *a, const G1Affine *b, G1Affine *o) { *o = to_affine(add(G1Xyzz::from_affine(*a), G1Xyzz::from_affine(*b))); }
void hc_g1_dbl(const G1Affine *a, G1Affine *o) { *o = to_affine(dbl(G1Xyzz::from_affine(*a))); }
void hc_g1_mul(const G1Affine *p, const uint32_t *k, int bits, G1Affine *o) { *o = to_affine(scalar_mul(*p, k, bits)); }
void hc_fq12_mul(const Fq12 *a, const Fq12 *b, Fq12 *o) { *o = mul(*a, *b); }
void hc_fq12_sqr(const Fq12 *a, Fq12 *o) { *o = sqr(*a); }
void hc_fq12_inv(const Fq12 *a, Fq12 *o) { *o = inv(*a); }
void hc_pairing_product(const G1Affine *ps, const G2Affine *qs, int n, Fq12 *out) {
    static G2Prepared prep[MI_MAX_PAIRS];
    const G2Prepared *qp[MI_MAX_PAIRS];
    for (int i = 0; i < n; i++) { g2_prepare_host(prep[i], qs[i]); qp[i] = &prep[i]; }
    PairingConsts pc;
    for (int e = 0; e < 3; e++)
        for (int k = 0; k < 12; k++) pc.frob[e][k] = fq_from64(MI_FROB[e][k]);
    *out = final_exponentiation(miller_product(ps, qp, n), pc);
}
void hc_glv_split(const uint32_t *s8, uint32_t *s1, uint32_t *s2) {
    uint32_t s[8], a[4], b[4];
    for (int i = 0; i < 8; i++) s[i] = s8[i];
    glv_split(s, a, b);
    for (int i = 0; i < 4; i++) { s1[i] = a[i]; s2[i] = b[i]; }
}
void hc_glv_digits(int w, const uint32_t *v4, int *out) {  // out[NW]
    uint32_t v[4] = {v4[0], v4[1], v4[2], v4[3]};
    if (w == 3) { GlvDigits<3> d; d.recode(v); for (int i = 0; i < GlvDigits<3>::NW; i++) out[i] = d.digit(i); }
    else { GlvDigits<4> d; d.recode(v); for (int i = 0; i < GlvDigits<4>::NW; i++) out[i] = d.digit(i); }
}
void hc_glv_mul(int w, const G1Affine *p, const uint32_t *k8, G1Affine *o) { *o = w == 3 ? glv_mul_host<3>(*p, k8) : glv_mul_host<4>(*p, k8); }
void hc_fq_sqr(const Fq *a, Fq *o) { *o = sqr(*a); }
// csrc/hostfold.hpp: what msm_finish_all folds with (64-bit limbs, the BMI2/ADX product when the CPU has it)
void hc_hf_mul(const Fq *a, const Fq *b, Fq *o) { *o = hostfold::fq(hostfold::mul(hostfold::h6(*a), hostfold::h6(*b))); }
void hc_hf_add(const Fq *a, const Fq *b, Fq *o) { *o = hostfold::fq(hostfold::add(hostfold::h6(*a), hostfold::h6(*b))); }
void hc_hf_sub(const Fq *a, const Fq *b, Fq *o) { *o = hostfold::fq(hostfold::sub(hostfold::h6(*a), hostfold::h6(*b))); }
int hc_hf_has_adx() {
#if defined(MI_HOSTFOLD_ADX)
    return hostfold::have_adx() ? 1 : 0;
#else
    return 0;
#endif
}
// sum_g 2^g h[g] over n per-bit sums given as affine points ((0,0): none), split at `cut`: low part + shifted high part
void hc_hf_fold(const G1Affine *h, int n, int cut, G1Affine *o) {
    std::vector<G1Xyzz> v((size_t)n);
    for (int i = 0; i < n; i++) v[(size_t)i] = G1Xyzz::from_affine(h[i]);
    const hostfold::PX lo = hostfold::fold_bits(v.data(), 0, (size_t)cut);
    const hostfold::PX hi = hostfold::shift(hostfold::fold_bits(v.data(), (size_t)cut, (size_t)n), (size_t)cut);
    *o = to_affine(hostfold::xyzz(hostfold::add(lo, hi)));
}
int hc_fq_sqrt(const Fq *a, Fq *o) { return fq_sqrt(*a, *o) ? 1 : 0; }
int hc_g1_decompress(const uint8_t *in, int check, G1Affine *o) { return g1_decompress_one(in, check != 0, *o) ? 1 : 0; }
void hc_g1_compress(const G1Affine *p, uint8_t *out) { g1_compress_one(*p, out); }
int hc_g1_in_subgroup(const G1Affine *p) { return g1_in_subgroup(*p) ? 1 : 0; }
int hc_constants_ok() {
    // the 32-bit limb tables in field.hpp must equal the generated 64-bit constants
    for (int i = 0; i < 6; i++) {
        if ((((uint64_t)FqP::MOD[2 * i + 1] << 32) | FqP::MOD[2 * i]) != MI_FQ_MODULUS[i]) return 1;
        if ((((uint64_t)FqP::ONE[2 * i + 1] << 32) | FqP::ONE[2 * i]) != MI_FQ_ONE[i]) return 2;
        if ((((uint64_t)FqP::R2[2 * i + 1] << 32) | FqP::R2[2 * i]) != MI_FQ_R2[i]) return 3;
    }
    for (int i = 0; i < 4; i++) {
        if ((((uint64_t)FrP::MOD[2 * i + 1] << 32) | FrP::MOD[2 * i]) != MI_FR_MODULUS[i]) return 4;
        if ((((uint64_t)FrP::ONE[2 * i + 1] << 32) | FrP::ONE[2 * i]) != MI_FR_ONE[i]) return 5;
        if ((((uint64_t)FrP::R2[2 * i + 1] << 32) | FrP::R2[2 * i]) != MI_FR_R2[i]) return 6;
    }
    if ((uint32_t)MI_FQ_INV != 0xffffffffu || (uint32_t)MI_FR_INV != 0xffffffffu) return 7;
    return 0;
}
}
