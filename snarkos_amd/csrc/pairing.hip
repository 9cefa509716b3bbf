// Pairing product on the device: kernel + host drivers.  Device/host arithmetic lives in tower.hpp.
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "api.hpp"
#include "tower.hpp"
#include "pairing_wg.hpp"

namespace mi {

struct PairingArgs {
    G1Affine p[MI_MAX_PAIRS];
    const G2Prepared *q[MI_MAX_PAIRS];
    int n;
    int verdict_only;  // the caller asks "is the product one?" only: pw_final_exp's form without the inversion in Fq
};

// one workgroup of PW_THREADS lanes per pairing product (pairing_wg.hpp); merged: the Miller factors of pairing_lines_kernel
__global__ void __launch_bounds__(PW_THREADS) pairing_product_kernel(PairingArgs args, const PairingConsts *__restrict__ pc,
                                                                     Fq12 *out, const Fq *merged) {
    extern __shared__ uint4 pw_lds[];  // PW_LDS_BYTES
    LShared &S = *(LShared *)pw_lds;
    pw_pairing_product(S, args.p, args.q, args.n, pc, out, merged, args.verdict_only != 0);
}
// first launch of the two-launch form: workgroup g computes the merged Miller factor of steps 3g .. 3g + 2 over all
// the pairs (pw_miller_group) into merged[g][12]
__global__ void __launch_bounds__(PW_THREADS) pairing_lines_kernel(PairingArgs args, const PairingConsts *__restrict__ pc, Fq *merged) {
    extern __shared__ uint4 pw_lds[];  // PW_LDS_BYTES
    LShared &S = *(LShared *)pw_lds;
    __shared__ const G2Prepared *live_q[MI_MAX_PAIRS];
    __shared__ int live_n_s;
    const int live_n = pw_live_pairs(S, args.p, args.q, args.n, live_q, &live_n_s, pc);
    if (live_n == 0) return;  // uniform: the product kernel sets one itself
    pw_miller_group(S, live_q, live_n, blockIdx.x, merged + (size_t)blockIdx.x * 12);
}

// Several independent products in ONE launch each (a workgroup per product; the line launch adds a grid dimension):
// products that become ready together -- the batches of a lane pass, a burst -- need one product stream, not one each.
struct PairingArgsN {
    int count;
    PairingArgs a[MI_VS_MAX_PRODUCTS_PER_TICKET];
};
__global__ void __launch_bounds__(PW_THREADS) pairing_product_multi_kernel(PairingArgsN args, const PairingConsts *__restrict__ pc,
                                                                           Fq12 *out, const Fq *merged, size_t merged_stride) {
    extern __shared__ uint4 pw_lds[];  // PW_LDS_BYTES
    LShared &S = *(LShared *)pw_lds;
    const PairingArgs &a = args.a[blockIdx.x];
    pw_pairing_product(S, a.p, a.q, a.n, pc, out + blockIdx.x, merged ? merged + blockIdx.x * merged_stride : nullptr, a.verdict_only != 0);
}
__global__ void __launch_bounds__(PW_THREADS) pairing_lines_multi_kernel(PairingArgsN args, const PairingConsts *__restrict__ pc, Fq *merged,
                                                                         size_t merged_stride) {
    extern __shared__ uint4 pw_lds[];  // PW_LDS_BYTES
    LShared &S = *(LShared *)pw_lds;
    __shared__ const G2Prepared *live_q[MI_MAX_PAIRS];
    __shared__ int live_n_s;
    const PairingArgs &a = args.a[blockIdx.y];
    const int live_n = pw_live_pairs(S, a.p, a.q, a.n, live_q, &live_n_s, pc);
    if (live_n == 0) return;  // uniform
    pw_miller_group(S, live_q, live_n, blockIdx.x, merged + blockIdx.y * merged_stride + (size_t)blockIdx.x * 12);
}

struct PairingState {  // per device (Context::mod)
    DevBuf consts;
    bool ready = false;
    ~PairingState() { consts.release(); }
};
static Fq host_fq_from_limbs(const uint64_t *l) {
    Fq r;
    for (int i = 0; i < 6; i++) { r.l[2 * i] = (uint32_t)l[i]; r.l[2 * i + 1] = (uint32_t)(l[i] >> 32); }
    return r;
}
const PairingConsts *pairing_consts(Context &cx, hipStream_t st) {
    PairingState &s = cx.mod<PairingState>("pairing");
    if (!s.ready) {
        PairingConsts h;
        for (int e = 0; e < 3; e++)
            for (int k = 0; k < 12; k++) h.frob[e][k] = host_fq_from_limbs(MI_FROB[e][k]);
        // D[12 + i] = 2^(32 i) (i < 12), D[24] = 2^384 mod q; below 12, D[m - 1] = D[m] / 2^32 mod q by 32 halvings
        for (int m = 12; m < 24; m++) {
            h.dtab[m] = Fq::zero();
            h.dtab[m].l[m - 12] = 1u;
        }
        h.dtab[24] = Fq::one();  // R mod q as an integer
        for (int m = 11; m >= 0; m--) {
            Fq v = h.dtab[m + 1];
            for (int b = 0; b < 32; b++) {
                uint32_t carry = 0;
                if (v.l[0] & 1u)
                    for (int i = 0; i < 12; i++) v.l[i] = addc32(v.l[i], FqP::MOD[i], carry, &carry);
                for (int i = 0; i < 11; i++) v.l[i] = (v.l[i] >> 1) | (v.l[i + 1] << 31);
                v.l[11] = (v.l[11] >> 1) | (carry << 31);
            }
            h.dtab[m] = v;
        }
        void *d = s.consts.get(sizeof h);
        MI_HIP(hipMemcpyAsync(d, &h, sizeof h, hipMemcpyHostToDevice, st));
        MI_HIP(hipStreamSynchronize(st));
        s.ready = true;
    }
    return (const PairingConsts *)s.consts.p;
}

// product over prepared (device-resident) G2 points; P's are host values passed as kernel arguments
// enqueue one pairing product on `st`; the result lands in d_out (device)
// Two launches: 21 workgroups per couple prepare the merged Miller factors (pairing_lines_kernel), the product kernel
// is 48 rounds (159 for two couples) shorter: 0.87 -> 0.76 ms for a product somebody waits for; in the streamed
// verifier, where eight products are in flight, the product kernel runs 0.78 ms instead of 0.93 and the throughput is
// the same within noise (tools/ab_pairing_launches.sh).  unit names the caller's product slot (its workspace for the
// factors; -1: the synchronous path).  MI_SNARK_PAIRING_LAUNCHES=1 forces the one-launch form.
void pairing_launch(Context &cx, const G1Affine *ps, const G2Prepared *const *d_qs, size_t n, Fq12 *d_out, hipStream_t st, int unit,
                    bool verdict_only) {
    PairingArgs a;
    memset(&a, 0, sizeof a);
    a.n = (int)n;
    a.verdict_only = verdict_only ? 1 : 0;
    for (size_t i = 0; i < n; i++) { a.p[i] = ps[i]; a.q[i] = d_qs[i]; }
    const PairingConsts *pc = pairing_consts(cx, st);
    static const int force = getenv("MI_SNARK_PAIRING_LAUNCHES") ? atoi(getenv("MI_SNARK_PAIRING_LAUNCHES")) : 0;
    const bool two_launches = force != 1;
    Fq *merged = nullptr;
    cx.raise_lds_limit((const void *)pairing_product_kernel, (int)PW_LDS_BYTES);
    if (two_launches && n > 0) {
        char nm[32];
        snprintf(nm, sizeof nm, "pairing.m%d", unit < 0 ? 99 : unit);
        merged = (Fq *)cx.buf(nm, sizeof(Fq) * 12 * PW_MILLER_GROUPS);
        cx.raise_lds_limit((const void *)pairing_lines_kernel, (int)PW_LDS_BYTES);
        cx.prof.begin("pairing_lines", st, n);
        hipLaunchKernelGGL(pairing_lines_kernel, dim3(PW_MILLER_GROUPS), dim3(PW_THREADS), PW_LDS_BYTES, st, a, pc, merged);
        cx.prof.end("pairing_lines", st);
    }
    cx.prof.begin("pairing", st, n);  // the product kernel alone (what the kernel trace calls pairing_product_kernel)
    hipLaunchKernelGGL(pairing_product_kernel, dim3(1), dim3(PW_THREADS), PW_LDS_BYTES, st, a, pc, d_out, (const Fq *)merged);
    cx.prof.end("pairing", st);
    MI_HIP(hipGetLastError());
}

// `count` products at once: product k pairs ps[k * MI_MAX_PAIRS + i] with d_qs[i], i < n (the same prepared G2 points and
// the same number of pairs for all: one verifier context); results in d_out[0 .. count): one, or something that is not
// (verdict form: nobody reads these values for anything but is_one)
void pairing_launch_multi(Context &cx, const G1Affine *ps, const G2Prepared *const *d_qs, size_t n, int count, Fq12 *d_out, hipStream_t st,
                          int unit) {
    if (count == 1) return pairing_launch(cx, ps, d_qs, n, d_out, st, unit, true);
    if (count < 1 || count > MI_VS_MAX_PRODUCTS_PER_TICKET || n == 0) throw HipError{hipErrorInvalidValue, "pairing_launch_multi", __LINE__};
    PairingArgsN a;
    memset(&a, 0, sizeof a);
    a.count = count;
    for (int k = 0; k < count; k++) {
        a.a[k].n = (int)n;
        a.a[k].verdict_only = 1;
        for (size_t i = 0; i < n; i++) {
            a.a[k].p[i] = ps[(size_t)k * MI_MAX_PAIRS + i];
            a.a[k].q[i] = d_qs[i];
        }
    }
    const PairingConsts *pc = pairing_consts(cx, st);
    const size_t stride = (size_t)12 * PW_MILLER_GROUPS;  // Fq per product
    char nm[32];
    snprintf(nm, sizeof nm, "pairing.mm%d", unit < 0 ? 99 : unit);
    Fq *merged = (Fq *)cx.buf(nm, sizeof(Fq) * stride * MI_VS_MAX_PRODUCTS_PER_TICKET);
    cx.raise_lds_limit((const void *)pairing_product_multi_kernel, (int)PW_LDS_BYTES);
    cx.raise_lds_limit((const void *)pairing_lines_multi_kernel, (int)PW_LDS_BYTES);
    cx.prof.begin("pairing_lines", st, n * count);
    hipLaunchKernelGGL(pairing_lines_multi_kernel, dim3(PW_MILLER_GROUPS, count), dim3(PW_THREADS), PW_LDS_BYTES, st, a, pc, merged,
                       stride);
    cx.prof.end("pairing_lines", st);
    cx.prof.begin("pairing", st, n * count);
    hipLaunchKernelGGL(pairing_product_multi_kernel, dim3(count), dim3(PW_THREADS), PW_LDS_BYTES, st, a, pc, d_out, (const Fq *)merged, stride);
    cx.prof.end("pairing", st);
    MI_HIP(hipGetLastError());
}

void pairing_product_prepared(Context &cx, const G1Affine *ps, const G2Prepared *const *d_qs, size_t n, Fq12 *out_host,
                                     hipStream_t st, bool verdict_only) {
    Fq12 *d_out = (Fq12 *)cx.buf("pairing.out", sizeof(Fq12));
    pairing_launch(cx, ps, d_qs, n, d_out, st, -1, verdict_only);
    Fq12 *h = (Fq12 *)cx.pin(sizeof(Fq12));
    MI_HIP(hipMemcpyAsync(h, d_out, sizeof(Fq12), hipMemcpyDeviceToHost, st));
    MI_HIP(hipStreamSynchronize(st));
    *out_host = *h;
}

// arbitrary G2 points given on the host: prepare (host), upload, run
void pairing_product_host_args(Context &cx, const G1Affine *ps, const G2Affine *qs, size_t n, Fq12 *out_host,
                                      hipStream_t st, bool verdict_only) {
    std::vector<G2Prepared> prep(n);
    for (size_t i = 0; i < n; i++) g2_prepare_host(prep[i], qs[i]);
    G2Prepared *d = (G2Prepared *)cx.buf("pairing.adhoc_q", sizeof(G2Prepared) * n);
    MI_HIP(hipMemcpyAsync(d, prep.data(), sizeof(G2Prepared) * n, hipMemcpyHostToDevice, st));
    MI_HIP(hipStreamSynchronize(st));
    const G2Prepared *dq[MI_MAX_PAIRS];
    for (size_t i = 0; i < n; i++) dq[i] = d + i;
    pairing_product_prepared(cx, ps, dq, n, out_host, st, verdict_only);
}

}  // namespace mi
