// Batched Schnorr signature verification over Edwards-BLS12 on gfx950.
//
// What the BFT layer checks for every signer of a batch certificate -- reference node/bft/src/helpers/storage.rs:465
// (Storage::check_certificate), node/bft/src/gateway.rs:1391 (the handshake's verify_bytes), account/src/lib.rs:76-88 --
// through snarkVM's console::account::Signature [RECALL: the crate is not in /root/reference; parity unpinned]:
//   g_r = response G + challenge pk_sig;   challenge == H8(g_r.x, pk_sig.x, pr_sig.x, address.x, message)
//   address == pk_sig + pr_sig + H4(pk_sig.x, pr_sig.x) G
// on the twisted Edwards curve -x^2 + y^2 = 1 + 3021 x^2 y^2 over Fr of BLS12-377 (complete addition law: no branches),
// H4 / H8 = one permutation of a Poseidon sponge over Fr of width 5 / 9.  The signature carries (challenge, response),
// not the nonce commitment, so there is no random-linear-combination shortcut: g_r has to be rebuilt per signature to
// be hashed.  What a certificate's hundreds of signatures offer is data parallelism, and per signature three
// independent chains that run as three ROLES side by side (a wave executes one role):
//   role 0   challenge * pk_sig      variable base: 2-bit windows over a table {P, 2P, 3P} in registers, 250 doublings
//   role 1   response * G            fixed base: 63 mixed additions from a 63 x 15 table of multiples of G (HBM / L2)
//   role 2   H4 -> sk_prf * G (fixed base) + pk_sig + pr_sig == address, and the well-formedness of the record
// then sixteen lanes per signature add the halves of g_r, normalise x (one inversion) and run H8 with the state spread
// over nine of them.
// Extended coordinates (X, Y, Z, T), add-2008-hwcd-3 / dbl-2008-hwcd with a = -1; Fr on 8 x 32-bit limbs (field.hpp).
#include "api.hpp"
#include "ed_constants.hpp"

namespace mi {

MI_HD Fr fr_from_u64x4(const uint64_t *c) {
    Fr r;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        r.l[2 * i] = (uint32_t)c[i];
        r.l[2 * i + 1] = (uint32_t)(c[i] >> 32);
    }
    return r;
}
struct EdExt {
    Fr x, y, z, t;
};
struct EdNiels {  // an affine point prepared for mixed additions: y + x, y - x, 2 d x y
    Fr yp, ym, td;
};
MI_HD EdExt ed_identity() { return EdExt{Fr::zero(), Fr::one(), Fr::one(), Fr::zero()}; }
MI_HD EdExt ed_add(const EdExt &p, const EdExt &q, const Fr &d2) {
    const Fr a = mul(sub(p.y, p.x), sub(q.y, q.x)), b = mul(add(p.y, p.x), add(q.y, q.x));
    const Fr c = mul(mul(p.t, q.t), d2), d = dbl(mul(p.z, q.z));
    const Fr e = sub(b, a), f = sub(d, c), g = add(d, c), h = add(b, a);
    return EdExt{mul(e, f), mul(g, h), mul(f, g), mul(e, h)};
}
MI_HD EdExt ed_add_niels(const EdExt &p, const EdNiels &q) {  // 7 products
    const Fr a = mul(sub(p.y, p.x), q.ym), b = mul(add(p.y, p.x), q.yp);
    const Fr c = mul(p.t, q.td), d = dbl(p.z);
    const Fr e = sub(b, a), f = sub(d, c), g = add(d, c), h = add(b, a);
    return EdExt{mul(e, f), mul(g, h), mul(f, g), mul(e, h)};
}
MI_HD EdExt ed_dbl(const EdExt &p) {
    const Fr a = sqr(p.x), b = sqr(p.y), c = dbl(sqr(p.z));
    const Fr d = neg(a);
    const Fr e = sub(sub(sqr(add(p.x, p.y)), a), b);
    const Fr g = add(d, b), f = sub(g, c), h = sub(d, b);
    return EdExt{mul(e, f), mul(g, h), mul(f, g), mul(e, h)};
}

#define ED_FB_WINDOWS 63  // 4-bit windows of a 251-bit scalar
struct EdConsts {
    Fr d, d2;
    Fr gx, gy;
    uint32_t order[8];
    Fr ark5[39][5], mds5[5][5];
    Fr ark9[39][9], mds9[9][9];
    EdNiels fb[ED_FB_WINDOWS][15];  // fb[w][d - 1] = d 16^w G
};
struct EdState {
    DevBuf consts;
    bool ready = false;
    ~EdState() { consts.release(); }
};
static const EdConsts *ed_consts(Context &cx, hipStream_t st) {
    EdState &s = cx.mod<EdState>("schnorr.consts");
    if (s.ready) return (const EdConsts *)s.consts.p;
    std::unique_ptr<EdConsts> h(new EdConsts());
    h->d = fr_from_u64x4(MI_ED_D);
    h->d2 = fr_from_u64x4(MI_ED_2D);
    h->gx = fr_from_u64x4(MI_ED_GEN[0]);
    h->gy = fr_from_u64x4(MI_ED_GEN[1]);
    for (int i = 0; i < 4; i++) {
        h->order[2 * i] = (uint32_t)MI_ED_ORDER[i];
        h->order[2 * i + 1] = (uint32_t)(MI_ED_ORDER[i] >> 32);
    }
    for (int r = 0; r < 39; r++) {
        for (int i = 0; i < 5; i++) h->ark5[r][i] = fr_from_u64x4(MI_ED_ARK5[r][i]);
        for (int i = 0; i < 9; i++) h->ark9[r][i] = fr_from_u64x4(MI_ED_ARK9[r][i]);
    }
    for (int i = 0; i < 5; i++)
        for (int j = 0; j < 5; j++) h->mds5[i][j] = fr_from_u64x4(MI_ED_MDS5[i][j]);
    for (int i = 0; i < 9; i++)
        for (int j = 0; j < 9; j++) h->mds9[i][j] = fr_from_u64x4(MI_ED_MDS9[i][j]);
    // the fixed-base table (host arithmetic, once per device): window bases 16^w G by doubling, their multiples by
    // addition, each normalised to affine for the mixed addition
    EdExt base{h->gx, h->gy, Fr::one(), mul(h->gx, h->gy)};
    for (int w = 0; w < ED_FB_WINDOWS; w++) {
        EdExt acc = base;
        for (int d = 1; d <= 15; d++) {
            const Fr zi = inv(acc.z), x = mul(acc.x, zi), y = mul(acc.y, zi);
            h->fb[w][d - 1] = EdNiels{add(y, x), sub(y, x), mul(mul(x, y), h->d2)};
            acc = ed_add(acc, base, h->d2);
        }
        for (int k = 0; k < 4; k++) base = ed_dbl(base);
    }
    MI_HIP(hipMemcpyAsync(s.consts.get(sizeof(EdConsts)), h.get(), sizeof(EdConsts), hipMemcpyHostToDevice, st));
    MI_HIP(hipStreamSynchronize(st));
    s.ready = true;
    return (const EdConsts *)s.consts.p;
}

// ---- device helpers
__device__ __forceinline__ Fr frl(const void *p) {
    const uint4 *q = reinterpret_cast<const uint4 *>(p);
    const uint4 a = q[0], b = q[1];
    Fr r;
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
    r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
    return r;
}
__device__ __forceinline__ void frs(void *p, const Fr &v) {
    uint4 *q = reinterpret_cast<uint4 *>(p);
    q[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    q[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
}
__device__ __forceinline__ bool limbs8_below(const uint32_t *v, const uint32_t *m) {
    bool lt = false, decided = false;
    for (int k = 7; k >= 0; k--)
        if (!decided && v[k] != m[k]) {
            lt = v[k] < m[k];
            decided = true;
        }
    return lt;
}
__device__ __attribute__((noinline)) Fr pow17(Fr x) {
    const Fr x2 = sqr(x), x4 = sqr(x2), x8 = sqr(x4), x16 = sqr(x8);
    return mul(x16, x);
}
// one permutation (W = 5 or 9); state in registers
template <int W>
__device__ void ed_poseidon(Fr (&s)[W], const Fr (*ark)[W], const Fr (*mds)[W]) {
#pragma unroll 1
    for (int r = 0; r < 39; r++) {
#pragma unroll
        for (int i = 0; i < W; i++) s[i] = add(s[i], ark[r][i]);
        if (r < 4 || r >= 35) {
#pragma unroll
            for (int i = 0; i < W; i++) s[i] = pow17(s[i]);
        } else {
            s[0] = pow17(s[0]);
        }
        Fr n[W];
#pragma unroll
        for (int i = 0; i < W; i++) {  // (unrolled: the products are calls, and a runtime index into n[] would go through scratch memory)
            Fr acc = Fr::zero();
#pragma unroll
            for (int j = 0; j < W; j++) acc = add(acc, mulr(mds[i][j], s[j]));
            n[i] = acc;
        }
#pragma unroll
        for (int i = 0; i < W; i++) s[i] = n[i];
    }
}
// low 250 bits of a canonical value
__device__ __forceinline__ void scalar250(uint32_t out[8], const Fr &mont) {
    const Fr c = from_mont(mont);
#pragma unroll
    for (int k = 0; k < 8; k++) out[k] = c.l[k];
    out[7] &= 0x03ffffffu;
}
// k G by the fixed-base table, k < 2^252 (eight words, top nibble zero)
__device__ EdExt ed_fixed_base(const uint32_t k[8], const EdConsts *__restrict__ kc) {
    EdExt acc = ed_identity();
#pragma unroll 1
    for (int w = 0; w < ED_FB_WINDOWS; w++) {
        const uint32_t d = (k[w >> 3] >> (4 * (w & 7))) & 15u;
        // digit 0 adds nothing: the lanes that have one idle through the addition (complete formulas would allow adding
        // the identity; a predicate is cheaper than a table entry for it)
        if (d) {
            const EdNiels *e = &kc->fb[w][d - 1];
            acc = ed_add_niels(acc, EdNiels{frl(&e->yp), frl(&e->ym), frl(&e->td)});
        }
    }
    return acc;
}
__device__ __forceinline__ bool ed_on_curve(const Fr &x, const Fr &y, const Fr &d) {
    const Fr xx = sqr(x), yy = sqr(y);
    return sub(yy, xx) == add(mul(mul(xx, yy), d), Fr::one());
}

// record words: address x, y (0..15), message (16..23), challenge (24..31), response (32..39), pk_sig (40..55), pr_sig (56..71)
#define SIG_WORDS 72
// role = blockIdx.y.  halves[2 i + role] (roles 0, 1) = that half of g_r; flags[i] = well-formed and address identity (role 2)
__global__ void __launch_bounds__(64) schnorr_mults_kernel(const uint32_t *__restrict__ recs, size_t n, const EdConsts *__restrict__ kc,
                                                            EdExt *__restrict__ halves, uint8_t *__restrict__ flags) {
    const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const uint32_t *r = recs + i * SIG_WORDS;
    const int role = blockIdx.y;
    if (role == 0) {  // challenge * pk_sig: 125 windows of two bits, most significant first
        const Fr d2 = frl(&kc->d2);
        const Fr px = frl(r + 40), py = frl(r + 48);
        const EdExt p1{px, py, Fr::one(), mul(px, py)};
        const EdExt p2 = ed_dbl(p1), p3 = ed_add(p2, p1, d2);
        uint32_t c[8];
#pragma unroll
        for (int k = 0; k < 8; k++) c[k] = r[24 + k];
        EdExt acc = ed_identity();
#pragma unroll 1
        for (int w = 125; w >= 0; w--) {
            acc = ed_dbl(ed_dbl(acc));
            const uint32_t dgt = (c[w >> 4] >> (2 * (w & 15))) & 3u;
            if (dgt) {
                EdExt t;
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    t.x.l[k] = dgt == 1 ? p1.x.l[k] : (dgt == 2 ? p2.x.l[k] : p3.x.l[k]);
                    t.y.l[k] = dgt == 1 ? p1.y.l[k] : (dgt == 2 ? p2.y.l[k] : p3.y.l[k]);
                    t.z.l[k] = dgt == 1 ? p1.z.l[k] : (dgt == 2 ? p2.z.l[k] : p3.z.l[k]);
                    t.t.l[k] = dgt == 1 ? p1.t.l[k] : (dgt == 2 ? p2.t.l[k] : p3.t.l[k]);
                }
                acc = ed_add(acc, t, d2);
            }
        }
        halves[2 * i] = acc;
    } else if (role == 1) {  // response * G
        uint32_t s[8];
#pragma unroll
        for (int k = 0; k < 8; k++) s[k] = r[32 + k];
        halves[2 * i + 1] = ed_fixed_base(s, kc);
    } else {  // well-formedness and the address identity
        const Fr d = frl(&kc->d), d2 = frl(&kc->d2);
        const Fr ax = frl(r), ay = frl(r + 8), px = frl(r + 40), py = frl(r + 48), qx = frl(r + 56), qy = frl(r + 64);
        bool ok = true;
        for (int f = 0; f < 3; f++) ok = ok && limbs8_below(r + 8 * f, FrP::MOD);            // address, message
        for (int f = 5; f < 9; f++) ok = ok && limbs8_below(r + 8 * f, FrP::MOD);            // pk_sig, pr_sig
        ok = ok && limbs8_below(r + 24, kc->order) && limbs8_below(r + 32, kc->order);      // challenge, response below s
        ok = ok && ed_on_curve(ax, ay, d) && ed_on_curve(px, py, d) && ed_on_curve(qx, qy, d);
        Fr st[5] = {Fr::zero(), px, qx, Fr::zero(), Fr::zero()};
        st[0].l[0] = 2;  // capacity: rate 2^64 + inputs = 4 * 2^64 + 2 (canonical words 0 and 2)
        st[0].l[2] = 4;
        st[0] = to_mont(st[0]);
        ed_poseidon<5>(st, kc->ark5, kc->mds5);
        uint32_t prf[8];
        scalar250(prf, st[1]);
        EdExt acc = ed_fixed_base(prf, kc);
        acc = ed_add(acc, EdExt{px, py, Fr::one(), mul(px, py)}, d2);
        acc = ed_add(acc, EdExt{qx, qy, Fr::one(), mul(qx, qy)}, d2);
        ok = ok && mul(ax, acc.z) == acc.x && mul(ay, acc.z) == acc.y;
        flags[i] = ok ? 1 : 0;
    }
}
// SIXTEEN lanes per signature (four signatures to a wave): all of them add the halves of g_r and normalise x in
// lockstep (the same time as one lane doing it), then the width-9 permutation of the challenge hash runs with lane L < 9
// holding state word L: a round is the S-box (five products, every lane in step), nine products for the lane's row of
// the MDS matrix and nine 32-byte shuffles that bring the other words in -- ~14 products in sequence per round instead
// of the 86 / 94 a single lane walks: 39 rounds x 14 against 3 700.
__global__ void __launch_bounds__(64) schnorr_finish_kernel(const uint32_t *__restrict__ recs, size_t n, const EdConsts *__restrict__ kc,
                                                             const EdExt *__restrict__ halves, const uint8_t *__restrict__ flags,
                                                             uint8_t *__restrict__ verdicts) {
    const int tid = threadIdx.x, L = tid & 15, grp = tid & ~15;
    const size_t i = ((size_t)blockIdx.x * 64 + tid) >> 4;
    const bool live = i < n;
    const size_t ii = live ? i : 0;
    const uint32_t *r = recs + ii * SIG_WORDS;
    const EdExt g = ed_add(halves[2 * ii], halves[2 * ii + 1], frl(&kc->d2));
    const Fr gx = mul(g.x, invr(g.z));  // Z is never zero on a complete curve
    // lane L's state word: capacity (8 * 2^64 + 5 inputs), g_r.x, pk_sig.x, pr_sig.x, address.x, message, zeros
    Fr s = Fr::zero();
    if (L == 0) {
        s.l[0] = 5;
        s.l[2] = 8;
        s = to_mont(s);
    } else if (L == 1) s = gx;
    else if (L == 2) s = frl(r + 40);
    else if (L == 3) s = frl(r + 56);
    else if (L == 4) s = frl(r);
    else if (L == 5) s = frl(r + 16);
    const int row = L < 9 ? L : 0;
    Fr m[9];
#pragma unroll
    for (int j = 0; j < 9; j++) m[j] = frl(&kc->mds9[row][j]);
#pragma unroll 1
    for (int rd = 0; rd < 39; rd++) {
        s = add(s, frl(&kc->ark9[rd][row]));
        const Fr p = pow17(s);
        if (rd < 4 || rd >= 35 || L == 0) s = p;
        Fr acc = Fr::zero();
#pragma unroll
        for (int j = 0; j < 9; j++) {
            Fr sj;
#pragma unroll
            for (int k = 0; k < 8; k++) sj.l[k] = (uint32_t)__shfl((int)s.l[k], grp + j, 64);
            acc = add(acc, mulr(m[j], sj));
        }
        s = acc;
    }
    if (L == 1 && live) {
        uint32_t c[8];
        scalar250(c, s);
        bool ok = flags[i] != 0;
#pragma unroll
        for (int k = 0; k < 8; k++) ok = ok && c[k] == r[24 + k];
        verdicts[i] = ok ? 1 : 0;
    }
}

void schnorr_verify_device(Context &cx, const void *d_recs, size_t n, uint8_t *d_verdicts, hipStream_t st) {
    if (!n) return;
    const EdConsts *kc = ed_consts(cx, st);
    // temporaries of an asynchronous entry point come from the scratch pool (one entry, guarded by an event on `st`)
    const size_t halves_bytes = sizeof(EdExt) * 2 * n;
    ScratchPool::Entry *scr = cx.scratch.acquire(halves_bytes + n, st);
    EdExt *halves = (EdExt *)scr->buf.p;
    uint8_t *flags = (uint8_t *)scr->buf.p + halves_bytes;
    cx.prof.begin("schnorr", st, n);
    hipLaunchKernelGGL(schnorr_mults_kernel, dim3((unsigned)cdiv(n, 64), 3), dim3(64), 0, st, (const uint32_t *)d_recs, n, kc, halves, flags);
    hipLaunchKernelGGL(schnorr_finish_kernel, dim3((unsigned)cdiv(n, 4)), dim3(64), 0, st, (const uint32_t *)d_recs, n, kc,
                       (const EdExt *)halves, (const uint8_t *)flags, d_verdicts);
    cx.prof.end("schnorr", st);
    MI_HIP(hipGetLastError());
    ScratchPool::release(scr, st);
}

}  // namespace mi

using namespace mi;

extern "C" {

int32_t mi_schnorr_verify_batch(const mi_sig_record *recs, size_t n, uint8_t *verdicts) {
    if (!recs || !n || !verdicts) return MI_ERR_BAD_ARGS;
    return guarded([&](Context &cx) -> int32_t {
        void *d = cx.buf("schnorr.in", sizeof(mi_sig_record) * n);
        uint8_t *dv = (uint8_t *)cx.buf("schnorr.out", n);
        MI_HIP(hipMemcpyAsync(d, recs, sizeof(mi_sig_record) * n, hipMemcpyHostToDevice, cx.stream));
        schnorr_verify_device(cx, d, n, dv, cx.stream);
        MI_HIP(hipMemcpyAsync(verdicts, dv, n, hipMemcpyDeviceToHost, cx.stream));
        MI_HIP(hipStreamSynchronize(cx.stream));
        return MI_OK;
    });
}
int32_t mi_schnorr_verify_batch_device(const void *d_recs, size_t n, void *d_verdicts, void *stream) {
    if (!d_recs || !n || !d_verdicts) return MI_ERR_BAD_ARGS;
    return guarded_ptr(d_recs, [&](Context &cx) -> int32_t {
        schnorr_verify_device(cx, d_recs, n, (uint8_t *)d_verdicts, (hipStream_t)stream);
        return MI_OK;
    });
}

}  // extern "C"
