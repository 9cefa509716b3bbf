// KZG10 batch_check and the VarunaShaped-v1 batch verifier on gfx950.
//
// Replaces, for the consensus-side batch (Consensus::try_advance_to_next_block ->
// Ledger::check_next_block -> VM::check_transactions -> Varuna::verify_batch -> SonicKZG10 -> KZG10):
//   polycommit::kzg10::KZG10::batch_check                         
//   polycommit::sonic_pc::SonicKZG10::{accumulate_elems,check_elems} (row A6) and the Fr / Fiat-Shamir
//   work of snark::varuna::VarunaSNARK::verify_batch              (row A7), on the synthetic proof shape
//   "VarunaShaped-v1"  with a SHA-256 stand-in transcript (row A9 is not reproduced: the
//   Poseidon parameters of snarkVM are not available offline).
// The reference verifies one transaction at a time (rayon over transactions, one pairing product per
// proof).  Here the whole batch becomes ONE random-linear-combination MSM plus ONE two-pair pairing
// product; only if that aggregate check fails are proofs checked one by one to name the bad ones.
#include <map>
#include <thread>
#include <vector>
#include "api.hpp"
#include "tower.hpp"
#include "pairing_wg.hpp"
#include "glv.hpp"
#include "codec28.hpp"

namespace mi {

// ---------------------------------------------------------------- small Fr helpers
__device__ __forceinline__ Fr fr_load(const void *p) {
    const uint4 *q = reinterpret_cast<const uint4 *>(p);
    uint4 a = q[0], b = q[1];
    Fr r;
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
    r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
    return r;
}
__device__ __forceinline__ void fr_store(void *p, const Fr &v) {
    uint4 *q = reinterpret_cast<uint4 *>(p);
    q[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    q[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
}
// 128-bit little-endian integer -> Montgomery Fr
__device__ __forceinline__ Fr fr_from_u128(const uint32_t w[4]) {
    Fr t = Fr::zero();
    t.l[0] = w[0]; t.l[1] = w[1]; t.l[2] = w[2]; t.l[3] = w[3];
    return to_mont(t);
}
__device__ __forceinline__ void copy_g1(G1Affine *dst, const G1Affine *src) {
    const uint4 *s = reinterpret_cast<const uint4 *>(src);
    uint4 *d = reinterpret_cast<uint4 *>(dst);
#pragma unroll
    for (int k = 0; k < 6; k++) d[k] = s[k];
}

// ---------------------------------------------------------------- SHA-256 (stand-in transcript)
__device__ __forceinline__ uint32_t ror32(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
__constant__ uint32_t SHA_K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98,
    0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786,
    0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8,
    0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13,
    0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819,
    0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a,
    0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7,
    0xc67178f2};
__device__ __forceinline__ void sha256_compress(uint32_t st[8], uint32_t w[16]) {
    uint32_t a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
#pragma unroll
    for (int i = 0; i < 64; i++) {
        if (i >= 16) {
            uint32_t w15 = w[(i - 15) & 15], w2 = w[(i - 2) & 15];
            uint32_t s0 = ror32(w15, 7) ^ ror32(w15, 18) ^ (w15 >> 3);
            uint32_t s1 = ror32(w2, 17) ^ ror32(w2, 19) ^ (w2 >> 10);
            w[i & 15] = w[i & 15] + s0 + w[(i - 7) & 15] + s1;
        }
        uint32_t t1 = h + (ror32(e, 6) ^ ror32(e, 11) ^ ror32(e, 25)) + ((e & f) ^ (~e & g)) + SHA_K[i] + w[i & 15];
        uint32_t t2 = (ror32(a, 2) ^ ror32(a, 13) ^ ror32(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
        h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += h;
}
// SHA-256( prefix(32 bytes, given as 8 big-endian words) || a[0..na) || b[0..nb) ), a/b little-endian
// words in memory (i.e. raw bytes).  Output: 8 big-endian state words.
__device__ void sha256_prefixed(uint32_t out[8], const uint32_t prefix[8], const uint32_t *__restrict__ a, uint32_t na,
                                const uint32_t *__restrict__ b, uint32_t nb) {
    uint32_t st[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    const uint32_t total = 8 + na + nb;                 // message words
    const uint32_t nblocks = (total + 1 + 2 + 15) / 16;  // + 0x80 word + 64-bit length
    // a / b are 16-byte aligned and na / nb multiples of 4 (asserted by the callers' layout): the message is read four
    // words at a time (one 16-byte load per lane instead of four scattered 4-byte ones; lanes are 2256 bytes apart), and
    // a block's four loads are issued BEFORE the previous block is compressed: every block is a new cache line for
    // every lane, and the ~1 400 instructions of a compression hide its latency
    auto fetch = [&](uint32_t blk, int q) -> uint4 {
        const uint32_t base = blk * 16 + 4 * (uint32_t)q;
        if (base >= 8 && base < 8 + na) return *reinterpret_cast<const uint4 *>(a + (base - 8));
        if (base >= 8 + na && base < total) return *reinterpret_cast<const uint4 *>(b + (base - 8 - na));
        return make_uint4(0, 0, 0, 0);
    };
    uint4 nxt[4];
#pragma unroll
    for (int q = 0; q < 4; q++) nxt[q] = fetch(0, q);
    for (uint32_t blk = 0; blk < nblocks; blk++) {
        uint4 cur[4];
#pragma unroll
        for (int q = 0; q < 4; q++) cur[q] = nxt[q];
        if (blk + 1 < nblocks) {
#pragma unroll
            for (int q = 0; q < 4; q++) nxt[q] = fetch(blk + 1, q);
        }
        uint32_t w[16];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t base = blk * 16 + 4 * q;
            uint32_t v[4];
            if (blk == 0 && q < 2) {
#pragma unroll
                for (int t = 0; t < 4; t++) v[t] = prefix[4 * q + t];
            } else if (base < total) {
                const uint4 x = cur[q];
                v[0] = __builtin_bswap32(x.x); v[1] = __builtin_bswap32(x.y); v[2] = __builtin_bswap32(x.z); v[3] = __builtin_bswap32(x.w);
            } else {
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    const uint32_t i = base + t;
                    v[t] = i == total ? 0x80000000u : (i == nblocks * 16 - 1 ? total * 32u : 0u);  // bit length < 2^32
                }
            }
#pragma unroll
            for (int t = 0; t < 4; t++) w[4 * q + t] = v[t];
        }
        sha256_compress(st, w);
    }
#pragma unroll
    for (int i = 0; i < 8; i++) out[i] = st[i];
}

// polynomial p (0..20: 9 proof commitments then 12 vk commitments) opens at query 0 (beta) or 1 (gamma): the fixed table of v1
__host__ __device__ __forceinline__ int vs_query_of(int p) {
    if (p < MI_VS_PROOF_COMMS) return p < 5 ? 0 : 1;
    return (p - MI_VS_PROOF_COMMS) < 6 ? 0 : 1;
}


// ---------------------------------------------------------------- VarunaShaped-v1 context
struct VsConsts {
    Fr omega16[MI_VS_INPUTS];  // w_16^k
    Fr n_inv;                  // 1/16
    uint32_t tag[8];           // big-endian words of the 32-byte domain tag
    G1Affine g;
    Fq tag_elem;               // the tag as an absorbed Fq element (its 32 bytes as Montgomery limbs)
    G1Affine gamma_g;          // v2 with hiding
    uint32_t fs_kind;          // 0: SHA-256 transcript, MI_VS_FS_POSEIDON: Poseidon sponge over Fq
    // ---- the shape (VarunaShaped-v2: include/mi355x_snark.h, mi_vs_shape; v1 = the fixed table, nothing bounded or hidden)
    uint32_t proof_words;      // record stride in 32-bit words: 564 (mi_vs_proof) or 796 (mi_vs_proof2)
    uint32_t rand_bytes;       // caller randomizers per proof: 32 (two) or 48 (three)
    uint32_t hiding, n_bounds;
    uint32_t n_bounded;        // proof polynomials with a degree bound
    uint32_t per_proof;        // H-side points per proof: 9 commitments + 2 witnesses + n_bounded shifted commitments
    uint32_t class_count[MI_VS_MAX_BOUNDS];
    uint8_t query_of[MI_VS_EVALS];
    uint8_t lc_kind[MI_VS_EVALS];  // v2: bit 0: times z_beta^16 - 1, bit 1: times z_gamma^16 - 1
    uint32_t max_inst;             // v2 with instances: 1 .. MI_VS_MAX_INSTANCES instances per proof (records: mi_vs_proof3)
    uint32_t inst_slot;            // first of the 2 (max_inst - 1) H-side slots of the further instances' commitments
    int8_t bound_class[MI_VS_PROOF_COMMS];
    uint8_t _pad[2];
    // ---- VarunaShaped-v3
    uint32_t max_circ;             // 1 .. MI_VS_MAX_CIRCUITS distinct circuits per proof (records: mi_vs_proof4)
    uint32_t need_inv;             // some lc_kind carries bit 2: 1 / (z_beta - z_gamma)
    uint32_t wire;                 // records arrive in the wire form (MI_VS_WIRE) and are decoded ahead of the transcript
    uint32_t wire_bytes;           // their stride
};
// The wire form of the context's record kind (include/mi355x_snark.h, MI_VS_WIRE) as the decode kernels walk it: runs of
// raw bytes and of Fr fields, and the point slots the SHAPE uses (shifted commitments of unbounded polynomials are left
// out here; instances and circuits beyond a record's own counts are decided per record).  cls: 0 always used, 2 + j
// the j-th further instance, 8 + d the d-th further circuit.
struct VsWireMap {
    uint32_t n_pts, n_runs;
    uint32_t off_ni, off_nc;       // wire offsets of n_instances / n_circuits (0: the kind has none)
    uint32_t rec_bytes, wire_bytes, max_inst, max_circ;
    uint16_t pt_wire[32], pt_rec[32];
    uint8_t pt_cls[32];
    struct Run { uint16_t wire, rec, count; uint8_t type, cls; } run[24];  // type 0: raw bytes (count = bytes), 1: Fr fields
    // parts of the record that no run and no point slot covers (slots the shape never uses): the decode zeroes them
    uint32_t n_gaps;
    struct Gap { uint16_t rec, bytes; } gap[16];
};
// terms / chains of one proof in the per-proof fallback (see vs_terms_kernel): its own H-side points, the commitments of
// up to max_circ circuits, G (and gamma_g), the two witnesses on the tau-H side, the class accumulators' points
__host__ __device__ __forceinline__ uint32_t vs_terms_of(uint32_t per_proof, uint32_t hiding, uint32_t n_bounded, uint32_t max_circ) {
    return per_proof + MI_VS_VK_COMMS * max_circ + 1 + hiding + 2 + n_bounded;
}
__host__ __device__ __forceinline__ uint32_t vs_chains_of(uint32_t per_proof, uint32_t hiding, uint32_t n_bounded, uint32_t max_circ) {
    return 2 * vs_terms_of(per_proof, hiding, n_bounded, max_circ) - 2;
}
// the wire layout of the context's record kind (v1 records when the context has no shape)
static void vs_build_wire_map(VsWireMap &m, const VsConsts &hc, bool has_shape) {
    memset(&m, 0, sizeof m);
    const int kind = !has_shape ? 1 : (hc.max_circ > 1 ? 4 : (hc.max_inst > 1 ? 3 : 2));
    uint32_t w = 0, r = 0;
    auto raw = [&](uint32_t bytes, uint8_t cls) {
        m.run[m.n_runs++] = VsWireMap::Run{(uint16_t)w, (uint16_t)r, (uint16_t)bytes, 0, cls};
        w += bytes;
        r += bytes;
    };
    auto frs = [&](uint32_t count, uint8_t cls) {
        m.run[m.n_runs++] = VsWireMap::Run{(uint16_t)w, (uint16_t)r, (uint16_t)count, 1, cls};
        w += 32 * count;
        r += 32 * count;
    };
    auto pts = [&](uint32_t count, uint8_t cls, const int8_t *bound_class) {
        for (uint32_t j = 0; j < count; j++) {
            const bool used = !bound_class || (bound_class[j] >= 0 && bound_class[j] < (int)hc.n_bounds);
            if (used) {
                m.pt_wire[m.n_pts] = (uint16_t)w;
                m.pt_rec[m.n_pts] = (uint16_t)r;
                m.pt_cls[m.n_pts] = cls;
                m.n_pts++;
            }
            w += 48;
            r += 96;
        }
    };
    raw(16, 0); frs(MI_VS_INPUTS, 0); pts(MI_VS_PROOF_COMMS, 0, nullptr); frs(MI_VS_EVALS, 0); pts(2, 0, nullptr);
    if (kind >= 2) { frs(2, 0); pts(MI_VS_PROOF_COMMS, 0, hc.bound_class); }
    if (kind >= 3) {
        m.off_ni = w;
        raw(16, 0);
        for (uint32_t j = 0; j < MI_VS_EXTRA_INSTANCES; j++) {
            const uint8_t cls = (uint8_t)(2 + j);
            if (j + 1 < hc.max_inst) { frs(MI_VS_INPUTS, cls); pts(2, cls, nullptr); frs(2, cls); }
            else { w += 32 * MI_VS_INPUTS + 96 + 64; r += sizeof(mi_vs_instance); }  // beyond the shape: never used, stays zero
        }
    }
    if (kind >= 4) {
        m.off_nc = w;
        raw(16, 0);
        for (uint32_t d = 0; d + 1 < MI_VS_MAX_CIRCUITS; d++) {
            const uint8_t cls = (uint8_t)(8 + d);
            if (d + 1 < hc.max_circ) { raw(16, cls); frs(MI_VS_VK_COMMS, cls); }
            else { w += sizeof(mi_vs_circuit_part); r += sizeof(mi_vs_circuit_part); }
        }
    }
    m.wire_bytes = w;
    m.rec_bytes = r;
    m.max_inst = hc.max_inst;
    m.max_circ = hc.max_circ;
    // what is left uncovered (16-byte granules: every field is a multiple of 16 bytes at a multiple of 16)
    std::vector<uint8_t> covered(r / 16, 0);
    for (uint32_t k = 0; k < m.n_runs; k++) {
        const uint32_t bytes = m.run[k].type ? 32u * m.run[k].count : m.run[k].count;
        for (uint32_t b = 0; b < bytes; b += 16) covered[(m.run[k].rec + b) / 16] = 1;
    }
    for (uint32_t k = 0; k < m.n_pts; k++)
        for (uint32_t b = 0; b < 96; b += 16) covered[(m.pt_rec[k] + b) / 16] = 1;
    for (uint32_t g = 0; g < covered.size();) {
        if (covered[g]) { g++; continue; }
        uint32_t e = g;
        while (e < covered.size() && !covered[e]) e++;
        if (m.n_gaps >= 16) throw std::runtime_error("wire map: too many gaps");
        m.gap[m.n_gaps++] = VsWireMap::Gap{(uint16_t)(16 * g), (uint16_t)(16 * (e - g))};
        g = e;
    }
}
// where one proof's operands go (device pointers into the lane's workspace)
struct VsOut {
    G1Affine *pts1, *pts2, *ptsb[MI_VS_MAX_BOUNDS];
    uint32_t *sc1, *sc2, *scb[MI_VS_MAX_BOUNDS];
    Fr *vk_scal, *g_scal, *h_scal;
    uint8_t *alg_ok;
    uint32_t *circuit_of;
    const uint8_t *dec_ok;  // wire records: per-proof decode flags
};
// A lane pass over SEVERAL batches (mi_vs_accumulate_begin_multi): the blocks from b[k].first_block on serve further
// batch k (block ranges ascend with k)
#define VS_MORE (MI_VS_MAX_BATCHES_PER_PASS - 1)
struct VsMore {
    int count = 0;  // further batches in this pass (0: single)
    struct Batch {
        const void *proofs;
        const uint8_t *rand;
        size_t n;
        uint32_t first_block;  // derive and transcript kernels both give a block 64 proofs
        const uint32_t *chal;  // Poseidon transcript: this batch's challenges
        VsOut o;
    } b[VS_MORE];
    // which further batch a block belongs to (-1: the first batch); wave-uniform
    __device__ __forceinline__ int of_block(uint32_t block) const {
        int k = -1;
        for (int j = 0; j < count; j++)
            if (block >= b[j].first_block) k = j;
        return k;
    }
};
#define VS_SLOTS MI_VS_MAX_TICKETS
#define VS_ACC_DEPTH MI_VS_MAX_OPEN_ACCUMULATIONS  // accumulations (derive + both MSMs) that may be enqueued before the oldest is folded
struct VsSlot {  // one in-flight pairing product (pipelined finalize)
    uint64_t ticket = 0;
    bool busy = false;
    uint64_t fails = 0;
    int count = 1;                                           // products of this ticket
    uint64_t fails_n[MI_VS_MAX_PRODUCTS_PER_TICKET] = {};    // algebraic failures per product
    hipEvent_t done = nullptr;
    hipStream_t stream = nullptr;  // each slot has its own stream: products of different batches run side by side
    Fq12 *h_out = nullptr;  // pinned
};
// The keys of one verifier.  The host copy is the master; every device that is handed work for this verifier gets
// its own resident copy on first use (VsKeys, kept in that device's Context), so one verifier context serves all GPUs
// of the process.
struct VsCtx {
    uint64_t id = 0;
    uint32_t flags = 0;
    size_t n_circuits = 0;
    mi_kzg_vk vk;
    std::vector<mi_vs_circuit_vk> circuits;
    G2Prepared h_prep, tauh_prep;  // prepared once on the host (as the reference prepares prepared_h / prepared_beta_h)
    G2Prepared shift_prep[MI_VS_MAX_BOUNDS];  // v2: the G2 partners of the degree-bound classes (prepared_shift_power)
    VsConsts consts;
    VsWireMap wire_map;
};
struct VsKeys {  // one device's copy
    DevBuf circuits, h_prep, tauh_prep, consts, shift_prep[MI_VS_MAX_BOUNDS], wire_map;
    VsWireMap wm;  // host copy
    size_t n_circuits = 0;
    bool fs_poseidon = false;
    VsConsts shape;  // host copy of what consts holds
    ~VsKeys() {
        circuits.release();
        wire_map.release();
        h_prep.release();
        tauh_prep.release();
        consts.release();
        for (auto &b : shift_prep) b.release();
    }
};
struct VsKeyTable {
    std::map<uint64_t, std::unique_ptr<VsKeys>> of;
};
struct VsAcc {  // an accumulation whose kernels are enqueued but whose host fold is still to come
    bool open = false;
    uint64_t handle = 0, ctx = 0;
    size_t n = 0;
    int count = 1;                                   // batches in this pass
    size_t ns[MI_VS_MAX_BATCHES_PER_PASS] = {};      // their sizes (ns[0] == n)
    size_t nbs[MI_VS_MAX_BATCHES_PER_PASS][MI_VS_MAX_BOUNDS] = {};  // their degree-bound class sizes
    hipStream_t st = nullptr;
    MsmJob j[4];         // H side, tau-H side, class 0, class 1 (pair pass: j[0] carries both batches' two MSMs)
    size_t nb[MI_VS_MAX_BOUNDS] = {0, 0};
    bool fused = false;  // j[0] carries all MSMs
    bool closing = false;
    uint8_t *h_alg = nullptr;
};
// The pipeline of one DEVICE -- accumulation lanes and product slots, each with its stream, workspaces (named per
// lane in the Context) and pinned result buffer.  It belongs to the device, not to a verifier context: lanes index
// shared workspaces, so two verifier contexts must never hold the same lane, and HIP keeps a destroyed stream's queue
// and scratch (measured: 45 MB of device memory per create/destroy cycle when every context made its own 20 streams).
// All streams are created at once, lanes first, then product slots.  Measured on the box: HIP spreads streams over
// at most 24 hardware queues in creation order; with this order 12 lanes + 8 slots ran side by side (4.3e6 proofs/s
// in round 1), created lazily in order of first use (lane, slot, lane, ...) the same streams reached 2.7e6, and with
// more than ~20 of our streams on the device a pairing kernel ends up queued in front of other batches' MSMs
// (10 lanes + 12 slots: half the throughput of 10 + 10).  A lane pass carries two batches under load
// (mi_vs_accumulate_begin_multi) and a ticket up to four products (mi_vs_finalize_async_multi), so the split is 12
// lanes + 8 slots (measured, 400 / 20 batches: 10 + 10 7.4e6 / 3.7e6 proofs/s, 12 + 8 7.8e6 / 3.85e6, 14 + 6 8.0e6 /
// 2.7e6: a burst needs the slots).
// MI_SNARK_VS_LANES (1..14, default 12) / MI_SNARK_VS_SLOTS (1..16, default 8), read once, trim the counts when the
// process needs queues for something else (RCCL, other libraries).
struct VsPipe {
    int lanes = 12, nslots = 8;
    VsAcc acc[VS_ACC_DEPTH];
    hipStream_t acc_st[VS_ACC_DEPTH] = {};  // one stream per accumulation lane
    hipEvent_t acc_ev[VS_ACC_DEPTH] = {};
    VsSlot slots[VS_SLOTS];  // pairing products run on the slots' streams, beside the next batches' accumulation
    uint64_t next_acc = 2, next_ticket = 1;
    // every lane has run one accumulation of at least this footprint (workspaces, pinned buffers, hardware queues exist):
    // the MSM points and the record bytes of the largest pass primed so far.  A later context with a larger shape (or
    // larger batches) primes again, once, instead of growing each lane's ~25 workspaces in mid-stream (every regrow
    // drains the device: ADVICE r2).
    size_t primed_points = 0, primed_bytes = 0;
    VsPipe() {  // the device is current (Context::mod is only called under guarded_on)
        if (const char *e = getenv("MI_SNARK_VS_LANES")) {
            const int x = atoi(e);
            if (x >= 1 && x <= VS_ACC_DEPTH) lanes = x;
        }
        // (Stream priorities do nothing good here: the first 4 or 6 lanes at the highest priority change neither the
        // 400-batch nor the 20-batch rate, all 12 collapse the stream to 4.5e5 proofs/s.)
        for (int p = 0; p < lanes; p++) {
            MI_HIP(hipStreamCreateWithFlags(&acc_st[p], hipStreamNonBlocking));
            MI_HIP(hipEventCreateWithFlags(&acc_ev[p], hipEventDisableTiming));
        }
        if (const char *e = getenv("MI_SNARK_VS_SLOTS")) {  // product slots (1..16): fewer streams when queues are scarce
            const int x = atoi(e);
            if (x >= 1 && x <= VS_SLOTS) nslots = x;
        }
        // (Product streams at the device's highest stream priority were tried -- a product is one workgroup somebody
        // waits for -- and cost a quarter of the throughput: 7.3e6 -> 5.4e6 proofs/s.)
        for (int i = 0; i < nslots; i++) {
            VsSlot &sl = slots[i];
            MI_HIP(hipStreamCreateWithFlags(&sl.stream, hipStreamNonBlocking));
            MI_HIP(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
            MI_HIP(hipHostMalloc((void **)&sl.h_out, sizeof(Fq12) * MI_VS_MAX_PRODUCTS_PER_TICKET, hipHostMallocDefault));
        }
        // HIP creates a stream's hardware queue at its first use (milliseconds each): do that now, in creation order,
        // not inside the first block a node verifies
        void *scratch = nullptr;
        MI_HIP(hipMalloc(&scratch, 256));
        for (int p = 0; p < lanes; p++) MI_HIP(hipMemsetAsync(scratch, 0, 4, acc_st[p]));
        for (int i = 0; i < nslots; i++) MI_HIP(hipMemsetAsync((char *)scratch + 64, 0, 4, slots[i].stream));
        for (int p = 0; p < lanes; p++) MI_HIP(hipStreamSynchronize(acc_st[p]));
        for (int i = 0; i < nslots; i++) MI_HIP(hipStreamSynchronize(slots[i].stream));
        MI_HIP(hipFree(scratch));
    }
    ~VsPipe() {
        for (auto &sl : slots) {
            if (sl.stream) (void)hipStreamSynchronize(sl.stream);
            if (sl.done) (void)hipEventDestroy(sl.done);
            if (sl.stream) (void)hipStreamDestroy(sl.stream);
            if (sl.h_out) (void)hipHostFree(sl.h_out);
        }
        for (int p = 0; p < VS_ACC_DEPTH; p++) {
            if (acc_st[p]) {
                (void)hipStreamSynchronize(acc_st[p]);
                (void)hipStreamDestroy(acc_st[p]);
            }
            if (acc_ev[p]) (void)hipEventDestroy(acc_ev[p]);
        }
    }
    void sync() {
        for (auto &sl : slots)
            if (sl.stream) MI_HIP(hipStreamSynchronize(sl.stream));
        for (int p = 0; p < VS_ACC_DEPTH; p++)
            if (acc_st[p]) MI_HIP(hipStreamSynchronize(acc_st[p]));
    }
};
static VsPipe &vs_pipe(Context &cx) { return cx.mod<VsPipe>("vs.pipe"); }

// ---- verifier contexts: process-wide table of host masters
static std::mutex g_vs_mu;
static std::map<uint64_t, std::shared_ptr<VsCtx>> &vs_ctxs() {
    static std::map<uint64_t, std::shared_ptr<VsCtx>> m;
    return m;
}
static uint64_t g_vs_next = 1;
static std::shared_ptr<VsCtx> vs_find(uint64_t id) {
    std::lock_guard<std::mutex> lk(g_vs_mu);
    auto it = vs_ctxs().find(id);
    return it == vs_ctxs().end() ? nullptr : it->second;
}
bool vs_strides(uint64_t id, size_t *proof_bytes, size_t *rand_bytes) {
    std::shared_ptr<VsCtx> v = vs_find(id);
    if (!v) return false;
    *proof_bytes = v->consts.wire ? (size_t)v->consts.wire_bytes : (size_t)v->consts.proof_words * 4;
    *rand_bytes = v->consts.rand_bytes;
    return true;
}
// records of the context's kind -> wire records, on the host (the inverse of the decode kernels; tests, benches and a
// caller that holds decoded records use it)
int32_t vs_wire_encode_host(uint64_t id, const void *records, size_t n, uint8_t *wire_out) {
    std::shared_ptr<VsCtx> v = vs_find(id);
    if (!v) return MI_ERR_BAD_ARGS;
    const VsWireMap &wm = v->wire_map;
    memset(wire_out, 0, (size_t)wm.wire_bytes * n);
    for (size_t i = 0; i < n; i++) {
        const uint8_t *r = (const uint8_t *)records + (size_t)wm.rec_bytes * i;
        uint8_t *w = wire_out + (size_t)wm.wire_bytes * i;
        uint32_t m1 = 0, d1 = 0;
        if (wm.off_ni) {
            uint32_t ni;
            memcpy(&ni, r + sizeof(mi_vs_proof2), 4);
            if (ni >= 1 && ni <= wm.max_inst) m1 = ni - 1;
        }
        if (wm.off_nc) {
            uint32_t nc;
            memcpy(&nc, r + sizeof(mi_vs_proof3), 4);
            if (nc >= 1 && nc <= wm.max_circ) d1 = nc - 1;
        }
        auto used = [&](uint32_t cls) { return cls == 0 || (cls >= 8 ? cls - 8 < d1 : cls - 2 < m1); };
        for (uint32_t k = 0; k < wm.n_runs; k++) {
            const VsWireMap::Run &run = wm.run[k];
            if (!used(run.cls)) continue;
            if (run.type == 0) memcpy(w + run.wire, r + run.rec, run.count);
            else
                for (uint32_t j = 0; j < run.count; j++) {
                    Fr x;
                    memcpy(&x, r + run.rec + 32 * j, 32);
                    x = from_mont(x);
                    memcpy(w + run.wire + 32 * j, &x, 32);
                }
        }
        for (uint32_t k = 0; k < wm.n_pts; k++) {
            if (!used(wm.pt_cls[k])) continue;
            G1Affine pt;
            memcpy(&pt, r + wm.pt_rec[k], sizeof pt);
            g1_compress_one(pt, w + wm.pt_wire[k]);
        }
    }
    return MI_OK;
}
void verify_release_keys() {
    std::lock_guard<std::mutex> lk(g_vs_mu);
    vs_ctxs().clear();
}
// this device's copy of the keys (uploaded on first use; the device lock is held)
static VsKeys &vs_keys(Context &cx, const VsCtx &v) {
    VsKeyTable &t = cx.mod<VsKeyTable>("vs.keys");
    auto it = t.of.find(v.id);
    if (it != t.of.end()) return *it->second;
    std::unique_ptr<VsKeys> k(new VsKeys());
    k->n_circuits = v.n_circuits;
    k->fs_poseidon = (v.flags & MI_VS_FS_POSEIDON) != 0;
    hipStream_t st = cx.stream;
    MI_HIP(hipMemcpyAsync(k->circuits.get(sizeof(mi_vs_circuit_vk) * v.n_circuits), v.circuits.data(),
                          sizeof(mi_vs_circuit_vk) * v.n_circuits, hipMemcpyHostToDevice, st));
    MI_HIP(hipMemcpyAsync(k->h_prep.get(sizeof(G2Prepared)), &v.h_prep, sizeof(G2Prepared), hipMemcpyHostToDevice, st));
    MI_HIP(hipMemcpyAsync(k->tauh_prep.get(sizeof(G2Prepared)), &v.tauh_prep, sizeof(G2Prepared), hipMemcpyHostToDevice, st));
    MI_HIP(hipMemcpyAsync(k->consts.get(sizeof(VsConsts)), &v.consts, sizeof(VsConsts), hipMemcpyHostToDevice, st));
    k->shape = v.consts;
    k->wm = v.wire_map;
    MI_HIP(hipMemcpyAsync(k->wire_map.get(sizeof(VsWireMap)), &v.wire_map, sizeof(VsWireMap), hipMemcpyHostToDevice, st));
    for (uint32_t c = 0; c < v.consts.n_bounds; c++)
        MI_HIP(hipMemcpyAsync(k->shift_prep[c].get(sizeof(G2Prepared)), &v.shift_prep[c], sizeof(G2Prepared), hipMemcpyHostToDevice, st));
    MI_HIP(hipStreamSynchronize(st));  // sources are the master's members: alive, but later users run on other streams
    (void)vs_pipe(cx);                 // all of the device's streams exist before the first launch (creation order)
    (void)pairing_consts(cx, st);      // the Frobenius table, before any pipelined launch
    VsKeys &ref = *k;
    t.of[v.id] = std::move(k);
    return ref;
}

int32_t vs_ctx_create(Context &cx, const mi_kzg_vk *vk, const mi_vs_key_ext *ext, const mi_vs_shape *shape,
                      const mi_vs_circuit_vk *circuits, size_t n_circuits, uint32_t flags, uint64_t *out) {
    if (shape) {
        if (shape->n_bounds > MI_VS_MAX_BOUNDS || shape->hiding > 1 || (shape->query_of[0] & 1) || (shape->query_of[1] & 1)) return MI_ERR_BAD_ARGS;
        if (shape->lc_kind[0] || shape->lc_kind[1]) return MI_ERR_BAD_ARGS;  // the public-input identity reads p_0, p_1 as they are
        for (int p = 0; p < MI_VS_EVALS; p++)
            if (shape->lc_kind[p] > 7) return MI_ERR_BAD_ARGS;
        if ((shape->hiding || shape->n_bounds) && !ext) return MI_ERR_BAD_ARGS;
        if (shape->max_instances > MI_VS_MAX_INSTANCES) return MI_ERR_BAD_ARGS;
        if (shape->max_circuits > MI_VS_MAX_CIRCUITS || (shape->max_circuits > 1 && shape->max_circuits > shape->max_instances)) return MI_ERR_BAD_ARGS;
    }
    if ((flags & MI_VS_WIRE) && !shape) {}  // v1 records have a wire form too
    std::shared_ptr<VsCtx> v(new VsCtx());
    v->n_circuits = n_circuits;
    v->flags = flags;
    v->vk = *vk;
    v->circuits.assign(circuits, circuits + n_circuits);
    G2Affine qa;
    memcpy(&qa, &vk->h, sizeof qa);
    g2_prepare_host(v->h_prep, qa);
    memcpy(&qa, &vk->tau_h, sizeof qa);
    g2_prepare_host(v->tauh_prep, qa);
    VsConsts &hc = v->consts;
    Fr w = host_omega(4);
    hc.omega16[0] = Fr::one();
    for (int k = 1; k < MI_VS_INPUTS; k++) hc.omega16[k] = mul(hc.omega16[k - 1], w);
    hc.n_inv = inv(from_u32<FrP>(MI_VS_INPUTS));
    static const char TAG[32] = "MI355X-VarunaShaped-v1";
    for (int i = 0; i < 8; i++)
        hc.tag[i] = ((uint32_t)(uint8_t)TAG[4 * i] << 24) | ((uint32_t)(uint8_t)TAG[4 * i + 1] << 16) |
                    ((uint32_t)(uint8_t)TAG[4 * i + 2] << 8) | (uint32_t)(uint8_t)TAG[4 * i + 3];
    memcpy(&hc.g, &vk->g, sizeof hc.g);
    hc.tag_elem = Fq::zero();
    memcpy(hc.tag_elem.l, TAG, 32);
    hc.fs_kind = flags & MI_VS_FS_POSEIDON;
    memset(hc._pad, 0, sizeof hc._pad);
    hc.max_circ = 1;
    hc.need_inv = 0;
    hc.wire = (flags & MI_VS_WIRE) ? 1 : 0;
    hc.wire_bytes = 0;
    memset(hc.lc_kind, 0, sizeof hc.lc_kind);
    hc.gamma_g = G1Affine::inf();
    // ---- the shape: v1 = the fixed query table, nothing hidden, nothing bounded, 2256-byte records, two randomizers
    hc.proof_words = sizeof(mi_vs_proof) / 4;
    hc.rand_bytes = 32;
    hc.hiding = hc.n_bounds = hc.n_bounded = 0;
    hc.class_count[0] = hc.class_count[1] = 0;
    for (int p = 0; p < MI_VS_EVALS; p++) hc.query_of[p] = (uint8_t)vs_query_of(p);
    for (int j = 0; j < MI_VS_PROOF_COMMS; j++) hc.bound_class[j] = -1;
    if (shape) {
        hc.proof_words = sizeof(mi_vs_proof2) / 4;
        hc.rand_bytes = 48;
        hc.hiding = shape->hiding;
        hc.n_bounds = shape->n_bounds;
        memcpy(hc.query_of, shape->query_of, sizeof hc.query_of);
        memcpy(hc.lc_kind, shape->lc_kind, sizeof hc.lc_kind);
        for (int j = 0; j < MI_VS_PROOF_COMMS; j++) {
            const int c = shape->bound_class[j];
            hc.bound_class[j] = (c >= 0 && c < (int)shape->n_bounds) ? (int8_t)c : (int8_t)-1;
            if (hc.bound_class[j] >= 0) {
                hc.n_bounded++;
                hc.class_count[c]++;
            }
        }
        if (shape->hiding) memcpy(&hc.gamma_g, &ext->gamma_g, sizeof hc.gamma_g);
        for (uint32_t c = 0; c < shape->n_bounds; c++) {
            memcpy(&qa, &ext->shift_h[c], sizeof qa);
            g2_prepare_host(v->shift_prep[c], qa);
        }
    }
    hc.max_inst = shape && shape->max_instances > 1 ? shape->max_instances : 1;
    hc.inst_slot = MI_VS_PROOF_COMMS + 2 + hc.n_bounded;
    hc.per_proof = hc.inst_slot + 2 * (hc.max_inst - 1);
    if (hc.max_inst > 1) hc.proof_words = sizeof(mi_vs_proof3) / 4;
    hc.max_circ = shape && shape->max_circuits > 1 ? shape->max_circuits : 1;
    if (hc.max_circ > 1) hc.proof_words = sizeof(mi_vs_proof4) / 4;
    for (int p = 0; p < MI_VS_EVALS; p++)
        if (hc.lc_kind[p] & 4) hc.need_inv = 1;
    // the per-proof fallback sums a proof's chains in one workgroup of at most 128 lanes per proof (vs_sums_kernel)
    if (vs_chains_of(hc.per_proof, hc.hiding, hc.n_bounded, hc.max_circ) > 128) return MI_ERR_BAD_ARGS;
    vs_build_wire_map(v->wire_map, hc, shape != nullptr);
    hc.wire_bytes = v->wire_map.wire_bytes;
    {
        std::lock_guard<std::mutex> lk(g_vs_mu);
        v->id = g_vs_next++;
        vs_ctxs()[v->id] = v;
    }
    (void)vs_keys(cx, *v);  // resident on the calling device now; on the others at their first batch
    *out = v->id;
    return MI_OK;
}
// called once per device (capi.hip walks the device list); the master goes with the primary's call
int32_t vs_ctx_destroy(Context &cx, uint64_t id) {
    std::shared_ptr<VsCtx> v = vs_find(id);
    if (!v) return MI_ERR_BAD_ARGS;
    VsKeyTable &t = cx.mod<VsKeyTable>("vs.keys");
    auto it = t.of.find(id);
    if (it != t.of.end()) {
        MI_HIP(hipStreamSynchronize(cx.stream));
        VsPipe &pp = vs_pipe(cx);
        pp.sync();  // nothing of this verifier's may still be running here
        for (auto &a : pp.acc)
            if (a.open && a.ctx == id && !a.closing) a.open = false;  // abandoned accumulations give their lanes back
        cx.lane_cv.notify_all();
        t.of.erase(it);
    }
    if (cx.index == 0) {
        std::lock_guard<std::mutex> lk(g_vs_mu);
        vs_ctxs().erase(id);
    }
    return MI_OK;
}
// ---------------------------------------------------------------- K: transcript + Fr work, one lane per proof
// Writes, for proof i:
//   pts1[i*11 + k]   = C_k (k < 9), W_beta, W_gamma          sc1[i*11 + k] = r_q xi^k | r_q z_q   (canonical)
//   pts2[i*2 + q]    = W_q                                    sc2[i*2 + q]  = r_q                  (canonical)
//   vk_scal[i*12+k]  = r_q xi^(9+k)  (Montgomery)             g_scal[i] = -(sum_q r_q val_q)       (Montgomery)
//   alg_ok[i]        = public-input consistency v_0 == x_hat(z_beta) + (z_beta^16 - 1) v_1
// Two waves per 64 proofs.  Wave 0 walks the transcript (four chained SHA-256, ~140 us: the serial spine) and then the
// opening-challenge loop that needs xi; wave 1 meanwhile copies the points into MSM order and, as soon as z_beta is
// known (after the second hash), runs the public-input check.  z_beta crosses through LDS at the one barrier.
// (a Fq / Fr field of a proof as delivered: canonical range checks on the raw limbs)
__device__ __forceinline__ bool vs_limbs_below(const uint32_t *__restrict__ v, const uint32_t *mod, int nl) {
    bool lt = false, decided = false;
    for (int k = nl - 1; k >= 0; k--) {
        const uint32_t a = v[k], m = mod[k];
        if (!decided && a != m) {
            lt = a < m;
            decided = true;
        }
    }
    return lt;
}
// y^2 == x^3 + 1, or the (0,0) encoding of infinity; coordinates below q
__device__ __attribute__((noinline)) bool vs_point_ok(const uint32_t *__restrict__ pt) {
    Fq x, y;
    uint32_t any = 0;
#pragma unroll
    for (int k = 0; k < 12; k++) {
        x.l[k] = pt[k];
        y.l[k] = pt[12 + k];
        any |= x.l[k] | y.l[k];
    }
    if (!vs_limbs_below(x.l, FqP::MOD, 12) || !vs_limbs_below(y.l, FqP::MOD, 12)) return false;
    if (!any) return true;
    return sqrq(y) == add(mulq(sqrq(x), x), Fq::one());
}

// The instance / circuit structure a record announces: m1 further instances, d1 further circuits (ids into pcirc), or --
// when a count or an id is out of range -- an invalid proof that keeps its first instance and circuit only (the oracle's
// vs3_record).  pw: the record's words; parts sit 4 words behind the n_circuits header, 100 words each.
#define VS_PART_WORDS (sizeof(mi_vs_circuit_part) / 4)
__device__ __forceinline__ bool vs_record_counts(const uint32_t *__restrict__ pw, const VsConsts *__restrict__ kc, size_t n_circuits,
                                                 uint32_t &m1, uint32_t &d1, uint32_t pcirc[MI_VS_MAX_CIRCUITS - 1]) {
    const uint32_t MI = kc->max_inst, MC = kc->max_circ;
    m1 = d1 = 0;
#pragma unroll
    for (int d = 0; d < MI_VS_MAX_CIRCUITS - 1; d++) pcirc[d] = 0xffffffffu;
    bool count_ok = true;
    uint32_t ni = 1;
    if (MI > 1) {
        ni = pw[sizeof(mi_vs_proof2) / 4];
        count_ok = ni >= 1 && ni <= MI;
    }
    if (MC > 1) {
        const uint32_t nc = pw[sizeof(mi_vs_proof3) / 4];
        const uint32_t *parts = pw + sizeof(mi_vs_proof3) / 4 + 4;
        bool ok = count_ok && nc >= 1 && nc <= MC && nc <= ni && pw[0] < n_circuits;
        uint32_t c_prev = pw[0], f_prev = 0;
        for (uint32_t d = 0; d + 1 < MC; d++)
            if (ok && d + 1 < nc) {
                const uint32_t c = parts[VS_PART_WORDS * d], f = parts[VS_PART_WORDS * d + 1];
                ok = c > c_prev && c < n_circuits && f > f_prev && f < ni;
                c_prev = c;
                f_prev = f;
                pcirc[d] = c;
            }
        count_ok = ok;
        d1 = ok ? nc - 1 : 0;
        if (!ok) {
#pragma unroll
            for (int d = 0; d < MI_VS_MAX_CIRCUITS - 1; d++) pcirc[d] = 0xffffffffu;
        }
    }
    if (count_ok) m1 = ni - 1;
    return count_ok;
}

// FS = 0: the SHA-256 transcript is walked here; FS = 1: the challenges come from `chal` (n x 3 x 128 bits: z_beta,
// z_gamma, xi), written by vs_fs_poseidon_kernel
template <int FS>
__global__ void __launch_bounds__(128) vs_derive_kernel(const void *__restrict__ proofs, size_t n,
                                                        const uint8_t *__restrict__ rand128,
                                                        const mi_vs_circuit_vk *__restrict__ circuits, size_t n_circuits,
                                                        const VsConsts *__restrict__ kc, VsOut o, const uint32_t *__restrict__ chal,
                                                        VsMore more) {
    __shared__ uint32_t zb_lds[64][4];
    __shared__ uint32_t inv_ok_lds[64];
    const int lane = threadIdx.x & 63, role = threadIdx.x >> 6;  // role 0: transcript + scalars, role 1: points + check
    uint32_t block = blockIdx.x;
    const int mb = more.of_block(block);
    if (mb >= 0) {  // block-uniform: this block belongs to a further batch of the pass
        block -= more.b[mb].first_block;
        proofs = more.b[mb].proofs;
        n = more.b[mb].n;
        rand128 = more.b[mb].rand;
        o = more.b[mb].o;
        chal = more.b[mb].chal;
    }
    const size_t i = (size_t)block * 64 + lane;
    const bool live = i < n;
    const uint8_t *const dec_ok = o.dec_ok;  // wire records: the decode kernels' per-proof flags (nullptr otherwise)
    const uint32_t PW = kc->proof_words, PP = kc->per_proof, RB = kc->rand_bytes, NB = kc->n_bounds;
    const bool v2 = PW != sizeof(mi_vs_proof) / 4;
    const uint32_t *pw = reinterpret_cast<const uint32_t *>(proofs) + (live ? i : 0) * (size_t)PW;
    G1Affine *const pts1 = o.pts1, *const pts2 = o.pts2;
    uint32_t *const sc1 = o.sc1, *const sc2 = o.sc2, *const circuit_of = o.circuit_of;
    Fr *const vk_scal = o.vk_scal, *const g_scal = o.g_scal;
    uint8_t *const alg_ok = o.alg_ok;
    const uint32_t *extra = pw + sizeof(mi_vs_proof) / 4;  // v2: random_v[2] (16 words), shifted[9] (216 words)
    uint32_t circuit = pw[0];
    if (circuit >= n_circuits) circuit = 0;  // malformed index: still well-defined, proof is rejected below
    const bool bad_circuit = pw[0] >= n_circuits;
    const uint32_t *digest = reinterpret_cast<const uint32_t *>(circuits[circuit].digest);
    const uint32_t *inputs = pw + 4;                 // 16 x 8 words
    const uint32_t *comms = inputs + 128;            // 9 x 24 words
    const uint32_t *evals = comms + 216;             // 21 x 8 words
    // records with instances (mi_vs_proof3): the count, then MI_VS_EXTRA_INSTANCES x {16 inputs, 2 commitments, 2
    // evaluations} = 192 words each.  A count out of range fails the proof and leaves only its first instance.
    // records of several circuits (mi_vs_proof4): the count, then MI_VS_MAX_CIRCUITS - 1 parts {id, first instance, pad,
    // 12 evaluations} = 100 words each
    const uint32_t MI = kc->max_inst, ISLOT = kc->inst_slot, MC = kc->max_circ;
    const uint32_t *xinst = pw + sizeof(mi_vs_proof2) / 4 + 4;
    const uint32_t *parts = pw + sizeof(mi_vs_proof3) / 4 + 4;
    uint32_t m1, d1, pcirc[MI_VS_MAX_CIRCUITS - 1];
    const bool count_ok = vs_record_counts(pw, kc, n_circuits, m1, d1, pcirc);
    uint32_t h[8], zw[2][4], xw[4];
    bool well_formed = true;
    if (role == 0) {
        if (live && FS == 0) {
            uint32_t tag[8];
#pragma unroll
            for (int k = 0; k < 8; k++) tag[k] = kc->tag[k];
            sha256_prefixed(h, tag, digest, 8, inputs, 128 + 48);
            for (uint32_t d = 0; d < d1; d++)  // every further circuit: its header (id, first instance) and digest
                sha256_prefixed(h, h, parts + VS_PART_WORDS * d, 4, reinterpret_cast<const uint32_t *>(circuits[pcirc[d]].digest), 8);
            for (uint32_t j = 0; j < m1; j++) sha256_prefixed(h, h, xinst + 192 * j, 176, nullptr, 0);  // inputs, A_j, B_j
            sha256_prefixed(h, h, comms + 48, 72, nullptr, 0);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                zw[0][k] = __builtin_bswap32(h[k]);
                zb_lds[lane][k] = zw[0][k];
            }
        }
    } else if (live) {
        // ---- what snarkVM's deserialiser guarantees and raw coordinates do not: points on the curve (or (0,0)),
        // coordinates below q, Fr fields below r (include/mi355x_snark.h, PRECONDITIONS ON POINTS)
        for (int k = 0; k < MI_VS_PROOF_COMMS; k++) well_formed = well_formed && vs_point_ok(comms + 24 * k);
        for (int q = 0; q < 2; q++) well_formed = well_formed && vs_point_ok(evals + 168 + 24 * q);
        for (int k = 0; k < MI_VS_INPUTS; k++) well_formed = well_formed && vs_limbs_below(inputs + 8 * k, FrP::MOD, 8);
        for (int k = 0; k < MI_VS_EVALS; k++) well_formed = well_formed && vs_limbs_below(evals + 8 * k, FrP::MOD, 8);
        if (v2) {
            for (int q = 0; q < 2; q++) well_formed = well_formed && vs_limbs_below(extra + 8 * q, FrP::MOD, 8);
            for (int j = 0; j < MI_VS_PROOF_COMMS; j++) well_formed = well_formed && vs_point_ok(extra + 16 + 24 * j);
        }
        for (uint32_t j = 0; j < m1; j++) {
            const uint32_t *x = xinst + 192 * j;
            for (int k = 0; k < MI_VS_INPUTS; k++) well_formed = well_formed && vs_limbs_below(x + 8 * k, FrP::MOD, 8);
            for (int c = 0; c < 2; c++) well_formed = well_formed && vs_point_ok(x + 128 + 24 * c) && vs_limbs_below(x + 176 + 8 * c, FrP::MOD, 8);
        }
        for (uint32_t d = 0; d < d1; d++)
            for (int k = 0; k < MI_VS_VK_COMMS; k++) well_formed = well_formed && vs_limbs_below(parts + VS_PART_WORDS * d + 4 + 8 * k, FrP::MOD, 8);
        if (dec_ok) well_formed = well_formed && dec_ok[i] != 0;
        for (uint32_t j = 0; j + 1 < MI; j++)  // the further instances' commitments: their H-side slots (infinity when absent)
            for (uint32_t c = 0; c < 2; c++) {
                G1Affine pt = G1Affine::inf();
                if (j < m1) copy_g1(&pt, reinterpret_cast<const G1Affine *>(xinst + 192 * j + 128 + 24 * c));
                copy_g1(pts1 + i * PP + ISLOT + 2 * j + c, &pt);
            }
        circuit_of[i * MC] = circuit;
        for (uint32_t d = 0; d + 1 < MC; d++) circuit_of[i * MC + 1 + d] = pcirc[d];  // 0xffffffff: no such circuit in this proof
        // ---- points into MSM order, randomizers as the tau-H side's scalars
        const G1Affine *pc = reinterpret_cast<const G1Affine *>(comms);
        for (int k = 0; k < MI_VS_PROOF_COMMS; k++) copy_g1(pts1 + i * PP + k, pc + k);
        const G1Affine *wpt = reinterpret_cast<const G1Affine *>(evals + 168);
#pragma unroll
        for (int q = 0; q < 2; q++) {
            copy_g1(pts1 + i * PP + MI_VS_PROOF_COMMS + q, wpt + q);
            copy_g1(pts2 + i * 2 + q, wpt + q);
        }
        if (NB) {  // degree-bounded polynomials: S_j joins the H side, C_j its class's accumulator
            const G1Affine *sh = reinterpret_cast<const G1Affine *>(extra + 16);
            uint32_t b = 0, pos[MI_VS_MAX_BOUNDS] = {0, 0};
            for (int j = 0; j < MI_VS_PROOF_COMMS; j++) {
                const int c = kc->bound_class[j];
                if (c < 0 || c >= (int)NB) continue;  // uniform
                copy_g1(pts1 + i * PP + MI_VS_PROOF_COMMS + 2 + b, sh + j);
                copy_g1(o.ptsb[c] + i * kc->class_count[c] + pos[c], pc + j);
                b++;
                pos[c]++;
            }
        }
        const uint32_t *rw = reinterpret_cast<const uint32_t *>(rand128 + (size_t)RB * i);
        uint32_t *os = sc2 + i * 16;
#pragma unroll
        for (int k = 0; k < 4; k++) { os[k] = rw[k]; os[4 + k] = 0; os[8 + k] = rw[4 + k]; os[12 + k] = 0; }
    }
    __syncthreads();
    bool ok = false;  // role 1: the proof's algebraic verdict
    if (role == 0 && live) {
        if (FS == 0) {
            sha256_prefixed(h, h, comms + 120, 96, nullptr, 0);
#pragma unroll
            for (int k = 0; k < 4; k++) zw[1][k] = __builtin_bswap32(h[k]);
            sha256_prefixed(h, h, evals, 168, nullptr, 0);
            for (uint32_t j = 0; j < m1; j++) sha256_prefixed(h, h, xinst + 192 * j + 176, 16, nullptr, 0);  // a_j, b_j
            for (uint32_t d = 0; d < d1; d++) sha256_prefixed(h, h, parts + VS_PART_WORDS * d + 4, 96, nullptr, 0);  // its 12 evaluations
#pragma unroll
            for (int k = 0; k < 4; k++) xw[k] = __builtin_bswap32(h[k]);
        } else {
            const uint4 c0 = reinterpret_cast<const uint4 *>(chal)[3 * i], c1 = reinterpret_cast<const uint4 *>(chal)[3 * i + 1],
                        c2 = reinterpret_cast<const uint4 *>(chal)[3 * i + 2];
            zw[0][0] = c0.x; zw[0][1] = c0.y; zw[0][2] = c0.z; zw[0][3] = c0.w;
            zw[1][0] = c1.x; zw[1][1] = c1.y; zw[1][2] = c1.z; zw[1][3] = c1.w;
            xw[0] = c2.x; xw[1] = c2.y; xw[2] = c2.z; xw[3] = c2.w;
        }
        Fr z[2] = {fr_from_u128(zw[0]), fr_from_u128(zw[1])};
        Fr xi = fr_from_u128(xw);
        // ---- randomizers
        const uint32_t *rw = reinterpret_cast<const uint32_t *>(rand128 + (size_t)RB * i);
        uint32_t r0[4] = {rw[0], rw[1], rw[2], rw[3]}, r1[4] = {rw[4], rw[5], rw[6], rw[7]};
        Fr r[2] = {fr_from_u128(r0), fr_from_u128(r1)};
        // ---- opening-challenge powers, combined values, MSM scalars (the query of each polynomial: the shape's table)
        Fr xip = Fr::one(), val[2] = {Fr::zero(), Fr::zero()};
        // the linear-combination table's challenge-dependent factors (v2): kap[kind] = (z_beta^16 - 1)^(kind & 1)
        // (z_gamma^16 - 1)^(kind >> 1)
        Fr kap1 = z[0], kap2 = z[1];
#pragma unroll
        for (int k = 0; k < 4; k++) { kap1 = sqr(kap1); kap2 = sqr(kap2); }
        kap1 = sub(kap1, Fr::one());
        kap2 = sub(kap2, Fr::one());
        const Fr kap3 = mul(kap1, kap2);
        // VarunaShaped-v3, kinds with bit 2: times 1 / (z_beta - z_gamma).  Every lane of the wave inverts its own
        // difference in lockstep (Pornin's binary GCD, branch-free): 64 inversions for the time of one; a Montgomery
        // batch across the pass's waves would add a launch and a grid-wide chain to save 0.1 % of the pass's work.
        Fr kinv = Fr::one();
        if (kc->need_inv) {  // uniform
            const Fr dz = sub(z[0], z[1]);
            inv_ok_lds[lane] = dz.is_zero() ? 0u : 1u;  // z_beta == z_gamma: the coefficient does not exist, the proof is invalid
            kinv = invr(dz);
        }
        auto lc_coef = [&](const Fr &power, uint32_t kind) {
            Fr coef = power;
            const uint32_t k3 = kind & 3u;
            if (k3) coef = mul(power, k3 == 1 ? kap1 : (k3 == 2 ? kap2 : kap3));
            if (kind & 4u) coef = mul(coef, kinv);
            return coef;
        };
        for (int k = 0; k < MI_VS_EVALS; k++) {
            const int q = kc->query_of[k] & 1;
            const Fr coef = lc_coef(xip, kc->lc_kind[k] & 7u);  // uniform kind
            Fr s = mul(q ? r[1] : r[0], coef);
            if (k < MI_VS_PROOF_COMMS) fr_store(sc1 + (i * PP + k) * 8, from_mont(s));
            else fr_store(vk_scal + (i * MC) * MI_VS_VK_COMMS + (k - MI_VS_PROOF_COMMS), s);
            const Fr tv = mul(coef, fr_load(evals + 8 * k));
            if (q) val[1] = add(val[1], tv);
            else val[0] = add(val[0], tv);
            xip = mul(xip, xi);
        }
        if (NB) {  // rho_j = r_3 xi^(21 + j): -rho_j S_j on the H side, +rho_j C_j in the class accumulator
            uint32_t r2w[4] = {rw[8], rw[9], rw[10], rw[11]};
            const Fr r3 = fr_from_u128(r2w);
            uint32_t b = 0, pos[MI_VS_MAX_BOUNDS] = {0, 0};
            for (int j = 0; j < MI_VS_PROOF_COMMS; j++) {
                const int c = kc->bound_class[j];
                if (c >= 0 && c < (int)NB) {  // uniform
                    const Fr rho = mul(r3, xip);
                    fr_store(sc1 + (i * PP + MI_VS_PROOF_COMMS + 2 + b) * 8, from_mont(neg(rho)));
                    fr_store(o.scb[c] + (i * kc->class_count[c] + pos[c]) * 8, from_mont(rho));
                    b++;
                    pos[c]++;
                }
                xip = mul(xip, xi);
            }
        }
        if (MI > 1) {  // the further instances' polynomials join the beta query with xi^(30 + 2j + c)
            if (!NB)
                for (int j = 0; j < MI_VS_PROOF_COMMS; j++) xip = mul(xip, xi);  // the powers the degree-bound terms would take
            for (uint32_t j = 0; j + 1 < MI; j++)
                for (uint32_t c = 0; c < 2; c++) {
                    Fr s = Fr::zero();
                    if (j < m1) {
                        s = mul(r[0], xip);
                        val[0] = add(val[0], mul(xip, fr_load(xinst + 192 * j + 176 + 8 * c)));
                        xip = mul(xip, xi);
                    }
                    fr_store(sc1 + (i * PP + ISLOT + 2 * j + c) * 8, from_mont(s));
                }
        }
        if (MC > 1) {  // the further circuits' polynomials: xi^(36 + 12 d + k), the first circuit's query and lc tables
            for (uint32_t j = 0; j < MI_VS_EXTRA_INSTANCES; j++)
                if (j >= m1) xip = mul(mul(xip, xi), xi);  // past the absent instances' powers: xi^36
            for (uint32_t d = 0; d + 1 < MC; d++)
                for (int k = 0; k < MI_VS_VK_COMMS; k++) {
                    const int pp = MI_VS_PROOF_COMMS + k, q = kc->query_of[pp] & 1;
                    Fr s = Fr::zero();
                    if (d < d1) {
                        const Fr coef = lc_coef(xip, kc->lc_kind[pp] & 7u);
                        s = mul(q ? r[1] : r[0], coef);
                        const Fr tv = mul(coef, fr_load(parts + VS_PART_WORDS * d + 4 + 8 * k));
                        if (q) val[1] = add(val[1], tv);
                        else val[0] = add(val[0], tv);
                        xip = mul(xip, xi);
                    }
                    fr_store(vk_scal + (i * MC + 1 + d) * MI_VS_VK_COMMS + k, s);
                }
        }
        Fr gs = Fr::zero(), hs = Fr::zero();
#pragma unroll
        for (int q = 0; q < 2; q++) {
            fr_store(sc1 + (i * PP + MI_VS_PROOF_COMMS + q) * 8, from_mont(mul(r[q], z[q])));
            gs = sub(gs, mul(r[q], val[q]));
            if (kc->hiding) hs = sub(hs, mul(r[q], fr_load(extra + 8 * q)));  // KZGProof.random_v of query q
        }
        fr_store(g_scal + i, gs);
        if (kc->hiding) fr_store(o.h_scal + i, hs);
    }
    if (role == 1 && live) {
    // ---- role 1: public-input consistency, inversion-free.  With d_k = z - w^k and prod_k d_k = z^16 - 1:
    //   v_0 = x_hat(z) + (z^16-1) v_1,  x_hat(z) = (z^16-1)/16 * sum_k in_k w^k / d_k
    //   <=>  16 (v_0 - (z^16-1) v_1) = sum_k in_k w^k prod_{j != k} d_j          (for z^16 != 1)
    // The right-hand side takes prefix/suffix products: 4 * 16 multiplications instead of a field inversion
    // (which is ~380 sequential multiplications per lane).
    uint32_t zbw[4];
    if (FS == 0) {
#pragma unroll
        for (int k = 0; k < 4; k++) zbw[k] = zb_lds[lane][k];
    } else {
        const uint4 c0 = reinterpret_cast<const uint4 *>(chal)[3 * i];
        zbw[0] = c0.x; zbw[1] = c0.y; zbw[2] = c0.z; zbw[3] = c0.w;
    }
    const Fr zb = fr_from_u128(zbw);
    Fr zn = zb;
#pragma unroll
    for (int k = 0; k < 4; k++) zn = sqr(zn);
    Fr znm1 = sub(zn, Fr::one());
    ok = !znm1.is_zero() && !bad_circuit && well_formed && count_ok;  // z^16 = 1 rejects, as in the oracle
    Fr pre[MI_VS_INPUTS], acc = Fr::one();
    for (int k = 0; k < MI_VS_INPUTS; k++) {  // pre[k] = prod_{j < k} d_j
        pre[k] = acc;
        acc = mul(acc, sub(zb, fr_load(&kc->omega16[k])));
    }
    for (uint32_t inst = 0; inst <= m1; inst++) {  // every instance's own identity (one instance: one pass)
        const uint32_t *in_p = inst ? xinst + 192 * (inst - 1) : inputs;
        const uint32_t *v0 = inst ? in_p + 176 : evals, *v1 = inst ? in_p + 184 : evals + 8;
        Fr rhs = Fr::zero(), suf = Fr::one();
        for (int k = MI_VS_INPUTS - 1; k >= 0; k--) {  // suf = prod_{j > k} d_j
            Fr wk = fr_load(&kc->omega16[k]);
            rhs = add(rhs, mul(mul(fr_load(in_p + 8 * k), wk), mul(pre[k], suf)));
            suf = mul(suf, sub(zb, wk));
        }
        Fr lhs = sub(fr_load(v0), mul(znm1, fr_load(v1)));
#pragma unroll
        for (int k = 0; k < 4; k++) lhs = dbl(lhs);  // * 16
        ok = ok && (lhs == rhs);
    }
    }
    if (kc->need_inv) {  // uniform: the scalar wave's "z_beta != z_gamma" reaches the checking wave
        __syncthreads();
        if (role == 1 && live) ok = ok && inv_ok_lds[lane] != 0;
    }
    if (role == 1 && live) alg_ok[i] = ok ? 1 : 0;
}

// ---------------------------------------------------------------- Poseidon Fiat-Shamir transcript (row A9 on the path)
// 37 permutations per proof (include/mi355x_snark.h, MI_VS_FS_POSEIDON), each 39 rounds
// of (round constants, x^17, 3x3 MDS): one lane per proof would walk 37 x 626 dependent Fq products (~30 ms).  Here a
// QUAD of lanes owns a proof -- lane l < 3 holds state element l -- and a round costs six product slots instead of
// fourteen: in a partial round lane 0 walks the S-box chain (x^2, x^4, x^8, x^16, x^17) while lanes 1 and 2 spend the
// first three of those slots on their MDS columns (mds[i][l] s_l, i = 0..2: same instruction, other operands); one
// more slot multiplies the new s_0 into every row; the partial sums cross the quad through LDS.  A full round is the
// S-box on all three lanes and three slots for the rows.  Sixteen proofs per wave, one wave per workgroup.
__device__ __forceinline__ Fq fq_sel(bool c, const Fq &a, const Fq &b) {
    Fq r;
#pragma unroll
    for (int k = 0; k < 12; k++) r.l[k] = c ? a.l[k] : b.l[k];
    return r;
}
__device__ __forceinline__ Fq28 f28_sel(bool c, const Fq28 &a, const Fq28 &b) {
    Fq28 r;
#pragma unroll
    for (int k = 0; k < 14; k++) r.l[k] = c ? a.l[k] : b.l[k];
    return r;
}
// element e of the transcript of a proof with m1 further instances (73 + 38 m1 elements; oracle: fs_element): a run
// of 16 / 32 / 48 bytes of the record taken as Montgomery limbs (zero-extended).  xinst: the further instances (192
// words each: 16 inputs, two commitments, two evaluations); they sit after C_1 (36 elements each) and at the end (2).
// VarunaShaped-v3: d1 further circuits put {header, digest} after the first circuit's digest (2 elements each) and
// their 12 evaluations at the very end; parts: 100 words each, pcirc: their ids.
__device__ __forceinline__ Fq vs_fs_element(const mi_vs_proof *__restrict__ p, const uint32_t *__restrict__ digest,
                                            const VsConsts *__restrict__ kc, const uint32_t *__restrict__ xinst, int m1, int e,
                                            int d1, const uint32_t *__restrict__ parts, const uint32_t *pcirc,
                                            const mi_vs_circuit_vk *__restrict__ circuits) {
    const uint32_t *pw = reinterpret_cast<const uint32_t *>(p);
    const uint32_t *src = kc->tag_elem.l;
    int words = 8;
    if (e >= 2 && e < 2 + 2 * d1) {
        const int d = (e - 2) >> 1;
        if ((e - 2) & 1) { src = reinterpret_cast<const uint32_t *>(circuits[pcirc[d]].digest); words = 8; }
        else { src = parts + VS_PART_WORDS * d; words = 4; }
        e = -1;
    } else if (e >= 2 + 2 * d1) {
        e -= 2 * d1;
        if (e >= 73 + 38 * m1) {  // the further circuits' evaluations
            const int f = e - 73 - 38 * m1;
            src = parts + VS_PART_WORDS * (f / MI_VS_VK_COMMS) + 4 + 8 * (f % MI_VS_VK_COMMS);
            words = 8;
            e = -1;
        }
    }
    if (e < 0) {}
    else if (e == 0) { src = kc->tag_elem.l; words = 8; }
    else if (e == 1) { src = digest; words = 8; }
    else if (e < 34) { src = pw + 4 + 4 * (e - 2); words = 4; }      // halves of the 16 inputs
    else if (e < 38) { src = pw + 132 + 12 * (e - 34); words = 12; }  // coordinates of C_0, C_1
    else if (e < 38 + 36 * m1) {
        const int f = e - 38, j = f / 36, k = f - 36 * j;
        const uint32_t *x = xinst + 192 * j;
        if (k < 32) { src = x + 4 * k; words = 4; }                   // halves of the instance's inputs
        else { src = x + 128 + 12 * (k - 32); words = 12; }           // coordinates of A_j, B_j
    } else if (e < 52 + 36 * m1) { src = pw + 132 + 12 * (e - 36 * m1 - 34); words = 12; }  // coordinates of C_2..C_8
    else if (e < 73 + 36 * m1) { src = pw + 348 + 8 * (e - 36 * m1 - 52); words = 8; }      // evaluations
    else { const int f = e - 73 - 36 * m1; src = xinst + 192 * (f >> 1) + 176 + 8 * (f & 1); words = 8; }  // a_j, b_j
    Fq r;
#pragma unroll
    for (int q = 0; q < 3; q++) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (4 * q < words) v = *reinterpret_cast<const uint4 *>(src + 4 * q);
        r.l[4 * q] = v.x; r.l[4 * q + 1] = v.y; r.l[4 * q + 2] = v.z; r.l[4 * q + 3] = v.w;
    }
    return r;
}
#define VS_FS_BLOCKS 37
#define VS_FS_ELEMS 73
// 64 proofs = 4 waves, one per SIMD.  128 (two waves per SIMD; -DVS_FS_PROOFS_PER_BLOCK=128) was measured: the kernel
// takes 16.3 ms instead of 9.3 and the streamed rate drops from 1.2e6 to 0.8e6 proofs/s -- one wave of this chain
// already keeps its SIMD's multiplier busy.
#ifndef VS_FS_PROOFS_PER_BLOCK
#define VS_FS_PROOFS_PER_BLOCK 64
#endif
__global__ void __launch_bounds__(4 * VS_FS_PROOFS_PER_BLOCK) vs_fs_poseidon_kernel(const void *__restrict__ proofs, size_t n,
                                                             const mi_vs_circuit_vk *__restrict__ circuits, size_t n_circuits,
                                                             const VsConsts *__restrict__ kc, const PoseidonConsts *__restrict__ pc,
                                                             uint32_t *__restrict__ chal, VsMore more) {
    // The permutation runs in the reduced-radix arithmetic of field28.hpp (14 signed 28-bit limbs: one instruction per
    // limb product, a dedicated squaring for the S-box chain, carry-free additions): the state never leaves it between
    // the first absorb and the last squeeze; constants and absorbed elements enter as they are stored.
    // Exchange buffers, TWO sets used alternately (round parity): a round writes its set, one barrier, reads it; the next
    // round's writes go to the other set, whose readers passed the barrier before -- one barrier per round, not two.
    // xs: the state of each proof (full rounds), or lane 0's s_0^16 in slot 0 (partial); xt: partial rounds, lane l's
    // products mds[i][l] s_l (l = 1, 2) and, in [2], lane 3's mds[i][0] s_0.  Dynamic LDS: 2 x 43 KB is more than half
    // a compute unit's LDS, which is also what keeps a CU to one batch (see the launch).
    struct FsExchange {
        Fq28 xs[VS_FS_PROOFS_PER_BLOCK][3];
        Fq28 xt[VS_FS_PROOFS_PER_BLOCK][3][3];
    };
    extern __shared__ uint4 fs_lds[];
    FsExchange *const ex2 = reinterpret_cast<FsExchange *>(fs_lds);
    int par = 0;
    const int t = threadIdx.x, ps = t >> 2, l = t & 3, lc = l < 3 ? l : 0;
    // the pass's batches are laid out in units of 64 proofs (VsMore::first_block, shared with the derive kernel): a
    // block of this kernel takes VS_FS_PROOFS_PER_BLOCK / 64 consecutive units, each 4 waves
    uint32_t block = blockIdx.x * (VS_FS_PROOFS_PER_BLOCK / 64) + (uint32_t)(ps >> 6);
    const int mb = more.of_block(block);
    if (mb >= 0) {  // uniform over the unit's waves: a further batch of the pass
        block -= more.b[mb].first_block;
        proofs = more.b[mb].proofs;
        n = more.b[mb].n;
        chal = const_cast<uint32_t *>(more.b[mb].chal);
    }
    const size_t i = (size_t)block * 64 + (size_t)(ps & 63);
    const mi_vs_proof *p = reinterpret_cast<const mi_vs_proof *>(reinterpret_cast<const uint32_t *>(proofs) + (i < n ? i : 0) * (size_t)kc->proof_words);
    uint32_t circuit = reinterpret_cast<const uint32_t *>(p)[0];
    if (circuit >= n_circuits) circuit = 0;
    const uint32_t *digest = reinterpret_cast<const uint32_t *>(circuits[circuit].digest);
    // this lane's constants: its MDS column (partial rounds), its row (full rounds)
    Fq28 mcol[3], mrow[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        mcol[k] = load28(pc->mds[k][lc]);
        mrow[k] = load28(pc->mds[lc][k]);
    }
    // further instances (mi_vs_proof3): every proof absorbs 38 more elements per instance; the block walks the shape's
    // maximum number of permutations (the barriers are block-wide), a proof stops absorbing and squeezing at its own end
    // (several circuits, mi_vs_proof4: 14 more per further circuit)
    const uint32_t MI = kc->max_inst, MC = kc->max_circ;
    const uint32_t *xinst = reinterpret_cast<const uint32_t *>(p) + sizeof(mi_vs_proof2) / 4 + 4;
    const uint32_t *parts = reinterpret_cast<const uint32_t *>(p) + sizeof(mi_vs_proof3) / 4 + 4;
    uint32_t um1, ud1, pcirc[MI_VS_MAX_CIRCUITS - 1];
    (void)vs_record_counts(reinterpret_cast<const uint32_t *>(p), kc, n_circuits, um1, ud1, pcirc);
    const int m1 = (int)um1, d1 = (int)ud1;
    const int n_el = VS_FS_ELEMS + 38 * m1 + 14 * d1, my_blocks = VS_FS_BLOCKS + 19 * m1 + 7 * d1,
              all_blocks = VS_FS_BLOCKS + 19 * ((int)MI - 1) + 7 * ((int)MC - 1);
    const int b_beta = 21 + 18 * m1 + d1, b_gamma = 25 + 18 * m1 + d1;
    Fq28 s;
#pragma unroll
    for (int k = 0; k < 14; k++) s.l[k] = 0;
    for (int b = 0; b < all_blocks; b++) {
        // ---- absorb: lane 1 takes element 2b, lane 2 element 2b + 1 (the last block holds one)
        const int e = 2 * b + (l == 2 ? 1 : 0);
        if ((l == 1 || l == 2) && e < n_el) s = add28(s, load28(vs_fs_element(p, digest, kc, xinst, m1, e, d1, parts, pcirc, circuits)));
        // ---- permutation
        for (int r = 0; r < MI_POSEIDON_ROUNDS; r++) {
            s = normalise28(add28(s, load28(pc->ark[r][lc])));
            const bool full = r < MI_POSEIDON_FULL_HALF || r >= MI_POSEIDON_FULL_HALF + MI_POSEIDON_PARTIAL;  // uniform
            if (full) {
                const Fq28 x2 = sqr28(s), x4 = sqr28(x2), x8 = sqr28(x4), x16 = sqr28(x8);
                s = mul28(x16, s);
                FsExchange &X = ex2[par];
                par ^= 1;
                if (l < 3) X.xs[ps][l] = s;
                __syncthreads();
                const Fq28 a0 = X.xs[ps][0], a1 = X.xs[ps][1], a2 = X.xs[ps][2];
                s = add28(add28(mul28(mrow[0], a0), mul28(mrow[1], a1)), mul28(mrow[2], a2));
            } else {
                // Lane 0 squares its way to s_0^16; lanes 1, 2 run their MDS columns beside it, and lane 3 -- which
                // mirrors lane 0's state -- the column of s_0 itself: mds[i][0] s_0^17 = (mds[i][0] s_0) s_0^16, so the
                // product x^16 x leaves the chain (four products, the exchange, one product: five in sequence, not six)
                const bool sb = l == 0;
                const Fq28 x2 = mul28(f28_sel(sb, s, mcol[0]), s);
                const Fq28 x4 = mul28(f28_sel(sb, x2, mcol[1]), f28_sel(sb, x2, s));
                const Fq28 x8 = mul28(f28_sel(sb, x4, mcol[2]), f28_sel(sb, x4, s));
                const Fq28 x16 = sqr28(x8);   // lanes 1, 2, 3: an idle slot (result unused)
                FsExchange &X = ex2[par];
                par ^= 1;
                if (l == 0) X.xs[ps][0] = x16;
                else {
                    const int col = l == 3 ? 2 : l - 1;
                    X.xt[ps][col][0] = x2;
                    X.xt[ps][col][1] = x4;
                    X.xt[ps][col][2] = x8;
                }
                __syncthreads();
                s = add28(add28(mul28(X.xs[ps][0], X.xt[ps][2][lc]), X.xt[ps][0][lc]), X.xt[ps][1][lc]);
            }
        }
        // ---- squeeze: the low 128 bits of the canonical value of the first rate element
        if ((b == b_beta || b == b_gamma || b == my_blocks - 1) && l == 1 && i < n) {
            const Fq c = from_mont(store28(s));
            const int which = b == b_beta ? 0 : (b == b_gamma ? 1 : 2);
            reinterpret_cast<uint4 *>(chal)[3 * i + which] = make_uint4(c.l[0], c.l[1], c.l[2], c.l[3]);
        }
    }
}

// one block per merged slot: slot < n_circuits*12 -> sum of vk_scal over the proofs of that circuit; then the sum of
// g_scal (G), then -- with hiding -- the sum of h_scal (gamma_g).  Output: canonical scalar at sc1[n * per_proof + slot],
// point copied alongside.
__global__ void __launch_bounds__(256) vs_merge_kernel(const Fr *__restrict__ vk_scal, const Fr *__restrict__ g_scal,
                                                       const Fr *__restrict__ h_scal, const uint32_t *__restrict__ circuit_of, size_t n,
                                                       const mi_vs_circuit_vk *__restrict__ circuits, size_t n_circuits,
                                                       const VsConsts *__restrict__ kc, G1Affine *__restrict__ pts1,
                                                       uint32_t *__restrict__ sc1, VsMore more) {
    __shared__ uint32_t lds[8 * 256];
    if (blockIdx.y) {  // a further batch of the pass
        const VsMore::Batch &mb = more.b[blockIdx.y - 1];
        vk_scal = mb.o.vk_scal;
        g_scal = mb.o.g_scal;
        h_scal = mb.o.h_scal;
        circuit_of = mb.o.circuit_of;
        n = mb.n;
        pts1 = mb.o.pts1;
        sc1 = mb.o.sc1;
    }
    const size_t slot = blockIdx.x, n_vk_slots = n_circuits * MI_VS_VK_COMMS;
    const int t = threadIdx.x;
    Fr acc = Fr::zero();
    if (slot < n_vk_slots) {
        const uint32_t c = (uint32_t)(slot / MI_VS_VK_COMMS), k = (uint32_t)(slot % MI_VS_VK_COMMS);
        const size_t MC = kc->max_circ;  // a proof's scalars for circuit c sit in the slot of whichever of its circuits that is
        for (size_t e = t; e < n * MC; e += 256)
            if (circuit_of[e] == c) acc = add(acc, fr_load(vk_scal + e * MI_VS_VK_COMMS + k));
    } else {
        const Fr *src = slot == n_vk_slots ? g_scal : h_scal;
        for (size_t i = t; i < n; i += 256) acc = add(acc, fr_load(src + i));
    }
    for (int stride = 128; stride >= 1; stride >>= 1) {
        if (t >= stride && t < 2 * stride)
#pragma unroll
            for (int k = 0; k < 8; k++) lds[k * 256 + (t - stride)] = acc.l[k];
        __syncthreads();
        if (t < stride) {
            Fr o;
#pragma unroll
            for (int k = 0; k < 8; k++) o.l[k] = lds[k * 256 + t];
            acc = add(acc, o);
        }
        __syncthreads();
    }
    if (t == 0) {
        const size_t dst = n * kc->per_proof + slot;
        fr_store(sc1 + dst * 8, from_mont(acc));
        if (slot < n_vk_slots)
            copy_g1(pts1 + dst, reinterpret_cast<const G1Affine *>(circuits[slot / MI_VS_VK_COMMS].comms) + slot % MI_VS_VK_COMMS);
        else
            copy_g1(pts1 + dst, slot == n_vk_slots ? &kc->g : &kc->gamma_g);
    }
}

// ---------------------------------------------------------------- per-proof fallback (names the bad proofs)
// A batch whose aggregate check failed is taken apart on the device, without a host round trip between the steps:
//   1. vs_terms_kernel    every term scalar * point of every proof (glv.hpp: two half-length chains per 253-bit scalar,
//                         signed fixed windows; one lane per chain, ~1.8 k Fq products deep)
//   2. vs_sums_kernel     the proof's accumulators (total_c, -total_w, the classes') from its terms, affine
//   3. vs_check_kernel    group testing: the pairing product of VS_GROUP proofs' summed accumulators (the randomizers
//                         are already in the terms, so a group passes only if all its proofs do), then
//   4. vs_check_kernel    one pairing product per proof, for the proofs of failing groups only (the other blocks
//                         exit at once).
// Terms of a proof, in order (PP = per_proof, H = hiding):
//   [0, PP)            its own H-side points (commitments, witnesses times z, shifted commitments) with sc1
//   [PP, PP + 12 MC)   the commitments of the proof's circuits (MC = the shape's max_circuits; unused slots: infinity) with vk_scal
//   PP + 12 MC         G with g_scal;   the next (if H): gamma_g with h_scal
//   next 2             W_q with r_q (the tau-H side, 128-bit scalars: one chain each)
//   then               the class accumulators' C_j with rho_j, class 0 first
// and its chains: 2 per term (s1 P, s2 phi(P)) except the two 128-bit ones.
#define VS_GLV_W 3
#define VS_GROUP 4
struct VsTermsIn {
    const G1Affine *pts1, *ptsb[MI_VS_MAX_BOUNDS];
    const uint32_t *sc1, *sc2, *scb[MI_VS_MAX_BOUNDS];
    const Fr *vk_scal, *g_scal, *h_scal;
    const uint32_t *circuit_of;
};
__global__ void __launch_bounds__(256) vs_terms_kernel(size_t n, VsTermsIn in, const mi_vs_circuit_vk *__restrict__ circuits,
                                                      const VsConsts *__restrict__ kc, G1Affine *__restrict__ tables,
                                                      G1Xyzz *__restrict__ terms) {
    const uint32_t PP = kc->per_proof, H = kc->hiding, MC = kc->max_circ, CH = vs_chains_of(PP, H, kc->n_bounded, MC);
    const uint32_t NVK = MI_VS_VK_COMMS * MC, n_c = PP + NVK + 1 + H;
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * CH) return;
    const size_t i = t / CH;
    const uint32_t c2 = (uint32_t)(t % CH);
    uint32_t k, half;
    if (c2 < 2 * n_c) { k = c2 >> 1; half = c2 & 1; }
    else if (c2 < 2 * n_c + 2) { k = n_c + (c2 - 2 * n_c); half = 2; }  // 128-bit scalar, one chain
    else { k = n_c + 2 + ((c2 - 2 * n_c - 2) >> 1); half = c2 & 1; }
    G1Affine P;
    Fr v;
    if (k < PP) {
        copy_g1(&P, in.pts1 + i * PP + k);
        v = fr_load(in.sc1 + (i * PP + k) * 8);
    } else if (k < PP + NVK) {
        const uint32_t kk = k - PP, circ = in.circuit_of[i * MC + kk / MI_VS_VK_COMMS];
        P = G1Affine::inf();  // a circuit slot the proof does not use (0xffffffff)
        v = Fr::zero();
        if (circ != 0xffffffffu) {
            copy_g1(&P, reinterpret_cast<const G1Affine *>(circuits[circ].comms) + kk % MI_VS_VK_COMMS);
            v = from_mont(fr_load(in.vk_scal + i * NVK + kk));
        }
    } else if (k == PP + NVK) {
        copy_g1(&P, &kc->g);
        v = from_mont(fr_load(in.g_scal + i));
    } else if (H && k == PP + NVK + 1) {
        copy_g1(&P, &kc->gamma_g);
        v = from_mont(fr_load(in.h_scal + i));
    } else if (k < n_c + 2) {
        const uint32_t q = k - n_c;
        copy_g1(&P, in.pts1 + i * PP + MI_VS_PROOF_COMMS + q);
        v = fr_load(in.sc2 + (i * 2 + q) * 8);
    } else {
        uint32_t e = k - (n_c + 2);
        const uint32_t c = e < kc->class_count[0] ? 0u : 1u;
        if (c) e -= kc->class_count[0];
        copy_g1(&P, in.ptsb[c] + i * kc->class_count[c] + e);
        v = fr_load(in.scb[c] + (i * kc->class_count[c] + e) * 8);
    }
    uint32_t sv[4];
    if (half == 2) {
#pragma unroll
        for (int j = 0; j < 4; j++) sv[j] = v.l[j];
    } else {
        uint32_t s8[8], s1[4], s2[4];
#pragma unroll
        for (int j = 0; j < 8; j++) s8[j] = v.l[j];
        glv_split(s8, s1, s2);
#pragma unroll
        for (int j = 0; j < 4; j++) sv[j] = half ? s2[j] : s1[j];
        if (half) P.x = mulq(P.x, glv_beta());  // phi(P); (0, 0) stays (0, 0)
    }
    G1Xyzz acc = G1Xyzz::inf();
    if (!P.is_inf() && (sv[0] | sv[1] | sv[2] | sv[3])) {
        G1Affine *tab = tables + t * (1u << (VS_GLV_W - 1));
        glv_build_table<VS_GLV_W>(tab, P);
        acc = glv_chain28<VS_GLV_W>(tab, sv);
    }
    store_xyzz(&terms[t], acc);
}

// block b (VS_GROUP proofs, LP = 64 or 128 lanes each: one wave per SIMD at 1024 proofs, so the sums run at a lone
// wave's latency): segmented tree sums of every proof's chains in LDS -> ps[i][4] affine (total_c, -total_w, class 0,
// class 1; unused ones infinity) and the same accumulators summed over the group -> gs[b][4].  The group's sums sit on
// the lanes next to the first proof's segment heads, so that all inversions of a wave are one lockstep call.
template <uint32_t LP>
__global__ void __launch_bounds__(VS_GROUP * LP) vs_sums_kernel(size_t n, const G1Xyzz *__restrict__ terms,
                                                                const VsConsts *__restrict__ kc, G1Affine *__restrict__ ps,
                                                                G1Affine *__restrict__ gs) {
    // sums in the reduced-radix arithmetic of field28.hpp (224 B per point in LDS, plus a flag: infinity)
    extern __shared__ uint4 sums_lds[];  // VS_GROUP * LP G1Xyzz28, then VS_GROUP * LP flags
    G1Xyzz28 *sh = reinterpret_cast<G1Xyzz28 *>(sums_lds);
    uint32_t *shf = reinterpret_cast<uint32_t *>(sh + VS_GROUP * LP);
    const uint32_t pl = threadIdx.x / LP, l = threadIdx.x % LP;
    const size_t first = (size_t)blockIdx.x * VS_GROUP, i = first + pl;
    const uint32_t PP = kc->per_proof, H = kc->hiding, CH = vs_chains_of(PP, H, kc->n_bounded, kc->max_circ);
    const uint32_t n_c = PP + MI_VS_VK_COMMS * kc->max_circ + 1 + H;
    const uint32_t lo1 = 2 * n_c, lo2 = lo1 + 2, lo3 = lo2 + 2 * kc->class_count[0];
    uint32_t seg, lo, hi;
    if (l < lo1) { seg = 0; lo = 0; hi = lo1; }
    else if (l < lo2) { seg = 1; lo = lo1; hi = lo2; }
    else if (l < lo3) { seg = 2; lo = lo2; hi = lo3; }
    else { seg = 3; lo = lo3; hi = CH; }
    const bool live = i < n && l < CH;
    G1Xyzz28 mine;
    bool have = false;
    if (live) {
        const G1Xyzz v = load_xyzz(&terms[i * CH + l]);
        have = !v.is_inf();
        mine = xyzz28_from(v);
    }
    sh[threadIdx.x] = mine;
    shf[threadIdx.x] = have ? 1u : 0u;
    const uint32_t pos = l - lo;
#pragma unroll 1
    for (uint32_t stride = 1; stride < LP; stride <<= 1) {
        __syncthreads();
        const bool act = live && (pos & (2 * stride - 1)) == 0 && l + stride < hi;
        if (act) xyzz_acc28(mine, have, sh[threadIdx.x + stride], shf[threadIdx.x + stride] != 0);
        __syncthreads();
        if (act) {
            sh[threadIdx.x] = mine;
            shf[threadIdx.x] = have ? 1u : 0u;
        }
    }
    __syncthreads();
    if (pl == 0 && live && pos == 0) {  // the group's sum of this accumulator, parked on the neighbouring lane's slot
        G1Xyzz28 g = mine;
        bool hg = have;
        for (uint32_t p = 1; p < VS_GROUP; p++)
            if (first + p < n) xyzz_acc28(g, hg, sh[p * LP + l], shf[p * LP + l] != 0);
        sh[threadIdx.x + 1] = g;
        shf[threadIdx.x + 1] = hg ? 1u : 0u;
    }
    __syncthreads();
    const bool head = live && pos == 0, ghead = pl == 0 && live && pos == 1;
    if (head || ghead) {
        if (ghead) {
            mine = sh[threadIdx.x];
            have = shf[threadIdx.x] != 0;
        }
        G1Affine pa = G1Affine::inf();
        if (have) pa = to_affine(xyzz28_store(mine));
        if (seg == 1) pa = neg(pa);
        copy_g1(head ? &ps[i * MI_MAX_PAIRS + seg] : &gs[(size_t)blockIdx.x * MI_MAX_PAIRS + seg], &pa);
    }
    if (l < MI_MAX_PAIRS && i < n) {  // segments of no chain at all (contexts without that class)
        const uint32_t slo[4] = {0, lo1, lo2, lo3}, shi[4] = {lo1, lo2, lo3, CH};
        if (slo[l] >= shi[l]) {
            G1Affine z = G1Affine::inf();
            copy_g1(&ps[i * MI_MAX_PAIRS + l], &z);
            if (pl == 0) copy_g1(&gs[(size_t)blockIdx.x * MI_MAX_PAIRS + l], &z);
        }
    }
}

// The pairing products of the fallback, workgroup-cooperative (PW_THREADS lanes), over 2 + n_bounds pairs.
struct VsG2Keys {
    const G2Prepared *q[MI_MAX_PAIRS];  // h, tau h, shift_h[0], shift_h[1]
};
// block b: group b's summed accumulators.  A group whose product is one has only valid proofs (verdicts written here);
// otherwise, or when a member failed the algebraic checks (its points may be anything, so the sum says nothing), the
// members that are still undecided go on the list of proofs to check one by one.
__global__ void __launch_bounds__(PW_THREADS) vs_group_check_kernel(size_t n, const G1Affine *__restrict__ gs, VsG2Keys keys,
                                                                    const VsConsts *__restrict__ kc,
                                                                    const PairingConsts *__restrict__ pc,
                                                                    const uint8_t *__restrict__ alg_ok, uint8_t *__restrict__ verdicts,
                                                                    uint32_t *__restrict__ todo, uint32_t *__restrict__ todo_count) {
    extern __shared__ uint4 pw_lds[];  // PW_LDS_BYTES
    LShared &S = *(LShared *)pw_lds;
    __shared__ G1Affine ps[MI_MAX_PAIRS];
    __shared__ const G2Prepared *qs[MI_MAX_PAIRS];
    __shared__ Fq12 result;
    const size_t b = blockIdx.x, first = b * VS_GROUP, last = first + VS_GROUP < n ? first + VS_GROUP : n;
    const int K = 2 + (int)kc->n_bounds, tid = threadIdx.x;
    bool alg = true;
    for (size_t i = first; i < last; i++) alg = alg && alg_ok[i];
    bool ok = false;
    if (alg) {  // uniform
        if (tid < K) {
            copy_g1(&ps[tid], &gs[b * MI_MAX_PAIRS + tid]);
            qs[tid] = keys.q[tid];
        }
        __syncthreads();
        pw_pairing_product(S, ps, qs, K, pc, &result, nullptr, true);
        __syncthreads();
        ok = result.is_one();
    }
    if (tid < (int)(last - first)) {
        const size_t i = first + tid;
        if (ok) verdicts[i] = 1;
        else if (!alg_ok[i]) verdicts[i] = 0;
        else todo[atomicAdd(todo_count, 1u)] = (uint32_t)i;
    }
}
// a fixed grid walks the list: launching one block per proof would spend ~1 us of dispatch per block, most of them to
// find nothing to do
__global__ void __launch_bounds__(PW_THREADS) vs_proof_check_kernel(const G1Affine *__restrict__ sums, VsG2Keys keys,
                                                                    const VsConsts *__restrict__ kc,
                                                                    const PairingConsts *__restrict__ pc, uint8_t *__restrict__ verdicts,
                                                                    const uint32_t *__restrict__ todo,
                                                                    const uint32_t *__restrict__ todo_count) {
    extern __shared__ uint4 pw_lds[];  // PW_LDS_BYTES
    LShared &S = *(LShared *)pw_lds;
    __shared__ G1Affine ps[MI_MAX_PAIRS];
    __shared__ const G2Prepared *qs[MI_MAX_PAIRS];
    __shared__ Fq12 result;
    const int K = 2 + (int)kc->n_bounds, tid = threadIdx.x;
    const uint32_t count = *todo_count;
    for (uint32_t idx = blockIdx.x; idx < count; idx += gridDim.x) {  // uniform
        const size_t i = todo[idx];
        if (tid < K) {
            copy_g1(&ps[tid], &sums[i * MI_MAX_PAIRS + tid]);
            qs[tid] = keys.q[tid];
        }
        __syncthreads();
        pw_pairing_product(S, ps, qs, K, pc, &result, nullptr, true);
        __syncthreads();
        if (tid == 0) verdicts[i] = result.is_one() ? 1 : 0;
        __syncthreads();
    }
}

// ---------------------------------------------------------------- wire records -> records (MI_VS_WIRE)
// The batches of a pass, as the decode kernels see them: batch k's wire records, where its decoded records and flags
// go, and the index of its first proof among all proofs of the pass.
struct VsWireJobs {
    int count = 0;
    struct B { const uint8_t *wire; uint8_t *rec; uint8_t *dec_ok; uint32_t n, first; } b[MI_VS_MAX_BATCHES_PER_PASS];
    __device__ __forceinline__ int of(uint32_t g) const {
        int k = 0;
        for (int j = 1; j < count; j++)
            if (g >= b[j].first) k = j;
        return k;
    }
};
__device__ __forceinline__ void vs_wire_counts(const uint8_t *__restrict__ w, const VsWireMap *__restrict__ wm, uint32_t &m1, uint32_t &d1) {
    m1 = d1 = 0;
    if (wm->off_ni) {
        const uint32_t ni = *reinterpret_cast<const uint32_t *>(w + wm->off_ni);
        if (ni >= 1 && ni <= wm->max_inst) m1 = ni - 1;
    }
    if (wm->off_nc) {
        const uint32_t nc = *reinterpret_cast<const uint32_t *>(w + wm->off_nc);
        if (nc >= 1 && nc <= wm->max_circ) d1 = nc - 1;
    }
}
__device__ __forceinline__ bool vs_wire_used(uint32_t cls, uint32_t m1, uint32_t d1) {
    return cls == 0 || (cls >= 8 ? cls - 8 < d1 : cls - 2 < m1);
}
// a lane per proof: headers as they are, Fr fields canonical -> Montgomery (not below r: the proof's flag is cleared and
// the field is zero).  Every byte of the decoded record is written by this kernel or the points kernel (runs the record
// does not use and the gaps of the map are zeroed here): no memset of the record buffer is needed.
__global__ void __launch_bounds__(64) vs_wire_fields_kernel(VsWireJobs jobs, uint32_t n_all, const VsWireMap *__restrict__ wm) {
    const uint32_t g = blockIdx.x * 64 + threadIdx.x;
    if (g >= n_all) return;
    const VsWireJobs::B &jb = jobs.b[jobs.of(g)];
    const uint32_t i = g - jb.first;
    const uint8_t *w = jb.wire + (size_t)i * wm->wire_bytes;
    uint8_t *r = jb.rec + (size_t)i * wm->rec_bytes;
    uint32_t m1, d1;
    vs_wire_counts(w, wm, m1, d1);
    bool ok = true;
    const uint4 zero4 = make_uint4(0, 0, 0, 0);
    for (uint32_t k = 0; k < wm->n_runs; k++) {
        const VsWireMap::Run run = wm->run[k];
        const bool used = vs_wire_used(run.cls, m1, d1);
        if (run.type == 0) {
            for (uint32_t b = 0; b < run.count; b += 16)
                *reinterpret_cast<uint4 *>(r + run.rec + b) = used ? *reinterpret_cast<const uint4 *>(w + run.wire + b) : zero4;
        } else {
            for (uint32_t j = 0; j < run.count; j++) {
                Fr v = Fr::zero();
                if (used) {
                    const Fr c = fr_load(w + run.wire + 32 * j);
                    if (vs_limbs_below(c.l, FrP::MOD, 8)) v = to_mont(c);
                    else ok = false;
                }
                fr_store(r + run.rec + 32 * j, v);
            }
        }
    }
    for (uint32_t k = 0; k < wm->n_gaps; k++)
        for (uint32_t b = 0; b < wm->gap[k].bytes; b += 16) *reinterpret_cast<uint4 *>(r + wm->gap[k].rec + b) = zero4;
    jb.dec_ok[i] = ok ? 1 : 0;
}
// TWO waves per 64 (proof, point slot of the shape) pairs: 48 bytes -> the affine point (codec28.hpp: one wave takes
// the square root, the other runs the exact subgroup test beside it, both on 28-bit limbs); anything that is not the
// encoding of a subgroup point clears the proof's flag and leaves infinity
__global__ void __launch_bounds__(128) vs_wire_points_kernel(VsWireJobs jobs, uint32_t n_all, const VsWireMap *__restrict__ wm, SqrtDev T) {
    extern __shared__ int32_t c28_lds[];  // C28_LDS_BYTES
    const uint32_t g = blockIdx.x * 64 + (threadIdx.x & 63), NP = wm->n_pts;
    const uint8_t *src = nullptr;
    uint32_t i = 0, slot = 0;
    int k = 0;
    const bool in_range = g < n_all * NP;
    if (in_range) {
        const uint32_t gp = g / NP;
        slot = g - gp * NP;
        k = jobs.of(gp);
        i = gp - jobs.b[k].first;
        const uint8_t *w = jobs.b[k].wire + (size_t)i * wm->wire_bytes;
        uint32_t m1, d1;
        vs_wire_counts(w, wm, m1, d1);
        if (vs_wire_used(wm->pt_cls[slot], m1, d1)) src = w + wm->pt_wire[slot];  // record strides and offsets are multiples of 16
    }
    G1Affine pt;
    const bool ok = g1_decode28(src, true, T, c28_lds, pt);
    if (threadIdx.x < 64 && in_range) {  // the root wave holds the verdict and the point (infinity when there is none)
        const VsWireJobs::B &jb = jobs.b[k];
        copy_g1(reinterpret_cast<G1Affine *>(jb.rec + (size_t)i * wm->rec_bytes + wm->pt_rec[slot]), &pt);
        if (src && !ok) jb.dec_ok[i] = 0;  // only ever cleared here; the fields kernel (earlier on the stream) set it
    }
}

// ---------------------------------------------------------------- host drivers
struct VsWork {
    VsOut o;
    size_t n1, n2, nb[MI_VS_MAX_BOUNDS];  // points of the H side, the tau-H side, the class accumulators
};

static void vs_work_sizes(const VsKeys &v, size_t n, VsWork &w) {
    const VsConsts &hc = v.shape;  // host copy of the device constants
    w.n1 = n * hc.per_proof + v.n_circuits * MI_VS_VK_COMMS + 1 + hc.hiding;
    w.n2 = n * 2;
    for (int c = 0; c < MI_VS_MAX_BOUNDS; c++) w.nb[c] = c < (int)hc.n_bounds ? n * hc.class_count[c] : 0;
}
static size_t vs_work_total(const VsWork &w) { return w.n1 + w.n2 + w.nb[0] + w.nb[1]; }
// May these batches share one lane pass?  A launch sequence carries MSM_MAX_PROBLEMS problems (a batch brings two, plus
// one per degree-bound class) and the small-MSM path must hold all of them.
static bool vs_multi_ok(const VsKeys &v, const size_t *ns, int count) {
    if (count < 1 || count > MI_VS_MAX_BATCHES_PER_PASS) return false;
    if (count == 1) return true;
    size_t total = 0;
    int problems = 0;
    for (int k = 0; k < count; k++) {
        VsWork w;
        vs_work_sizes(v, ns[k], w);
        total += vs_work_total(w);
        problems += 2 + (w.nb[0] ? 1 : 0) + (w.nb[1] ? 1 : 0);
    }
    return problems <= MSM_MAX_PROBLEMS && total <= msm_small_limit();
}
// May these batches be folded into ONE aggregate (a pass that ends with one partial)?  The whole pass is then one batch
// of sum n proofs on the small-MSM path.
static bool vs_merged_ok(const VsKeys &v, const size_t *ns, int count) {
    if (count < 2 || count > MI_VS_MAX_BATCHES_PER_PASS) return false;
    size_t n_all = 0;
    for (int k = 0; k < count; k++) n_all += ns[k];
    VsWork w;
    vs_work_sizes(v, n_all, w);
    return vs_work_total(w) <= msm_small_limit();
}
// the most batches of n proofs that one pass takes
static int vs_max_per_pass(const VsKeys &v, size_t n) {
    size_t ns[MI_VS_MAX_BATCHES_PER_PASS];
    for (int k = 0; k < MI_VS_MAX_BATCHES_PER_PASS; k++) ns[k] = n;
    int best = 1;
    for (int c = 2; c <= MI_VS_MAX_BATCHES_PER_PASS; c++)
        if (vs_multi_ok(v, ns, c)) best = c;
    return best;
}

// derive (+ merge) of the pass's batches, one launch each for all of them: the further batches' operands follow the
// first's in the lane's operand arrays, so all their MSMs are one launch sequence.  ws[k] describes batch k.
// merged: the batches become ONE batch of sum n proofs -- every output of the derive kernel is indexed by the proof's
// number, so batch k writes behind the proofs before it (its VsOut is the first one's with every pointer moved on) and
// everything after the derive kernel (merge, MSMs, fold) sees a single batch, described by ws[0].
static void vs_run_derive(Context &cx, VsKeys &v, const void *const *d_proofs_in, const size_t *ns, const void *const *d_rand, int count,
                          hipStream_t st, int lane, VsWork *ws, bool merged = false) {
    const VsConsts &hc = v.shape;
    // every operand array exists once per lane: several passes are in flight on different streams
    char nm[32];
    auto W = [&](const char *name) {
        snprintf(nm, sizeof nm, "vs%d.%s", lane, name);
        return nm;
    };
    size_t total = 0, cap = 0, n_all = 0;
    for (int k = 0; k < count; k++) {
        vs_work_sizes(v, ns[k], ws[k]);
        total += vs_work_total(ws[k]);
        if (ns[k] > cap) cap = ns[k];
        n_all += ns[k];
    }
    if (merged) {
        vs_work_sizes(v, n_all, ws[0]);
        total = vs_work_total(ws[0]);
        cap = n_all;
    }
    // one array for all MSMs' operands (per batch: H side, tau-H side, then the class accumulators): they run as ONE
    // launch sequence when they fit the small-MSM path
    G1Affine *pts = (G1Affine *)cx.buf(W("pts"), sizeof(G1Affine) * total);
    uint32_t *sc = (uint32_t *)cx.buf(W("sc"), 32 * total);
    size_t at = 0;
    static const char *const suffix[MI_VS_MAX_BATCHES_PER_PASS] = {"", ".b", ".c", ".d", ".e", ".f", ".g", ".h"};
    for (int k = 0; k < (merged ? 1 : count); k++) {
        VsWork &x = ws[k];
        VsOut &o = x.o;
        o.pts1 = pts + at;
        o.sc1 = sc + 8 * at;
        o.pts2 = o.pts1 + x.n1;
        o.sc2 = o.sc1 + 8 * x.n1;
        o.ptsb[0] = o.pts2 + x.n2;
        o.scb[0] = o.sc2 + 8 * x.n2;
        o.ptsb[1] = o.ptsb[0] + x.nb[0];
        o.scb[1] = o.scb[0] + 8 * x.nb[0];
        at += vs_work_total(x);
        char nn[24];
        snprintf(nn, sizeof nn, "vk_scal%s", suffix[k]);
        o.vk_scal = (Fr *)cx.buf(W(nn), sizeof(Fr) * cap * MI_VS_VK_COMMS * hc.max_circ);
        snprintf(nn, sizeof nn, "g_scal%s", suffix[k]);
        o.g_scal = (Fr *)cx.buf(W(nn), sizeof(Fr) * 2 * cap);
        o.h_scal = o.g_scal + cap;
        snprintf(nn, sizeof nn, "alg_ok%s", suffix[k]);
        o.alg_ok = (uint8_t *)cx.buf(W(nn), cap);
        snprintf(nn, sizeof nn, "circuit_of%s", suffix[k]);
        o.circuit_of = (uint32_t *)cx.buf(W(nn), 4 * cap * hc.max_circ);
        o.dec_ok = nullptr;
    }
    if (merged) {
        size_t before = ns[0];  // proofs ahead of batch k
        for (int k = 1; k < count; k++) {
            const VsOut &b = ws[0].o;
            VsOut &o = ws[k].o;
            o.pts1 = b.pts1 + before * hc.per_proof;
            o.sc1 = b.sc1 + 8 * before * hc.per_proof;
            o.pts2 = b.pts2 + 2 * before;
            o.sc2 = b.sc2 + 16 * before;
            for (int c = 0; c < MI_VS_MAX_BOUNDS; c++) {
                o.ptsb[c] = b.ptsb[c] + before * hc.class_count[c];
                o.scb[c] = b.scb[c] + 8 * before * hc.class_count[c];
            }
            o.vk_scal = b.vk_scal + before * MI_VS_VK_COMMS * hc.max_circ;
            o.g_scal = b.g_scal + before;
            o.h_scal = b.h_scal + before;
            o.alg_ok = b.alg_ok + before;
            o.circuit_of = b.circuit_of + before * hc.max_circ;
            o.dec_ok = nullptr;
            before += ns[k];
        }
    }
    // wire records: decoded into the lane's record buffer ahead of everything else (one launch each for all batches)
    const void *recs[MI_VS_MAX_BATCHES_PER_PASS];
    for (int k = 0; k < count; k++) recs[k] = d_proofs_in[k];
    if (hc.wire) {
        const VsWireMap &wm = v.wm;
        uint8_t *exp = (uint8_t *)cx.buf(W("records"), (size_t)wm.rec_bytes * n_all);
        uint8_t *dok = (uint8_t *)cx.buf(W("dec_ok"), n_all);
        VsWireJobs jobs;
        jobs.count = count;
        size_t first = 0;
        for (int k = 0; k < count; k++) {
            jobs.b[k] = VsWireJobs::B{(const uint8_t *)d_proofs_in[k], exp + first * wm.rec_bytes, dok + first, (uint32_t)ns[k], (uint32_t)first};
            recs[k] = jobs.b[k].rec;
            ws[k].o.dec_ok = jobs.b[k].dec_ok;
            first += ns[k];
        }
        cx.prof.begin("vs_wire_decode", st, n_all);
        hipLaunchKernelGGL(vs_wire_fields_kernel, dim3((unsigned)cdiv(n_all, 64)), dim3(64), 0, st, jobs, (uint32_t)n_all, (const VsWireMap *)v.wire_map.p);
        hipLaunchKernelGGL(vs_wire_points_kernel, dim3((unsigned)cdiv(n_all * wm.n_pts, 64)), dim3(128), C28_LDS_BYTES, st, jobs, (uint32_t)n_all,
                           (const VsWireMap *)v.wire_map.p, sqrt_tables(cx, st));
        cx.prof.end("vs_wire_decode", st);
        stats().points_decompressed += n_all * wm.n_pts;
    }
    const void *const *d_proofs = recs;
    VsOut &o = ws[0].o;
    VsMore more;
    more.count = count - 1;
    unsigned derive_blocks = (unsigned)cdiv(ns[0], 64);
    for (int k = 1; k < count; k++) {
        VsMore::Batch &mb = more.b[k - 1];
        mb.proofs = d_proofs[k];
        mb.rand = (const uint8_t *)d_rand[k];
        mb.n = ns[k];
        mb.first_block = derive_blocks;
        mb.chal = nullptr;
        mb.o = ws[k].o;
        derive_blocks += (unsigned)cdiv(ns[k], 64);
    }
    const size_t n = ns[0];
    const uint32_t *chal = nullptr;
    if (v.fs_poseidon) {
        uint32_t *ch = (uint32_t *)cx.buf(W("chal"), 48 * n_all);  // every batch of the pass
        size_t off = n;
        for (int k = 1; k < count; k++) {
            more.b[k - 1].chal = ch + 12 * off;
            off += ns[k];
        }
        const PoseidonConsts *pc = poseidon_consts(cx, st);
        cx.prof.begin("vs_fs_poseidon", st, n_all);
        // 64 proofs (4 waves, one per SIMD) per block, and a block asks for more than half a CU's LDS: the transcript
        // kernels of the batches in flight then never share a CU.  Their loop body (~70 KB of unrolled products) is
        // larger than the instruction cache; waves of different batches on one CU sit at different places in it and
        // fetch against each other (4 concurrent kernels each ran 2.2x slower before).
        // (the kernel's two exchange sets are 2 x 43 KB of dynamic LDS: more than half a CU's by themselves)
        static const size_t fs_pad = 2 * (size_t)VS_FS_PROOFS_PER_BLOCK * 12 * sizeof(Fq28);
        cx.raise_lds_limit((const void *)vs_fs_poseidon_kernel, (int)fs_pad);
        static_assert(VS_FS_PROOFS_PER_BLOCK % 64 == 0, "VsMore::first_block counts units of 64 proofs");
        hipLaunchKernelGGL(vs_fs_poseidon_kernel, dim3(cdiv(derive_blocks, VS_FS_PROOFS_PER_BLOCK / 64)), dim3(4 * VS_FS_PROOFS_PER_BLOCK), fs_pad, st, d_proofs[0], n,
                           (const mi_vs_circuit_vk *)v.circuits.p, v.n_circuits, (const VsConsts *)v.consts.p, pc, ch, more);
        cx.prof.end("vs_fs_poseidon", st);
        chal = ch;
    }
    cx.prof.begin("vs_derive", st, n_all);
    if (v.fs_poseidon)
        hipLaunchKernelGGL(vs_derive_kernel<1>, dim3(derive_blocks), dim3(128), 0, st, d_proofs[0], n, (const uint8_t *)d_rand[0],
                           (const mi_vs_circuit_vk *)v.circuits.p, v.n_circuits, (const VsConsts *)v.consts.p, o, chal, more);
    else
        hipLaunchKernelGGL(vs_derive_kernel<0>, dim3(derive_blocks), dim3(128), 0, st, d_proofs[0], n, (const uint8_t *)d_rand[0],
                           (const mi_vs_circuit_vk *)v.circuits.p, v.n_circuits, (const VsConsts *)v.consts.p, o, chal, more);
    cx.prof.end("vs_derive", st);
    hipLaunchKernelGGL(vs_merge_kernel, dim3((unsigned)(v.n_circuits * MI_VS_VK_COMMS + 1 + hc.hiding), (unsigned)(merged ? 1 : count)), dim3(256), 0, st,
                       o.vk_scal, o.g_scal, o.h_scal, o.circuit_of, merged ? n_all : n, (const mi_vs_circuit_vk *)v.circuits.p, v.n_circuits,
                       (const VsConsts *)v.consts.p, o.pts1, o.sc1, more);
    MI_HIP(hipGetLastError());
}
// one batch
static VsWork vs_run_derive(Context &cx, VsKeys &v, const void *d_proofs, size_t n, const void *d_rand, hipStream_t st, int lane) {
    VsWork w;
    vs_run_derive(cx, v, &d_proofs, &n, &d_rand, 1, st, lane, &w);
    return w;
}

static G1Jac host_jac(const mi_g1_jac &j) {
    G1Jac r;
    memcpy(&r, &j, sizeof r);
    return r;
}
static G1Affine host_jac_to_affine_normalised(const mi_g1_jac &j) {
    G1Jac r = host_jac(j);
    if (r.z.is_zero()) return G1Affine::inf();
    return G1Affine{r.x, r.y};  // library outputs always carry z = 1
}

// handles and tickets name their device and their lane / slot: (sequence << 16) | (device index << 8) | lane
static inline uint64_t vs_make_id(uint64_t seq, int dev_index, int unit) { return (seq << 16) | ((uint64_t)dev_index << 8) | (uint64_t)unit; }
static void vs_release_lane(Context &cx, VsAcc &a) {
    a.open = a.closing = false;
    cx.lane_cv.notify_all();
}

// All MSMs of one accumulation on the lane's stream: the H side (253-bit scalars), the tau-H side (the 128-bit
// randomizers) and, with degree bounds, one accumulator per class.  When everything fits the small-MSM path they share
// ONE launch sequence: each further problem is another set of windows (msm_enqueue's split points).
static bool vs_enqueue_msms(Context &cx, const VsWork &w, hipStream_t st, int lane, MsmJob j[4], int force_c) {
    const size_t total = w.n1 + w.n2 + w.nb[0] + w.nb[1];
    if (total <= msm_small_limit()) {
        const size_t s0 = w.n1, s1 = w.nb[0] ? s0 + w.n2 : 0, s2 = w.nb[1] ? s1 + w.nb[0] : 0;
        msm_enqueue(cx, w.o.pts1, w.o.sc1, total, st, 254, 8 + 4 * lane, &j[0], force_c, lane, MsmSplits(s0, s1, s2));
        return true;
    }
    msm_enqueue(cx, w.o.pts2, w.o.sc2, w.n2, st, 128, 8 + 4 * lane + 1, &j[1], 0, lane);
    for (int c = 0; c < MI_VS_MAX_BOUNDS; c++)
        if (w.nb[c]) msm_enqueue(cx, w.o.ptsb[c], w.o.scb[c], w.nb[c], st, 254, 8 + 4 * lane + 2 + c, &j[2 + c], 0, lane);
    msm_enqueue(cx, w.o.pts1, w.o.sc1, w.n1, st, 254, 8 + 4 * lane, &j[0], 0, lane);
    return false;
}

// all batches of a pass: per batch (H side, tau-H side, class accumulators) -- every one a problem of one launch sequence
static void vs_enqueue_msms_multi(Context &cx, const VsWork *ws, int count, hipStream_t st, int lane, MsmJob j[4], int force_c) {
    MsmSplits sp;
    int np = 0;
    size_t at = 0;
    auto problem = [&](size_t len) {
        if (np > 0) {
            if (np - 1 >= MSM_MAX_PROBLEMS - 1) throw HipError{hipErrorInvalidValue, "pass carries too many MSMs", __LINE__};
            sp.at[np - 1] = at;
        }
        at += len;
        np++;
    };
    for (int k = 0; k < count; k++) {
        if (ws[0].o.pts1 + at != ws[k].o.pts1) throw HipError{hipErrorInvalidValue, "pass operand layout", __LINE__};
        problem(ws[k].n1);
        problem(ws[k].n2);
        for (int c = 0; c < MI_VS_MAX_BOUNDS; c++)
            if (ws[k].nb[c]) problem(ws[k].nb[c]);
    }
    msm_enqueue(cx, ws[0].o.pts1, ws[0].o.sc1, at, st, 254, 8 + 4 * lane, &j[0], force_c, lane, sp);
}

// accumulate one shard: partial sums + algebraic failures (+ optional per-proof alg flags to host)
// accumulate = begin (enqueue derive + both MSMs, no host synchronisation) + end (wait, fold on the host).
// VS_ACC_DEPTH begin/end pairs may be open at once per device: every lane has its own stream, device workspaces and
// pinned result buffers, so the next batches' kernels run while batch k is folded.  When every lane is taken the call
// WAITS for one (the device lock is released meanwhile): a synchronous verify beside a saturated stream is late, not
// refused.
// msm = false: derive only (the caller wants the lane's workspaces for the per-proof fallback, not the aggregate)
// more: further batches for the same pass (vs_multi_ok shapes only): each has its own operands, accumulators and partial
struct VsBatches {
    int count = 1;
    bool merged = false;  // fold the batches into ONE aggregate: the pass ends with one partial over all their proofs
    const void *proofs[MI_VS_MAX_BATCHES_PER_PASS] = {};
    const void *rand[MI_VS_MAX_BATCHES_PER_PASS] = {};
    size_t n[MI_VS_MAX_BATCHES_PER_PASS] = {};
};
static int32_t vs_accumulate_begin_impl(Context &cx, const VsCtx &vc, const void *d_proofs, size_t n, const void *d_rand,
                                        hipStream_t st, uint64_t *handle, VsWork *work_out, bool msm = true,
                                        const VsBatches *all = nullptr) {
    VsKeys &v = vs_keys(cx, vc);
    VsPipe &pp = vs_pipe(cx);
    VsBatches one;
    one.proofs[0] = d_proofs;
    one.rand[0] = d_rand;
    one.n[0] = n;
    const VsBatches &bt = all ? *all : one;
    const bool multi = bt.count > 1, merged = multi && bt.merged;
    if (multi && (!msm || !(merged ? vs_merged_ok(v, bt.n, bt.count) : vs_multi_ok(v, bt.n, bt.count)))) return MI_ERR_BAD_ARGS;
    int par = -1;
    for (;;) {
        for (int p = 0; p < pp.lanes; p++)
            if (!pp.acc[p].open) {
                par = p;
                break;
            }
        if (par >= 0) break;
        if (!tl_ctx_lock) return MI_ERR_AGAIN;
        cx.lane_cv.wait(*tl_ctx_lock);  // ends (any thread) and context destruction signal it
    }
    const int per_pass_prime = vs_max_per_pass(v, n);
    size_t need_points = 0, need_bytes = 0;
    {
        VsWork wq;
        vs_work_sizes(v, n, wq);
        need_points = vs_work_total(wq) * (size_t)per_pass_prime;
        need_bytes = (size_t)v.shape.proof_words * 4 * n * (size_t)per_pass_prime;
        {  // what priming covers: the largest pass of either form for batches of n proofs
            size_t ns4[MI_VS_MAX_BATCHES_PER_PASS];
            for (int k = 0; k < MI_VS_MAX_BATCHES_PER_PASS; k++) ns4[k] = n;
            for (int c = 2; c <= MI_VS_MAX_BATCHES_PER_PASS; c++)
                if (vs_merged_ok(v, ns4, c)) {
                    vs_work_sizes(v, n * (size_t)c, wq);
                    if (vs_work_total(wq) > need_points) need_points = vs_work_total(wq);
                    if ((size_t)v.shape.proof_words * 4 * n * (size_t)c > need_bytes) need_bytes = (size_t)v.shape.proof_words * 4 * n * (size_t)c;
                }
        }
        if (multi) {
            size_t tot = 0, nb = 0, n_sum = 0;
            for (int k = 0; k < bt.count; k++) {
                vs_work_sizes(v, bt.n[k], wq);
                tot += vs_work_total(wq);
                nb += (size_t)v.shape.proof_words * 4 * bt.n[k];
                n_sum += bt.n[k];
            }
            if (merged) {  // one batch of the sum: the merged slots (vk, G) once
                vs_work_sizes(v, n_sum, wq);
                tot = vs_work_total(wq);
            }
            if (tot > need_points) need_points = tot;
            if (nb > need_bytes) need_bytes = nb;
        }
    }
    if (need_points > pp.primed_points || need_bytes > pp.primed_bytes) {
        // The first batch a device sees runs once on EVERY lane (results discarded): each lane's ~25 workspaces, its
        // pinned result buffers and its events then exist, sized for this batch shape, before the pipeline fills --
        // otherwise the first dozen batches of a stream each pay the allocations of the lane they land on.
        if (need_points > pp.primed_points) pp.primed_points = need_points;
        if (need_bytes > pp.primed_bytes) pp.primed_bytes = need_bytes;
        HostSpan hs("begin.prime");
        for (int p = 0; p < pp.lanes; p++) {
            if (pp.acc[p].open) continue;  // the lane this call takes is primed too: its real pass follows on the same stream
            hipStream_t sp = pp.acc_st[p];
            MI_HIP(hipEventRecord(pp.acc_ev[p], st));
            MI_HIP(hipStreamWaitEvent(sp, pp.acc_ev[p], 0));
            // a shape that may run passes of several batches is primed with the largest (the batch repeated): the
            // larger workspaces exist before a stream under load starts sharing passes
            const int per_pass = per_pass_prime;
            VsBatches rep;
            rep.count = per_pass;
            for (int k = 0; k < per_pass; k++) {
                rep.proofs[k] = d_proofs;
                rep.rand[k] = d_rand;
                rep.n[k] = n;
            }
            VsWork ws[MI_VS_MAX_BATCHES_PER_PASS];
            vs_run_derive(cx, v, rep.proofs, rep.n, rep.rand, per_pass, sp, p, ws);
            char nm[32];
            snprintf(nm, sizeof nm, "vs.alg_pin%d", p);
            (void)cx.pin_named(nm, MI_VS_MAX_BATCHES_PER_PASS * n);
            MsmJob jj[4];
            if (per_pass > 1) vs_enqueue_msms_multi(cx, ws, per_pass, sp, p, jj, 0);
            else (void)vs_enqueue_msms(cx, ws[0], sp, p, jj, 0);
            // the merged form, with as many batches as IT takes (its problems do not multiply with the batches): its
            // per-proof arrays are longer
            int merged_count = 0;
            for (int k = 0; k < MI_VS_MAX_BATCHES_PER_PASS; k++) {
                rep.proofs[k] = d_proofs;
                rep.rand[k] = d_rand;
                rep.n[k] = n;
                if (k >= 1 && vs_merged_ok(v, rep.n, k + 1)) merged_count = k + 1;
            }
            if (merged_count) {
                vs_run_derive(cx, v, rep.proofs, rep.n, rep.rand, merged_count, sp, p, ws, true);
                (void)vs_enqueue_msms(cx, ws[0], sp, p, jj, 0);
            }
        }
        for (int p = 0; p < pp.lanes; p++)
            if (!pp.acc[p].open) MI_HIP(hipStreamSynchronize(pp.acc_st[p]));
    }
    const uint64_t hd = vs_make_id(pp.next_acc++, cx.index, par);
    VsAcc &a = pp.acc[par];
    // Each lane has its own stream, so the next batches' derive and MSMs run beside this one's.  The inputs were
    // produced on the caller's stream: order ours behind it.
    hipStream_t sa = pp.acc_st[par];
    {
        HostSpan hs("begin.order");
        MI_HIP(hipEventRecord(pp.acc_ev[par], st));
        MI_HIP(hipStreamWaitEvent(sa, pp.acc_ev[par], 0));
    }
    VsWork ws[MI_VS_MAX_BATCHES_PER_PASS];
    {
        HostSpan hs("begin.derive");
        vs_run_derive(cx, v, bt.proofs, bt.n, bt.rand, bt.count, sa, par, ws, merged);
    }
    const VsWork &w = ws[0];
    size_t n_max = 0, n_all = 0;
    for (int k = 0; k < bt.count; k++) {
        if (bt.n[k] > n_max) n_max = bt.n[k];
        n_all += bt.n[k];
    }
    char nm[32];
    snprintf(nm, sizeof nm, "vs.alg_pin%d", par);
    const size_t pin_want = MI_VS_MAX_BATCHES_PER_PASS * n_max;  // the same size whatever the pass carries: no regrowth in a stream
    a.h_alg = (uint8_t *)cx.pin_named(nm, pin_want > n_all ? pin_want : n_all);
    if (msm) {
        {
            HostSpan hs("begin.alg_copy");
            size_t off = 0;
            if (merged) MI_HIP(hipMemcpyAsync(a.h_alg, ws[0].o.alg_ok, n_all, hipMemcpyDeviceToHost, sa));  // one array
            for (int k = 0; k < bt.count && !merged; k++) {  // ahead of the MSMs: their done-event covers the copies
                MI_HIP(hipMemcpyAsync(a.h_alg + off, ws[k].o.alg_ok, bt.n[k], hipMemcpyDeviceToHost, sa));
                off += bt.n[k];
            }
        }
        HostSpan hs_msm("begin.msm_enqueue");
        static const int vs_c = getenv("MI_SNARK_VS_C") ? atoi(getenv("MI_SNARK_VS_C")) : 0;  // window width override (experiments)
        if (multi && !merged) {
            vs_enqueue_msms_multi(cx, ws, bt.count, sa, par, a.j, vs_c);
            a.fused = true;
        } else {
            a.fused = vs_enqueue_msms(cx, w, sa, par, a.j, vs_c);
        }
    }
    a.nb[0] = w.nb[0];
    a.nb[1] = w.nb[1];
    a.n = merged ? n_all : n;
    a.count = merged ? 1 : bt.count;  // a merged pass ends like a single batch: one partial
    for (int k = 0; k < bt.count; k++) {
        a.ns[k] = bt.n[k];
        a.nbs[k][0] = ws[k].nb[0];
        a.nbs[k][1] = ws[k].nb[1];
    }
    if (merged) a.ns[0] = n_all;
    a.st = sa;
    a.handle = hd;
    a.ctx = vc.id;
    a.open = true;
    if (work_out) *work_out = ws[0];
    *handle = hd;
    return MI_OK;
}
// keep_lane: the caller goes on using the lane's workspaces (per-proof fallback) and releases it itself
// count: the batches the caller expects the pass to hold (out[0 .. count)); a pass has to be ended with its own count
static int32_t vs_accumulate_end_impl(Context &cx, uint64_t handle, mi_vs_partial *out, bool keep_lane = false, int count = 1) {
    if ((handle & 0xff) >= VS_ACC_DEPTH || (int)((handle >> 8) & 0xff) != cx.index) return MI_ERR_BAD_ARGS;
    VsAcc &a = vs_pipe(cx).acc[handle & 0xff];
    if (!a.open || a.handle != handle || a.closing) return MI_ERR_BAD_ARGS;
    if (a.count != count) return MI_ERR_BAD_ARGS;
    a.closing = true;  // the lane stays taken until the fold is done; the device lock is not held meanwhile
    // host folds (~250 doublings + ~250 additions for 254-bit scalars, half that for the randomizer side), each
    // split over two pool threads
    uint64_t fails = 0, fails_more[MI_VS_MAX_BATCHES_PER_PASS] = {};
    try {
        Unlocked unlocked;
        const G1Jac inf = G1Jac{Fq::one(), Fq::one(), Fq::zero()};
        memcpy(&out->bound[0], &inf, sizeof inf);
        memcpy(&out->bound[1], &inf, sizeof inf);
        if (count > 1) {
            mi_g1_jac *outs[MSM_MAX_PROBLEMS] = {};
            int np = 0;
            size_t off = 0;
            for (int k = 0; k < count; k++) {
                memcpy(&out[k].bound[0], &inf, sizeof inf);
                memcpy(&out[k].bound[1], &inf, sizeof inf);
                outs[np++] = &out[k].total_c;
                outs[np++] = &out[k].total_w;
                for (int c = 0; c < MI_VS_MAX_BOUNDS; c++)
                    if (a.nbs[k][c]) outs[np++] = &out[k].bound[c];
            }
            msm_finish_all(cx, &a.j[0], a.st, outs, np);  // waits for the pass's event: the flags below are there after it
            for (int k = 0; k < count; k++) {
                for (size_t i = 0; i < a.ns[k]; i++) fails_more[k] += a.h_alg[off + i] ? 0 : 1;
                off += a.ns[k];
            }
        } else if (a.fused) {
            msm_finish(cx, &a.j[0], a.st, &out->total_c, &out->total_w, a.nb[0] ? &out->bound[0] : nullptr,
                       a.nb[1] ? &out->bound[a.nb[0] ? 1 : 0] : nullptr);
            if (a.nb[1] && !a.nb[0]) {  // only class 1 in use: it was the third problem
                out->bound[1] = out->bound[0];
                memcpy(&out->bound[0], &inf, sizeof inf);
            }
        } else {
            msm_finish(cx, &a.j[1], a.st, &out->total_w);
            for (int c = 0; c < MI_VS_MAX_BOUNDS; c++)
                if (a.nb[c]) msm_finish(cx, &a.j[2 + c], a.st, &out->bound[c]);
            msm_finish(cx, &a.j[0], a.st, &out->total_c);
        }
        if (count == 1)
            for (size_t i = 0; i < a.n; i++) fails += a.h_alg[i] ? 0 : 1;
    } catch (...) {
        vs_release_lane(cx, a);
        throw;
    }
    out->n_proofs = a.n;
    out->n_alg_fail = fails;
    if (count > 1)
        for (int k = 0; k < count; k++) {
            out[k].n_proofs = a.ns[k];
            out[k].n_alg_fail = fails_more[k];
        }
    if (keep_lane) a.closing = false;  // still open: nobody else can take it
    else vs_release_lane(cx, a);
    return MI_OK;
}
int32_t vs_accumulate_begin(Context &cx, uint64_t id, const void *d_proofs, size_t n, const void *d_rand, hipStream_t st,
                            uint64_t *handle) {
    std::shared_ptr<VsCtx> v = vs_find(id);
    if (!v) return MI_ERR_BAD_ARGS;
    return vs_accumulate_begin_impl(cx, *v, d_proofs, n, d_rand, st, handle, nullptr);
}
int32_t vs_accumulate_end(Context &cx, uint64_t id, uint64_t handle, mi_vs_partial *out) {
    if (!vs_find(id)) return MI_ERR_BAD_ARGS;
    return vs_accumulate_end_impl(cx, handle, out);
}
int32_t vs_accumulate_begin_multi(Context &cx, uint64_t id, int count, const void *const *d_proofs, const size_t *n,
                                  const void *const *d_rand, hipStream_t st, uint64_t *handle) {
    std::shared_ptr<VsCtx> v = vs_find(id);
    if (!v || count < 1 || count > MI_VS_MAX_BATCHES_PER_PASS) return MI_ERR_BAD_ARGS;
    VsBatches bt;
    bt.count = count;
    for (int k = 0; k < count; k++) {
        bt.proofs[k] = d_proofs[k];
        bt.rand[k] = d_rand[k];
        bt.n[k] = n[k];
    }
    return vs_accumulate_begin_impl(cx, *v, d_proofs[0], n[0], d_rand[0], st, handle, nullptr, true, &bt);
}
int32_t vs_accumulate_begin_merged(Context &cx, uint64_t id, int count, const void *const *d_proofs, const size_t *n,
                                   const void *const *d_rand, hipStream_t st, uint64_t *handle) {
    std::shared_ptr<VsCtx> v = vs_find(id);
    if (!v || count < 2 || count > MI_VS_MAX_BATCHES_PER_PASS) return MI_ERR_BAD_ARGS;
    VsBatches bt;
    bt.count = count;
    bt.merged = true;
    for (int k = 0; k < count; k++) {
        bt.proofs[k] = d_proofs[k];
        bt.rand[k] = d_rand[k];
        bt.n[k] = n[k];
    }
    return vs_accumulate_begin_impl(cx, *v, d_proofs[0], n[0], d_rand[0], st, handle, nullptr, true, &bt);
}
int32_t vs_accumulate_end_multi(Context &cx, uint64_t id, uint64_t handle, mi_vs_partial *outs, int count) {
    if (!vs_find(id) || count < 1 || count > MI_VS_MAX_BATCHES_PER_PASS) return MI_ERR_BAD_ARGS;
    return vs_accumulate_end_impl(cx, handle, outs, false, count);
}
int32_t vs_batches_per_pass(Context &cx, uint64_t id, size_t n) {
    std::shared_ptr<VsCtx> v = vs_find(id);
    if (!v || !n) return MI_ERR_BAD_ARGS;
    return vs_max_per_pass(vs_keys(cx, *v), n);
}
// the most batches of n proofs that one MERGED pass takes (its MSM problems do not multiply with the batches: shapes
// with degree-bound classes merge more batches than they can carry side by side)
int32_t vs_batches_per_merged_pass(Context &cx, uint64_t id, size_t n) {
    std::shared_ptr<VsCtx> v = vs_find(id);
    if (!v || !n) return MI_ERR_BAD_ARGS;
    size_t ns[MI_VS_MAX_BATCHES_PER_PASS];
    for (int k = 0; k < MI_VS_MAX_BATCHES_PER_PASS; k++) ns[k] = n;
    int best = 1;
    for (int c = 2; c <= MI_VS_MAX_BATCHES_PER_PASS; c++)
        if (vs_merged_ok(vs_keys(cx, *v), ns, c)) best = c;
    return best;
}
int32_t vs_accumulate_device(Context &cx, uint64_t id, const void *d_proofs, size_t n, const void *d_rand,
                             mi_vs_partial *out, hipStream_t st) {
    std::shared_ptr<VsCtx> v = vs_find(id);
    if (!v) return MI_ERR_BAD_ARGS;
    uint64_t hd;
    int32_t s = vs_accumulate_begin_impl(cx, *v, d_proofs, n, d_rand, st, &hd, nullptr);
    if (s != MI_OK) return s;
    return vs_accumulate_end_impl(cx, hd, out);
}

// sum the shards' partials on the host: the points of the pairing check (total_c, -total_w, the class accumulators) and
// the count of algebraic failures
static void vs_sum_partials(const mi_vs_partial *parts, size_t np, G1Affine ps[MI_MAX_PAIRS], uint64_t *fails) {
    if (np == 1) {  // one rank: the partial's points are the pairing's (normalised already: no addition, no inversion --
                    // the general path below spent two host inversions per product on the finisher's one thread)
        ps[0] = host_jac_to_affine_normalised(parts[0].total_c);
        ps[1] = neg(host_jac_to_affine_normalised(parts[0].total_w));
        for (int c = 0; c < MI_VS_MAX_BOUNDS; c++) ps[2 + c] = host_jac_to_affine_normalised(parts[0].bound[c]);
        *fails = parts[0].n_alg_fail;
        return;
    }
    G1Xyzz t[MI_MAX_PAIRS] = {G1Xyzz::inf(), G1Xyzz::inf(), G1Xyzz::inf(), G1Xyzz::inf()};
    uint64_t f = 0;
    for (size_t k = 0; k < np; k++) {
        t[0] = add_mixed(t[0], host_jac_to_affine_normalised(parts[k].total_c));
        t[1] = add_mixed(t[1], host_jac_to_affine_normalised(parts[k].total_w));
        for (int c = 0; c < MI_VS_MAX_BOUNDS; c++) t[2 + c] = add_mixed(t[2 + c], host_jac_to_affine_normalised(parts[k].bound[c]));
        f += parts[k].n_alg_fail;
    }
    for (int k = 0; k < MI_MAX_PAIRS; k++) ps[k] = to_affine(t[k]);
    ps[1] = neg(ps[1]);
    *fails = f;
}
// the G2 side: h, tau h and the degree-bound classes' partners; returns the number of pairs
static int vs_g2_keys(VsKeys &v, const G2Prepared *qs[MI_MAX_PAIRS]) {
    qs[0] = (const G2Prepared *)v.h_prep.p;
    qs[1] = (const G2Prepared *)v.tauh_prep.p;
    for (uint32_t c = 0; c < v.shape.n_bounds; c++) qs[2 + c] = (const G2Prepared *)v.shift_prep[c].p;
    for (uint32_t c = v.shape.n_bounds; c < MI_VS_MAX_BOUNDS; c++) qs[2 + c] = nullptr;
    return 2 + (int)v.shape.n_bounds;
}

int32_t vs_finalize(Context &cx, uint64_t id, const mi_vs_partial *parts, size_t np, uint8_t *all_ok, hipStream_t st) {
    std::shared_ptr<VsCtx> vc = vs_find(id);
    if (!vc) return MI_ERR_BAD_ARGS;
    VsKeys &v = vs_keys(cx, *vc);
    G1Affine ps[MI_MAX_PAIRS];
    uint64_t fails = 0;
    vs_sum_partials(parts, np, ps, &fails);
    const G2Prepared *qs[MI_MAX_PAIRS];
    const int K = vs_g2_keys(v, qs);
    Fq12 f;
    pairing_product_prepared(cx, ps, qs, K, &f, st, true);
    stats().pairing_products++;
    stats().pairing_pairs += (uint64_t)K;
    *all_ok = (f.is_one() && fails == 0) ? 1 : 0;
    return MI_OK;
}

// pipelined finalize: sum the partials on the host, enqueue the pairing product on one of the device's product
// streams and return a ticket; the caller goes on to accumulate the next batch while the product runs
static int32_t vs_slot_wait(VsSlot &sl, uint8_t *all_ok, int count = 1) {
    if (count != sl.count) return MI_ERR_BAD_ARGS;
    {
        Unlocked unlocked;  // other threads keep enqueueing and folding while this one waits for the product
        MI_HIP(hipEventSynchronize(sl.done));
    }
    if (all_ok)
        for (int k = 0; k < count; k++) all_ok[k] = (sl.h_out[k].is_one() && sl.fails_n[k] == 0) ? 1 : 0;
    sl.busy = false;
    return MI_OK;
}
// count products in one ticket: product k sums parts[k * np .. (k + 1) * np) (np ranks each)
int32_t vs_finalize_async_multi(Context &cx, uint64_t id, const mi_vs_partial *parts, size_t np, int count, uint64_t *ticket) {
    std::shared_ptr<VsCtx> vc = vs_find(id);
    if (!vc || count < 1 || count > MI_VS_MAX_PRODUCTS_PER_TICKET) return MI_ERR_BAD_ARGS;
    VsKeys &v = vs_keys(cx, *vc);
    VsPipe &pp = vs_pipe(cx);
    int slot = -1;
    for (int i = 0; i < pp.nslots; i++)
        if (!pp.slots[i].busy) {
            slot = i;
            break;
        }
    if (slot < 0) return MI_ERR_AGAIN;  // every product stream is busy: wait for a ticket first
    const uint64_t tk = vs_make_id(pp.next_ticket++, cx.index, slot);
    VsSlot &sl = pp.slots[slot];
    HostSpan hs("finalize_async");
    G1Affine ps[MI_VS_MAX_PRODUCTS_PER_TICKET * MI_MAX_PAIRS];
    {
        HostSpan hs_sum("finalize_async.sum");
        for (int k = 0; k < count; k++) {
            uint64_t fails = 0;
            vs_sum_partials(parts + (size_t)k * np, np, ps + (size_t)k * MI_MAX_PAIRS, &fails);
            sl.fails_n[k] = fails;
        }
    }
    const G2Prepared *qs[MI_MAX_PAIRS];
    const int K = vs_g2_keys(v, qs);
    {
        HostSpan hs_launch("finalize_async.launch");
        pairing_launch_multi(cx, ps, qs, K, count, sl.h_out, sl.stream, slot);  // the kernels store their results in pinned host memory
    }
    stats().pairing_products += (uint64_t)count;
    stats().pairing_pairs += (uint64_t)count * (uint64_t)K;
    MI_HIP(hipEventRecord(sl.done, sl.stream));
    sl.ticket = tk;
    sl.busy = true;
    sl.fails = sl.fails_n[0];
    sl.count = count;
    *ticket = tk;
    return MI_OK;
}
int32_t vs_finalize_async(Context &cx, uint64_t id, const mi_vs_partial *parts, size_t np, uint64_t *ticket) {
    return vs_finalize_async_multi(cx, id, parts, np, 1, ticket);
}
int32_t vs_finalize_wait_multi(Context &cx, uint64_t id, uint64_t ticket, uint8_t *all_ok, int count) {
    if (!vs_find(id)) return MI_ERR_BAD_ARGS;
    if ((ticket & 0xff) >= VS_SLOTS || (int)((ticket >> 8) & 0xff) != cx.index) return MI_ERR_BAD_ARGS;
    VsSlot &sl = vs_pipe(cx).slots[ticket & 0xff];
    if (!sl.busy || sl.ticket != ticket) return MI_ERR_BAD_ARGS;
    const int32_t st = vs_slot_wait(sl, all_ok, count);
    cx.lane_cv.notify_all();  // vs_wait_product_slot
    return st;
}
// MI_OK as soon as one of the device's product slots is free, MI_ERR_AGAIN when none was for timeout_ms: what a driver
// whose mi_vs_finalize_async came back MI_ERR_AGAIN waits on when the tickets in flight are not its own (another
// stream, or a caller of the split entry points, on the same device)
int32_t vs_wait_product_slot(Context &cx, int timeout_ms) {
    VsPipe &pp = vs_pipe(cx);
    const auto until = std::chrono::steady_clock::now() + std::chrono::milliseconds(timeout_ms < 0 ? 0 : timeout_ms);
    for (;;) {
        for (int i = 0; i < pp.nslots; i++)
            if (!pp.slots[i].busy) return MI_OK;
        if (!tl_ctx_lock || cx.lane_cv.wait_until(*tl_ctx_lock, until) == std::cv_status::timeout) return MI_ERR_AGAIN;
    }
}
int32_t vs_finalize_wait(Context &cx, uint64_t id, uint64_t ticket, uint8_t *all_ok) {
    return vs_finalize_wait_multi(cx, id, ticket, all_ok, 1);
}

// name the invalid proofs of a batch whose aggregate check failed: terms, sums, group products, per-proof products, all
// on the lane's stream and workspaces (the lane is still held by the caller); one wait at the end
static void vs_identify(Context &cx, VsKeys &v, const VsWork &w, size_t n, int lane, hipStream_t st, uint8_t *verdicts) {
    char nm[32];
    const uint32_t CH = vs_chains_of(v.shape.per_proof, v.shape.hiding, v.shape.n_bounded, v.shape.max_circ);
    const size_t n_groups = cdiv(n, VS_GROUP);
    snprintf(nm, sizeof nm, "vs%d.terms", lane);
    G1Xyzz *terms = (G1Xyzz *)cx.buf(nm, sizeof(G1Xyzz) * n * CH);
    snprintf(nm, sizeof nm, "vs%d.tables", lane);
    G1Affine *tables = (G1Affine *)cx.buf(nm, sizeof(G1Affine) * n * CH * (1u << (VS_GLV_W - 1)));
    snprintf(nm, sizeof nm, "vs%d.sums", lane);
    G1Affine *sums = (G1Affine *)cx.buf(nm, sizeof(G1Affine) * (n + n_groups) * MI_MAX_PAIRS);
    G1Affine *gsums = sums + n * MI_MAX_PAIRS;
    snprintf(nm, sizeof nm, "vs%d.verdicts", lane);
    const size_t todo_off = (n + 15) / 16 * 16;
    uint8_t *d_verdicts = (uint8_t *)cx.buf(nm, todo_off + 4 * (n + 1));
    uint32_t *d_todo = (uint32_t *)(d_verdicts + todo_off), *d_count = d_todo + n;
    VsTermsIn in;
    in.pts1 = w.o.pts1; in.sc1 = w.o.sc1; in.sc2 = w.o.sc2;
    for (int c = 0; c < MI_VS_MAX_BOUNDS; c++) { in.ptsb[c] = w.o.ptsb[c]; in.scb[c] = w.o.scb[c]; }
    in.vk_scal = w.o.vk_scal; in.g_scal = w.o.g_scal; in.h_scal = w.o.h_scal; in.circuit_of = w.o.circuit_of;
    const VsConsts *kc = (const VsConsts *)v.consts.p;
    MI_HIP(hipMemsetAsync(d_count, 0, 4, st));
    cx.prof.begin("vs_terms", st, n);
    hipLaunchKernelGGL(vs_terms_kernel, dim3(cdiv(n * CH, 256)), dim3(256), 0, st, n, in, (const mi_vs_circuit_vk *)v.circuits.p, kc,
                       tables, terms);
    cx.prof.end("vs_terms", st);
    if (CH > 128) throw std::runtime_error("vs_identify: shape has more chains per proof than vs_sums_kernel lanes");
    const uint32_t LP = CH <= 64 ? 64 : 128;
    const size_t sums_lds = (sizeof(G1Xyzz28) + 4) * VS_GROUP * LP;
    cx.prof.begin("vs_sums", st, n);
    if (LP == 64) {
        cx.raise_lds_limit((const void *)vs_sums_kernel<64>, (int)sums_lds);
        hipLaunchKernelGGL(vs_sums_kernel<64>, dim3((unsigned)n_groups), dim3(VS_GROUP * 64), sums_lds, st, n, terms, kc, sums, gsums);
    } else {
        cx.raise_lds_limit((const void *)vs_sums_kernel<128>, (int)sums_lds);
        hipLaunchKernelGGL(vs_sums_kernel<128>, dim3((unsigned)n_groups), dim3(VS_GROUP * 128), sums_lds, st, n, terms, kc, sums, gsums);
    }
    cx.prof.end("vs_sums", st);
    VsG2Keys keys;
    const G2Prepared *qs[MI_MAX_PAIRS];
    (void)vs_g2_keys(v, qs);
    for (int k = 0; k < MI_MAX_PAIRS; k++) keys.q[k] = qs[k];
    cx.raise_lds_limit((const void *)vs_group_check_kernel, (int)PW_LDS_BYTES);
    cx.raise_lds_limit((const void *)vs_proof_check_kernel, (int)PW_LDS_BYTES);
    const PairingConsts *pc = pairing_consts(cx, st);
    cx.prof.begin("vs_groups", st, n);
    hipLaunchKernelGGL(vs_group_check_kernel, dim3((unsigned)n_groups), dim3(PW_THREADS), PW_LDS_BYTES, st, n, gsums, keys, kc, pc,
                       w.o.alg_ok, d_verdicts, d_todo, d_count);
    cx.prof.end("vs_groups", st);
    cx.prof.begin("vs_per_proof", st, n);
    const unsigned grid = (unsigned)std::min<size_t>(n, (size_t)cx.n_cus);
    hipLaunchKernelGGL(vs_proof_check_kernel, dim3(grid), dim3(PW_THREADS), PW_LDS_BYTES, st, sums, keys, kc, pc, d_verdicts,
                       (const uint32_t *)d_todo, (const uint32_t *)d_count);
    MI_HIP(hipGetLastError());
    cx.prof.end("vs_per_proof", st);
    snprintf(nm, sizeof nm, "vs.verdict_pin%d", lane);
    uint8_t *h = (uint8_t *)cx.pin_named(nm, n);
    MI_HIP(hipMemcpyAsync(h, d_verdicts, n, hipMemcpyDeviceToHost, st));
    {
        Unlocked unlocked;
        MI_HIP(hipStreamSynchronize(st));
    }
    memcpy(verdicts, h, n);
    stats().fallback_batches++;
    for (size_t i = 0; i < n; i++) stats().proofs_rejected += h[i] ? 0 : 1;
}

int32_t vs_verify_batch_device(Context &cx, uint64_t id, const void *d_proofs, size_t n, const void *d_rand,
                               uint8_t *verdicts, uint8_t *all_ok, hipStream_t st) {
    std::shared_ptr<VsCtx> vc = vs_find(id);
    if (!vc) return MI_ERR_BAD_ARGS;
    StatTimer timer;
    mi_vs_partial part;
    VsWork w;
    uint64_t hd;
    int32_t s = vs_accumulate_begin_impl(cx, *vc, d_proofs, n, d_rand, st, &hd, &w);
    if (s != MI_OK) return s;
    // the lane (its stream and workspaces: w points into them) stays ours until the verdicts are known
    s = vs_accumulate_end_impl(cx, hd, &part, true);
    if (s != MI_OK) return s;
    VsAcc &a = vs_pipe(cx).acc[hd & 0xff];
    try {
        s = vs_finalize(cx, id, &part, 1, all_ok, a.st);
        if (s == MI_OK) {
            if (*all_ok) memset(verdicts, 1, n);
            else vs_identify(cx, vs_keys(cx, *vc), w, n, (int)(hd & 0xff), a.st, verdicts);  // aggregate check failed
            stats().batches_verified++;
            stats().proofs_verified += n;
        }
    } catch (...) {
        vs_release_lane(cx, a);
        throw;
    }
    vs_release_lane(cx, a);
    return s;
}

// the verdict vector of a batch KNOWN to fail (a streamed batch whose all_ok came back 0: the stream keeps no per-proof
// state): derive and the per-proof fallback, without the aggregate MSMs and product
int32_t vs_identify_device(Context &cx, uint64_t id, const void *d_proofs, size_t n, const void *d_rand, uint8_t *verdicts,
                           hipStream_t st) {
    std::shared_ptr<VsCtx> vc = vs_find(id);
    if (!vc) return MI_ERR_BAD_ARGS;
    if (n == 0) return MI_OK;
    VsWork w;
    uint64_t hd;
    const int32_t s = vs_accumulate_begin_impl(cx, *vc, d_proofs, n, d_rand, st, &hd, &w, false);
    if (s != MI_OK) return s;
    VsAcc &a = vs_pipe(cx).acc[hd & 0xff];
    try {
        vs_identify(cx, vs_keys(cx, *vc), w, n, (int)(hd & 0xff), a.st, verdicts);
    } catch (...) {
        vs_release_lane(cx, a);
        throw;
    }
    vs_release_lane(cx, a);
    return MI_OK;
}

// ---------------------------------------------------------------- one block over all devices of the process
// SURVEY.md 8e inside ONE process (the reference node is one process: reference cli/src/commands/start.rs:616-642,
// the verifier is entered from Consensus::try_advance_to_next_block, reference node/consensus/src/lib.rs:499): the
// block's proofs are cut into contiguous shards, one host thread per device stages and accumulates its shard
// (derive + both MSMs + host fold -> one 304-byte partial), the partials are all-gathered over RCCL (one grouped
// ncclAllGather over the devices' communicators; on the host when RCCL is not available: they are host values), the
// primary device runs the single pairing product, and only if that fails every device names the invalid proofs of
// its own shard.
struct VsShardJob {
    Context *cx = nullptr;
    const void *proofs = nullptr, *rand = nullptr;
    size_t n = 0, first = 0;  // first: index of the shard's first proof in the block
    uint64_t handle = 0;
    bool lane_held = false;
    VsWork w;
    mi_vs_partial part;
    int32_t status = MI_OK;
};
int32_t vs_verify_block(uint64_t id, const void *const *proofs, const void *const *rand, const size_t *n_per_shard,
                        int n_shards, bool device_ptrs, uint8_t *verdicts, uint8_t *all_ok) {
    std::shared_ptr<VsCtx> vc = vs_find(id);
    if (!vc) return MI_ERR_BAD_ARGS;
    std::vector<VsShardJob> jobs;
    size_t total = 0;
    for (int s = 0; s < n_shards; s++) {
        if (n_per_shard[s] == 0) {
            continue;  // fewer proofs than devices
        }
        VsShardJob j;
        j.proofs = proofs[s];
        j.rand = rand[s];
        j.n = n_per_shard[s];
        j.first = total;
        j.cx = device_ptrs ? context_of_ptr(proofs[s]) : context_at(s);
        if (!j.cx) return MI_ERR_NO_DEVICE;
        total += j.n;
        jobs.push_back(j);
    }
    if (jobs.empty()) return MI_ERR_BAD_ARGS;
    for (size_t a = 0; a < jobs.size(); a++)
        for (size_t b = 0; b < a; b++)
            if (jobs[a].cx == jobs[b].cx && !device_ptrs) return MI_ERR_BAD_ARGS;
    auto on_all = [&](const std::function<void(VsShardJob &)> &f) {  // one host thread per shard beyond the first
        std::vector<std::thread> th;
        for (size_t k = 1; k < jobs.size(); k++) th.emplace_back([&, k] { f(jobs[k]); });
        f(jobs[0]);
        for (auto &t : th) t.join();
    };
    auto release = [&](VsShardJob &j) {
        if (!j.lane_held) return;
        (void)guarded_on(j.cx, [&](Context &cx) -> int32_t {
            vs_release_lane(cx, vs_pipe(cx).acc[j.handle & 0xff]);
            return MI_OK;
        });
        j.lane_held = false;
    };
    // ---- phase 1: every shard's partial accumulator
    on_all([&](VsShardJob &j) {
        j.status = guarded_on(j.cx, [&](Context &cx) -> int32_t {
            const void *dp = j.proofs, *dr = j.rand;
            hipStream_t st = cx.stream;
            if (!device_ptrs) {
                const size_t pb = vc->consts.wire ? (size_t)vc->consts.wire_bytes : (size_t)vc->consts.proof_words * 4, rb = vc->consts.rand_bytes;
                void *p = cx.buf("blk.in_proofs", pb * j.n);
                void *r = cx.buf("blk.in_rand", rb * j.n);
                MI_HIP(hipMemcpyAsync(p, j.proofs, pb * j.n, hipMemcpyHostToDevice, st));
                MI_HIP(hipMemcpyAsync(r, j.rand, rb * j.n, hipMemcpyHostToDevice, st));
                dp = p;
                dr = r;
            }
            int32_t s = vs_accumulate_begin_impl(cx, *vc, dp, j.n, dr, st, &j.handle, &j.w);
            if (s != MI_OK) return s;
            j.lane_held = true;
            s = vs_accumulate_end_impl(cx, j.handle, &j.part, true);
            if (s != MI_OK) j.lane_held = false;  // the end released it
            return s;
        });
    });
    int32_t status = MI_OK;
    for (auto &j : jobs)
        if (j.status != MI_OK) status = j.status;
    // ---- the exchange: all shards' partials to the primary (RCCL all-gather; host values if there is no RCCL)
    std::vector<mi_vs_partial> parts(jobs.size());
    if (status == MI_OK) {
        bool gathered = false;
        std::shared_ptr<Comm> comm = jobs.size() > 1 && !device_ptrs ? comm_local() : nullptr;
        if (comm && (int)jobs.size() == n_contexts()) {
            std::vector<const void *> mine(jobs.size());
            std::vector<void *> all(jobs.size(), nullptr);
            for (size_t k = 0; k < jobs.size(); k++) mine[jobs[k].cx->index] = &jobs[k].part;
            all[0] = parts.data();  // the primary's copy is the one that is used
            gathered = comm_all_gather(*comm, mine.data(), sizeof(mi_vs_partial), all.data()) == MI_OK;
        }
        if (!gathered)
            for (size_t k = 0; k < jobs.size(); k++) parts[k] = jobs[k].part;
        // ---- the single pairing product
        status = guarded([&](Context &cx) -> int32_t { return vs_finalize(cx, id, parts.data(), parts.size(), all_ok, cx.stream); });
    }
    // ---- phase 2, only when the aggregate failed: every device names the invalid proofs of its shard
    if (status == MI_OK) {
        if (*all_ok) {
            memset(verdicts, 1, total);
        } else {
            on_all([&](VsShardJob &j) {
                j.status = guarded_on(j.cx, [&](Context &cx) -> int32_t {
                    VsAcc &a = vs_pipe(cx).acc[j.handle & 0xff];
                    vs_identify(cx, vs_keys(cx, *vc), j.w, j.n, (int)(j.handle & 0xff), a.st, verdicts + j.first);
                    return MI_OK;
                });
            });
            for (auto &j : jobs)
                if (j.status != MI_OK) status = j.status;
        }
    }
    for (auto &j : jobs) release(j);
    return status;
}

// ---------------------------------------------------------------- KZG10::batch_check (with hiding: row A5)
// lane i: sc[i] = r_i, sc[n+i] = r_i z_i (canonical), rv[i] = r_i v_i, rr[i] = r_i rbar_i (Montgomery); sc2[i] = r_i
__global__ void __launch_bounds__(128) kzg_scalars_kernel(size_t n, const Fr *__restrict__ points,
                                                          const Fr *__restrict__ values, const Fr *__restrict__ random_v,
                                                          const uint8_t *__restrict__ rand128, uint32_t *__restrict__ sc,
                                                          uint32_t *__restrict__ sc2, Fr *__restrict__ rv, Fr *__restrict__ rr) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t *rw = reinterpret_cast<const uint32_t *>(rand128 + 16 * i);
    uint32_t r4[4] = {rw[0], rw[1], rw[2], rw[3]};
    Fr r = fr_from_u128(r4);
    uint32_t *o = sc + i * 8, *o2 = sc2 + i * 8;
#pragma unroll
    for (int k = 0; k < 4; k++) { o[k] = r4[k]; o[4 + k] = 0; o2[k] = r4[k]; o2[4 + k] = 0; }
    fr_store(sc + (n + i) * 8, from_mont(mul(r, fr_load(points + i))));
    fr_store(rv + i, mul(r, fr_load(values + i)));
    if (random_v) fr_store(rr + i, mul(r, fr_load(random_v + i)));
}
// single block: out = -(sum v) canonical
__global__ void __launch_bounds__(256) kzg_gsum_kernel(size_t n, const Fr *__restrict__ rv, uint32_t *__restrict__ out) {
    __shared__ uint32_t lds[8 * 256];
    const int t = threadIdx.x;
    Fr acc = Fr::zero();
    for (size_t i = t; i < n; i += 256) acc = add(acc, fr_load(rv + i));
    for (int stride = 128; stride >= 1; stride >>= 1) {
        if (t >= stride && t < 2 * stride)
#pragma unroll
            for (int k = 0; k < 8; k++) lds[k * 256 + (t - stride)] = acc.l[k];
        __syncthreads();
        if (t < stride) {
            Fr o;
#pragma unroll
            for (int k = 0; k < 8; k++) o.l[k] = lds[k * 256 + t];
            acc = add(acc, o);
        }
        __syncthreads();
    }
    if (t == 0) fr_store(out, from_mont(neg(acc)));
}

// total_c = sum r_i (C_i + z_i w_i) - (sum r_i v_i) g - (sum r_i rbar_i) gamma_g, total_w = sum r_i w_i,
// e(total_c, h) e(-total_w, beta_h) == 1; gamma_g / random_v null: no hiding
int32_t kzg_batch_check_host(Context &cx, const mi_kzg_vk *vk, const mi_g1_affine *gamma_g, const mi_g1_affine *comms,
                             const mi_fr *points, const mi_fr *values, const mi_g1_affine *witnesses, const mi_fr *random_v,
                             size_t n, const uint8_t *rand128, uint8_t *ok_out) {
    hipStream_t st = cx.stream;
    const bool hiding = gamma_g != nullptr;
    const size_t n1 = 2 * n + 1 + (hiding ? 1 : 0);
    G1Affine *pts = (G1Affine *)cx.buf("kzg.pts", sizeof(G1Affine) * n1);
    uint32_t *sc = (uint32_t *)cx.buf("kzg.sc", 32 * n1);
    uint32_t *sc2 = (uint32_t *)cx.buf("kzg.sc2", 32 * n);
    Fr *dpts = (Fr *)cx.buf("kzg.points", sizeof(Fr) * n), *dval = (Fr *)cx.buf("kzg.values", sizeof(Fr) * n);
    Fr *drv = (Fr *)cx.buf("kzg.random_v", sizeof(Fr) * n);
    Fr *rv = (Fr *)cx.buf("kzg.rv", sizeof(Fr) * n), *rr = (Fr *)cx.buf("kzg.rr", sizeof(Fr) * n);
    uint8_t *dr = (uint8_t *)cx.buf("kzg.rand", 16 * n);
    MI_HIP(hipMemcpyAsync(pts, comms, sizeof(G1Affine) * n, hipMemcpyHostToDevice, st));
    MI_HIP(hipMemcpyAsync(pts + n, witnesses, sizeof(G1Affine) * n, hipMemcpyHostToDevice, st));
    MI_HIP(hipMemcpyAsync(pts + 2 * n, &vk->g, sizeof(G1Affine), hipMemcpyHostToDevice, st));
    if (hiding) {
        MI_HIP(hipMemcpyAsync(pts + 2 * n + 1, gamma_g, sizeof(G1Affine), hipMemcpyHostToDevice, st));
        MI_HIP(hipMemcpyAsync(drv, random_v, sizeof(Fr) * n, hipMemcpyHostToDevice, st));
    }
    MI_HIP(hipMemcpyAsync(dpts, points, sizeof(Fr) * n, hipMemcpyHostToDevice, st));
    MI_HIP(hipMemcpyAsync(dval, values, sizeof(Fr) * n, hipMemcpyHostToDevice, st));
    MI_HIP(hipMemcpyAsync(dr, rand128, 16 * n, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(kzg_scalars_kernel, dim3(cdiv(n, 128)), dim3(128), 0, st, n, dpts, dval, hiding ? (const Fr *)drv : (const Fr *)nullptr,
                       dr, sc, sc2, rv, rr);
    hipLaunchKernelGGL(kzg_gsum_kernel, dim3(1), dim3(256), 0, st, n, rv, sc + 2 * n * 8);
    if (hiding) hipLaunchKernelGGL(kzg_gsum_kernel, dim3(1), dim3(256), 0, st, n, rr, sc + (2 * n + 1) * 8);
    MI_HIP(hipGetLastError());
    mi_g1_jac tc, tw;
    msm_device(cx, pts, sc, n1, &tc, st);
    msm_device(cx, pts + n, sc2, n, &tw, st, 128);
    G1Affine ps[2] = {host_jac_to_affine_normalised(tc), neg(host_jac_to_affine_normalised(tw))};
    G2Affine qs[2];
    memcpy(&qs[0], &vk->h, sizeof(G2Affine));
    memcpy(&qs[1], &vk->tau_h, sizeof(G2Affine));
    Fq12 f;
    pairing_product_host_args(cx, ps, qs, 2, &f, st, true);
    *ok_out = f.is_one() ? 1 : 0;
    return MI_OK;
}

}  // namespace mi
