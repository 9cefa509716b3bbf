#!/usr/bin/env python3
"""Generate snarkos_amd/csrc/comba_gen.hpp: fully unrolled product-scanning Montgomery products for gfx950.

One inline-asm statement per (column, operand kind): every limb product is v_mad_u64_u32 (64-bit product added
into the 64-bit column accumulator, carry to VCC) + v_addc_co_u32 (carry into the overflow word).  hipcc pads
each asm statement with an s_nop, so a column's products share one statement (<= 30 asm operands).
"""
import os

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "snarkos_amd", "csrc", "comba_gen.hpp")


def gen(name, params, n, lanes, reduce=True):
    o = []
    a = o.append
    a("template <>")
    if reduce:
        a("__device__ __forceinline__ Fp<%s> mul_comba_gen<%s, %d>(const Fp<%s> &a, const Fp<%s> &b) {" % (params, params, lanes, params, params))
    else:
        # same product without the final conditional subtraction: result < q (1 + q/R) for canonical inputs
        a("__device__ __forceinline__ Fp<%s> mul_comba_nr_gen<%s>(const Fp<%s> &a, const Fp<%s> &b) {" % (params, params, params, params))
    a("    uint32_t m[%d], r[%d];" % (n, n))
    a("    uint64_t acc = 0, acc2 = 0;")
    a("    uint32_t ovf = 0, ovf2 = 0;")
    a("    (void)acc2; (void)ovf2;")
    for k in range(2 * n - 1):
        for kind in ("ab", "mp"):
            prods = []
            for i in range(n):
                j = k - i
                if kind == "ab" and 0 <= j < n:
                    prods.append((i, j))
                if kind == "mp" and 1 <= j < n and i < k:
                    prods.append((i, j))
            if not prods:
                continue
            # operands: %0 acc, %1 ovf, (%2 acc2, %3 ovf2), then inputs
            base = 4 if lanes == 2 else 2
            lines, ins = [], []
            for idx, (i, j) in enumerate(prods):
                x = base + 2 * idx
                tgt = (2, 3) if (lanes == 2 and idx % 2 == 1) else (0, 1)
                lines.append("v_mad_u64_u32 %%%d, vcc, %%%d, %%%d, %%%d" % (tgt[0], x, x + 1, tgt[0]))
                lines.append("v_addc_co_u32 %%%d, vcc, 0, %%%d, vcc" % (tgt[1], tgt[1]))
                if kind == "ab":
                    ins.append('"v"(a.l[%d]), "v"(b.l[%d])' % (i, j))
                else:
                    ins.append('"v"(m[%d]), "s"(%s::MOD[%d])' % (i, params, j))
            outs = '"+&v"(acc), "+&v"(ovf)' + (', "+&v"(acc2), "+&v"(ovf2)' if lanes == 2 else "")
            a('    asm("%s"' % "\\n\\t".join(lines))
            a("        : %s" % outs)
            a("        : %s" % ", ".join(ins))
            a('        : "vcc");')
        if lanes == 2:
            a("    { uint64_t s = acc + acc2; ovf += ovf2 + (uint32_t)(s < acc); acc = s; acc2 = 0; ovf2 = 0; }")
        if k < n:
            a("    { const uint32_t lo = (uint32_t)acc, hi = (uint32_t)(acc >> 32); m[%d] = 0u - lo;" % k)
            a("      acc = ((uint64_t)ovf << 32) + (uint64_t)hi + (uint64_t)(lo != 0); ovf = 0; }")
        else:
            a("    { r[%d] = (uint32_t)acc; acc = ((uint64_t)ovf << 32) | (acc >> 32); ovf = 0; }" % (k - n))
    a("    r[%d] = (uint32_t)acc;" % (n - 1))
    a("    Fp<%s> out;" % params)
    for i in range(n):
        a("    out.l[%d] = r[%d];" % (i, i))
    if reduce:
        a("    reduce_once(out);")
    a("    return out;")
    a("}")
    return "\n".join(o)


def gen_sqr(params, n):
    """a^2 R^-1: the cross products a_i a_j (i < j) are taken once against the doubled operand a2 = 2a (2a < 2^(32 n):
    both moduli leave at least 3 spare bits), the squares a_i^2 once: n (n + 1) / 2 limb products instead of n^2; the
    reduction half (m_i p_j) is the product's.
    Limb j of 2a is (a_j << 1) | (a_(j-1) >> 31): the bit that comes in from below belongs to the products with a_(j-1),
    and a_i times it is what the pair (i, j - 1) lost when a_(j-1) << 1 dropped its top bit -- except for i = j - 1,
    where no such pair exists (the square a_i^2 is taken once, not doubled): those neighbours use b2_j = a_j << 1."""
    o = []
    a = o.append
    a("template <>")
    a("__device__ __forceinline__ Fp<%s> sqr_comba_gen<%s>(const Fp<%s> &a) {" % (params, params, params))
    a("    uint32_t m[%d], r[%d], a2[%d], b2[%d];" % (n, n, n, n))
    a("    a2[0] = a.l[0] << 1; b2[0] = a2[0];")
    for j in range(1, n):
        a("    a2[%d] = __builtin_amdgcn_alignbit(a.l[%d], a.l[%d], 31); b2[%d] = a.l[%d] << 1;" % (j, j, j - 1, j, j))
    a("    uint64_t acc = 0;")
    a("    uint32_t ovf = 0;")
    for k in range(2 * n - 1):
        for kind in ("aa", "mp"):
            ins = []
            if kind == "aa":
                for i in range(n):
                    j = k - i
                    if i < j < n:
                        ins.append('"v"(a.l[%d]), "v"(%s[%d])' % (i, "b2" if j == i + 1 else "a2", j))
                if k % 2 == 0 and k // 2 < n:
                    ins.append('"v"(a.l[%d]), "v"(a.l[%d])' % (k // 2, k // 2))
            else:
                for i in range(n):
                    j = k - i
                    if 1 <= j < n and i < k:
                        ins.append('"v"(m[%d]), "s"(%s::MOD[%d])' % (i, params, j))
            if not ins:
                continue
            lines = []
            for idx in range(len(ins)):
                x = 2 + 2 * idx
                lines.append("v_mad_u64_u32 %%0, vcc, %%%d, %%%d, %%0" % (x, x + 1))
                lines.append("v_addc_co_u32 %1, vcc, 0, %1, vcc")
            a('    asm("%s"' % "\\n\\t".join(lines))
            a('        : "+&v"(acc), "+&v"(ovf)')
            a("        : %s" % ", ".join(ins))
            a('        : "vcc");')
        if k < n:
            a("    { const uint32_t lo = (uint32_t)acc, hi = (uint32_t)(acc >> 32); m[%d] = 0u - lo;" % k)
            a("      acc = ((uint64_t)ovf << 32) + (uint64_t)hi + (uint64_t)(lo != 0); ovf = 0; }")
        else:
            a("    { r[%d] = (uint32_t)acc; acc = ((uint64_t)ovf << 32) | (acc >> 32); ovf = 0; }" % (k - n))
    a("    r[%d] = (uint32_t)acc;" % (n - 1))
    a("    Fp<%s> out;" % params)
    for i in range(n):
        a("    out.l[%d] = r[%d];" % (i, i))
    a("    reduce_once(out);")
    a("    return out;")
    a("}")
    return "\n".join(o)


OUT28 = os.path.join(os.path.dirname(OUT), "f28_gen.hpp")


def gen_mul28(square):
    """Montgomery product / squaring on 14 signed 28-bit limbs (field28.hpp; arithmetic and bounds: tools/model_f28.py):
    ONE v_mad_i64_i32 per limb product, the column's carry is an arithmetic shift of the 64-bit accumulator."""
    n, w = 14, 28
    o = []
    a = o.append
    if square:
        a("__device__ __forceinline__ Fq28 sqr28_gen(const Fq28 &a) {")
        a("    int32_t a2[14];")
        for j in range(1, n):
            a("    a2[%d] = a.l[%d] * 2;" % (j, j))
    else:
        a("__device__ __forceinline__ Fq28 mul28_gen(const Fq28 &a, const Fq28 &b) {")
    a("    constexpr uint32_t MASK = (1u << 28) - 1u;")
    a("    int32_t m[14];")
    a("    Fq28 r;")
    a("    int64_t acc = 0;")
    for k in range(2 * n - 1):
        ab, mp = [], []
        for i in range(n):
            j = k - i
            if not 0 <= j < n:
                continue
            if not square:
                ab.append('"v"(a.l[%d]), "v"(b.l[%d])' % (i, j))
            elif i < j:
                ab.append('"v"(a.l[%d]), "v"(a2[%d])' % (i, j))
            elif i == j:
                ab.append('"v"(a.l[%d]), "v"(a.l[%d])' % (i, i))
        for i in range(min(k, n)):
            j = k - i
            if 1 <= j < n:
                mp.append('"v"(m[%d]), "s"(F28C.mod[%d])' % (i, j))
        for ins in (ab, mp):
            if not ins:
                continue
            assert len(ins) <= 14
            lines = ["v_mad_i64_i32 %%0, vcc, %%%d, %%%d, %%0" % (1 + 2 * x, 2 + 2 * x) for x in range(len(ins))]
            a('    asm("%s"' % "\\n\\t".join(lines))
            a('        : "+&v"(acc)')
            a("        : %s" % ", ".join(ins))
            a('        : "vcc");')
        if k < n:
            a("    { const uint32_t lo = (uint32_t)acc & MASK; m[%d] = (int32_t)((0u - lo) & MASK); acc = (acc + m[%d]) >> 28; }" % (k, k))
        else:
            a("    { r.l[%d] = (int32_t)((uint32_t)acc & MASK); acc >>= 28; }" % (k - n))
    a("    r.l[13] = (int32_t)acc;")
    a("    return r;")
    a("}")
    return "\n".join(o)


def sqr_columns_model(a, n):
    """the integer the squaring's product half accumulates, column by column, exactly as gen_sqr emits it
    (tests/test_comba_model.py checks it equals a * a)"""
    al = [(a >> (32 * i)) & 0xFFFFFFFF for i in range(n)]
    a2 = [((al[j] << 1) | (al[j - 1] >> 31 if j else 0)) & 0xFFFFFFFF for j in range(n)]
    b2 = [(al[j] << 1) & 0xFFFFFFFF for j in range(n)]
    tot = 0
    for k in range(2 * n - 1):
        col = 0
        for i in range(n):
            j = k - i
            if i < j < n:
                col += al[i] * (b2[j] if j == i + 1 else a2[j])
        if k % 2 == 0 and k // 2 < n:
            col += al[k // 2] ** 2
        tot += col << (32 * k)
    return tot


def main():
    parts = ["// GENERATED by tools/gen_comba.py -- do not edit.",
             "// Product-scanning (Comba) Montgomery products for gfx950, see field.hpp.",
             "#pragma once",
             "namespace mi {",
             "template <class P, int LANES> __device__ __forceinline__ Fp<P> mul_comba_gen(const Fp<P> &a, const Fp<P> &b);",
             "template <class P> __device__ __forceinline__ Fp<P> mul_comba_nr_gen(const Fp<P> &a, const Fp<P> &b);",
             "template <class P> __device__ __forceinline__ Fp<P> sqr_comba_gen(const Fp<P> &a);"]
    for lanes in (1, 2):
        parts.append(gen("Fq", "FqP", 12, lanes))
        parts.append(gen("Fr", "FrP", 8, lanes))
    parts.append(gen("Fq", "FqP", 12, 1, reduce=False))
    parts.append(gen_sqr("FqP", 12))
    parts.append(gen_sqr("FrP", 8))
    parts.append("}  // namespace mi")
    open(OUT, "w").write("\n".join(parts) + "\n")
    print("wrote", OUT)
    parts28 = ["// GENERATED by tools/gen_comba.py -- do not edit.",
               "// Montgomery product and squaring on 14 signed 28-bit limbs for gfx950, see field28.hpp.",
               "#pragma once", "namespace mi {", gen_mul28(False), gen_mul28(True), "}  // namespace mi"]
    open(OUT28, "w").write("\n".join(parts28) + "\n")
    print("wrote", OUT28)


if __name__ == "__main__":
    main()
