"""CPU tier: the Rust shim crate (integration/snarkvm-algorithms-hip) cannot drift from the C ABI.

ffi.rs is generated from include/mi355x_snark.h (tools/gen_rust_ffi.py): the committed file must equal a fresh
rendering, and -- independently of the generator's own rendering -- every `pub fn` of its extern block must match the
header's prototype of the same name argument by argument (count, pointer depth, constness, base type), every struct
field by field, and the library must export each of them.  The INTEGRATION.md stub is the same text."""
import ctypes as C
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_rust_ffi as G  # noqa: E402

FFI = os.path.join(ROOT, "integration", "snarkvm-algorithms-hip", "src", "ffi.rs")
C2R = {"int32_t": "i32", "uint32_t": "u32", "uint64_t": "u64", "size_t": "usize", "uint8_t": "u8", "double": "f64", "char": "c_char",
       "void": "c_void"}


def _c_shape(cty):
    """C type text -> (base, [constness of each pointer level's pointee, innermost first])"""
    toks = cty.replace("*", " * ").split()
    const = False
    base, levels = None, []
    for t in toks:
        if t == "const":
            const = True
        elif t == "*":
            levels.append(const)
            const = False
        else:
            base = t
    return C2R.get(base, base), levels


def _rust_shape(rty):
    levels = []
    rty = rty.strip()
    while rty.startswith("*"):
        m = re.match(r"\*(const|mut)\s+(.*)", rty)
        levels.append(m.group(1) == "const")
        rty = m.group(2)
    return rty, levels[::-1]


def _rust_decls():
    src = open(FFI).read()
    block = src[src.index('extern "C" {'):]
    fns = {}
    for m in re.finditer(r"pub fn (mi_[a-z0-9_]+)\((.*?)\) -> ([^;]+);", block):
        args = [a.split(":", 1) for a in m.group(2).split(", ") if a]
        fns[m.group(1)] = ([(n.strip(), t.strip()) for n, t in args], m.group(3).strip())
    return src, fns


def test_ffi_rs_is_a_fresh_rendering_of_the_header():
    assert open(FFI).read() == G.render(), "run python tools/gen_rust_ffi.py"


def test_every_prototype_matches_argument_by_argument():
    _, _, fn_typedefs, functions = G.parse_header()
    src, fns = _rust_decls()
    assert len(functions) >= 50 and sorted(fns) == sorted(f[0] for f in functions)
    fn_names = {t[0] for t in fn_typedefs}
    for name, ret, args in functions:
        rargs, rret = fns[name]
        assert len(rargs) == len(args), name
        assert _rust_shape(rret) == _c_shape(ret), (name, ret, rret)
        for (cn, cty), (rn, rty) in zip(args, rargs):
            assert cn == rn, (name, cn, rn)
            if cty in fn_names:
                assert rty == cty
                continue
            assert _rust_shape(rty) == _c_shape(cty), (name, cn, cty, rty)
    # the callback type
    m = re.search(r"pub type mi_vs_exchange_fn = Option<unsafe extern \"C\" fn\((.*?)\) -> i32>;", src)
    (tname, tret, targs), = fn_typedefs
    rargs = [a.split(":", 1)[1].strip() for a in m.group(1).split(", ")]
    assert tname == "mi_vs_exchange_fn" and tret == "int32_t" and len(rargs) == len(targs) == 5
    for (_, cty), rty in zip(targs, rargs):
        assert _rust_shape(rty) == _c_shape(cty)


def test_struct_layouts_match_the_python_loader_and_the_header():
    from snarkos_amd import lib as L
    _, structs, _, _ = G.parse_header()
    defines = dict(G.parse_header()[0])
    size = {"uint64_t": 8, "uint32_t": 4, "uint8_t": 1, "int8_t": 1}
    for name, fields in structs:  # declaration order: members are defined before use; all members are 8-byte multiples or u8/u32 tails
        total = 0
        for _, ty, arr in fields:
            n = int(defines.get(arr, arr)) if arr else 1
            total += size[ty] * n
        size[name] = total
    assert size["mi_g1_affine"] == 96 and size["mi_g1_jac"] == 144 and size["mi_g2_affine"] == 192 and size["mi_fq12"] == 576
    assert size["mi_kzg_vk"] == L.KZG_VK_BYTES and size["mi_vs_circuit_vk"] == L.VS_CIRCUIT_VK_BYTES
    assert size["mi_vs_proof"] == L.VS_PROOF_BYTES and size["mi_vs_partial"] == L.VS_PARTIAL_BYTES
    # the record kinds a v2 / v3 context reads (the safe layer sizes its slices by these), signatures, counters
    assert size["mi_vs_proof2"] == L.VS2_PROOF_BYTES == 3184 and size["mi_vs_proof3"] == L.VS3_PROOF_BYTES == 5504
    assert size["mi_vs_proof4"] == L.VS4_PROOF_BYTES == 6320 and size["mi_vs_circuit_part"] == 400
    assert size["mi_vs_shape"] == L.VS2_SHAPE_BYTES and size["mi_vs_key_ext"] == L.VS2_KEY_EXT_BYTES
    assert size["mi_sig_record"] == L.SIG_RECORD_BYTES == 288 and size["mi_stats"] == 128
    rs = open(os.path.join(ROOT, "integration", "snarkvm-algorithms-hip", "src", "lib.rs")).read()
    assert "mi_vs_strides" in rs and "proofs: &[u8]" in rs and "fn count(&self, proofs: &[u8], randomizers128: &[u8])" in rs
    src = open(FFI).read()
    for name, fields in structs:
        body = re.search(r"pub struct %s \{(.*?)\n\}" % name, src, flags=re.S).group(1)
        rf = re.findall(r"pub (\w+): ([^,]+),", body)
        assert [f[0] for f in fields] == [f[0] for f in rf], name


def test_library_exports_what_the_crate_links_against():
    from snarkos_amd import lib as L
    l = C.CDLL(L.LIB_PATH)
    _, fns = _rust_decls()
    for name in fns:
        assert hasattr(l, name), name
    assert sorted(fns) == sorted(L.EXPORTS)


def test_integration_md_embeds_the_generated_extern_block():
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    src = open(FFI).read()
    block = src[src.index('#[link(name = "mi355x_snark")]'):]
    assert block.strip() in md, "INTEGRATION.md's extern block is stale: paste integration/snarkvm-algorithms-hip/src/ffi.rs"
