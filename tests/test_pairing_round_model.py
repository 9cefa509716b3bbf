"""CPU tier: the big-integer model of the pairing kernel's rounds (tools/model_pround.py; device code: csrc/pairing_wg.hpp,
generated multiply-accumulate blocks in pround_gen.hpp): every bound the HIP code relies on with worst-case operands, the
general round against the schoolbook product in Fq[w]/(w^12 + 5), the cyclotomic squaring round against the square of
(scaled) cyclotomic-subgroup elements, and the round's term table in the header against the one the model derives from
the Granger-Scott formulas.  The device code itself is pinned by the -m gpu pairing / verifier parity tests."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import model_pround  # noqa: E402


def test_general_round_model():
    stats = model_pround.self_test(n_random=12, seed=3)
    assert stats["max_F_bits"] <= 416


def test_cyclotomic_squaring_round_model():
    assert model_pround.self_test_csqr(n_random=2, seed=4)


def test_squaring_table_in_the_header_is_the_models():
    src = open(os.path.join(ROOT, "snarkos_amd", "csrc", "pairing_wg.hpp")).read()
    body = src[src.index("PW_CSQR_TABLE[13][4] = {"):]
    body = body[:body.index("};")]
    rows = []
    for line in body.splitlines()[1:]:
        m = re.match(r"\s*/\*\s*(\d+)[^*]*\*/\s*\{(.*)\},\s*$", line)
        if not m:
            continue
        terms = []
        for item in re.findall(r"PW_CS\((\d+), (\d+), (\d+), (\d+)\)|(?<![\w(])0(?![\w)])", m.group(2)):
            terms.append(tuple(int(x) for x in item) if item[0] else None)
        rows.append((int(m.group(1)), terms))
    assert [k for k, _ in rows] == list(range(13))
    want = model_pround.csqr_table()
    for k, terms in rows:
        assert len(terms) == 4, (k, terms)
        assert terms[3] == want[k][3] or k == 12, k  # the scale's term sits on lane g = 3
        got = sorted(t for t in terms[:3] if t) if k < 12 else sorted(t for t in terms if t)
        exp = sorted(t for t in want[k][:3] if t) if k < 12 else sorted(t for t in want[k] if t)
        assert got == exp, (k, got, exp)


def test_generated_header_is_current():
    import gen_pround
    before = open(gen_pround.OUT).read()
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_pround.py")], stdout=subprocess.DEVNULL)
    assert open(gen_pround.OUT).read() == before
