"""CPU tier: the limb identity behind the dedicated squaring (tools/gen_comba.py gen_sqr -> comba_gen.hpp sqr_comba_gen):
cross products against the doubled operand, neighbours against a_j << 1, squares once.  The device code itself is pinned
by every -m gpu parity test (the squaring is on all their paths)."""
import os
import random
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import gen_comba  # noqa: E402


def test_squaring_columns_sum_to_the_square():
    rnd = random.Random(5)
    for n, bits in ((12, 378), (8, 254)):  # operands up to 2q / 2r: the doubled operand still fits n limbs
        cases = [0, 1, (1 << bits) - 1, (1 << (bits - 1)), int("8" * (bits // 4), 16), int("f" * 8 + "0" * 8, 16) << 64]
        cases += [rnd.getrandbits(bits) for _ in range(500)]
        for a in cases:
            assert gen_comba.sqr_columns_model(a, n) == a * a


def test_generated_header_is_current():
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    before = open(gen_comba.OUT).read()
    subprocess.check_call([sys.executable, os.path.join(root, "tools", "gen_comba.py")], stdout=subprocess.DEVNULL)
    assert open(gen_comba.OUT).read() == before
