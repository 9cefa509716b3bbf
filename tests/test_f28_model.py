"""CPU tier: the big-integer model of the reduced-radix (14 signed 28-bit limbs) field arithmetic of msm_accumulate
(snarkos_amd/csrc/field28.hpp, generated product in f28_gen.hpp): accumulator and limb bounds with worst-case operands,
values against plain modular arithmetic, the mixed addition against the affine group law, the degenerate cases flagged.
The device code itself is pinned by the -m gpu MSM / verifier parity tests."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import model_f28  # noqa: E402


def test_model_self_test():
    stats = model_f28.self_test(n=40, seed=9)
    assert stats["max_acc_bits"] <= 36


def test_constants_match_the_header():
    """2^392 mod q and the 28-bit limbs of q as field28.hpp derives them at compile time (checked here by value through the
    model's own derivation: the header's constexpr code is exercised by the build and the GPU tests)"""
    assert model_f28.PL[0] == 1 and model_f28.val(model_f28.PL) == model_f28.Q
    assert model_f28.val(model_f28.ONE) == (1 << 392) % model_f28.Q
    assert model_f28.val(model_f28.TO32) == 1 << 384


def test_generated_header_is_current():
    import gen_comba
    before = open(gen_comba.OUT28).read()
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_comba.py")], stdout=subprocess.DEVNULL)
    assert open(gen_comba.OUT28).read() == before
