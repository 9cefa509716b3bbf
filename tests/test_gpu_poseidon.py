"""-m gpu: Poseidon sponge over Fq (row A9, own parameters), many sponges per launch, bit-exact vs the oracle."""
import numpy as np
import pytest

from helpers import rng_points

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("m,k", [(0, 1), (1, 1), (2, 2), (5, 3), (16, 4)])
def test_poseidon_hash_many(orc, mi, m, k):
    n = 200
    ins = rng_points(900 + m, (n * max(m, 1) + 1) // 2 + 1).reshape(-1, 6)[: n * m].reshape(n, m, 6)  # field elements < q
    got = mi.PoseidonSponge.hash_many(np.ascontiguousarray(ins), k)
    want = orc.poseidon_hash_many(np.ascontiguousarray(ins), k, threads=8)
    assert (got == want).all()


def test_poseidon_golden(mi):
    import golden_util as G
    import orc_py as O
    d = G.load("poseidon.json")
    for c in d["sponge"]:
        ins = np.stack([O.fq_from_int(G.h(x)) for x in c["inputs"]]).reshape(1, -1, 6) if c["inputs"] else np.zeros((1, 0, 6), dtype=np.uint64)
        out = mi.PoseidonSponge.hash_many(ins, len(c["outputs"]))
        assert [O.fq_to_int(x) for x in out[0]] == [G.h(x) for x in c["outputs"]]
