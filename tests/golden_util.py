"""Load tests/golden/*.json (hex strings) into ABI-layout arrays."""
import json
import os

import numpy as np

import orc_py as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def h(s):
    return int(s, 16)


def pt(p):
    return None if p is None else (h(p[0]), h(p[1]))


def g1_arr(points):
    return np.stack([O.g1_from_ints(pt(p)) for p in points])


def g2_pt(q):
    return ((h(q[0][0]), h(q[0][1])), (h(q[1][0]), h(q[1][1])))


def scalars_arr(hexes):
    return np.stack([O.int_to_limbs(h(s), 4) for s in hexes])


def fr_arr(hexes):
    return O.fr_from_ints([h(s) for s in hexes])


def rand_arr(hexes):
    return np.frombuffer(b"".join(h(r).to_bytes(16, "little") for r in hexes), dtype=np.uint8).copy()
