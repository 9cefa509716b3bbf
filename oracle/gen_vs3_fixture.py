#!/usr/bin/env python3
"""Write tests/golden/vs3_wire_24.npz: 24 VarunaShaped-v3 proofs (several circuits per proof, up to four instances,
inverse-challenge coefficients, hiding, two degree-bound classes; a quarter of them spoilt) made by the C oracle's
synthetic prover (toy SRS, seed 0xA1E00006), TOGETHER WITH what the independent Python big-integer restatement
(oracle/pyref.py) says about them: its per-proof verdicts, its wire encoding of every record (compressed points,
canonical Fr) and the decode flags of a damaged copy.  The C oracle and the HIP library are both checked against this
file, so the v3 relation and the wire codec are pinned by a committed pyref output, not only by live agreement.
Run:  python oracle/gen_vs3_fixture.py   (about a minute: pyref's pairings are big-integer Python)"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import orc_py as O  # noqa: E402
import pyref as P  # noqa: E402

V3_LC = [0, 0, 1, 4, 3, 0, 5, 3, 6] + [7, 1, 0, 4] * 3
N, NC = 24, 5
sh = O.vs2_shape(None, None, True, 2, lc_kind=V3_LC, max_instances=4, max_circuits=3)
b = O.VS2Batch(0xA1E00006, N, NC, 256, shape=sh)
assert 0 < b.expected.sum() < N and (b.verify_cpu(8) == b.expected).all()
shd = {"query_of": [int(x) for x in sh[:21]], "bound_class": [int(x) for x in sh[21:30].view(np.int8)], "hiding": bool(sh[30]),
       "n_bounds": int(sh[31]), "lc_kind": [int(x) for x in sh[32:53]], "max_instances": int(sh[53]), "max_circuits": int(sh[54])}
vkb, ext = b.vk.tobytes(), b.ext.tobytes()
tau_h = O.g2_to_ints(np.frombuffer(vkb[288:480], dtype=np.uint64))
gamma_g = O.g1_to_ints(np.frombuffer(ext[:96], dtype=np.uint64))
shift_h = [O.g2_to_ints(np.frombuffer(ext[96 + 192 * c:288 + 192 * c], dtype=np.uint64)) for c in range(2)]
circuits = [P.vs_parse_circuit(b.circuits[i * 1184:(i + 1) * 1184].tobytes()) for i in range(NC)]
RB, WB = 6320, 5072
verdicts, wire = np.zeros(N, dtype=np.uint8), np.zeros(N * WB, dtype=np.uint8)
for i in range(N):
    rec = b.proofs[i * RB:(i + 1) * RB].tobytes()
    rs = tuple(int.from_bytes(b.randomizers[48 * i + 16 * t:48 * i + 16 * t + 16].tobytes(), "little") for t in range(3))
    verdicts[i] = P.vs2_verify_one(tau_h, gamma_g, shift_h, shd, circuits, P.vs4_parse_proof(rec), rs)
    wire[i * WB:(i + 1) * WB] = np.frombuffer(P.vs_wire_encode(rec, shd), dtype=np.uint8)
    print("proof", i, "pyref verdict", int(verdicts[i]), "generator's", int(b.expected[i]), flush=True)
assert (verdicts == b.expected).all()
# a damaged copy of the wire records: pyref's decode flags and decoded records
bad = wire.copy()
c0 = 16 + 512
bad[2 * WB + c0 + 48 * 5 + 9] ^= 0x08        # C_5 of proof 2: another x
bad[7 * WB + 16 + 32 * 3:7 * WB + 16 + 32 * 4] = 0xFF   # input 3 of proof 7: not below r
bad[11 * WB + c0 + 48 * 1 + 47] ^= 0x80      # C_1 of proof 11: the other root (a valid encoding of another point)
dec_ok, dec = np.zeros(N, dtype=np.uint8), np.zeros(N * RB, dtype=np.uint8)
for i in range(N):
    r, ok = P.vs_wire_decode(bad[i * WB:(i + 1) * WB].tobytes(), shd)
    dec[i * RB:(i + 1) * RB] = np.frombuffer(r, dtype=np.uint8)
    dec_ok[i] = ok
assert list(np.flatnonzero(dec_ok == 0)) == [2, 7]
out = os.path.join(os.path.dirname(HERE), "tests", "golden", "vs3_wire_24.npz")
np.savez_compressed(out, vk=b.vk, ext=b.ext, shape=sh, circuits=b.circuits, proofs=b.proofs, randomizers=b.randomizers,
                    pyref_verdicts=verdicts, pyref_wire=wire, damaged_wire=bad, pyref_dec_ok=dec_ok, pyref_decoded=dec)
print("wrote", out, os.path.getsize(out), "bytes")
