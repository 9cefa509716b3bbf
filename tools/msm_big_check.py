"""Exactness of a very large MSM without a CPU Pippenger: the generated points are k_i * G with k_i = splitmix64(seed + i) | 1,
so  sum_i s_i P_i = (sum_i s_i k_i mod r) * G  -- one big-integer dot product on the host, one scalar multiplication by
the oracle.  python tools/msm_big_check.py LOG_N   (ad-hoc; run on the GPU box)"""
import ctypes as C, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
from snarkos_amd import algorithms as A, lib as L
import orc_py as O, pyref as P
from helpers import jac_affine
L.init(0)
n = 1 << int(sys.argv[1]); seed = 0xB16
pts = torch.empty(n * 96, dtype=torch.uint8, device="cuda")
L.check(L.lib().mi_dbg_gen_points_device(C.c_uint64(seed), C.c_size_t(n), L.dptr(pts), L.stream_ptr()), "gen")
g = np.random.default_rng(5)
sc = g.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64); sc[:, 3] &= np.uint64((1 << 60) - 1)
d_sc = torch.from_numpy(sc.view(np.int64)).cuda()
t0 = time.perf_counter(); got = jac_affine(A.VariableBase.msm(pts, d_sc)); dt = time.perf_counter() - t0
def splitmix(x):
    x = (x + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return x ^ (x >> 31)
t = 0
for i in range(n):
    s = int(sc[i, 0]) | (int(sc[i, 1]) << 64) | (int(sc[i, 2]) << 128) | (int(sc[i, 3]) << 192)
    t += s * (splitmix(seed + i) | 1)
want = O.g1_mul(O.g1_generator(), t % P.R)
print("MSM 2^%s: first call %.1f ms, bit-exact vs (sum s_i k_i) * G: %s" % (sys.argv[1], dt * 1e3, bool((got == want).all())))
sys.exit(0 if (got == want).all() else 1)
