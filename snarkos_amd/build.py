"""Build libmi355x_snark.so (hand-written HIP for gfx950) in-tree with hipcc.

    python -m snarkos_amd.build [--force]

Each translation unit is compiled in parallel to an object file under snarkos_amd/csrc/_obj/ and
linked into snarkos_amd/libmi355x_snark.so.  hipcc cross-compiles gfx950 without a GPU.
"""
import concurrent.futures as cf
import os
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libmi355x_snark.so")
UNITS = ["capi", "msm", "ntt", "pairing", "verify", "stream", "comm", "schnorr"]
PLAIN_UNITS = ["cpu_probe"]  # host-only C++ built WITHOUT the x86-64-v3 / ADX flags (it is what checks for them)
# host side: x86-64-v3 (BMI2 mulx) + ADX for the 64-bit-limb host products; every MI355X host CPU has them
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-march=x86-64-v3", "-madx"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _headers():
    out = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    out.append(os.path.join(os.path.dirname(HERE), "include", "mi355x_snark.h"))
    return out


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def _compile(unit, extra):
    obj = os.path.join(OBJ, unit + ".o")
    t0 = time.time()
    if unit in PLAIN_UNITS:
        src = os.path.join(CSRC, unit + ".cpp")
        cmd = [HIPCC, "-O2", "-std=c++17", "-fPIC", "-Wall", "-x", "c++", "-c", src, "-o", obj]
    else:
        src = os.path.join(CSRC, unit + ".hip")
        cmd = [HIPCC] + FLAGS + extra + ["-c", src, "-o", obj]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1700)
    return unit, r.returncode, r.stdout, time.time() - t0


def build(force=False, verbose=True, extra_flags=()):
    os.makedirs(OBJ, exist_ok=True)
    hdrs = _headers()
    todo = [u for u in UNITS if force or _stale(os.path.join(OBJ, u + ".o"), [os.path.join(CSRC, u + ".hip")] + hdrs)]
    todo += [u for u in PLAIN_UNITS if force or _stale(os.path.join(OBJ, u + ".o"), [os.path.join(CSRC, u + ".cpp")])]
    if todo:
        with cf.ThreadPoolExecutor(max_workers=min(len(todo), os.cpu_count() or 4)) as ex:
            for unit, rc, out, dt in ex.map(lambda u: _compile(u, list(extra_flags)), todo):
                if verbose:
                    print("[build] %-8s %6.1fs rc=%d" % (unit, dt, rc), flush=True)
                if rc != 0:
                    sys.stderr.write(out)
                    raise RuntimeError("hipcc failed on %s.hip" % unit)
                elif verbose and "warning" in out:
                    sys.stderr.write(out)
    objs = [os.path.join(OBJ, u + ".o") for u in UNITS + PLAIN_UNITS]
    if todo or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl", "-lpthread"]
        subprocess.check_call(cmd, timeout=600)
        if verbose:
            print("[build] linked", LIB, flush=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
