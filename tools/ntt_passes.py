"""Round-trip time of a large NTT for the pass structures of csrc/ntt.hip: register radix-16 passes (the default above
2^20), LDS-tiled passes (MI_SNARK_NTT_LDS, an even number of <= 8-bit passes) and any pass count (MI_SNARK_NTT_NP).
Each configuration runs in a child process (the knobs are read per call but the tables are cached).  Checks that the
round trip returns the input.   python tools/ntt_passes.py [LOG_N]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 2 and sys.argv[2] == "child":
    import time, torch
    sys.path.insert(0, ROOT)
    from snarkos_amd import algorithms as A, lib as L
    L.init(0)
    log_n = int(sys.argv[1])
    g = torch.Generator(device="cpu").manual_seed(3)
    a = torch.randint(0, 1 << 62, (1 << log_n, 4), dtype=torch.int64, generator=g)
    a[:, 3] &= (1 << 58) - 1
    a = a.cuda()
    ref = a.clone()
    d = A.EvaluationDomain(1 << log_n)
    for _ in range(3):
        d.fft(a); d.ifft(a)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    R = 10
    for _ in range(R):
        d.fft(a); d.ifft(a)
    torch.cuda.synchronize()
    print("%.4f ms round trip, identity %s" % ((time.perf_counter() - t0) / R * 1e3, bool((a == ref).all().item())))
    sys.exit(0)
log_n = sys.argv[1] if len(sys.argv) > 1 else "22"
for name, env in (("register passes (default)", {}), ("LDS passes, 4", {"MI_SNARK_NTT_LDS": "1"}), ("LDS passes, 3", {"MI_SNARK_NTT_LDS": "1", "MI_SNARK_NTT_NP": "3"}),
                  ("LDS passes, 5", {"MI_SNARK_NTT_LDS": "1", "MI_SNARK_NTT_NP": "5"})):
    e = dict(os.environ); e.update(env)
    out = subprocess.run([sys.executable, os.path.abspath(__file__), log_n, "child"], env=e, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True).stdout.strip()
    print("%-28s %s" % (name, out), flush=True)
