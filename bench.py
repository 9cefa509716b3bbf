#!/usr/bin/env python3
"""Benchmark of the batch SNARK verification hot path on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Headline (BASELINE.json metric / configs[3], configs[4] at N = 8): proofs verified per second on a
batch of 1024 synthetic VarunaShaped-v1 proofs PER GPU (weak scaling: N GPUs verify N x 1024 proofs as
one block-validation batch: every rank accumulates its shard, the 592-byte partial accumulators are
all-gathered over RCCL, one pairing product decides the whole batch).  A "step" is one such batch.
Inputs (proofs, randomizers, keys) are resident in HBM before the timed region.

One JSON line on rank 0, with
  roofline     -- the dominant kernel of the timed region: algorithmic bytes / HIP-event time vs 8 TB/s
  cpu_baseline -- the CPU oracle (a port of the reference's per-transaction verify loop) on this host
  extra        -- MSM 2^20 points/s (configs[1]) and NTT 2^22 round trip (configs[2]) on the same GPU,
                  and the batch with 1 % corrupted proofs (fallback that names the bad proofs)
"""
import argparse
import json
import os
import sys
import time

# streams of this process (torch's, 8 accumulation lanes, 8 in-flight pairing products) should not share
# hardware queues: a 3 ms pairing kernel must not sit in front of the next batch's MSM kernels
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
# RCCL prints a version banner on stdout at init when NCCL_DEBUG asks for it: the driver reads ONE JSON line from there
os.environ["NCCL_DEBUG"] = os.environ.get("MI_BENCH_NCCL_DEBUG", "WARN")

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from snarkos_amd import algorithms as A  # noqa: E402
from snarkos_amd import lib as L  # noqa: E402
from snarkos_amd import sharding  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 TB/s measured copy)
PROOFS_PER_GPU = 1024
# algorithmic bytes per unit of each profiled kernel (DESIGN.md "Measurement")
ALG_BYTES = {
    "msm_accumulate": 128.0,        # per (point, scalar) pair: 96 B affine point + 32 B scalar
    "msm_bitsums": 192.0,           # per chunk partial (XYZZ, 4 x 48 B); their number per pass comes from the plan: see main()
    "vs_derive": 2256.0 + 32.0,     # per proof: the proof record + its two 128-bit randomizers
    "pairing": 96.0 + 13264.0,      # per pair: one G1 point + one prepared G2 line table
    "ntt_pass": 64.0,               # per element per pass: 32 B read + 32 B written
}


def load_fixture():
    d = np.load(os.path.join(ROOT, "tests", "golden", "vs_batch_1024.npz"))
    return d["vk"], d["circuits"], d["proofs"], d["randomizers"]


def sync_all(dist_on):
    torch.cuda.synchronize()
    if dist_on:
        torch.distributed.barrier()
    torch.cuda.synchronize()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-extra", action="store_true", help="skip the MSM / NTT / corrupted-batch side measurements")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--acc-depth", type=int, default=0,
                    help="accumulation lanes (lane passes on the GPU before the oldest is folded, <= 14); default 12, or 8 when RCCL needs queues too")
    ap.add_argument("--exchange-group", type=int, default=0,
                    help="batches per partial exchange when sharded (fewer, larger all-gathers); default 4 with a process group")
    ap.add_argument("--fold-threads", type=int, default=3, help="host threads folding finished accumulations")
    ap.add_argument("--products", type=int, default=0, help="pairing products in flight (<= 16); default 8, or 6 when RCCL needs queues too")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for rehearsals on one GPU)")
    ap.add_argument("--exchange", default="auto", choices=["auto", "native", "torch"],
                    help="partial exchange when sharded: the library's own RCCL all-gather (native) or the torch.distributed "
                         "callback; auto = native when RCCL can be initialised from the library, else torch")
    ap.add_argument("--single-process", action="store_true",
                    help="ONE process drives --gpus N devices: mi_init(device list), one 1024 x N-proof block per step through "
                         "mi_vs_verify_block_device (in-library sharding, RCCL exchange from C++, one pairing product)")
    args = ap.parse_args()
    if args.single_process:
        return single_process(args)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.acc_depth <= 0:
        # HIP maps streams onto 24 hardware queues and the pipeline falls off a cliff when its streams share them: 20
        # streams are the library's budget.  Under load a lane pass carries two batches and a product ticket up to four
        # products (measured, 400 / 20 batches: 10 + 10 -> 7.4e6 / 3.7e6 proofs/s, 12 + 8 -> 7.8e6 / 3.85e6, 14 + 6 ->
        # 8.0e6 / 2.7e6, 10 + 12 -> 3.6e6); with a process group RCCL's own streams must fit too
        args.acc_depth = 12 if world == 1 else 8  # measured with an RCCL group up: 12 -> 0.64, 10 -> 0.29, 8 -> 0.31 ms/batch
    os.environ["MI_SNARK_VS_LANES"] = str(args.acc_depth)  # read when the verifier context is created
    if args.products <= 0:
        args.products = 8 if world == 1 else 6  # 8 lanes + 6 product streams: room for RCCL's own queues below the 24-queue cliff
    os.environ["MI_SNARK_VS_SLOTS"] = str(args.products)
    if args.exchange_group <= 0:
        args.exchange_group = 4 if world > 1 else 1
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist_on = world > 1
    dev_index = local_rank % max(1, torch.cuda.device_count())  # rehearsals may put several ranks on one GPU
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=dev)
        else:
            torch.distributed.init_process_group(args.backend)
    L.init(dev_index)
    gather_dev = dev if args.backend == "nccl" else None

    vk, circuits, proofs, rnd = load_fixture()
    # every rank holds its own shard of the block's proofs; randomizers differ per rank (caller RNG)
    rnd = rnd.copy()
    rnd[::16] ^= np.uint8(rank)
    d_proofs = torch.from_numpy(proofs).to(dev)
    d_rnd = torch.from_numpy(rnd).to(dev)
    verifier = A.VarunaShaped(vk, circuits)

    # The K batches are verified as a stream (chunks of a block / mempool traffic): up to 8 batches are accumulating
    # on their own stream lanes while earlier ones are folded on host threads and up to 8 product tickets (four products each at most) run on
    # their own streams.  Every batch is fully verified inside the timed region; "latency_ms" below is one batch on
    # its own, unpipelined.
    def gather_all(part):  # runs on verify_stream's finisher thread: every collective comes from that one thread
        torch.cuda.set_device(dev_index)
        part = np.ascontiguousarray(part, dtype=np.uint8)  # (count, 592): one collective per group of batches
        return sharding.gather_bytes(part, part.size, torch.distributed, gather_dev).reshape(world, -1, L.VS_PARTIAL_BYTES)

    gather = gather_all if dist_on else None
    # the exchange itself: RCCL called from the library (mi_comm_init_rank + mi_vs_exchange_rccl: an ncclAllGather from
    # the native driver's finisher thread); the unique id travels over torch.distributed once.  Falls back to the
    # torch.distributed callback when the library cannot bring RCCL up on every rank.
    comm = None
    if dist_on and args.exchange != "torch":
        comm = native_comm(rank, world, dev if args.backend == "nccl" else "cpu")
        if comm is None and args.exchange == "native":
            raise SystemExit("native RCCL exchange requested but not available on every rank")
    # (the transport behind a native communicator is the library's decision, taken by all ranks at mi_comm_init_rank:
    # RCCL when it comes up everywhere, the shared-memory all-gather between the node's processes otherwise)
    if not dist_on and args.exchange == "native":
        # one rank with the exchange path switched on (a communicator of world 1): what the grouped exchange of the
        # streaming driver costs by itself, apart from what several ranks do to each other
        comm = L.comm_init_rank(L.comm_unique_id(), 0, 1)
    exchange_kind = ("native-" + L.comm_transport(comm)) if comm is not None else ("torch.distributed" if dist_on else "none")
    # the stream driver merges the batches of a pass when nothing is exchanged between ranks (csrc/stream.hip)
    merged_passes = exchange_kind == "none" and os.environ.get("MI_SNARK_STREAM_MERGE", "1")[:1] != "0"
    stamps = []

    def run(k, stamp=False):
        oks = []
        for ok in verifier.verify_stream(((d_proofs, d_rnd) for _ in range(k)), None if comm is not None else gather,
                                         depth=args.products, acc_depth=args.acc_depth, fold_threads=args.fold_threads,
                                         exchange_group=args.exchange_group, comm=comm):
            oks.append(ok)
            if stamp:
                stamps.append(time.perf_counter())
        assert len(oks) == k and all(oks), "synthetic batch must verify"

    trace = (lambda m: print("[bench] " + m, file=sys.stderr, flush=True)) if os.environ.get("MI_BENCH_TRACE") else (lambda m: None)
    trace("warm-up stream")
    if args.warmup:
        run(args.warmup)
    trace("timed stream")
    # ---- timed region: exactly K batches, no profiling hooks
    L.prof_enable(0)
    sync_all(dist_on)
    st0 = L.stats_read()
    t0 = time.perf_counter()
    run(args.steps)
    sync_all(dist_on)
    dt = time.perf_counter() - t0
    st1 = L.stats_read()
    if dist_on:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    # ---- outside the timed region: the steady state of the same stream (pipeline full).  K batches include ~2 ms of
    # pipeline fill and drain -- a third of the timed region at K = 20 -- so a longer stream is run once more and the
    # verdict-to-verdict spacing of its middle half reported beside ms_per_step.
    trace("steady-state stream")
    run(max(args.steps, 240), stamp=True)
    sync_all(dist_on)
    trace("bracketed stream")
    lo, hi = len(stamps) // 4, len(stamps) - 1 - len(stamps) // 4
    steady = 1e3 * (stamps[hi] - stamps[lo]) / (hi - lo)
    # ---- the same K batches again with HIP-event brackets around the headline kernels, on the streams they run on
    # (the brackets cost host time, so they stay out of the timed region; kernel durations agree with
    # profiles/*_rocprofv3_kernel_stats.csv)
    L.prof_enable(1)
    for k in ALG_BYTES:
        L.prof_read(k)
    L.prof_read("msm_total")
    run(args.steps)
    sync_all(dist_on)
    L.prof_enable(False)
    trace("streams done")

    kernels = {}
    for k, per_unit in ALG_BYTES.items():
        ms, launches, units = L.prof_read(k)
        if launches:
            kernels[k] = {"total_ms": ms, "launches": launches, "avg_us": 1e3 * ms / launches,
                          "units_per_launch": units / launches,
                          "achieved_GBps": per_unit * units / (ms * 1e-3) / 1e9 if ms > 0 else None}
    ms_tot, n_msm, _ = L.prof_read("msm_total")
    if n_msm:
        kernels["msm_total"] = {"total_ms": ms_tot, "launches": n_msm, "avg_us": 1e3 * ms_tot / n_msm}

    if rank != 0:
        if dist_on:
            torch.distributed.destroy_process_group()
        return

    chip = chip_alu_view(kernels, steady, merged_passes, st1["pairing_products"] - st0["pairing_products"], args.steps)
    if chip and "msm_bitsums" in kernels:
        # the bit sums' unit is a chunk partial (192 B, read by the groups whose bit it carries; counted once): their number
        # per pass comes from the library's plan, the profiler only knows the launch's points
        kb = kernels["msm_bitsums"]
        kb["chunk_partials_per_launch"] = chip["plan"]["chunk_partials_per_pass"]
        kb["achieved_GBps"] = ALG_BYTES["msm_bitsums"] * kb["chunk_partials_per_launch"] / (kb["avg_us"] * 1e-6) / 1e9
    dom = max((k for k in kernels if k in ALG_BYTES and (k != "msm_bitsums" or chip)), key=lambda k: kernels[k]["total_ms"])
    pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    traffic = None
    if os.path.exists(pmc_path):
        try:
            traffic = json.load(open(pmc_path)).get({"msm_bitsums": "msm_bit_sums_queue"}.get(dom, dom))
        except Exception:
            traffic = None
    roofline = {"bound": "hbm", "kernel": dom, "achieved": kernels[dom]["achieved_GBps"], "peak": HBM_PEAK_GBPS,
                "unit": "GB/s", "frac": kernels[dom]["achieved_GBps"] / HBM_PEAK_GBPS,
                "avg_launch_us": kernels[dom]["avg_us"],
                "algorithmic_bytes_per_launch": ALG_BYTES[dom] * kernels[dom].get("chunk_partials_per_launch", kernels[dom]["units_per_launch"]),
                "traffic": traffic}
    if "msm_accumulate" in kernels:
        # SURVEY.md 8d's per-proof figure for the verifier (128 B x the proof's (point, scalar) pairs after vk merging):
        # the MSM accumulation kernel's view, beside the dominant-by-time kernel above
        ka = kernels["msm_accumulate"]
        roofline["msm_accumulate_view"] = {
            # a launch serves one lane pass = one or two batches: proofs per launch from the derive kernel's units
            "points_per_proof": ka["units_per_launch"] / (kernels["vs_derive"]["units_per_launch"] if "vs_derive" in kernels else PROOFS_PER_GPU),
            "proofs_per_launch": kernels["vs_derive"]["units_per_launch"] if "vs_derive" in kernels else PROOFS_PER_GPU,
            "algorithmic_bytes_per_launch": 128.0 * ka["units_per_launch"],
            "avg_launch_us": ka["avg_us"], "achieved": ka["achieved_GBps"], "frac": ka["achieved_GBps"] / HBM_PEAK_GBPS,
            "note": "integer-ALU bound: see extra.msm_2^20.accumulate_frac_of_measured_alu_peak for the same kernel at full occupancy"}

    out = {
        "metric": "varuna_proofs_verified_per_sec", "value": world * PROOFS_PER_GPU * args.steps / dt,
        "unit": "proofs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps, "steady_state_ms_per_step": steady,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "integer limbs, Montgomery (Fq 12x32-bit, and 14 signed 28-bit limbs in the MSM / transcript kernels; Fr 8x32-bit)", "data": "synthetic",
        "config": {"workload": "configs[3]: batch verify of 1024 synthetic VarunaShaped-v1 proofs per GPU "
                               "(KZG open + pairing check); N GPUs = one %d-proof block sharded, partials all-gathered over RCCL" % (world * PROOFS_PER_GPU),
                   "proofs_per_gpu": PROOFS_PER_GPU, "circuits": 8, "all_valid": True, "parallelism": "shard%d" % world,
                   "exchange": exchange_kind, "exchange_group": args.exchange_group,
                   # the all-gathers of the timed region on rank 0's finisher thread (native transports; the torch callback is
                   # outside the library's counters): how long a group's exchange held the thread that launches products
                   "exchange_calls": st1["exchange_calls"] - st0["exchange_calls"],
                   "exchange_wait_ms_per_group": (1e-6 * (st1["exchange_nanos"] - st0["exchange_nanos"]) / (st1["exchange_calls"] - st0["exchange_calls"])
                                                  if st1["exchange_calls"] > st0["exchange_calls"] else None),
                   "pipeline": "batches streamed: %d accumulation lanes, %d pairing products in flight, %d host fold threads"
                               % (args.acc_depth, args.products, args.fold_threads),
                   # without an exchange the driver folds the batches of a lane pass (up to four under load) into ONE
                   # aggregate: one MSM pair, one host fold, one pairing product for all of them
                   "merged_passes": merged_passes,
                   "pairing_products_in_timed_region": st1["pairing_products"] - st0["pairing_products"]},
        "roofline": roofline, "alu": alu_view(kernels), "alu_chip": chip, "kernels": kernels,
    }
    roofline["traffic_source"] = "profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/profile_round.sh, committed; not measured in this run)"
    if world == 1:
        trace("one batch on its own")
        verifier.verify_batch(d_proofs, d_rnd)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            ok, verdicts = verifier.verify_batch(d_proofs, d_rnd)
        torch.cuda.synchronize()
        out["latency_ms_one_batch_unpipelined"] = 1e3 * (time.perf_counter() - t0) / 10
        assert ok and verdicts.all()
        t0 = time.perf_counter()
        for _ in range(5):
            verifier.accumulate(d_proofs, d_rnd)
        out["accumulate_stage_ms"] = 1e3 * (time.perf_counter() - t0) / 5
        trace("latency done")

    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(vk, circuits, proofs, rnd)
    if world == 1 and not args.no_extra:
        out["extra"] = extras(dev, verifier, proofs, rnd)
        if not args.no_cpu_baseline:
            out["extra"]["cpu_side_by_side"] = cpu_side_by_side(dev)
    print(json.dumps(out), flush=True)
    if dist_on:
        torch.distributed.destroy_process_group()


# ---- integer-ALU view of the dominant kernel (the verifier is latency / ALU bound: SURVEY.md 8d)
# A pairing product is ONE workgroup on one compute unit: three waves carry the twelve coefficients of the Fq12 value, a
# fourth its scale (pairing_wg.hpp), after a short first launch of 21 workgroups that prepare the merged Miller factors
# (~30 us, bracketed and counted separately).  Work of the product kernel in 32x32-bit limb products (v_mad_u64_u32) on the
# 192 coefficient lanes: 127 general Fq12 rounds (80 Miller, 47 in the final exponentiation) x 179 limb products (108
# product + 48 fold + 23 Barrett) and the 315 cyclotomic squaring rounds of the hard part x 119 (12 to scale the operand
# + 36 product + 48 + 23).  Peak for the waves the kernel owns: tools/ubench at one wave per SIMD sustains 50.25 G Fq
# products/s chip-wide = 49.1 M products/s per wave = 14.1 G limb products/s per wave.
PAIRING_LIMB_PRODUCTS = 192 * (127 * 179 + 315 * 119)
PAIRING_WAVES = 4
WAVE_LIMB_PRODUCT_RATE = 50.25e9 / 1024 * 288


def alu_view(kernels):
    k = kernels.get("pairing")
    if not k:
        return None
    t = k["avg_us"] * 1e-6
    ach = PAIRING_LIMB_PRODUCTS / t
    return {"kernel": "pairing", "unit": "32x32-bit limb products/s", "work_per_launch": PAIRING_LIMB_PRODUCTS,
            "achieved": ach, "waves": PAIRING_WAVES, "peak_for_those_waves": PAIRING_WAVES * WAVE_LIMB_PRODUCT_RATE,
            "frac": ach / (PAIRING_WAVES * WAVE_LIMB_PRODUCT_RATE),
            "note": "one workgroup (4 waves, one CU: three carry the value, the fourth its scale) per product (its 21-workgroup line launch is bracketed separately); a launch carries "
                    "one to four products (a workgroup each: same duration, work and waves scale together, so the fraction is per "
                    "product); several launches run side by side in the streamed region, so avg_us includes their mutual slowdown"}


# ---- chip-level ALU view of the streamed steady state: how much of the device's integer throughput a batch uses.
# Work per batch in Fq-product equivalents on 28-bit limbs (one product = 14 x 14 v_mad_i64_i32 + reduction), counted from
# the launch's own numbers (points per pass from the profiler's units, window width / windows / chunk length from the
# library's plan):
#   accumulate  one mixed addition (8 products + 2 squarings) per non-zero digit: points x windows x (1 - 2^-c); the
#               128-bit randomizer side has ceil(129 / c) live windows
#   bit sums    per (window, bit) group one wave: ceil(matches / 64) queue additions + 6 tree levels, each a full XYZZ
#               addition (12 products + 2 squarings) on 64 lanes; matches = half the window's chunks -- or, for the long
#               lists of merged passes, half the entries of the list the gather stage leaves (one addition per chunk first)
#   derive      ~165 products per proof on 32-bit limbs (11 curve checks, ~250 Fr products, the SHA-256 transcript)
#   pairing     PAIRING_LIMB_PRODUCTS (above) / 288 per product, on 32-bit limbs
# against the rates tools/ubench.hip measures for a tight loop over the whole chip (8 waves/SIMD): 79.9 G products/s and
# 96.7 G squarings/s on 28-bit limbs, 60 G products/s on 32-bit limbs.
CHIP_RATE_28, CHIP_SQR_28, CHIP_RATE_32 = 79.9e9, 96.7e9, 60.0e9


def chip_alu_view(kernels, steady_ms, merged=False, products=None, steps=None):
    import ctypes as C
    if "msm_accumulate" not in kernels or "vs_derive" not in kernels:
        return None
    pts = kernels["msm_accumulate"]["units_per_launch"]        # points of all MSMs of a lane pass
    proofs = kernels["vs_derive"]["units_per_launch"]          # proofs of a lane pass (one or two batches)
    batches = proofs / PROOFS_PER_GPU
    c, nwin, chunk, small = C.c_int32(0), C.c_int32(0), C.c_int32(0), C.c_int32(0)
    L.check(L.lib().mi_dbg_msm_plan(C.c_size_t(int(round(pts))), C.c_int32(254), C.byref(c), C.byref(nwin), C.byref(chunk), C.byref(small)), "mi_dbg_msm_plan")
    c, nwin, chunk = c.value, nwin.value, chunk.value
    n_w = 2.0 * proofs                                           # the tau-H side: two witnesses per proof, 128-bit scalars
    n_h = pts - n_w
    nwin_w = -(-129 // c)
    live = 1.0 - 2.0 ** -c
    sq = CHIP_RATE_28 / CHIP_SQR_28                              # a squaring in product equivalents
    adds = (n_h * nwin + n_w * nwin_w) * live
    acc = adds * (8 + 2 * sq)
    buckets = 2 ** (c - 1)
    partials = [0.0]                                             # chunk partials of the pass (what the reduction reads)
    # long chunk lists are shortened first (csrc/msm.hip msm_gather_chunks_kernel, the library's own rule): one addition per
    # chunk partial, up to GATHER consecutive ones of a bucket per lane; the groups then sum the shorter list
    GATHER = 8
    # (msm.hip: points x windows / chunk length / all problems' windows >= 1000; a pass has two problems per aggregate)
    gather = pts / (2.0 * chunk * (1.0 if merged else max(1.0, batches))) >= 1000.0
    def groups_work(n_points, windows, times=1.0):               # per problem
        per_bucket = n_points * live / buckets
        chunks = buckets * max(1.0, per_bucket / chunk + 0.5)
        partials[0] += times * windows * chunks
        entries, extra = chunks, 0.0
        if gather:
            entries = buckets * max(1.0, chunks / buckets / GATHER + 0.5)
            extra = windows * chunks * (12 + 2 * sq)            # the gather stage's additions
        wave_adds = -(-(entries / 2) // 64) + 6
        return times * (windows * c * wave_adds * 64 * (12 + 2 * sq) + extra)
    if merged:  # the pass is ONE problem pair: its groups sum the chunks of all its batches
        bits = groups_work(n_h, nwin) + groups_work(n_w, nwin_w)
    else:
        bits = groups_work(n_h / batches, nwin, batches) + groups_work(n_w / batches, nwin_w, batches)
    derive = 165.0 * proofs * CHIP_RATE_28 / CHIP_RATE_32
    products_per_pass = batches if not (products and steps) else batches * products / float(steps)
    pairing = products_per_pass * PAIRING_LIMB_PRODUCTS / 288.0 * CHIP_RATE_28 / CHIP_RATE_32
    per_batch = (acc + bits + derive + pairing) / batches
    out = {"unit": "Fq products/s (28-bit-limb equivalents)", "work_per_batch": per_batch,
           "split_per_batch": {"msm_accumulate": acc / batches, "msm_bit_sums": bits / batches, "vs_derive": derive / batches, "pairing": pairing / batches},
           "plan": {"points_per_pass": pts, "batches_per_pass": batches, "c": c, "windows": nwin, "chunk": chunk,
                    "chunk_partials_per_pass": partials[0], "gather_stage": bool(gather)},
           "steady_state_ms_per_batch": steady_ms, "achieved": per_batch / (steady_ms * 1e-3), "peak": CHIP_RATE_28,
           "frac": per_batch / (steady_ms * 1e-3) / CHIP_RATE_28,
           "note": "the whole chip over the middle of the stream; counts are analytic (formulas above chip_alu_view in bench.py), the "
                   "host fold of the bit sums is not GPU work"}
    ka = kernels["msm_accumulate"]
    out["msm_accumulate_in_stream"] = {"avg_us": ka["avg_us"], "products_per_launch": acc, "achieved": acc / (ka["avg_us"] * 1e-6),
                                       "note": "several passes share the device: this launch's own rate, not a fraction of the chip"}
    if "msm_bitsums" in kernels:
        kb = kernels["msm_bitsums"]
        out["msm_bit_sums_in_stream"] = {"avg_us": kb["avg_us"], "products_per_launch": bits, "achieved": bits / (kb["avg_us"] * 1e-6),
                                         "waves": bits / (64 * (12 + 2 * sq)) / (-(-(1) // 1)) and None}
        out["msm_bit_sums_in_stream"].pop("waves")
    return out


def native_comm(rank, world, flag_device):
    """Bring up the library's own RCCL communicator on every rank (the unique id travels over torch.distributed once) and
    prove it with one all-gather before the timed region relies on it.  Each step runs on a watchdog thread: a rank where
    RCCL does not come up within the limit votes no, and every rank then keeps the torch.distributed callback."""
    import ctypes as C
    import threading

    state = {"comm": None, "ok": 0}

    def bring_up(uid):
        try:
            c = L.comm_init_rank(uid, rank, world)
            mine = np.full(8, rank, dtype=np.uint8)
            out = np.zeros(8 * world, dtype=np.uint8)
            L.check(L.lib().mi_comm_all_gather(C.c_uint64(c), L.ptr(mine), C.c_size_t(8), L.ptr(out)), "mi_comm_all_gather")
            if (out.reshape(world, 8) == np.arange(world, dtype=np.uint8)[:, None]).all():
                state["comm"], state["ok"] = c, 1
        except Exception:  # noqa: BLE001 -- any failure means: keep the torch exchange
            pass

    uid = [None]
    if rank == 0:
        try:
            uid[0] = L.comm_unique_id()
        except L.MiError:
            uid[0] = None
    torch.distributed.broadcast_object_list(uid, src=0)
    if uid[0] is not None:
        t = threading.Thread(target=bring_up, args=(uid[0],), daemon=True)
        t.start()
        t.join(90.0)
    flag = torch.tensor([state["ok"]], dtype=torch.int32, device=flag_device)
    torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
    return state["comm"] if int(flag.item()) == 1 else None


def single_process(args):
    """One process, N devices: a block of N x 1024 proofs per step, shards resident on their devices."""
    n_dev = args.gpus
    assert torch.cuda.device_count() >= n_dev, "not enough visible GPUs"
    L.init(list(range(n_dev)))
    vk, circuits, proofs, rnd = load_fixture()
    verifier = A.VarunaShaped(vk, circuits)
    shards = []
    for d in range(n_dev):
        r = rnd.copy()
        r[::16] ^= np.uint8(d)
        dev = torch.device("cuda", d)
        shards.append((torch.from_numpy(proofs).to(dev), torch.from_numpy(r).to(dev)))
    for d in range(n_dev):
        torch.cuda.synchronize(d)
    for _ in range(max(1, args.warmup)):
        ok, _ = verifier.verify_block_device(shards)
        assert ok
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ok, verdicts = verifier.verify_block_device(shards)
    dt = time.perf_counter() - t0
    assert ok and verdicts.all()
    out = {"metric": "varuna_proofs_verified_per_sec", "value": n_dev * PROOFS_PER_GPU * args.steps / dt, "unit": "proofs/s",
           "n_gpus": n_dev, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "integer limbs, Montgomery (Fq 12x32-bit, and 14 signed 28-bit limbs in the MSM / transcript kernels; Fr 8x32-bit)",
           "data": "synthetic",
           "config": {"workload": "configs[4] shape in ONE process: a %d-proof block per step, mi_vs_verify_block_device over %d "
                                  "device(s): one host thread and one 1024-proof shard per device, partials all-gathered (RCCL from the "
                                  "library), one pairing product; blocks are sequential (SURVEY.md 3.3), so this is block LATENCY, not "
                                  "the streamed rate" % (n_dev * PROOFS_PER_GPU, n_dev),
                      "proofs_per_gpu": PROOFS_PER_GPU, "parallelism": "single-process x%d" % n_dev}}
    print(json.dumps(out), flush=True)


def usable_cores():
    """Cores this process may actually use: affinity mask, capped by the cgroup CPU quota if there is one."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, q // per))
        except Exception:
            pass
    return n


def cpu_baseline(vk, circuits, proofs, rnd):
    """The oracle's port of the reference shape: one proof at a time, per-proof pairing product, one
    thread per core (the reference runs rayon over transactions on all cores)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orc_py as O

    cores = usable_cores()
    b = O.VSArrays(vk, circuits, proofs, rnd)
    b_small = O.VSArrays(vk, circuits, proofs[:64 * O.VS_PROOF_BYTES], rnd[:64 * 32])
    b_small.verify_cpu(cores)  # warm-up (fixed-base tables, page-in)
    times = []
    for _ in range(5):  # median of 5 after the warm-up
        t0 = time.perf_counter()
        v = b.verify_cpu(cores)
        times.append(time.perf_counter() - t0)
        assert v.all()
    dt = sorted(times)[2]
    # beside it, the GPU path's own algorithm on the CPU (one aggregated MSM pair + ONE pairing product per batch):
    # what the host cores reach when handed the same restructuring; not the reference's shape
    b.verify_aggregated(cores)
    t1 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        ok = b.verify_aggregated(cores)[0]
    dta = (time.perf_counter() - t1) / reps
    assert ok
    return {"value": b.n / dt, "unit": "proofs/s", "cores": cores, "kind": "port",
            "sample": "all 1024 proofs of the batch, oracle orc_vs_verify_batch_cpu, %d OpenMP threads, median of 5 runs, %.2f s each" % (cores, dt),
            "aggregated_algorithm_value": b.n / dta,
            "aggregated_note": "same oracle, batch folded into one MSM pair + one pairing product (the GPU path's algorithm), %d threads" % cores}


def cpu_side_by_side(dev):
    """configs[0] (the reference's CPU-runnable MSM, 2^16 points) and a 2^20 NTT: oracle on the host cores beside the
    HIP path, same inputs, results compared bit for bit."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ctypes as C
    import orc_py as O

    cores = usable_cores()
    out = {"cores": cores}
    n = 1 << 16
    pts = torch.empty(n * 96, dtype=torch.uint8, device=dev)
    L.check(L.lib().mi_dbg_gen_points_device(C.c_uint64(0xA1E00001), C.c_size_t(n), L.dptr(pts), L.stream_ptr()), "gen_points")
    g = torch.Generator(device="cpu").manual_seed(0xA1E00001)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from uniform_scalars import uniform_scalars
    sc = uniform_scalars(n, 0xA1E00001)
    d_sc = sc.to(dev)
    torch.cuda.synchronize()
    A.VariableBase.msm(pts, d_sc)
    t0 = time.perf_counter()
    for _ in range(5):
        got = A.VariableBase.msm(pts, d_sc)
    torch.cuda.synchronize()
    t_gpu = (time.perf_counter() - t0) / 5
    h_pts = pts.cpu().numpy().view(np.uint64).reshape(n, 12)
    h_sc = sc.numpy().view(np.uint64)
    O.msm_pippenger(h_pts[:1024], h_sc[:1024], cores)
    t0 = time.perf_counter()
    want = O.msm_pippenger(h_pts, h_sc, cores)
    t_cpu = time.perf_counter() - t0
    out["msm_2^16"] = {"gpu_ms": 1e3 * t_gpu, "cpu_ms": 1e3 * t_cpu, "gpu_points_per_s": n / t_gpu, "cpu_points_per_s": n / t_cpu,
                       "bit_exact": bool((got[:12] == want).all() and got[12:].any())}
    m = 1 << 20
    a = torch.randint(0, 1 << 62, (m, 4), dtype=torch.int64, generator=g)
    a[:, 3] &= (1 << 58) - 1
    d_a = a.to(dev)
    dom = A.EvaluationDomain(m)
    dom.fft(d_a); dom.ifft(d_a)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        dom.fft(d_a)
        dom.ifft(d_a)
    torch.cuda.synchronize()
    t_gpu = (time.perf_counter() - t0) / 10
    h_a = a.numpy().view(np.uint64)
    t0 = time.perf_counter()
    f = O.ntt(h_a, threads=cores)
    t_cpu = time.perf_counter() - t0
    dom.fft(d_a)
    torch.cuda.synchronize()
    out["ntt_2^20_forward"] = {"gpu_ms": 1e3 * t_gpu, "cpu_ms": 1e3 * t_cpu,
                               "bit_exact": bool((d_a.cpu().numpy().view(np.uint64) == f).all())}
    return out


def units_adds(n, nwin):
    """mixed additions of the bucket accumulation: one per non-zero digit (all but ~2^-16 of them at c = 16)"""
    return float(n) * nwin


def extras(dev, verifier, proofs, rnd):
    ex = {}
    st = torch.cuda.current_stream()
    # --- configs[1]: 2^20-point G1 MSM, inputs resident
    n = 1 << 20
    pts = torch.empty(n * 96, dtype=torch.uint8, device=dev)
    import ctypes as C
    L.check(L.lib().mi_dbg_gen_points_device(C.c_uint64(0xA1E00002), C.c_size_t(n), L.dptr(pts), L.stream_ptr(st)), "gen_points")
    g = torch.Generator(device="cpu").manual_seed(0xA1E00002)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from uniform_scalars import uniform_scalars
    sc = uniform_scalars(n, 0xA1E00002).to(dev)  # uniform in [0, r): the top window sees full 253-bit digits
    torch.cuda.synchronize()
    A.VariableBase.msm(pts, sc)
    L.prof_enable(True)
    L.prof_read("msm_accumulate"); L.prof_read("msm_total")
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        res = A.VariableBase.msm(pts, sc)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    ms, launches, units = L.prof_read("msm_accumulate")
    ms_t, l_t, _ = L.prof_read("msm_total")
    ex["msm_2^20"] = {"points_per_s": n / dt, "ms": 1e3 * dt, "accumulate_kernel_ms": ms / launches,
                      "gpu_pipeline_ms": ms_t / l_t,
                      "roofline_frac_hbm": 128.0 * n / dt / 1e9 / HBM_PEAK_GBPS,
                      # integer-ALU view of the dominant kernel: 8 Fq products + 2 squarings per mixed addition, one
                      # addition per (point, window) digit, against the rates measured in a tight loop for the reduced-radix
                      # arithmetic the kernel uses (tools/ubench, 8 waves/SIMD: 79.9 G products/s, 96.7 G squarings/s)
                      "accumulate_fq_products_per_s": 10.0 * units_adds(n, 16) / (ms / launches * 1e-3),
                      "accumulate_frac_of_measured_alu_peak": units_adds(n, 16) * (8.0 / 79.9e9 + 2.0 / 96.7e9) / (ms / launches * 1e-3),
                      "accumulate_vs_32bit_limb_product_rate": 10.0 * units_adds(n, 16) / (ms / launches * 1e-3) / 56.6e9,
                      "result_is_infinity": bool(not res[12:].any())}
    # --- 8f row 3: KZG10::commit of a 2^20-coefficient polynomial (Montgomery coefficients resident, powers resident):
    # coefficient conversion + the same MSM
    rb = A.ResidentBases(pts.cpu().numpy().view(np.uint64).reshape(n, 12))
    rb.commit(sc)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        rb.commit(sc)
    torch.cuda.synchronize()
    dtc = (time.perf_counter() - t0) / reps
    ex["kzg_commit_2^20"] = {"ms": 1e3 * dtc, "coefficients_per_s": n / dtc}
    rb.release()
    del pts, sc
    # --- configs[2]: NTT 2^22 forward + inverse round trip, in place in HBM
    log_n = 22
    a = torch.randint(0, 1 << 62, (1 << log_n, 4), dtype=torch.int64, generator=g)
    a[:, 3] &= (1 << 58) - 1
    a = a.to(dev)
    ref = a.clone()
    d = A.EvaluationDomain(1 << log_n)
    d.fft(a); d.ifft(a)
    torch.cuda.synchronize()
    assert torch.equal(a, ref), "ifft(fft(a)) != a"
    L.prof_read("ntt_pass")
    t0 = time.perf_counter()
    for _ in range(reps):
        d.fft(a); d.ifft(a)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    ms, launches, _ = L.prof_read("ntt_pass")
    L.prof_enable(False)
    assert torch.equal(a, ref)
    ex["ntt_2^22_roundtrip"] = {"ms": 1e3 * dt, "elements_per_s": 2 * (1 << log_n) / dt, "pass_kernel_avg_ms": ms / launches,
                                "passes_per_transform": launches // (2 * reps),
                                "roofline_frac_hbm": 2 * 64.0 * (1 << log_n) / dt / 1e9 / HBM_PEAK_GBPS}
    del a, ref
    # --- row A9: Poseidon sponge over Fq, 65536 independent sponges absorbing 64 elements each (32 permutations)
    npos, mpos = 1 << 16, 64
    x = torch.randint(0, 1 << 62, (npos, mpos, 6), dtype=torch.int64, generator=g)
    x[:, :, 5] &= (1 << 55) - 1  # < q
    x = x.to(dev)
    A.PoseidonSponge.hash_many(x, 2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    A.PoseidonSponge.hash_many(x, 2)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ex["poseidon_fq"] = {"sponges": npos, "absorbed_each": mpos, "ms": 1e3 * dt, "permutations_per_s": npos * (mpos // 2) / dt,
                         "fq_products_per_s": npos * (mpos // 2) * 626 / dt}
    del x
    # --- compressed points: the 11 264 commitments of a 1024-proof batch (9 + 2 per proof) decoded in one launch
    # (square root in Fq, sign, subgroup check), inputs resident; the points are the batch's own commitments
    try:
        pr = proofs.reshape(-1, L.VS_PROOF_BYTES)
        off_c, off_w = 16 + 512, 16 + 512 + 864 + 672
        pts_b = np.concatenate([pr[:, off_c:off_c + 864].reshape(-1, 96), pr[:, off_w:off_w + 192].reshape(-1, 96)])
        comp = A.G1Codec.compress(pts_b.view(np.uint64).reshape(-1, 12))
        d_comp = torch.from_numpy(comp).to(dev)
        codec = {"points": int(len(comp))}
        for check, key in ((True, "ms_with_subgroup_check"), (False, "ms_without_subgroup_check")):
            outp, okp = A.G1Codec.decompress(d_comp, check)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                outp, okp = A.G1Codec.decompress(d_comp, check)
            torch.cuda.synchronize()
            codec[key] = 1e3 * (time.perf_counter() - t0) / 5
        codec["all_decoded"] = bool(okp.all().item()) and bool((outp.cpu().numpy().view(np.uint64) == pts_b.view(np.uint64).reshape(-1, 12)).all())
        codec["points_per_s_checked"] = len(comp) / (1e-3 * codec["ms_with_subgroup_check"])
        ex["g1_decompress_batch_commitments"] = codec
    except Exception as e:  # noqa: BLE001
        ex["g1_decompress_batch_commitments"] = {"error": str(e)}
    # --- VarunaShaped-v3, the records as the node holds them (MI_VS_WIRE): the headline batch with every commitment a
    # 48-byte compressed point and every field element canonical, decoded INSIDE the timed stream's lane passes
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import wire_one
        ex["batch_1024_v3_compressed"] = wire_one.measure(dev)
    except Exception as e:  # noqa: BLE001
        ex["batch_1024_v3_compressed"] = {"error": str(e)}
    # --- the signatures of a subdag's batch certificates (100 certificates x 67 signers: the quorum of a
    # 100-validator committee), Schnorr over Edwards-BLS12, one call; the oracle on the host cores beside it
    try:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import orc_py as O
        nsig = 6700
        sb = O.SigBatch(0xA1E00005, nsig, 100, 0)
        d_sig = torch.from_numpy(sb.records).to(dev)
        outv = A.SchnorrBatch.verify(d_sig)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            outv = A.SchnorrBatch.verify(d_sig)
        torch.cuda.synchronize()
        dts = (time.perf_counter() - t0) / 5
        cores = usable_cores()
        t0 = time.perf_counter()
        cpu_v = sb.verify_cpu(cores, sb.records[:1024 * 288])
        dtc = time.perf_counter() - t0
        ex["schnorr_certificate_signatures"] = {"signatures": nsig, "ms": 1e3 * dts, "signatures_per_s": nsig / dts,
                                                "all_valid": bool(outv.all().item()) and bool(cpu_v.all()),
                                                "cpu_signatures_per_s": 1024 / dtc, "cpu_cores": cores,
                                                "note": "100 certificates x 67 signers; [RECALL] scheme, own generator / Poseidon constants: parity unpinned"}
    except Exception as e:  # noqa: BLE001
        ex["schnorr_certificate_signatures"] = {"error": str(e)}
    # --- the 1024-proof batch under the Poseidon Fiat-Shamir transcript (row A9 on the path: 37 permutations per proof)
    try:
        dp = np.load(os.path.join(ROOT, "tests", "golden", "vs_batch_1024_poseidon.npz"))
        vp = A.VarunaShaped(dp["vk"], dp["circuits"], transcript="poseidon")
        pp, pr = torch.from_numpy(dp["proofs"]).to(dev), torch.from_numpy(dp["randomizers"]).to(dev)
        ok, _ = vp.verify_batch(pp, pr)
        assert ok
        L.prof_enable(2)
        L.prof_read("vs_fs_poseidon")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            ok, _ = vp.verify_batch(pp, pr)
        torch.cuda.synchronize()
        lat = (time.perf_counter() - t0) / 5
        ms, launches, _ = L.prof_read("vs_fs_poseidon")
        L.prof_enable(0)
        nb = 240
        list(vp.verify_stream(((pp, pr) for _ in range(12))))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        oks = list(vp.verify_stream(((pp, pr) for _ in range(nb))))
        torch.cuda.synchronize()
        dts = time.perf_counter() - t0
        assert all(oks)
        ex["batch_1024_poseidon_fs"] = {
            "latency_ms_one_batch_unpipelined": 1e3 * lat, "transcript_kernel_ms": ms / max(1, launches),
            "permutations_per_proof": 37, "fq_products_per_batch": 1024 * 37 * 626,
            "streamed_proofs_per_s": 1024 * nb / dts, "streamed_ms_per_batch": 1e3 * dts / nb,
            "note": "Poseidon sponge over Fq (own constants), a quad of lanes per proof; the SHA-256 headline above does not pay this"}
        vp.close()
    except FileNotFoundError:
        pass
    # --- VarunaShaped-v2: the query-set table, hiding and two degree-bound classes (4 pairs, 4 fused MSMs, 3184-byte records)
    try:
        d2 = np.load(os.path.join(ROOT, "tests", "golden", "vs2_batch_1024.npz"))
        v2 = A.VarunaShaped(d2["vk"], d2["circuits"], shape=d2["shape"], key_ext=d2["ext"])
        p2, r2 = torch.from_numpy(d2["proofs"]).to(dev), torch.from_numpy(d2["randomizers"]).to(dev)
        ok, _ = v2.verify_batch(p2, r2)
        assert ok
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            ok, _ = v2.verify_batch(p2, r2)
        torch.cuda.synchronize()
        lat = (time.perf_counter() - t0) / 5
        nb = 200
        list(v2.verify_stream(((p2, r2) for _ in range(20))))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        oks = list(v2.verify_stream(((p2, r2) for _ in range(nb))))
        torch.cuda.synchronize()
        dts = time.perf_counter() - t0
        assert all(oks)
        ex["batch_1024_v2_shape"] = {
            "latency_ms_one_batch_unpipelined": 1e3 * lat, "streamed_proofs_per_s": 1024 * nb / dts,
            "streamed_ms_per_batch": 1e3 * dts / nb, "pairs_in_product": 4, "msm_points_per_proof": 9 + 2 + 4 + 4,
            "note": "hiding (gamma_g, random_v), degree-bound classes {5,6} and {7,8}, SHA-256 transcript"}
        v2.close()
    except FileNotFoundError:
        pass
    # --- the same 1024-proof batch with 1 % corrupted proofs: aggregate check fails, fallback names them
    bad = proofs.copy()
    idx = np.arange(5, 1024, 101)  # 11 proofs
    ev_off = 16 + 16 * 32 + 9 * 96
    for i in idx:
        bad[i * L.VS_PROOF_BYTES + ev_off + 32 * 7] ^= 1  # flip one bit of evaluation 7
    d_bad = torch.from_numpy(bad).to(dev)
    d_rnd = torch.from_numpy(rnd).to(dev)
    def timed(fn, reps=5):
        """median wall time of fn over reps calls after one untimed call (each call ends with its verdicts on the host)"""
        fn()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            r = fn()
            ts.append(time.perf_counter() - t0)
        return r, sorted(ts)[len(ts) // 2]

    (ok, verdicts), dt = timed(lambda: verifier.verify_batch(d_bad, d_rnd))
    want = np.ones(1024, dtype=np.uint8)
    want[idx] = 0
    ex["batch_1pct_corrupted"] = {"proofs_per_s": 1024 / dt, "ms": 1e3 * dt, "all_ok": ok,
                                  "verdicts_correct": bool((verdicts == want).all()), "n_bad": int(len(idx))}
    # the same fallback with one invalid proof and with every proof invalid (group testing: 1 and 256 failing groups),
    # and the verdict vector of a failed STREAMED batch (mi_vs_identify_device: no aggregate pass)
    for name, which in (("one_bad", np.array([517])), ("all_bad", np.arange(1024))):
        b2 = proofs.copy()
        for i in which:
            b2[i * L.VS_PROOF_BYTES + ev_off + 32 * 7] ^= 1
        d_b2 = torch.from_numpy(b2).to(dev)
        (ok2, v2), dt2 = timed(lambda: verifier.verify_batch(d_b2, d_rnd))
        w2 = np.ones(1024, dtype=np.uint8)
        w2[which] = 0
        ex["batch_1pct_corrupted"]["ms_" + name] = 1e3 * dt2
        ex["batch_1pct_corrupted"]["verdicts_correct"] &= bool((v2 == w2).all()) and not ok2
    v3, dt3 = timed(lambda: verifier.identify(d_bad, d_rnd))
    ex["batch_1pct_corrupted"]["ms_identify_only"] = 1e3 * dt3
    ex["batch_1pct_corrupted"]["verdicts_correct"] &= bool((v3 == want).all())
    return ex


if __name__ == "__main__":
    main()
