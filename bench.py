#!/usr/bin/env python3
"""Headline benchmark: seconds to the Canonical circular order + achieved HBM GB/s at
n = 32768 synthetic taxa (BASELINE.json `metric`, configs[3] run on the GPUs given).

A "step" = one complete circular-order computation (initial row sums, the whole
agglomeration loop, expansion) on a synthetic random symmetric matrix that is already
resident in HBM when the step starts.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--taxa 32768] [--seed 1]

N > 1 is launched by the driver through torch.distributed.run (one rank per GPU) and orders the SAME
matrix on all N GPUs (BASELINE.json's metric: one problem, strong scaling): every rank holds the matrix
(8 GiB of 288) and runs the whole event chain; only the ~860 base scans of the lookahead windows - the
part of the run that still streams the matrix, ~0.19 s of it - are sharded (tile index mod N), each followed
by ONE RCCL all-gather on the engine's stream (candidate records + the pairs every rank emitted for the
new window).  The other ~31 900 events are a chain of latency-bound steps that no exchange can shorten
(DESIGN.md "Multi-GPU"), so the curve is Amdahl-flat by construction.  If the RCCL communicator cannot be
created the ranks agree to let rank 0 compute alone (reported in config.parallelism).
FNN_BENCH_REPLICAS=1 instead runs N independent orders (seed + rank), one per GPU: that is NOT the
BASELINE metric and is reported under its own metric name (orders per second).

Outside the timed region the line also carries: `roofline` (the streaming kernel that is left, with its share
of the GPU time), `chain` (per-kernel averages of the latency-bound event chain from an extra run with HIP events
around every launch), `cpu_baseline` (the oracle on the host cores: the first events of the same matrix,
extrapolated, plus a fully measured run at n = 4096 on one thread and on all cores next to the engine's own
time at that size), and the comparison of the order with the oracle's golden (tests/golden/oracle_big.json).

Prints ONE JSON line on rank 0.
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6290 measured copy rate
PMC_SUMMARY = os.path.join("profiles", "r04", "r04_pmc_screen_windows_summary_n32768.json")
ISA_RESOURCES = os.path.join("profiles", "r04", "isa_resources.json")


def source_sha256():
    return hashlib.sha256(open(os.path.join(ROOT, "fastneighbornet_amd", "csrc", "fnn_hip.hip"), "rb").read()).hexdigest()


KCLASS = ["k_scan", "k_screen", "k_track", "k_decide", "k_update", "k_emit", "k_resolve", "k_finalize"]


def host_cores():
    # the GPU box gives one GPU's job a 16-core CPU share; more threads only oversubscribe it
    return min(len(os.sched_getaffinity(0)), 16)


def cpu_baseline(n, seed, total_entries, budget_s=15.0):
    """Oracle (C restatement, kind "port") on the host cores over a bounded sample of the
    SAME workload: the first events of the n-taxa run (each event scans ~n^2/2 entries).
    The sample's rate is extrapolated to the whole run's sum of scanned entries."""
    from oracle import nnet_oracle as O
    cores = host_cores()
    t0 = time.time()
    D = O.synth(n, seed, "uniform53")
    t_gen = time.time() - t0
    st = O.Stepper(D, threads=cores)
    del D
    ent = 0
    k = 0
    t0 = time.time()
    while time.time() - t0 < budget_s:
        ev = st.step()
        if ev is None:
            break
        ent += ev.entries
        k += 1
    dt = time.time() - t0
    st.close()
    rate = ent / dt  # entries / s
    return {
        "value": round(total_entries / rate, 2),
        "unit": "s",
        "cores": cores,
        "kind": "port",
        "sample": (f"oracle/nnet_oracle.c with its OpenMP scan on {cores} host threads: first {k} agglomeration "
                   f"events of the same n={n} seed={seed} matrix ({ent:.3e} matrix entries in {dt:.1f} s = "
                   f"{rate * 8 / 1e9:.2f} GB/s); value = whole-run seconds extrapolated as "
                   f"sum_t E_t / sample rate; host matrix generation {t_gen:.1f} s not counted"),
    }


def cpu_measured_small(api, dev_index, n=4096, seed=1):
    """A FULLY measured configuration (BASELINE.json configs[1]): the oracle's whole run at n = 4096 on one
    thread (the reference's -threads 1 semantics) and on all host cores, next to the engine's time there."""
    from oracle import nnet_oracle as O
    from fastneighbornet_amd._capi import Handle
    D = O.synth(n, seed, "uniform53")
    cores = host_cores()
    t0 = time.time()
    o_all, _, se = O.run(D, threads=cores, want_events=False)
    t_all = time.time() - t0
    t0 = time.time()
    o_one, _, _ = O.run(D, threads=1, want_events=False)
    t_one = time.time() - t0
    with Handle(api, n, device=dev_index) as h:
        h.synth(seed, "uniform53")
        h.run()                      # warm-up
        h.synth(seed, "uniform53")
        order, st = h.run()
    return {"n_taxa": n, "seed": seed, "oracle_1_thread_s": round(t_one, 2), "oracle_all_cores_s": round(t_all, 2),
            "cores": cores, "engine_s": round(st.t_total_s, 4), "sum_entries": int(se),
            "orders_identical": bool((o_all == o_one).all() and (order == o_one).all())}


def chain_kernel(st, chain, sec):
    """A roofline-style object for k_track, the kernel that dominates the run: algorithmic bytes per launch (the tracked
    pairs' 48-byte records + 64 B of node state each, and 16 B per swept entry: DESIGN.md section 3), its average duration
    from the extra run with HIP events around every launch, and its ISA resources from profiles/ (quoted only when they
    were extracted from this very source)."""
    if not chain or "k_track" not in chain["kernels"]:
        return None
    k = chain["kernels"]["k_track"]
    bytes_per_launch = (float(st.bytes_total) - float(st.scan_bytes) - float(st.plain_bytes)) / max(k["launches"], 1)
    out = {"kernel": "fnn::k_track<false, false> (the instantiation without the ComputeRx helper workgroups - inputs without exact ties - and without the fused update)", "bound": "instruction stream / fan-in latency (one wave's issue rate), not HBM",
           "launches": k["launches"], "avg_us": k["avg_us"], "share_of_kernel_time": k["share_of_kernel_time"],
           "bytes_per_launch": round(bytes_per_launch, 1),
           "achieved": round(bytes_per_launch / (k["avg_us"] * 1e-6) / 1e9, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
           "frac": round(bytes_per_launch / (k["avg_us"] * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4)}
    try:
        doc = json.load(open(os.path.join(ROOT, ISA_RESOURCES)))
        if doc.get("fnn_hip_sha256") == source_sha256():
            r = doc["kernels"]["void fnn::k_track<false, false>"]
            out.update({"vgpr": r["vgpr"], "scratch_bytes_per_lane": r["scratch_bytes_per_lane"], "lds_bytes_per_block": r["lds_bytes_per_block"],
                        "sgpr_spills": r["sgpr_spills"], "vgpr_spills": r["vgpr_spills"], "resources_from": ISA_RESOURCES})
        else:
            out["resources_from"] = f"{ISA_RESOURCES} is from another source (sha256 differs): not quoted"
    except (OSError, KeyError):
        out["resources_from"] = None
    return out


def amdahl(chain, sec, n_ranks):
    if not chain:
        return None
    k = chain["kernels"]
    shard = sum(k[c]["launches"] * k[c]["avg_us"] for c in ("k_screen", "k_emit", "k_resolve") if c in k) * 1e-6
    ex = chain.get("exchange")
    per_exchange = ((ex["allgather_us_avg"] + ex["merge_us_avg"]) * 1e-6) if ex else None
    # (this run's sharded kernels already ran on 1 / n_ranks of the tiles)
    out = {"sharded_kernel_seconds_of_this_run": round(shard, 4), "ranks_of_this_run": n_ranks}
    serial = sec - shard - (ex["allgather_calls"] * per_exchange if ex else 0.0)
    full = shard * n_ranks
    for n in (1, 2, 4, 8):
        t = serial + full / n
        if n > 1:
            t += (ex["allgather_calls"] if ex else k.get("k_screen", {"launches": 0})["launches"]) * (per_exchange if per_exchange else 40e-6)
        out[f"n{n}"] = round(t, 4)
    out["note"] = ("serial part + sharded part / N + exchanges x (all-gather + merge); the exchange cost is this run's measurement when N > 1, "
                   "else assumed 40 us (RCCL small-message latency over xGMI has not been measured on this pool)")
    return out


def config5(api, h, n, seed, order_ref):
    """order -> split weights -> Nexus document for the bench's matrix on this GPU; seconds per stage."""
    import ctypes as C
    import numpy as np
    import fastneighbornet_amd as fa
    h.synth(seed, "uniform53")
    t0 = time.perf_counter()
    D = h.matrix()                                   # fnn_get_matrix: the device-generated distances for the later stages
    t_get = time.perf_counter() - t0
    order, st = h.run()
    assert (order == order_ref).all()
    h.close()                                        # (the engine's 10 GiB go back before the solver sizes its factor)
    t0 = time.perf_counter()
    w, sw = fa.split_weights(D, order)
    t_w = time.perf_counter() - t0
    host = C.CDLL(os.path.join(os.path.dirname(fa.__file__), "libfastnn_host.so"))
    host.fnnh_write_nexus.argtypes = [C.c_char_p, C.c_int32, C.POINTER(C.c_double), C.c_char_p, C.POINTER(C.c_int32), C.POINTER(C.c_double)]
    host.fnnh_write_nexus.restype = C.c_int32
    names = b"".join((f"t{i + 1}".encode()).ljust(256, b"\0") for i in range(n))
    t0 = time.perf_counter()
    ns = host.fnnh_write_nexus(b"/dev/null", n, D.ctypes.data_as(C.POINTER(C.c_double)), names,
                               order.ctypes.data_as(C.POINTER(C.c_int32)), w.ctypes.data_as(C.POINTER(C.c_double)))
    t_doc = time.perf_counter() - t0
    assert ns == int((w > 1e-6).sum()) == sw["nsplits"]
    return {"order": round(st.t_total_s, 3), "weights": round(t_w, 2), "weights_device": round(sw["t_solve_s"], 2), "nexus": round(t_doc, 2),
            "total": round(st.t_total_s + t_w + t_doc, 2),
            "kkt": sw["kkt_violation"], "kkt_certified": bool(sw["certified"]),
            "kkt_note": "max(-min x, max |g| on x > 0, max -g on x = 0) / max|A^T d|, g = A^T (A x - d), evaluated on the device from the returned "
                        "weights with the implicit operators (independent of the factor the solver used); tests/test_split_weights.py::"
                        "test_config5_end_to_end_32768 repeats it on the host with the oracle's operators",
            "nsplits": int(ns), "route": sw["method"], "steps": int(sw["outer_iterations"]), "rebuilds": int(sw["refactorizations"]),
            "entered": int(sw["entered"]), "screened_out": int(sw["screened_out"]), "departed": int(sw["departed"]),
            "factor_capacity": int(sw["capacity"]), "free_set_peak": int(sw["free_set_peak"]), "hipmalloc_s": round(sw["t_alloc_s"], 2),
            "matrix_download_s": round(t_get, 2),
            "note": "order = fnn_run with the matrix resident; weights = fnn_split_weights_f64 wall clock incl. the upload of the host matrix; "
                    "nexus = printNexusFromWeights to /dev/null on the host cores"}


def golden_check(n, seed, order):
    """The order against the oracle's golden for this (n, seed), generated in the build container."""
    try:
        doc = json.load(open(os.path.join(ROOT, "tests", "golden", "oracle_big.json")))
    except OSError:
        return None
    for c in doc["cases"]:
        if (c["n"], c["seed"], c["dist"]) == (n, seed, "uniform53"):
            return hashlib.sha256(order.tobytes()).hexdigest() == c["order_sha256"]
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--taxa", "--n", dest="n", type=int, default=32768)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-chain", action="store_true", help="skip the extra run that times every kernel of the event chain")
    ap.add_argument("--no-config5", action="store_true", help="skip BASELINE config 5 (order + split weights + Nexus document, ~100 s, outside the timed region)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    import torch
    dist = None
    same_gpu = os.environ.get("FNN_BENCH_SAME_GPU") == "1"  # rehearsal on a one-GPU box (gloo transport)
    dev_index = 0 if same_gpu else local_rank
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(dev_index)
        if same_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))

    import ctypes as C
    import fastneighbornet_amd as fa
    from fastneighbornet_amd import distributed as fd
    from fastneighbornet_amd._capi import Handle
    api = fa.api()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    n = args.n
    h = None
    parallelism = "single GPU"
    sharded = False
    replicas = world > 1 and os.environ.get("FNN_BENCH_REPLICAS") == "1"
    shard_mode = world > 1 and not replicas
    if replicas:
        parallelism = (f"REPLICAS (not the BASELINE metric): {world} independent orders (seed + rank), one per GPU, "
                       f"no data-path collective")
    if shard_mode:
        ok, why = 1, ""
        try:
            h = Handle(api, n, device=dev_index)
            if same_gpu:
                fd.init_gloo(h, dist)
            else:
                fd.init_rccl(h, dist, torch.device("cuda", dev_index))
        except Exception as e:  # all ranks must agree on the fallback (init_rccl itself is symmetric)
            ok, why = 0, str(e)
        flag = torch.tensor([ok], dtype=torch.int32, device="cpu" if same_gpu else "cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        sharded = int(flag.item()) == 1
        if sharded:
            parallelism = (f"one problem on {world} ranks: matrix and event chain replicated, the base scans of the lookahead "
                           f"windows sharded (tile index mod {world}) with one all-gather each "
                           f"({'gloo host callback' if same_gpu else 'RCCL on the engine stream'}): candidate records + emitted pairs")
        else:
            if h is not None:
                h.close()
                h = None
            parallelism = f"rank 0 computes alone (communicator setup failed: {why or 'on another rank'})"
    worker = sharded or replicas or rank == 0
    if worker and h is None:
        h = Handle(api, n, device=dev_index)
    if worker:
        api.set_scan_timing(h._h, 1)

    seed = args.seed + (rank if replicas else 0)

    def one_step():
        h.synth(seed, "uniform53")  # matrix generated in HBM (2-3 ms at n = 32768)
        return h.run()

    for _ in range(args.warmup):
        if worker:
            one_step()
    barrier()
    t0 = time.perf_counter()
    last = None
    for _ in range(args.steps):
        if worker:
            last = one_step()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if same_gpu else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- outside the timed region: every kernel of the event chain between HIP events (all ranks of a sharded run
    # take part: the exchange is collective)
    chain = None
    if worker and not args.no_chain:
        api.set_scan_timing(h._h, 2)
        _, st2 = one_step()
        ms = (C.c_double * 8)()
        cnt = (C.c_int64 * 8)()
        api.get_kernel_times(h._h, ms, cnt)
        api.set_scan_timing(h._h, 1)
        tot = sum(ms)
        xms = (C.c_double * 2)()
        xcnt = (C.c_int64 * 2)()
        api.get_exchange_times(h._h, xms, xcnt)
        chain = {
            "note": ("an extra, untimed run with HIP events around every launch (slower than the timed run by the event "
                     "records); a window event = k_track (tracked pairs + sweep, the exact row sum of the previous "
                     "event's cluster in a workgroup beside them, then Cx/Cy, 4-candidate choice and merge plan by the "
                     "last workgroup) + k_update; an event that scans adds k_screen/k_emit/k_resolve (or k_scan) + k_decide"),
            "run_s_with_event_records": round(st2.t_total_s, 4),
            "kernels": {KCLASS[c]: {"launches": int(cnt[c]), "avg_us": round(ms[c] * 1e3 / cnt[c], 2),
                                    "share_of_kernel_time": round(ms[c] / tot, 4)} for c in range(8) if cnt[c] > 0},
            # several ranks: the one collective of the path (all-gather after a sharded base scan) and the merge behind it
            "exchange": ({"allgather_calls": int(xcnt[0]), "allgather_us_avg": round(xms[0] * 1e3 / xcnt[0], 2),
                          "merge_us_avg": round(xms[1] * 1e3 / max(xcnt[1], 1), 2)} if xcnt[0] > 0 else None),
        }
    # ---- several ranks: every rank's order, not only rank 0's (the replicated event chain rests on identical decisions)
    ranks_orders = None
    if dist is not None and not replicas:
        mine = hashlib.sha256(last[0].tobytes()).hexdigest() if (worker and last is not None) else None
        ranks_orders = [None] * world
        dist.all_gather_object(ranks_orders, mine)
    if dist is not None:
        dist.barrier()

    if rank == 0:
        order, st = last
        sec = elapsed / args.steps
        fp64_equiv = 8.0 * float(st.sum_entries)     # BASELINE.md's 8 * sum E_t, for reference
        assert sorted(order[1:].tolist()) == list(range(1, n + 1)) and order[0] == 0 and order[1] == 1
        gold = golden_check(n, seed, order) if not replicas else None
        assert gold is not False, "the circular order differs from the oracle's golden (tests/golden/oracle_big.json)"
        ranks_seen = None
        if ranks_orders is not None:
            seen = [s for s in ranks_orders if s is not None]
            ranks_seen = len(seen)
            assert len(set(seen)) == 1 and seen[0] == hashlib.sha256(order.tobytes()).hexdigest(), \
                f"the ranks returned different circular orders: {ranks_orders}"
            assert not sharded or ranks_seen == world, f"only {ranks_seen} of {world} ranks reported an order"
        share = args.gpus if sharded else 1  # this rank's launches cover 1/share of the entries
        # dominant streaming kernel: the bf16 screening pass.  With lookahead windows the timed launches
        # are the host-scheduled base scans (kernel name k_screen<true, true>)
        have_screen = st.scan_launches > 0
        k_bytes = float(st.scan_bytes) if have_screen else float(st.plain_bytes)
        k_time = st.t_scan_s if have_screen else st.t_plain_s
        k_launches = int(st.scan_launches if have_screen else st.plain_launches)
        scan_gbps = k_bytes / share / max(k_time, 1e-12) / 1e9
        # `traffic` is a PMC figure collected in separate rocprofv3 passes (the guide's recipe) and kept under profiles/: it is
        # quoted only while the kernels it was collected from are the ones this run executes (same sha256 of fnn_hip.hip)
        traffic, traffic_src = None, None
        pmc = os.path.join(ROOT, PMC_SUMMARY)
        if n == 32768 and args.gpus == 1 and have_screen and os.path.exists(pmc):
            doc = json.load(open(pmc))
            if doc.get("fnn_hip_sha256") == source_sha256():
                traffic = round(doc["hbm_bytes_per_launch_avg"], 1)
                traffic_src = doc["fnn_hip_sha256"][:16]
        if replicas:
            head = {"metric": f"orders per second, n={n} taxa, {world} independent replicas (NOT the BASELINE metric)",
                    "value": round(world / sec, 4), "unit": "orders/s", "higher_is_better": True, "scaling": "weak"}
        else:
            head = {"metric": f"sec to circular order, n={n} taxa (+ achieved HBM GB/s)", "value": round(sec, 4), "unit": "s",
                    "higher_is_better": False, "scaling": "strong" if world > 1 else "weak"}
        out = {
            **head,
            "n_gpus": args.gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(sec * 1e3, 2),
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"{n} synthetic taxa (SplitMix64 uniform53, seed {args.seed}), -mode Canonical, "
                            f"{args.gpus} MI355X",
                "parallelism": parallelism,
                "n_taxa": n,
                "events": int(st.n_events),
                "sum_entries": int(st.sum_entries),
                "order_matches_oracle_golden": gold,
                # several ranks: how many ranks computed an order; all of them are compared with rank 0's (and so with the golden)
                "rccl_ranks_seen": ranks_seen, "all_ranks_same_order": (True if ranks_seen else None),
                "fp64_every_event_bytes": int(fp64_equiv),
                "bytes_read_by_all_scan_work": int(st.bytes_total),
                "lookahead_windows": {"events_with_a_scan": int(st.n_base_scans), "events_served_by_a_window": int(st.n_window_hits),
                                      "windows_that_could_not_certify": int(st.n_window_fails),
                                      "tracked_pairs_per_window": round(st.window_pairs / max(st.n_base_scans, 1), 1)},
                "screening": {"events": int(st.n_screen_events), "rescanned_units_32x512": int(st.n_rescan_units)},
            },
            "hbm_gbps_whole_run": round(float(st.bytes_total) / sec / 1e9, 1),
            "hbm_frac_whole_run": round(float(st.bytes_total) / sec / 1e9 / HBM_PEAK_GBPS, 4),
            # SURVEY.md 8(d)'s second figure, B_scan / t_order with B_scan = 8 B x sum_t E_t (what a scan of the fp64
            # matrix in every event would read).  NOT physical traffic: the windows avoid reading most of it.
            "b_scan_over_t_order_gbps": round(float(fp64_equiv) / sec / 1e9, 1),
            "us_per_event": round((st.t_agglom_s) / max(st.n_events, 1) * 1e6, 2),
            "rx_decisions": {"certified_from_approximate_row_sums": int(st.n_rx_certified), "exact_sequential_sums": int(st.n_rx_exact)},
            "phases_s": {"init": round(st.t_init_s, 4), "agglomerate": round(st.t_agglom_s, 4),
                         "expand": round(st.t_expand_s, 4), "screen_kernel_sum": round(st.t_scan_s, 4),
                         "plain_scan_kernel_sum": round(st.t_plain_s, 4)},
            "roofline": {
                "kernel": ("fnn::k_screen<true, true> (bf16 screening pass; the scheduled base scans of the lookahead windows)"
                           if have_screen else "fnn::k_scan<true> (plain fp64 scan)"),
                "bound": "hbm",
                "achieved": round(scan_gbps, 1),
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": round(scan_gbps / HBM_PEAK_GBPS, 4),
                "traffic": traffic,
                "traffic_note": (f"HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this kernel "
                                 f"(FETCH_SIZE x2 per the gfx950 correction): {PMC_SUMMARY}, collected from fnn_hip.hip sha256 {traffic_src}... "
                                 f"= the source of this run" if traffic else
                                 "null: no PMC summary under profiles/ was collected from the kernels of this build (sha256 of fnn_hip.hip differs or the file is absent)"),
                # the kernel this roofline describes is a small part of the run: the rest is the latency-bound event
                # chain (see `chain`), for which seconds-to-order and us per event are the yardsticks
                "time_share": round(k_time / max(sec, 1e-12), 4),
                "launches": k_launches,
                "avg_launch_us": round(k_time / max(k_launches, 1) * 1e6, 2),
                "algorithmic_bytes_per_launch_avg": round(k_bytes / share / max(k_launches, 1), 1),
                "algorithmic_bytes_note": "per launch: E_t = m(m-1)/2 - (m-c) entries at 2 B (the bf16 copy) + the fp64 rescans of "
                                          "the candidate units (32 x 512 x 8 B each)",
                "per_gpu": args.gpus > 1,
                "plain_fp64_scan": {"kernel": "fnn::k_scan<true> (events below the screening threshold of min(2048, n/4) live nodes)",
                                    "launches": int(st.plain_launches),
                                    "achieved": round(float(st.plain_bytes) / max(st.t_plain_s, 1e-12) / 1e9, 1),
                                    "avg_launch_us": round(st.t_plain_s / max(st.plain_launches, 1) * 1e6, 2)},
            },
            "chain": chain,
            # what N ranks can gain at best (Amdahl): only the base scans' kernels (k_screen, k_emit, k_resolve) are sharded; every
            # exchange adds its all-gather + merge.  From this run's own per-kernel times; the curve is flat by construction.
            "expected_amdahl_s": amdahl(chain, sec, args.gpus if sharded else 1),
            # the latency-bound kernel that dominates the run (not HBM-bound: priced here only so that the line describes it)
            "chain_kernel": chain_kernel(st, chain, sec),
            "n_handover_retries": int(st.n_handover_retries),
            "n_stalled_events": int(st.n_stalled_events),
            "n_sweeps_exact": int(st.n_sweeps_exact),
            "windows_that_could_not_certify": int(st.n_window_fails),
            "source_sha256": {"fnn_hip.hip": source_sha256()[:16]},
        }
        try:
            g = C.c_double(0.0)
            if api.stream_probe(dev_index, 4 << 30, 5, C.byref(g)) == 0:
                out["roofline"]["measured_stream_read_gbps"] = round(g.value, 1)
        except Exception:
            pass
        # ---- BASELINE.json configs[4], outside the timed region: the same 32768 taxa end to end - order, circular split
        # weights (fnn_split_weights_f64: block active-set method, with the solver's own Kuhn-Tucker certificate), Nexus
        # document (the routine the CLI runs, to /dev/null) - FastNN.java:369-491
        if n == 32768 and args.gpus == 1 and not args.no_config5:
            try:
                out["config5_e2e_s"] = config5(api, h, n, seed, order)
            except Exception as e:  # reported, never required for the headline
                out["config5_e2e_s"] = {"error": f"{type(e).__name__}: {e}"}
        if h is not None:
            h.close()
            h = None
        if not args.no_cpu_baseline and args.gpus == 1:
            try:
                out["cpu_baseline"] = cpu_baseline(n, seed, int(st.sum_entries))
                out["cpu_baseline"]["measured_n4096"] = cpu_measured_small(api, dev_index)
            except Exception as e:  # the baseline is reported, never required
                out.setdefault("cpu_baseline", {"value": None, "unit": "s", "cores": 0, "kind": "port"})
                out["cpu_baseline"]["error"] = f"{type(e).__name__}: {e}"
            # the reference itself (Java): only where an operator has supplied a JDK and the reference sources
            try:
                import importlib.util
                spec = importlib.util.spec_from_file_location("java_baseline", os.path.join(ROOT, "tests", "golden", "java", "java_baseline.py"))
                jb = importlib.util.module_from_spec(spec)
                spec.loader.exec_module(jb)
                out["cpu_baseline"]["java_reference"] = jb.run_if_available(ROOT)
            except Exception as e:
                out["cpu_baseline"]["java_reference"] = f"Java baseline unavailable ({type(e).__name__}: {e})"
        print(json.dumps(out), flush=True)
    if h is not None:
        h.close()  # every rank releases its engine (and its RCCL communicator) before the group goes away
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
