#!/usr/bin/env python3
"""Headline benchmark: seconds to the Canonical circular order + achieved HBM GB/s at
n = 32768 synthetic taxa (BASELINE.json `metric`, configs[3] run on the GPUs given).

A "step" = one complete circular-order computation (initial row sums, the whole
agglomeration loop, expansion) on a synthetic random symmetric matrix that is already
resident in HBM when the step starts.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--n 32768] [--seed 1]

N > 1 is launched by the driver through torch.distributed.run (one rank per GPU).  The
matrix row-sharding over several GPUs is not built yet (DESIGN.md "Multi-GPU"): with
N > 1 rank 0 computes the single problem instance on its GPU while the other ranks wait
at the barrier, so the reported time is a true whole-job time ("scaling": "strong").

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6290 measured copy rate


def cpu_baseline(n, seed, total_entries, budget_s=20.0):
    """Oracle (C restatement, kind "port") on the host cores over a bounded sample of the
    SAME workload: the first events of the n-taxa run (each event scans ~n^2/2 entries).
    The sample's rate is extrapolated to the whole run's sum of scanned entries."""
    from oracle import nnet_oracle as O
    # the GPU box gives one GPU's job a 16-core CPU share; more threads only oversubscribe it
    cores = min(len(os.sched_getaffinity(0)), 16)
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n", type=int, default=32768)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import fastneighbornet_amd as fa
    from fastneighbornet_amd._capi import Handle
    api = fa.api()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    n = args.n
    worker = (rank == 0)  # the single problem instance lives on rank 0's GPU (see module docstring)
    h = None
    if worker:
        h = Handle(api, n, device=local_rank)
        api.set_scan_timing(h._h, 1)

    def one_step():
        h.synth(args.seed, "uniform53")  # matrix generated in HBM (2-3 ms at n = 32768)
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
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        order, st = last
        sec = elapsed / args.steps
        scan_bytes = float(st.scan_bytes)
        assert sorted(order[1:].tolist()) == list(range(1, n + 1)) and order[0] == 0 and order[1] == 1
        scan_gbps = scan_bytes / max(st.t_scan_s, 1e-12) / 1e9
        out = {
            "metric": f"sec to circular order, n={n} taxa (+ achieved HBM GB/s)",
            "value": round(sec, 4),
            "unit": "s",
            "n_gpus": args.gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(sec * 1e3, 2),
            "higher_is_better": False,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"{n} synthetic taxa (SplitMix64 uniform53, seed {args.seed}), -mode Canonical, "
                            f"{args.gpus} MI355X" + (" (rank 0 computes; row-sharding not built yet)" if args.gpus > 1 else ""),
                "n_taxa": n,
                "events": int(st.n_events),
                "sum_entries": int(st.sum_entries),
                "algorithmic_bytes": int(scan_bytes),
            },
            "hbm_gbps_whole_run": round(scan_bytes / sec / 1e9, 1),
            "hbm_frac_whole_run": round(scan_bytes / sec / 1e9 / (HBM_PEAK_GBPS * max(args.gpus, 1)), 4),
            "phases_s": {"init": round(st.t_init_s, 4), "agglomerate": round(st.t_agglom_s, 4),
                         "expand": round(st.t_expand_s, 4), "scan_kernel_sum": round(st.t_scan_s, 4)},
            "roofline": {
                "kernel": "fnn::k_scan",
                "bound": "hbm",
                "achieved": round(scan_gbps, 1),
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": round(scan_gbps / HBM_PEAK_GBPS, 4),
                "traffic": None,
                "launches": int(st.scan_launches),
                "avg_launch_us": round(st.t_scan_s / max(st.scan_launches, 1) * 1e6, 2),
                "algorithmic_bytes_per_launch_avg": round(scan_bytes / max(st.scan_launches, 1), 1),
            },
        }
        try:
            import ctypes as C
            g = C.c_double(0.0)
            if api.stream_probe(local_rank, 4 << 30, 5, C.byref(g)) == 0:
                out["roofline"]["measured_stream_read_gbps"] = round(g.value, 1)
        except Exception:
            pass
        if h is not None:
            h.close()
        if not args.no_cpu_baseline and args.gpus == 1:
            try:
                out["cpu_baseline"] = cpu_baseline(n, args.seed, int(st.sum_entries))
            except Exception as e:  # the baseline is reported, never required
                out["cpu_baseline"] = {"value": None, "unit": "s", "cores": 0, "kind": "port",
                                       "sample": f"failed: {e}"}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
