"""Summarise rocprofv3 --pmc passes over the scan kernel into profiles/<round>/pmc_scan_summary.json.

usage: pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <algorithmic_bytes_total> <out.json> [kernel substring]

Corrections follow MI355X_MICROARCH.md (HBM): FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950
FETCH_SIZE reports exactly half of the bytes of a wide (16 B/lane) coalesced streaming read, so
the read side is doubled; WRITE_SIZE is exact for streaming stores."""
import csv
import hashlib
import json
import os
import sys


KERNEL = None


def load(path, name):
    vals = []
    with open(path) as f:
        for r in csv.DictReader(f):
            ok = (KERNEL in r["Kernel_Name"]) if KERNEL else ("k_scan" in r["Kernel_Name"] or "k_screen" in r["Kernel_Name"])
            if r["Counter_Name"] == name and ok:
                vals.append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    vals.sort()
    return [v for _, v in vals]


def main():
    global KERNEL
    fetch_csv, write_csv, algo_total, out = sys.argv[1], sys.argv[2], float(sys.argv[3]), sys.argv[4]
    KERNEL = sys.argv[5] if len(sys.argv) > 5 else None
    fetch = load(fetch_csv, "FETCH_SIZE")
    write = load(write_csv, "WRITE_SIZE") if write_csv != "-" else []
    n = len(fetch)
    fetch_b = sum(fetch) * 1024.0 * 2.0
    write_b = sum(write) * 1024.0
    res = {
        "kernel": KERNEL or "fnn::k_screen<true> (+ fnn::k_scan<true> for m < 8192)",
        "launches": n,
        "FETCH_SIZE_KiB_sum_raw": sum(fetch),
        "WRITE_SIZE_KiB_sum_raw": sum(write),
        "correction": "read bytes = FETCH_SIZE * 1024 * 2 (gfx950 counts 128-B requests at 64 B); write bytes = WRITE_SIZE * 1024",
        "hbm_read_bytes_total": fetch_b,
        "hbm_write_bytes_total": write_b,
        "hbm_bytes_per_launch_avg": (fetch_b + write_b) / max(n, 1),
        "algorithmic_bytes_total": algo_total,
        "algorithmic_bytes_per_launch_avg": algo_total / max(n, 1),
        "traffic_over_algorithmic": (fetch_b + write_b) / algo_total if algo_total else None,
        "first_launch_read_bytes": fetch[0] * 2048.0 if fetch else None,
        # the kernels these counters were collected from: bench.py quotes `traffic` only while this still is the source it runs
        "fnn_hip_sha256": hashlib.sha256(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fastneighbornet_amd", "csrc",
                                                           "fnn_hip.hip"), "rb").read()).hexdigest(),
    }
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
