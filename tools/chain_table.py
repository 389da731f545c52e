"""Join a rocprofv3 --kernel-trace CSV with the engine's event log: per event kind (2-/3-/4-way) and
launch-sequence type (window event / scheduled scan event / end game), the average duration of every
kernel of the sequence and of the gaps between consecutive kernels.

usage: chain_table.py <kernel_trace.csv> <events.npz> <out.json> [out.md]

A launch sequence starts at each k_track (or at k_scan/k_rx_fill where no k_track is launched) and ends
at its k_update.  Sequences whose k_update returns at once (< 2.5 us: stalled launch sequences, loop
already ended) are counted apart; the remaining ones map 1:1, in order, onto the event log."""
import csv
import json
import sys

import numpy as np


def short(name):
    for k in ("k_track", "k_screen", "k_emit", "k_resolve", "k_scan", "k_rx_fill", "k_decide4", "k_decide", "k_update",
              "k_finalize", "k_chain_flush", "k_init", "k_prep_screen", "k_synth", "k_events"):
        if k in name:
            return k
    return name.split("(")[0]


def main():
    trace, evf, out = sys.argv[1], sys.argv[2], sys.argv[3]
    rows = []
    with open(trace) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    ev = np.load(evf)
    kinds, ms = ev["kind"], ev["m"]
    # split into launch sequences
    seqs, cur = [], None
    for s, e, k in rows:
        if k in ("k_init", "k_prep_screen", "k_synth", "k_chain_flush", "k_events"):
            continue
        starts = k == "k_track" or (cur is None) or ("k_update" in cur["k"]) or ("k_finalize" in cur["k"] and k != "k_finalize")
        if k == "k_finalize" and cur is not None and "k_update" in cur["k"]:
            starts = False
        if starts:
            cur = {"k": {}, "order": [], "t0": s}
            seqs.append(cur)
        cur["k"][k] = (s, e)
        cur["order"].append((k, s, e))
    live = [q for q in seqs if "k_update" in q["k"] and q["k"]["k_update"][1] - q["k"]["k_update"][0] >= 2500]
    idle = [q for q in seqs if q not in live] if len(seqs) < 200000 else []
    res = {"sequences": len(seqs), "live_sequences": len(live), "events_in_log": int(len(kinds)),
           "note": "live sequences are matched in order to the event log; a mismatch in count is reported, the table then uses the shorter prefix"}
    nmatch = min(len(live), len(kinds))
    groups = {}
    prev_of = {id(q): (seqs[i - 1]["order"][-1][2] if i > 0 else None) for i, q in enumerate(seqs)}
    for i in range(nmatch):
        q = live[i]
        typ = "scheduled_scan" if "k_screen" in q["k"] else ("end_game_fp64_scan" if "k_scan" in q["k"] else "window")
        key = (typ, int(kinds[i]))
        g = groups.setdefault(key, {"count": 0, "dur": {}, "gap": {}, "span": 0.0, "m_sum": 0.0, "lead_gap": 0.0})
        g["count"] += 1
        g["m_sum"] += float(ms[i])
        o = q["order"]
        for j, (k, s, e) in enumerate(o):
            g["dur"][k] = g["dur"].get(k, 0.0) + (e - s)
            if j > 0:
                gk = o[j - 1][0] + "->" + k
                g["gap"][gk] = g["gap"].get(gk, 0.0) + (s - o[j - 1][2])
        g["span"] += o[-1][2] - o[0][1]
        if prev_of[id(q)] is not None:
            g["lead_gap"] += o[0][1] - prev_of[id(q)]
    table = []
    for (typ, kind), g in sorted(groups.items()):
        c = g["count"]
        table.append({"sequence": typ, "kind": {2: "2-way", 3: "3-way", 4: "4-way", 5: "finish"}.get(kind, str(kind)),
                      "events": c, "avg_m": round(g["m_sum"] / c, 1),
                      "kernel_us": {k: round(v / c / 1e3, 2) for k, v in g["dur"].items()},
                      "gap_us": {k: round(v / c / 1e3, 2) for k, v in g["gap"].items()},
                      "gap_before_sequence_us": round(g["lead_gap"] / c / 1e3, 2),
                      "first_start_to_last_end_us": round(g["span"] / c / 1e3, 2)})
    res["table"] = table
    if idle:
        d = {}
        for q in idle:
            for k, s, e in q["order"]:
                d.setdefault(k, []).append(e - s)
        res["idle_sequences"] = {"count": len(idle), "kernel_us": {k: round(float(np.mean(v)) / 1e3, 2) for k, v in d.items()}}
    tot = rows[-1][1] - rows[0][0]
    busy = sum(e - s for s, e, _ in rows)
    res["trace_span_s"] = round(tot / 1e9, 4)
    res["kernel_busy_s"] = round(busy / 1e9, 4)
    res["gaps_total_s"] = round((tot - busy) / 1e9, 4)
    json.dump(res, open(out, "w"), indent=1)
    if len(sys.argv) > 4:
        with open(sys.argv[4], "w") as f:
            f.write("| sequence | kind | events | avg m | kernels (us) | gaps (us) | gap before | span (us) |\n|---|---|---|---|---|---|---|---|\n")
            for t in table:
                f.write(f"| {t['sequence']} | {t['kind']} | {t['events']} | {t['avg_m']} | "
                        + ", ".join(f"{k} {v}" for k, v in t["kernel_us"].items()) + " | "
                        + ", ".join(f"{k} {v}" for k, v in t["gap_us"].items()) + f" | {t['gap_before_sequence_us']} | {t['first_start_to_last_end_us']} |\n")
    print(json.dumps({k: v for k, v in res.items() if k != "table"}))
    for t in table:
        print(t)


if __name__ == "__main__":
    main()
