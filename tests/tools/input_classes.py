"""Run time and window statistics of the default mode per input class (tests/inputs.py) - the evidence beyond the bench's one
matrix that VERDICT r02 asked for.   usage: tests/tools/input_classes.py [n ...]  -> markdown table on stdout"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastneighbornet_amd as fa
from fastneighbornet_amd._capi import Handle
from oracle import nnet_oracle as O
import inputs

CASES = [("uniform53", 1), ("uniform53", 2), ("uniform53", 3), ("dec4", 1), ("tree", 5), ("treenoise", 6), ("neg", 1)]
if os.environ.get("CASES"):  # e.g. CASES=neg,tree
    CASES = [c for c in CASES if c[0] in os.environ["CASES"].split(",")]


def main():
    a = fa.api()
    sizes = [int(x) for x in sys.argv[1:]] or [8192]
    print("| taxa | input class (seed) | s to order | events with a scan | events served by a window | windows that could not certify | "
          "4-candidate choices certified / exact | exact re-sweeps | stalled events | hand-over rereads |")
    print("|---|---|---|---|---|---|---|---|---|---|", flush=True)
    for n in sizes:
        for dist, seed in CASES:
            with Handle(a, n) as h:
                if dist in inputs.DEVICE_DISTS:
                    h.synth(seed, dist)
                else:
                    D = inputs.make(n, dist, seed, O)
                    h.set_matrix(D)
                    del D
                order, st = h.run()
                if n <= 8192:  # (first-touch costs of a process: time the second run, as the bench does)
                    if dist in inputs.DEVICE_DISTS:
                        h.synth(seed, dist)
                        order, st = h.run()
            print(f"| {n} | {dist} ({seed}) | {st.t_total_s:.3f} | {st.n_base_scans} | {st.n_window_hits} | {st.n_window_fails} | "
                  f"{st.n_rx_certified} / {st.n_rx_exact} | {st.n_sweeps_exact} | {st.n_stalled_events} | {st.n_handover_retries} |", flush=True)


if __name__ == "__main__":
    main()
