"""Quick on-GPU timing of whole runs at a few sizes (development aid, not the bench)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastneighbornet_amd as fa
from fastneighbornet_amd._capi import Handle

a = fa.api()
sizes = [int(x) for x in sys.argv[1:]] or [4096]
for n in sizes:
    with Handle(a, n) as h:
        a.set_scan_timing(h._h, 1)
        h.synth(1, "uniform53")
        t = time.time()
        order, st = h.run()
        dt = time.time() - t
        if n <= 8192 and not os.environ.get("FNN_TICKS"):
            # small problems: the first run of a process pays first-touch costs (~10 ms); bench.py times after a warm-up
            # run, so print that figure as well
            first = st.t_total_s
            h.synth(1, "uniform53")
            order, st = h.run()
            print(f"n={n}: first run of the process {first:.3f}s, second run {st.t_total_s:.3f}s", flush=True)
        import ctypes as C
        tk = (C.c_int64 * 8)()
        a._fn("debug_event_ticks", C.c_int32, [C.c_void_p, C.POINTER(C.c_int64)])
        a.debug_event_ticks(h._h, tk)
        if st.n_window_hits:
            print("k_track us/window event (last workgroup, thread 0): " + " ".join(f"{nm}={tk[i] / 100.0 / st.n_window_hits:.2f}" for i, nm in
                  enumerate(["prologue", "pairs", "sweep", "reduce", "arrive", "records", "verdict", "tail"])), flush=True)
        a._fn("debug_decide_ticks", C.c_int32, [C.c_void_p, C.POINTER(C.c_int64)])
        a.debug_decide_ticks(h._h, tk)
        if st.n_window_hits:
            print("  decide step in the tail: " + " ".join(f"{nm}={tk[i] / 100.0 / st.n_window_hits:.2f}" for i, nm in
                  enumerate(["loads", "pick+choice", "plan", "replay"])), flush=True)
        a._fn("debug_plan_ticks", C.c_int32, [C.c_void_p, C.POINTER(C.c_int64)])
        a.debug_plan_ticks(h._h, tk)
        if st.n_events and tk[0]:
            print("  merge plan (all decide steps): " + " ".join(f"{nm}={tk[i] / 100.0 / st.n_events:.2f}" for i, nm in
                  enumerate(["candidates+choice", "ids+counters", "slot_operations", "window_bookkeeping"])), flush=True)
        a._fn("debug_update_ticks", C.c_int32, [C.c_void_p, C.POINTER(C.c_int64)])
        a.debug_update_ticks(h._h, tk)
        if tk[0] or tk[4]:
            print("k_update us/event: " + " ".join(f"{nm}={tk[i] / 100.0 / st.n_events:.2f}" for i, nm in
                  enumerate(["sp_state", "sp_load", "sp_phases", "sp_tail", "bulk_state", "bulk_-", "bulk_columns", "bulk_tail"])), flush=True)
        if os.environ.get("FNN_TICKS"):
            wt = (C.c_int64 * 768)()
            a._fn("debug_update_wg_ticks", C.c_int32, [C.c_void_p, C.POINTER(C.c_int64)])
            a.debug_update_wg_ticks(h._h, wt)
            used = [w for w in range(256) if wt[512 + w]]
            if used:
                # per workgroup: average start and end, relative to the involved slots' workgroup (slot 255: it takes part in every event)
                ref = wt[255] / wt[512 + 255]
                rows = [(w, wt[512 + w], (wt[w] / wt[512 + w] - ref) / 100.0, (wt[256 + w] / wt[512 + w] - ref) / 100.0) for w in used]
                print("k_update per workgroup (events: us from the start of the involved slots' workgroup to this one's start - end; the averages are over "
                      "different sets of events, compare workgroups with similar counts): "
                      + " ".join(f"wg{w}({c}):{a0:+.2f}..{a1:.2f}" for w, c, a0, a1 in rows[:8] + rows[-10:]), flush=True)
        gb = st.scan_bytes / 1e9
        print(f"n={n} total={st.t_total_s:.3f}s init={st.t_init_s:.4f} agglom={st.t_agglom_s:.3f} "
              f"scan={st.t_scan_s:.3f}s events={st.n_events} sumE/n^3={st.sum_entries / n**3:.4f} "
              f"scan_GBps={gb / max(st.t_scan_s, 1e-9):.1f} whole_GBps={gb / st.t_total_s:.1f} "
              f"per_event_overhead_us={(st.t_agglom_s - st.t_scan_s) / max(st.n_events, 1) * 1e6:.1f} "
              f"rx_certified={st.n_rx_certified} rx_exact={st.n_rx_exact} screen_events={st.n_screen_events} "
              f"rescan_units_per_event={st.n_rescan_units / max(st.n_screen_events, 1):.1f} "
              f"base_scans={st.n_base_scans} window_hits={st.n_window_hits} window_fails={st.n_window_fails} "
              f"pairs_per_window={st.window_pairs / max(st.n_base_scans, 1):.0f} bytes_total={st.bytes_total / 1e12:.3f}TB "
              f"timed_launches={st.scan_launches} timed_screen_bytes={st.scan_bytes} plain_launches={st.plain_launches} exact_sweeps={st.n_sweeps_exact} stalled={st.n_stalled_events} handover_retries={st.n_handover_retries}", flush=True)
