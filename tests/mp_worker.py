"""Worker of the multi-process tests: rank `RANK` of `WORLD_SIZE` over gloo.  Runs the engine
(emulation on CPU, or the HIP library with every rank on GPU 0) with the scan sharded over
the ranks and writes the order + event count for the parent test to compare."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    backend, n, seed, dist_name, out_dir = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
    import torch.distributed as dist
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    from fastneighbornet_amd import distributed as fd
    from fastneighbornet_amd._capi import Handle
    if backend == "bootstrap":  # the symmetric RCCL bootstrap alone (no device needed): every rank must raise if any rank cannot load the library
        import torch
        try:
            fd.bootstrap_rccl(dist, torch.device("cpu"))
            res = {"raised": False}
        except RuntimeError as e:
            res = {"raised": True, "message": str(e)}
        json.dump(res, open(os.path.join(out_dir, f"rank{rank}.json"), "w"))
        dist.barrier()
        dist.destroy_process_group()
        return
    from oracle import nnet_oracle as O
    if backend == "emu":
        import ctypes as C
        from fastneighbornet_amd._capi import Api
        lib = C.CDLL(os.path.join(ROOT, "tests", "emu", "build", "libfnn_emu.so"))
        api = Api(lib, "emu_")
    else:
        import fastneighbornet_amd as fa
        api = fa.api()
    import inputs
    D = inputs.make(n, dist_name, seed, O)   # uniform53 / dec4 (SplitMix64) and the host-generated classes: tree, treenoise, neg
    with Handle(api, n, record_events=True) as h:
        fd.init_gloo(h, dist)
        h.set_matrix(D)
        order, st = h.run()
        ev = h.events()
    json.dump({"order": order.tolist(), "n_events": int(st.n_events), "sum_entries": int(st.sum_entries),
               "kinds": ev["kind"].tolist(), "x": ev["x_id"].tolist(), "y": ev["y_id"].tolist(),
               "window_hits": int(st.n_window_hits), "base_scans": int(st.n_base_scans), "rx_exact": int(st.n_rx_exact),
               "screen_events": int(st.n_screen_events)},
              open(os.path.join(out_dir, f"rank{rank}.json"), "w"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
