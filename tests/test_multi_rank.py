"""The N > 1 path: ONE problem on several ranks.  Every rank holds the matrix and runs the whole event chain;
with lookahead windows (the shipped mode) only the base scans are sharded over the ranks (tile index mod world)
and followed by one exchange - candidate records + the pairs each rank emitted for the new window; without
windows every event's scan is sharded and exchanges its candidate records.  On CPU (gloo, world_size 2 and 3)
through the emulation; on the GPU box with every rank on the one GPU (host-callback transport; RCCL itself needs
distinct GPUs and is exercised by the driver's multi-GPU bench)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_ranks(backend, world, n, seed, dist_name, tmp_path, port, extra_env=None):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world))
    env.update(extra_env or {})
    procs = []
    for r in range(world):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "mp_worker.py"), backend, str(n),
                                       str(seed), dist_name, str(tmp_path)], env=e, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o)
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
    return [json.load(open(os.path.join(str(tmp_path), f"rank{r}.json"))) for r in range(world)]


def check(res, oracle, n, seed, dist_name):
    import inputs
    o_ref, ev_ref, se = oracle.run(inputs.make(n, dist_name, seed, oracle))
    for r in res:
        assert r["order"] == o_ref.tolist()
        assert r["n_events"] == len(ev_ref) and r["sum_entries"] == se
        assert r["kinds"] == ev_ref["kind"].tolist()
        assert r["x"] == ev_ref["x_id"].tolist() and r["y"] == ev_ref["y_id"].tolist()


@pytest.mark.parametrize("world,n,seed,dist_name", [(2, 70, 1, "uniform53"), (2, 45, 2, "dec4"), (3, 64, 3, "uniform53"),
                                                    (2, 300, 4, "uniform53"), (3, 257, 5, "dec4")])
def test_emulation_over_gloo(emu_api, oracle, tmp_path, world, n, seed, dist_name):
    """Windows on (the emulation screens from 8 taxa on): sharded base scans, one exchange per base scan."""
    res = run_ranks("emu", world, n, seed, dist_name, tmp_path, 29511 + world + n)
    check(res, oracle, n, seed, dist_name)
    assert all(r["window_hits"] > 0 and r["base_scans"] > 0 for r in res), [(r["window_hits"], r["base_scans"]) for r in res]
    assert len({(r["window_hits"], r["base_scans"]) for r in res}) == 1  # the ranks stay in step


@pytest.mark.parametrize("world,n,seed,env", [(2, 90, 6, {"FNN_LA_K": "-1"}),          # no windows: every event's scan sharded
                                              (3, 120, 7, {"FNN_LA_PCAP": "40"}),       # tracked list overflows after the merge
                                              (2, 150, 8, {"FNN_LA_K": "3", "FNN_LA_TARGET": "8"})])
def test_emulation_over_gloo_modes(emu_api, oracle, tmp_path, world, n, seed, env):
    res = run_ranks("emu", world, n, seed, "uniform53", tmp_path, 29711 + world + n, env)
    check(res, oracle, n, seed, "uniform53")


@pytest.mark.parametrize("world,n,seed,dist_name,env", [
    (2, 220, 5, "tree", None),                              # exact ties of Q everywhere: the tie-breaks and the exact ComputeRx sums decide
    (3, 150, 6, "tree", {"FNN_LA_K": "4"}),
    (2, 200, 7, "treenoise", None),
    (2, 180, 1, "neg", {"FNN_EMU_PLAIN_NEG": "1"}),         # the PRODUCT's path for negative entries: plain scan of every event, sharded + exchanged
    (3, 130, 2, "neg", {"FNN_EMU_PLAIN_NEG": "1"}),
    (2, 160, 3, "neg", None)])                               # (the emulation's own default: mixed-sign screening brackets, windows refused)
def test_emulation_over_gloo_input_classes(emu_api, oracle, tmp_path, world, n, seed, dist_name, env):
    """Several ranks on the input classes that leave the fast path (tests/inputs.py): a tie-rich tree metric, tree + noise,
    and a matrix with negative entries - with FNN_EMU_PLAIN_NEG the emulation follows the product (fnn_engine.h: begin turns
    screening and windows off for such a matrix; every event's scan is then sharded by tile index mod world and its candidate
    records exchanged: the one mode in which sharding every event pays)."""
    res = run_ranks("emu", world, n, seed, dist_name, tmp_path, 29311 + 7 * world + n, env)
    check(res, oracle, n, seed, dist_name)
    assert len({(r["window_hits"], r["base_scans"], r["rx_exact"]) for r in res}) == 1  # the ranks stay in step
    if dist_name == "neg":
        assert all(r["window_hits"] == 0 for r in res)
        if env:
            assert all(r["screen_events"] == 0 and r["base_scans"] == 0 for r in res), [(r["screen_events"], r["base_scans"], r["n_events"]) for r in res]
        else:
            assert all(r["screen_events"] > 0 for r in res)
    if dist_name == "tree":
        assert all(r["rx_exact"] > 0 for r in res)


def test_a_give_up_on_one_rank_is_an_error_on_all_ranks(emu_api, tmp_path):
    """Several ranks with windows rely on identical decisions everywhere.  A window event that ONE rank gives up (on the GPU:
    a record of k_track's fan-in that could not be read back; here injected, FNN_FAULT_GIVEUP=rank:event) makes that rank
    scan and exchange at an event where the others do not: the exchange blocks carry their event number, and a block
    of another event is an error on every rank - never a silently different order, never a hang."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29893", WORLD_SIZE="2", FNN_FAULT_GIVEUP="1:40")
    procs = []
    for r in range(2):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "mp_worker.py"), "emu", "300", "4", "uniform53", str(tmp_path)],
                                      env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=120)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise AssertionError("the ranks hang after a one-sided give-up")
        outs.append(o)
    assert all(p.returncode != 0 for p in procs), outs
    assert all("code 12" in o for o in outs), outs


def test_an_error_on_one_rank_stops_every_rank_at_the_same_round_trip(emu_api, tmp_path):
    """An internal-consistency error in ONE rank's device state (injected: FNN_FAULT_ERROR=rank:event writes code 99 into rank 1's
    state after that event) would stop that rank alone and leave the others waiting in their next exchange.  The ranks gather
    their error words at every host round trip (fnn_engine.h: pull_state_ranks): both must raise, both naming rank 1 - no hang."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29871", WORLD_SIZE="2", FNN_FAULT_ERROR="1:40", FNN_BATCH="16")
    procs = []
    for r in range(2):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "mp_worker.py"), "emu", "300", "4", "uniform53", str(tmp_path)],
                                      env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=120)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise AssertionError("the ranks hang after an error on one of them")
        outs.append(o)
    assert all(p.returncode != 0 for p in procs), outs
    assert all("code 99 on rank 1" in o for o in outs), outs


def test_rccl_bootstrap_fails_on_every_rank_when_one_rank_cannot_load_the_library(tmp_path):
    """distributed.bootstrap_rccl is symmetric: rank 1 is pointed at a library that does not exist; BOTH ranks must raise,
    and the message must name rank 1 (no rank may go on to ncclCommInitRank and hang there)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29877", WORLD_SIZE="2")
    procs = []
    for r in range(2):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        if r == 1:
            e["FNN_RCCL_PATH"] = "/nonexistent/librccl.so"
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "mp_worker.py"), "bootstrap", "0", "0", "-", str(tmp_path)],
                                      env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    for p in procs:
        o, _ = p.communicate(timeout=300)
        assert p.returncode == 0, o
    res = [json.load(open(os.path.join(str(tmp_path), f"rank{r}.json"))) for r in range(2)]
    assert all(r["raised"] for r in res), res
    assert all("rank 1" in r["message"] and "/nonexistent/librccl.so" in r["message"] for r in res), res


@pytest.mark.gpu
@pytest.mark.parametrize("world,n,seed,dist_name,screen", [(2, 600, 1, "uniform53", False), (3, 300, 2, "dec4", False),
                                                        (2, 1500, 3, "uniform53", True), (3, 700, 4, "dec4", True)])
def test_hip_ranks_sharing_one_gpu(hip_api, oracle, tmp_path, world, n, seed, dist_name, screen):
    # screen=True: the bf16 screening pass + k_resolve forced on at small n: lookahead windows with sharded base scans
    # (k_emit into the exchange block, k_merge); screen=False: no screening copy, every event's scan sharded
    extra = {"FNN_SCREEN_MIN_N": "8", "FNN_SCREEN_MIN_M": "64"} if screen else None
    res = run_ranks("hip", world, n, seed, dist_name, tmp_path, 29611 + world + (7 if screen else 0), extra)
    check(res, oracle, n, seed, dist_name)


@pytest.mark.gpu
def test_hip_ranks_negative_entries_shard_every_scan(hip_api, tmp_path):
    """The product's several-rank path for a matrix with negative entries (4096 taxa, uniform53 - 0.25): no screening copy
    in use, no windows; EVERY event's plain fp64 scan is sharded by tile index mod world and its candidate records are
    exchanged.  Two ranks on the one GPU, against the oracle's golden (tests/golden/big_4096_neg_s1.npz)."""
    import numpy as np
    res = run_ranks("hip", 2, 4096, 1, "neg", tmp_path, 29655)
    z = np.load(os.path.join(ROOT, "tests", "golden", "big_4096_neg_s1.npz"))
    for r in res:
        assert r["order"] == z["order"].tolist()
        assert r["kinds"] == z["traj"][:, 6].tolist() and r["x"] == z["traj"][:, 4].tolist() and r["y"] == z["traj"][:, 5].tolist()
        assert r["window_hits"] == 0 and r["screen_events"] == 0


@pytest.mark.gpu
@pytest.mark.parametrize("torch_first", [True, False])
def test_rccl_plumbing_single_rank(oracle, torch_first):
    """RCCL needs one GPU per rank, so on a one-GPU box only a 1-rank communicator can be made;
    FNN_COMM_FORCE keeps the exchange path on: dlopen, unique id, ncclCommInitRank, one
    ncclAllGather per event on the engine's stream, the all-gather of the ranks' status words at every host round trip,
    ncclCommDestroy."""
    code = r'''
import os, sys
sys.path.insert(0, os.environ["FNN_ROOT"]); sys.path.insert(0, os.path.join(os.environ["FNN_ROOT"], "tests"))
import ctypes as C
if os.environ["FNN_TORCH_FIRST"] == "1":
    import torch   # bench.py order: torch's HIP runtime serves both
import fastneighbornet_amd as fa
fa.api()
from fastneighbornet_amd import distributed as fd
from fastneighbornet_amd._capi import Handle
from oracle import nnet_oracle as O
a = fa.api()
n = 500
D = O.synth(n, 3)
o_ref, _, _ = O.run(D)
buf = (C.c_uint8 * 128)()
path = fd.rccl_path()
with Handle(a, n) as h:   # HIP is initialised before RCCL is touched, as in bench.py
    a.check(a.comm_unique_id(buf, path.encode() if path else None))
    h.comm_init_rccl(1, 0, bytes(buf), path)
    h.set_matrix(D)
    order, st = h.run()
assert (order == o_ref).all()
# the ranks' status words travel over the same communicator before every host round trip: an error injected into the device
# state (code 99 after event 40) must come back through ncclAllGather and be reported with its rank
os.environ["FNN_FAULT_ERROR"] = "0:40"
os.environ["FNN_BATCH"] = "16"
with Handle(a, n) as h:
    a.check(a.comm_unique_id(buf, path.encode() if path else None))
    h.comm_init_rccl(1, 0, bytes(buf), path)
    h.set_matrix(D)
    try:
        h.run()
        raise SystemExit("the injected error was not reported")
    except RuntimeError as e:
        assert "code 99 on rank 0" in str(e), str(e)
print("RCCL_OK")
'''
    env = dict(os.environ, FNN_COMM_FORCE="1", FNN_ROOT=ROOT, FNN_TORCH_FIRST="1" if torch_first else "0")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, r.stdout + r.stderr
