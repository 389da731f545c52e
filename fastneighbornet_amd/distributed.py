"""Several GPUs of one node, one process per GPU (torch.distributed for the bootstrap only).

ONE problem on all ranks: every rank keeps the whole matrix and runs the whole event chain (the ranks stay in step
because every decision is a deterministic function of identical state).  With lookahead windows - the shipped mode
from 4096 taxa on - only the base scans are sharded (tile index mod world) and each is followed by ONE all-gather
(RCCL on the engine's stream): per rank 64 candidate records of 24 bytes plus the tracked-pair records it emitted,
a fixed block of 16 + 1536 + 48 * 65536 / world bytes (48-byte tracked-pair records).  Without windows every event's scan is sharded and exchanges
at most 64 candidate records per rank.  See include/fastnn.h.
"""
from __future__ import annotations

import os


def rccl_path() -> str | None:
    """The librccl that shares a HIP runtime with libfastnn_hip.so.

    PyTorch ships private copies of libamdhip64 / librccl.  If torch was imported before our
    library was loaded, our library runs on torch's HIP runtime (same soname) and torch's
    librccl is the matching one; otherwise ours runs on /opt/rocm's runtime and so must RCCL.
    Mixing them fails in ncclCommInitRank ("unhandled cuda error")."""
    import fastneighbornet_amd as fa
    fa.api()
    if fa.TORCH_LOADED_FIRST:
        import torch
        p = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        if os.path.exists(p):
            return p
    for p in ("/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"):
        if os.path.exists(p):
            return p
    return None


def bootstrap_rccl(dist, device):
    """The collective part of the RCCL bootstrap that needs no engine handle: returns (id bytes, library path).

    Symmetric by construction: every rank performs the same collectives in the same order whatever fails where -
    (1) every rank probes its librccl (dlopen + symbol check: no call into the library, nothing to tear down) and rank 0
    also creates the id; (2) ONE all-gather of {ok, reason} - so a failure on any rank is an exception on ALL ranks,
    naming the rank and its reason, instead of a hang; (3) broadcast of the id from rank 0."""
    import ctypes as C

    import torch

    from . import api
    a = api()
    rank, world = dist.get_rank(), dist.get_world_size()
    # A path the USER names (FNN_RCCL_PATH) is binding: a caller that names a library must not silently get another one.
    # The path this module discovers is only the first candidate: if that file exists but cannot be loaded, the default
    # search (librccl.so.1 on the loader's path) still gets its turn.
    user = os.environ.get("FNN_RCCL_PATH")
    cands = [user] if user else [rccl_path(), None]
    buf = (C.c_uint8 * 128)()
    ok_local, why, path = 0, "", cands[0]
    for cnd in dict.fromkeys(cands):
        try:
            a.check(a.comm_probe(cnd.encode() if cnd else None))
            ok_local, path = 1, cnd
            break
        except Exception as e:  # noqa: BLE001 - reported through the all-gather below
            why = (why + "; " if why else "") + f"{cnd or 'default search'}: {e}"
    if ok_local and rank == 0:
        try:
            a.check(a.comm_unique_id(buf, path.encode() if path else None))
        except Exception as e:  # noqa: BLE001
            ok_local, why = 0, str(e)
    reports = [None] * world
    dist.all_gather_object(reports, (ok_local, why))
    bad = [(r, w) for r, (o, w) in enumerate(reports) if not o]
    if bad:
        raise RuntimeError("RCCL bootstrap failed on rank " + "; rank ".join(f"{r}: {w}" for r, w in bad))
    t = torch.tensor(list(bytes(buf)), dtype=torch.uint8, device=device)
    dist.broadcast(t, src=0)
    return bytes(t.cpu().tolist()), path


def init_rccl(handle, dist, device) -> None:
    """Collective over the default process group: rank 0 creates the RCCL id, everybody joins (bootstrap_rccl, then
    ncclCommInitRank on every rank)."""
    uid, path = bootstrap_rccl(dist, device)
    handle.comm_init_rccl(dist.get_world_size(), dist.get_rank(), uid, path)


def init_gloo(handle, dist) -> None:
    """Test transport over a gloo process group (one host round trip per event)."""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()

    def allgather(send: bytes):
        t = torch.frombuffer(bytearray(send), dtype=torch.uint8)
        out = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(out, t)
        return [bytes(o.numpy().tobytes()) for o in out]

    handle.comm_init_host(world, rank, allgather)
