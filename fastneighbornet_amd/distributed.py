"""Several GPUs of one node, one process per GPU (torch.distributed for the bootstrap only).

ONE problem on all ranks: every rank keeps the whole matrix and runs the whole event chain (the ranks stay in step
because every decision is a deterministic function of identical state).  With lookahead windows - the shipped mode
from 4096 taxa on - only the base scans are sharded (tile index mod world) and each is followed by ONE all-gather
(RCCL on the engine's stream): per rank 64 candidate records of 24 bytes plus the tracked-pair records it emitted,
a fixed block of 16 + 1536 + 16 * 65536 / world bytes.  Without windows every event's scan is sharded and exchanges
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


def init_rccl(handle, dist, device) -> None:
    """Collective over the default process group: rank 0 creates the RCCL id, everybody joins.

    Symmetric by construction: every rank performs the same collectives in the same order whatever fails
    where - (1) broadcast of {status, id} from rank 0, (2) all-reduce of "my library loaded and has the
    symbols", and only if every rank is fine (3) ncclCommInitRank - so a failure on one rank surfaces as
    an exception on ALL ranks instead of a hang."""
    import ctypes as C

    import torch

    from . import api
    a = api()
    world, rank = dist.get_world_size(), dist.get_rank()
    path = rccl_path()
    buf = (C.c_uint8 * 128)()
    status, why = 1, ""
    if rank == 0:
        try:
            a.check(a.comm_unique_id(buf, path.encode() if path else None))
        except Exception as e:  # noqa: BLE001 - reported through the broadcast below
            status, why = 0, str(e)
    t = torch.tensor([status] + list(bytes(buf)), dtype=torch.uint8, device=device)
    dist.broadcast(t, src=0)
    host = t.cpu().tolist()
    ok_local = 1 if host[0] == 1 else 0
    if ok_local:
        try:  # can this rank load librccl at all?  (dlopen + symbol check, no communicator yet)
            probe = (C.c_uint8 * 128)()
            a.check(a.comm_unique_id(probe, path.encode() if path else None))
        except Exception as e:  # noqa: BLE001
            ok_local, why = 0, str(e)
    flag = torch.tensor([ok_local], dtype=torch.int32, device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) != 1:
        raise RuntimeError(f"RCCL bootstrap failed on at least one rank ({why or 'on another rank'})")
    handle.comm_init_rccl(world, rank, bytes(host[1:]), path)


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
