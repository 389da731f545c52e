"""Several GPUs of one node, one process per GPU (torch.distributed for the bootstrap only).

Every rank keeps the whole matrix and scans 1/world of each event's tiles; one 16-byte record
per rank is all-gathered per event (RCCL on the engine's stream).  See include/fastnn.h.
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
    """Collective over the default process group: rank 0 creates the RCCL id, everybody joins."""
    import ctypes as C

    import torch

    from . import api
    a = api()
    world, rank = dist.get_world_size(), dist.get_rank()
    path = rccl_path()
    buf = (C.c_uint8 * 128)()
    if rank == 0:
        a.check(a.comm_unique_id(buf, path.encode() if path else None))
    t = torch.tensor(list(bytes(buf)), dtype=torch.uint8, device=device)
    dist.broadcast(t, src=0)
    handle.comm_init_rccl(world, rank, bytes(t.cpu().tolist()), path)


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
