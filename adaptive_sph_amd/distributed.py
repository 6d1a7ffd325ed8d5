"""Slab decomposition glue: one process per GPU (torch.distributed is plumbing only -- it carries the RCCL
unique id and the launcher's barrier; the halo exchange itself is RCCL inside libsph_hip.so).

  slab_cuts          x-cuts that give every rank the same number of particles (quantiles of x)
  partition          indices of the particles each rank owns
  make_slab_context  context of this rank, configured, communicator initialised, particles uploaded
  make_loopback_group  k contexts in this process (ranks 0..k-1) for single-GPU verification
"""
from __future__ import annotations

from typing import List, Sequence

import numpy as np

from . import ffi

INF = float(np.finfo(np.float32).max)


def slab_cuts(x: np.ndarray, n_ranks: int) -> List[float]:
    """n_ranks+1 cut positions: -inf, quantiles of x, +inf.  Rank r owns cut[r] <= x < cut[r+1]."""
    xs = np.sort(np.asarray(x, dtype=np.float32))
    cuts = [-INF]
    for r in range(1, n_ranks):
        k = (len(xs) * r) // n_ranks
        # cut half-way between two distinct x values so that equal coordinates stay on one side
        lo = xs[max(k - 1, 0)]
        hi = xs[min(k, len(xs) - 1)]
        cuts.append(float(np.float32((np.float64(lo) + np.float64(hi)) * 0.5)) if hi > lo else float(hi))
    cuts.append(INF)
    return cuts


def partition(x: np.ndarray, cuts: Sequence[float]) -> List[np.ndarray]:
    x = np.asarray(x, dtype=np.float32)
    out = []
    for r in range(len(cuts) - 1):
        lo, hi = np.float32(cuts[r]), np.float32(cuts[r + 1])
        out.append(np.nonzero((x >= lo) & (x < hi))[0] if r + 1 < len(cuts) - 1 else np.nonzero(x >= lo)[0])
    return out


def _slab_capacity(n_total: int, n_ranks: int) -> int:
    # room for imbalance as the fluid moves (static cuts) plus the ghost layers
    return int(n_total / n_ranks * 2.5) + 65536


def make_slab_context(lib: ffi.SphLibrary, pos, mass, vel, planes, rank: int, world: int, local_rank: int) -> ffi.Context:
    import torch.distributed as dist
    import torch
    cuts = slab_cuts(pos[:, 0], world)
    mine = partition(pos[:, 0], cuts)[rank]
    ctx = ffi.Context(lib, _slab_capacity(len(mass), world), planes, device_id=local_rank)
    ctx.dist_configure(rank, world, cuts[rank], cuts[rank + 1])
    ctx.upload(mass[mine], pos[mine], vel[mine])
    if world > 1:
        ctx.upload_field("particle_id", mine.astype(np.uint32))
    # RCCL unique id: created on rank 0, broadcast through the launcher's process group
    buf = torch.zeros(128, dtype=torch.uint8)
    if rank == 0:
        import ctypes as C
        raw = (C.c_uint8 * 128)()
        rc = lib.comm_unique_id(raw)
        if rc != 0:
            raise ffi.SphError(rc, "ncclGetUniqueId failed")
        buf = torch.tensor(list(raw), dtype=torch.uint8)
    dev = torch.device("cuda", local_rank) if dist.get_backend() == "nccl" else torch.device("cpu")
    buf = buf.to(dev)
    dist.broadcast(buf, src=0)
    ctx.comm_init(bytes(buf.cpu().numpy().tolist()), rank, world)
    return ctx


def make_loopback_group(lib: ffi.SphLibrary, pos, mass, vel, planes, n_ranks: int, device_id: int = 0) -> List[ffi.Context]:
    cuts = slab_cuts(pos[:, 0], n_ranks)
    parts = partition(pos[:, 0], cuts)
    ctxs = []
    for r in range(n_ranks):
        c = ffi.Context(lib, _slab_capacity(len(mass), n_ranks), planes, device_id=device_id)
        c.dist_configure(r, n_ranks, cuts[r], cuts[r + 1])
        c.upload(mass[parts[r]], pos[parts[r]], vel[parts[r]])
        c.upload_field("particle_id", parts[r].astype(np.uint32))
        ctxs.append(c)
    return ctxs


def gather_by_id(contexts: Sequence[ffi.Context], name: str, n_total: int) -> np.ndarray:
    """Reassemble a field of all ranks into global particle order (loopback / rank-0 diagnostics)."""
    fid, dt, w = ffi.FIELDS[name]
    out = np.zeros((n_total, w) if w > 1 else (n_total,), dtype=dt)
    for c in contexts:
        ids = c.download("particle_id")
        out[ids] = c.download(name)
    return out
