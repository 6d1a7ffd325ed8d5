"""Slab decomposition glue: one process per GPU (torch.distributed is plumbing only -- it carries the RCCL
unique id and the launcher's barrier; the halo exchange itself is RCCL inside libsph_hip.so).

  slab_cuts          x-cuts that give every rank the same number of particles (quantiles of x)
  partition          indices of the particles each rank owns
  make_slab_context  context of this rank, configured, communicator initialised, particles uploaded
  make_loopback_group  k contexts in this process (ranks 0..k-1) for single-GPU verification
  group_single_step_adaptivity  single_step_adaptivity for a slab group, through a gather to one context and back
  ThreadedGroup      k ranks of this process, one host thread each, every rank calling sph_step by itself (verification of the per-rank code)
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


def pick_transport(world: int) -> str:
    """"rccl" -- one rank per GPU over xGMI, the fast path -- unless the launch puts several ranks on one device (RCCL refuses that:
    "Duplicate GPU detected") or SPH_TRANSPORT=shm asks for the host-staged shared-memory transport (sph_comm_init_shm)."""
    import os
    import torch
    want = os.environ.get("SPH_TRANSPORT", "")
    if want in ("shm", "rccl"):
        return want
    return "shm" if (torch.cuda.is_available() and world > torch.cuda.device_count()) else "rccl"


_SHM_SERIAL = [0]


def make_slab_context(lib: ffi.SphLibrary, pos, mass, vel, planes, rank: int, world: int, local_rank: int, transport: str = None) -> ffi.Context:
    """This rank's slab context of a `torch.distributed` launch (one process per rank): configured, communicator attached, particles
    uploaded.  torch.distributed only carries the RCCL unique id (or the shared-memory segment's name) and a barrier."""
    import os
    import torch.distributed as dist
    import torch
    transport = transport or pick_transport(world)
    cuts = slab_cuts(pos[:, 0], world)
    mine = partition(pos[:, 0], cuts)[rank]
    cap = _slab_capacity(len(mass), world)
    ctx = ffi.Context(lib, cap, planes, device_id=local_rank)
    ctx.dist_configure(rank, world, cuts[rank], cuts[rank + 1])
    ctx.upload(mass[mine], pos[mine], vel[mine])
    if world > 1:
        ctx.upload_field("particle_id", mine.astype(np.uint32))
    if transport == "shm":
        # rank 0 names and creates the segment, the others map it after the launcher's barrier; an outbox holds one message to one
        # x-neighbour: at most every particle of the slab as a 48-byte migrant record
        _SHM_SERIAL[0] += 1
        name = [f"/sph_shm_{os.getpid()}_{_SHM_SERIAL[0]}" if rank == 0 else None]
        dist.broadcast_object_list(name, src=0)
        per_side = max(1 << 22, cap * 48)
        if rank == 0:
            ctx.comm_init_shm(name[0], rank, world, per_side, True)
        dist.barrier()
        if rank != 0:
            ctx.comm_init_shm(name[0], rank, world, per_side, False)
        dist.barrier()
        return ctx
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


_ADAPT_FIELDS = ("h2", "h2_next", "level_estimation", "level_old", "particle_size_class")


def assemble_lists(ids: Sequence[np.ndarray], lists: Sequence, n: int):
    """The ranks' exported neighbour lists -- rank r: rows of its owned particles in the order of ids[r], (offsets, indices) with
    global ids as indices -- as ONE CSR in global index order: row i = the list of the particle with id i."""
    counts = np.zeros(n, np.int64)
    for i, (off, _) in zip(ids, lists):
        counts[np.asarray(i, np.int64)] = np.diff(np.asarray(off, np.int64))
    off_g = np.zeros(n + 1, np.int64)
    np.cumsum(counts, out=off_g[1:])
    idx_g = np.empty(int(off_g[-1]), np.uint32)
    for i, (off, idx) in zip(ids, lists):
        off = np.asarray(off, np.int64)
        lens = np.diff(off)
        # entry e of this rank's index array sits in row r(e); its place in the global CSR = start of that row's global row + (e - off[r])
        dst = np.repeat(off_g[np.asarray(i, np.int64)] - off[:-1], lens) + np.arange(len(idx))
        idx_g[dst] = idx
    return off_g, idx_g


def group_single_step_adaptivity(lib: ffi.SphLibrary, contexts: Sequence[ffi.Context], planes, P, dt: float, step_number: int,
                                 split_patterns=None, capacity: int = 0, log=None) -> dict:
    """single_step_adaptivity (simulation.rs:2732-2796) for the ranks of a slab decomposition that have just stepped.

    The reference decides sequentially over ALL particles in index order and a transfer's two partners may sit on different
    ranks, so the exact form goes through one place: the owned particles of every rank (mass, position, velocity, h2, h2_next,
    level values, size class) and their neighbour lists (global ids, ghosts included) are assembled in global index order -- the
    particle ids ARE the reference's Vec indices --, uploaded to ONE plain context, the same decisions (adaptivity.py) and the
    same device-side share / merge / split run there, and the result goes back: every rank is re-uploaded with the particles of
    its slab, their new index as id.  Same arithmetic as on a single context; the price is a gather and a scatter over PCIe per
    adaptive step (support lengths FromMass: the previous step's lambda sums do not travel).  In-process form (loopback group,
    or one process driving several GPUs); with one process per GPU the same gather / scatter goes through the launcher."""
    from .adaptivity import AdaptivityDriver
    if P.support_length_estimation != "FromMass":
        raise ValueError("group_single_step_adaptivity: support_length_estimation must be FromMass")
    ids = [c.download("particle_id") for c in contexts]
    n = int(sum(len(i) for i in ids))
    allid = np.concatenate(ids)
    if not np.array_equal(np.sort(allid), np.arange(n, dtype=allid.dtype)):
        raise ValueError("group_single_step_adaptivity: the particle ids of the ranks are not the indices 0 .. n-1")
    g = {f: gather_by_id(contexts, f, n) for f in ("mass", "position", "velocity") + _ADAPT_FIELDS}
    off_g, idx_g = assemble_lists(ids, [c.download_neighbors() for c in contexts], n)   # neighbour lists in index order
    T = ffi.Context(lib, capacity or max(2 * n, n + 65536), planes, device_id=0)
    try:
        T.upload(g["mass"], g["position"], g["velocity"])
        for f in _ADAPT_FIELDS:
            T.upload_field(f, g[f])
        info = AdaptivityDriver(T, split_patterns, log).single_step_adaptivity(P, dt, step_number, lists=(off_g.astype(np.uint32), idx_g))
        new = {f: T.download(f) for f in ("mass", "position", "velocity") + _ADAPT_FIELDS}
    finally:
        T.close()
    n_new = len(new["mass"])
    cuts = [contexts[0].dist_get_cuts()[0]] + [c.dist_get_cuts()[1] for c in contexts]
    parts = partition(new["position"][:, 0], [-INF] + [float(v) for v in cuts[1:-1]] + [INF])
    for c, mine in zip(contexts, parts):
        c.upload(new["mass"][mine], new["position"][mine], new["velocity"][mine])
        for f in _ADAPT_FIELDS:
            c.upload_field(f, new[f][mine])
        c.upload_field("particle_id", mine.astype(np.uint32))
    info["n_after"] = n_new
    return info


class ThreadedGroup:
    """k slab contexts of this process stepped the way the ranks of a multi-process run are: every rank calls sph_step on its OWN
    context from its OWN thread (a group of one member, rank-local branches and counts), the collectives meet in host memory
    (sph_thread_group_create / sph_comm_init_threads).  A collective that not every rank enters, ranks in different collectives or
    a send without a matching receive -- what would hang the RCCL transport -- comes back as an error."""

    def __init__(self, lib: ffi.SphLibrary, pos, mass, vel, planes, n_ranks: int, device_id: int = 0, capacities=None):
        import ctypes as C
        from concurrent.futures import ThreadPoolExecutor
        self.lib = lib
        self.group = C.c_void_p()
        rc = lib.thread_group_create(n_ranks, C.byref(self.group))
        if rc != 0:
            raise ffi.SphError(rc, "sph_thread_group_create failed")
        cuts = slab_cuts(pos[:, 0], n_ranks)
        parts = partition(pos[:, 0], cuts)
        self.contexts = []
        for r in range(n_ranks):
            cap = capacities[r] if capacities is not None and capacities[r] else _slab_capacity(len(mass), n_ranks)
            c = ffi.Context(lib, cap, planes, device_id=device_id)
            c.dist_configure(r, n_ranks, cuts[r], cuts[r + 1])
            c.comm_init_threads(self.group, r, n_ranks)
            c.upload(mass[parts[r]], pos[parts[r]], vel[parts[r]])
            c.upload_field("particle_id", parts[r].astype(np.uint32))
            self.contexts.append(c)
        self.pool = ThreadPoolExecutor(n_ranks)

    def step(self, p):
        """One sph_step per rank, concurrently (ctypes drops the GIL inside the call).  Returns the ranks' stats; raises the first
        rank's error after ALL ranks have returned."""
        futs = [self.pool.submit(c.step, p) for c in self.contexts]
        out, err = [], None
        for f in futs:
            try:
                out.append(f.result())
            except ffi.SphError as e:
                err = err or e
        if err:
            raise err
        return out

    def close(self):
        self.pool.shutdown(wait=True)
        for c in self.contexts:
            c.close()
        self.contexts = []
        if self.group:
            self.lib.thread_group_destroy(self.group)
            self.group = None
