"""Slab decomposition glue: one process per GPU (torch.distributed is plumbing only -- it carries the RCCL
unique id and the launcher's barrier; the halo exchange itself is RCCL inside libsph_hip.so).

  slab_cuts          x-cuts that give every rank the same number of particles (quantiles of x)
  partition          indices of the particles each rank owns
  make_slab_context  context of this rank, configured, communicator initialised, particles uploaded
  make_loopback_group  k contexts in this process (ranks 0..k-1) for single-GPU verification
  group_single_step_adaptivity  single_step_adaptivity for a slab group, through a gather to one context and back
  rank_single_step_adaptivity   the same with one process per rank: gather / scatter through the launcher's process group
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
    if want in ("shm", "rccl", "ipc"):   # ipc: the peer-mapped push transport (device to device, no RCCL launch per exchange) on top of shm
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
    if transport in ("shm", "ipc"):
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
        if transport == "ipc":
            # every rank exports its device inbox, the launcher all-gathers the 64-byte handles, every rank maps the others'.  A rank
            # whose export or mapping fails tells the others THROUGH the gather (None / a flag) instead of leaving them in it: every rank
            # then raises the same error
            try:
                mine_h, err = ctx.comm_ipc_export(per_side), None
            except ffi.SphError as e:
                mine_h, err = None, str(e)
            handles = [None] * world
            dist.all_gather_object(handles, (mine_h, err))
            bad = [(r, h[1]) for r, h in enumerate(handles) if h[0] is None]
            if bad:
                raise ffi.SphError(2, f"peer-mapped transport: rank {bad[0][0]} could not export its inbox: {bad[0][1]}")
            try:
                ctx.comm_init_ipc(b"".join(h[0] for h in handles), world)
                err = None
            except ffi.SphError as e:
                err = str(e)
            errs = [None] * world
            dist.all_gather_object(errs, err)
            bad = [(r, e) for r, e in enumerate(errs) if e]
            if bad:
                raise ffi.SphError(2, f"peer-mapped transport: rank {bad[0][0]} could not map its peers' inboxes: {bad[0][1]}")
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


class GatherContext:
    """The ONE plain context an adaptive step of a slab decomposition runs on (the reference decides sequentially over ALL particles
    in index order, and a transfer's two partners may sit on different ranks: exactness goes through one place).  Kept between
    adaptive steps -- allocating and freeing every per-particle buffer each time costs more than the step -- on the device the
    caller names.  MEMORY: the whole particle set of the decomposition must fit that ONE device next to the rank's own slab
    (~290 B per particle of capacity): scenes that need slabs to fit at all cannot take this path."""

    def __init__(self, lib: ffi.SphLibrary, planes, device_id: int = 0, split_patterns=None, log=None):
        self.lib, self.planes, self.device_id, self.split_patterns, self.log = lib, planes, device_id, split_patterns, log
        self.ctx = None
        self.driver = None

    def ensure(self, n: int, capacity: int = 0):
        from .adaptivity import AdaptivityDriver
        want = capacity or max(2 * n, n + 65536)
        if self.ctx is None or self.ctx.capacity < want:
            self.close()
            self.ctx = ffi.Context(self.lib, want, self.planes, device_id=self.device_id)
            self.driver = AdaptivityDriver(self.ctx, self.split_patterns, self.log)
        return self.ctx

    def close(self):
        if self.ctx is not None:
            self.ctx.close()
        self.ctx, self.driver = None, None


def _adapt_gathered(gc: GatherContext, P, dt: float, step_number: int, g: dict, ids: Sequence[np.ndarray], lists: Sequence, capacity: int = 0):
    """`g`: the fields of ALL particles in global index order; `ids` / `lists`: the ranks' ids and exported neighbour lists.
    -> (fields of the new vector in its index order, info)."""
    n = len(g["mass"])
    off_g, idx_g = assemble_lists(ids, lists, n)   # neighbour lists in index order
    if off_g[-1] >= 2 ** 32:
        raise ValueError(f"adaptive step of a slab decomposition: {int(off_g[-1])} neighbour-list entries do not fit the 32-bit CSR offsets of the partner search")
    T = gc.ensure(n, capacity)
    T.upload(g["mass"], g["position"], g["velocity"])
    for f in _ADAPT_FIELDS:
        T.upload_field(f, g[f])
    info = gc.driver.single_step_adaptivity(P, dt, step_number, lists=(off_g.astype(np.uint32), idx_g))
    new = {f: T.download(f) for f in ("mass", "position", "velocity") + _ADAPT_FIELDS}
    info["n_after"] = len(new["mass"])
    return new, info


def _reupload(c: ffi.Context, new: dict, mine: np.ndarray):
    c.upload(new["mass"][mine], new["position"][mine], new["velocity"][mine])
    for f in _ADAPT_FIELDS:
        c.upload_field(f, new[f][mine])
    c.upload_field("particle_id", mine.astype(np.uint32))


def group_single_step_adaptivity(lib: ffi.SphLibrary, contexts: Sequence[ffi.Context], planes, P, dt: float, step_number: int,
                                 split_patterns=None, capacity: int = 0, log=None, gather: GatherContext = None) -> dict:
    """single_step_adaptivity (simulation.rs:2732-2796) for the ranks of a slab decomposition that have just stepped.

    The reference decides sequentially over ALL particles in index order and a transfer's two partners may sit on different
    ranks, so the exact form goes through one place: the owned particles of every rank (mass, position, velocity, h2, h2_next,
    level values, size class) and their neighbour lists (global ids, ghosts included) are assembled in global index order -- the
    particle ids ARE the reference's Vec indices --, uploaded to ONE plain context (GatherContext: pass one to keep it between
    steps), the same decisions (adaptivity.py) and the same device-side share / merge / split run there, and the result goes
    back: every rank is re-uploaded with the particles of its slab, their new index as id.  Same arithmetic as on a single
    context; the price is a gather and a scatter over PCIe per adaptive step (support lengths FromMass: the previous step's lambda
    sums do not travel).  In-process form (loopback group, or one process driving several GPUs); rank_single_step_adaptivity is
    the same for one process per GPU."""
    if P.support_length_estimation != "FromMass":
        raise ValueError("group_single_step_adaptivity: support_length_estimation must be FromMass")
    ids = [c.download("particle_id") for c in contexts]
    n = int(sum(len(i) for i in ids))
    allid = np.concatenate(ids)
    if not np.array_equal(np.sort(allid), np.arange(n, dtype=allid.dtype)):
        raise ValueError("group_single_step_adaptivity: the particle ids of the ranks are not the indices 0 .. n-1")
    g = {f: gather_by_id(contexts, f, n) for f in ("mass", "position", "velocity") + _ADAPT_FIELDS}
    own = gather is None
    gc = gather or GatherContext(lib, planes, 0, split_patterns, log)
    try:
        new, info = _adapt_gathered(gc, P, dt, step_number, g, ids, [c.download_neighbors() for c in contexts], capacity)
    finally:
        if own:
            gc.close()
    cuts = [contexts[0].dist_get_cuts()[0]] + [c.dist_get_cuts()[1] for c in contexts]
    parts = partition(new["position"][:, 0], [-INF] + [float(v) for v in cuts[1:-1]] + [INF])
    for c, mine in zip(contexts, parts):
        _reupload(c, new, mine)
    return info


def _decide_on_gathered(lib, kind: str, g: dict, lists, P, dt: float):
    """find_share_partner_sequential / find_merge_partner_sequential over the WHOLE vector (fields in global index order)"""
    from .adaptivity import find_partners_native
    return find_partners_native(lib, kind, g["particle_size_class"], g["mass"], g["level_estimation"], g["position"], g["h2"], lists[0], lists[1], P, dt)


def group_single_step_adaptivity_on_slabs(lib: ffi.SphLibrary, contexts: Sequence[ffi.Context], P, dt: float, step_number: int, log=None) -> dict:
    """single_step_adaptivity (simulation.rs:2732-2796) for the ranks of a slab group that have just stepped, WITHOUT moving the
    particles: the DECISIONS are taken where the reference takes them -- sequentially over the whole vector, here on the host over
    the fields the searches read (size class, mass, level, position, h2: 21 B per particle) and the step's neighbour lists, gathered
    by global id -- and applied by every rank to its own slab (sph_group_adapt: the slab form of share / merge / split).  Same
    arithmetic, same indices as on one context; nothing is re-uploaded.  (Every context needs its split patterns:
    Context.set_split_patterns.)"""
    from .adaptivity import adapt_params
    if P.support_length_estimation != "FromMass":
        raise ValueError("group_single_step_adaptivity_on_slabs: support_length_estimation must be FromMass")
    p, ap = P.to_ffi(), adapt_params(P, dt)
    ids = [c.download("particle_id") for c in contexts]
    n = int(sum(len(i) for i in ids))
    info = {"n_before": n, "shares": 0, "merges": 0, "splits": 0}
    off_g, idx_g = assemble_lists(ids, [c.download_neighbors() for c in contexts], n)   # the step's lists, kept across the passes like self.neighs
    if off_g[-1] >= 2 ** 32:
        raise ValueError(f"adaptive step of a slab decomposition: {int(off_g[-1])} neighbour-list entries do not fit the 32-bit CSR offsets of the partner search")
    lists = (off_g.astype(np.uint32), idx_g)
    total_mass1 = float(sum(c.download("mass").sum(dtype=np.float64) for c in contexts))

    def gathered():
        for c in contexts:
            c.classify(p)
        return {f: gather_by_id(contexts, f, n) for f in ("particle_size_class", "mass", "level_estimation", "position", "h2")}

    if P.sharing:
        mp, mc = _decide_on_gathered(lib, "share", gathered(), lists, P, dt)
        info["shares"] = int(mc.sum())
        if log:
            log(f"SEQUENTIAL SHARE {info['shares']} shares")
        ffi.group_adapt(contexts, "share", p, ap, mp, mc)
    if step_number % 2 == 0:
        if P.merging:
            mp, mc = _decide_on_gathered(lib, "merge", gathered(), lists, P, dt)
            info["merges"] = int(mc.sum())
            if log:
                log(f"SEQUENTIAL MERGE {info['merges']} merges")
            ffi.group_adapt(contexts, "merge", p, ap, mp, mc)
    elif P.splitting:
        for c in contexts:
            c.classify(p)
        ffi.group_adapt(contexts, "split", p, ap)
        info["splits"] = int(sum(c.n for c in contexts)) - n
    total_mass2 = float(sum(c.download("mass").sum(dtype=np.float64) for c in contexts))
    if not abs(total_mass1 - total_mass2) <= 0.005:             # assert_ft_approx_eq(total_mass1, total_mass2, 0.005, "mass sum"), in f64 (adaptivity.py)
        raise AssertionError(f"mass sum: {total_mass1} vs {total_mass2}")
    info["n_after"] = int(sum(c.n for c in contexts))
    return info


def rank_single_step_adaptivity_on_slabs(ctx: ffi.Context, P, dt: float, step_number: int, root: int = 0) -> dict:
    """The same with ONE PROCESS PER RANK (the RCCL / shared-memory / thread transports): every rank sends `root` the fields the
    partner searches read and its exported neighbour lists through the launcher's process group, `root` decides over the whole
    vector and broadcasts merge_partner / merge_counter (6 B per particle), every rank applies them to its own slab (the slab form
    of sph_share_particles / sph_merge_particles / sph_split_particles: collective calls through the context's own transport).
    Every rank calls this after the same sph_step; a failure on the root is re-raised on every rank."""
    import torch.distributed as dist
    from .adaptivity import adapt_params
    if P.support_length_estimation != "FromMass":
        raise ValueError("rank_single_step_adaptivity_on_slabs: support_length_estimation must be FromMass")
    import time
    rank, world = dist.get_rank(), dist.get_world_size()
    p, ap = P.to_ffi(), adapt_params(P, dt)
    # where the adaptive step's time goes (this rank's clock): device -> host (lists, fields), the launcher's
    # gather / broadcast (pickled numpy arrays), the host's partner search on the root, the apply calls on the slabs
    tm = {"download_s": 0.0, "gather_broadcast_s": 0.0, "host_decide_s": 0.0, "apply_s": 0.0}
    t0 = time.perf_counter()
    lists_mine = ctx.download_neighbors()
    ids_mine = ctx.download("particle_id")
    tm["download_s"] += time.perf_counter() - t0
    state = {"lists": None, "n": 0}

    def decide(kind):
        """-> (merge_partner, merge_counter) of the whole vector on every rank"""
        t0 = time.perf_counter()
        ctx.classify(p)
        mine = {f: ctx.download(f) for f in ("particle_size_class", "mass", "level_estimation", "position", "h2")}
        tm["download_s"] += time.perf_counter() - t0
        mine["particle_id"] = ids_mine
        if state["lists"] is None:
            mine["lists"] = lists_mine
        parts = [None] * world if rank == root else None
        t0 = time.perf_counter()
        dist.gather_object(mine, parts, dst=root)
        tm["gather_broadcast_s"] += time.perf_counter() - t0
        out = [None]
        t0 = time.perf_counter()
        if rank == root:
            try:
                n = int(sum(len(q["particle_id"]) for q in parts))
                if state["lists"] is None:
                    off_g, idx_g = assemble_lists([q["particle_id"] for q in parts], [q["lists"] for q in parts], n)
                    if off_g[-1] >= 2 ** 32:
                        raise ValueError(f"{int(off_g[-1])} neighbour-list entries do not fit the 32-bit CSR offsets of the partner search")
                    state["lists"] = (off_g.astype(np.uint32), idx_g)
                g = {}
                for f in ("particle_size_class", "mass", "level_estimation", "position", "h2"):
                    a = np.zeros((n,) + parts[0][f].shape[1:], parts[0][f].dtype)
                    for q in parts:
                        a[q["particle_id"]] = q[f]
                    g[f] = a
                out = [_decide_on_gathered(ctx.lib, kind, g, state["lists"], P, dt)]
            except Exception as e:  # noqa: BLE001 -- re-raised on every rank below
                out = [(None, (type(e).__name__, e.status if isinstance(e, ffi.SphError) else None, str(e)))]
        else:
            state["lists"] = True   # (only the root keeps them)
        tm["host_decide_s"] += time.perf_counter() - t0
        t0 = time.perf_counter()
        dist.broadcast_object_list(out, src=root)
        tm["gather_broadcast_s"] += time.perf_counter() - t0
        if out[0][0] is None:
            kind_, code, msg = out[0][1]
            if code is not None:
                raise ffi.SphError(code, f"adaptive step failed on rank {root}: {msg}")
            raise RuntimeError(f"adaptive step failed on rank {root} ({kind_}): {msg}")
        return out[0]

    def total(v):
        t = [None] * world
        dist.all_gather_object(t, v)
        return sum(t)

    n_before = total(ctx.n)
    info = {"n_before": n_before, "shares": 0, "merges": 0, "splits": 0}
    m1 = total(float(ctx.download("mass").sum(dtype=np.float64)))
    def apply(f, *a):
        t0 = time.perf_counter()
        f(*a)
        tm["apply_s"] += time.perf_counter() - t0

    if P.sharing:
        mp, mc = decide("share")
        info["shares"] = int(mc.sum())
        apply(ctx.share_particles, p, ap, mp, mc)
    if step_number % 2 == 0:
        if P.merging:
            mp, mc = decide("merge")
            info["merges"] = int(mc.sum())
            apply(ctx.merge_particles, p, ap, mp, mc)
    elif P.splitting:
        apply(ctx.classify, p)
        apply(ctx.split_particles, p, ap)
    info["n_after"] = total(ctx.n)
    info["seconds"] = tm
    info["splits"] = max(0, info["n_after"] - n_before) if step_number % 2 == 1 else 0
    m2 = total(float(ctx.download("mass").sum(dtype=np.float64)))
    if not abs(m1 - m2) <= 0.005:
        raise AssertionError(f"mass sum: {m1} vs {m2}")
    return info


def rank_single_step_adaptivity(ctx: ffi.Context, gather: GatherContext, P, dt: float, step_number: int, capacity: int = 0, root: int = 0) -> dict:
    """The same adaptive step with ONE PROCESS PER RANK (the launch bench.py --gpus N and the RCCL / shared-memory transports use):
    every rank hands the fields adaptivity reads and its exported neighbour lists to `root` through the launcher's process group
    (torch.distributed gather_object: pickled numpy arrays -- host memory either way, the decisions are host code), `root`
    assembles them in global index order, runs the decisions and the device-side apply on its GatherContext (`gather`; None on
    the other ranks) and scatters every rank the particles of its slab, which it uploads with their new indices as ids.  Every
    rank calls this after the same sph_step; returns the step's counts (shares / merges / splits, n_after) on every rank."""
    import torch.distributed as dist
    if P.support_length_estimation != "FromMass":
        raise ValueError("rank_single_step_adaptivity: support_length_estimation must be FromMass")
    rank, world = dist.get_rank(), dist.get_world_size()
    mine = {f: ctx.download(f) for f in ("particle_id", "mass", "position", "velocity") + _ADAPT_FIELDS}
    mine["lists"] = ctx.download_neighbors()
    mine["cuts"] = ctx.dist_get_cuts()[:2]
    parts = [None] * world if rank == root else None
    dist.gather_object(mine, parts, dst=root)
    out = [None] * world
    if rank == root:
        # whatever goes wrong on the root between the gather and the scatter (the ids check, the 2^32-entries check, the mass-sum
        # assertion, an SphError of the gather context) must reach EVERY rank: the others sit in scatter_object_list meanwhile
        try:
            ids = [q["particle_id"] for q in parts]
            n = int(sum(len(i) for i in ids))
            allid = np.concatenate(ids)
            if not np.array_equal(np.sort(allid), np.arange(n, dtype=allid.dtype)):
                raise ValueError("rank_single_step_adaptivity: the particle ids of the ranks are not the indices 0 .. n-1")
            g = {}
            for f in ("mass", "position", "velocity") + _ADAPT_FIELDS:
                a = np.zeros((n,) + parts[0][f].shape[1:], parts[0][f].dtype)
                for q in parts:
                    a[q["particle_id"]] = q[f]
                g[f] = a
            new, info = _adapt_gathered(gather, P, dt, step_number, g, ids, [q["lists"] for q in parts], capacity)
            cuts = [-INF] + [float(q["cuts"][1]) for q in parts[:-1]] + [INF]
            for r, sel in enumerate(partition(new["position"][:, 0], cuts)):
                out[r] = ({f: new[f][sel] for f in new}, sel, info)
        except Exception as e:  # noqa: BLE001 -- re-raised on every rank below
            code = e.status if isinstance(e, ffi.SphError) else None
            out = [(None, None, (type(e).__name__, code, str(e)))] * world
    got = [None]
    dist.scatter_object_list(got, out if rank == root else None, src=root)
    if got[0][0] is None:
        kind, code, msg = got[0][2]
        if code is not None:
            raise ffi.SphError(code, f"adaptive step failed on rank {root}: {msg}")
        raise RuntimeError(f"adaptive step failed on rank {root} ({kind}): {msg}")
    new, sel, info = got[0]
    _reupload(ctx, {f: v for f, v in new.items()}, np.arange(len(sel)))
    ctx.upload_field("particle_id", sel.astype(np.uint32))
    return dict(info)


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
