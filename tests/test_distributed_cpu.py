"""N>1 on CPU: two gloo ranks run the slab decomposition's host logic (cuts, partition, ghost selection over TWO support
radii, all-reduced CFL term) and check with the CPU oracle the two properties the device-side halo exchange relies on: a rank's
owned + ghost particles reproduce the GLOBAL neighbour sets and densities of its owned particles, AND of its first ghost ring
(ghosts within one support radius of the cut) -- which is why those ghosts can compute their own pressure acceleration and a
Jacobi iteration needs one neighbour exchange.  (The device path itself is covered on the GPU by tests/test_gpu_slabs.py.)"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from adaptive_sph_amd import distributed as D, ffi, scene as sc
from adaptive_sph_amd.workloads import dam_break_params


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tests.oracle_harness import load_oracle, csr_sets
        olib = load_oracle()
        olib.lib.oracle_set_num_threads(2)
        scn = sc.dam_break_small(64, 32, 1 / 32)
        pos, mass, vel = sc.init_particles(scn)
        rng = np.random.default_rng(5)
        pos = (pos + rng.uniform(-0.2, 0.2, pos.shape).astype(np.float32) / 32).astype(np.float32)   # break the lattice
        planes = sc.boundary_planes(scn.boundary)
        p = dam_break_params().to_ffi()

        cuts = D.slab_cuts(pos[:, 0], world)
        parts = D.partition(pos[:, 0], cuts)
        assert sum(len(x) for x in parts) == len(mass) and len(np.unique(np.concatenate(parts))) == len(mass)
        mine = parts[rank]

        # all-reduced scalars of the step header: h_max (ghost width) and the CFL term
        h = np.float32(1.9) * np.sqrt(mass[mine] / np.float32(1.0) * np.float32(1 / np.pi), dtype=np.float32)
        t = torch.tensor([-float(h.max()), 1.0 / (float((vel[mine] ** 2).sum(1).max()) + 0.01)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        h_max = -t[0].item()
        assert abs(h_max - float((np.float32(1.9) * np.sqrt(mass * np.float32(1 / np.pi))).max())) < 1e-7
        support = 2.0 * h_max
        w = 2.0 * support      # two ghost rings

        # ghost layer: my owned particles within w of a cut go to that neighbour (send/recv over gloo)
        recv = []
        for nb, edge, side in ((rank - 1, cuts[rank] + w, "lo"), (rank + 1, cuts[rank + 1] - w, "hi")):
            if nb < 0 or nb >= world:
                continue
            sel = mine[pos[mine, 0] < edge] if side == "lo" else mine[pos[mine, 0] >= edge]
            cnt = torch.tensor([len(sel)])
            other = torch.zeros(1, dtype=torch.long)
            if rank < nb:
                dist.send(cnt, nb); dist.recv(other, nb)
            else:
                dist.recv(other, nb); dist.send(cnt, nb)
            out = torch.from_numpy(sel.astype(np.int64))
            inc = torch.zeros(int(other.item()), dtype=torch.long)
            if rank < nb:
                dist.send(out, nb); dist.recv(inc, nb)
            else:
                dist.recv(inc, nb); dist.send(out, nb)
            recv.append(inc.numpy())
        ghosts = np.concatenate(recv) if recv else np.zeros(0, np.int64)
        assert len(np.intersect1d(ghosts, mine)) == 0
        local = np.concatenate([mine, ghosts])

        # oracle on owned + ghosts vs oracle on everything
        full = ffi.Context(olib, len(mass), planes)
        full.upload(mass, pos, vel)
        full.step(p)
        loc = ffi.Context(olib, len(local), planes)
        loc.upload(mass[local], pos[local], vel[local])
        loc.step(p)
        n_own = len(mine)
        assert np.array_equal(loc.download("neighbor_count")[:n_own], full.download("neighbor_count")[mine])
        fo, fi = full.download_neighbors()
        lo, li = loc.download_neighbors()
        fs, ls = csr_sets(fo, fi), csr_sets(lo, li)
        for k in range(0, n_own, 7):
            assert np.array_equal(np.sort(local[ls[k]]), fs[mine[k]])
        d_loc, d_full = loc.download("density")[:n_own], full.download("density")[mine]
        assert np.abs(d_loc - d_full).max() <= 2e-6 * d_full.max()     # same neighbours, different summation order
        # first ghost ring: complete neighbourhoods as well (their neighbours lie inside the second ring)
        gx = pos[ghosts, 0]
        ring1 = np.zeros(len(ghosts), bool)
        if rank > 0:
            ring1 |= (gx < cuts[rank]) & (gx >= cuts[rank] - support)
        if rank + 1 < world:
            ring1 |= (gx >= cuts[rank + 1]) & (gx < cuts[rank + 1] + support)
        assert ring1.any() and not ring1.all()
        for k in np.nonzero(ring1)[0][::5]:
            assert np.array_equal(np.sort(local[ls[n_own + k]]), fs[ghosts[k]])
        assert np.array_equal(loc.download("neighbor_count")[n_own:][ring1], full.download("neighbor_count")[ghosts][ring1])
        q.put((rank, "ok", n_own, len(ghosts)))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, "fail: " + traceback.format_exc(), 0, 0))
    finally:
        dist.destroy_process_group()


def test_two_rank_slabs_gloo(oracle_lib):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for pr in procs:
        pr.join(timeout=60)
    for rank, status, n_own, n_ghost in sorted(res):
        assert status == "ok", status
        assert n_own > 0 and n_ghost > 0


def test_cuts_and_partition_properties():
    rng = np.random.default_rng(1)
    x = rng.normal(size=10007).astype(np.float32)
    for k in (1, 2, 3, 8):
        cuts = D.slab_cuts(x, k)
        assert len(cuts) == k + 1 and all(cuts[i] < cuts[i + 1] for i in range(k))
        parts = D.partition(x, cuts)
        assert sum(len(p) for p in parts) == len(x)
        assert len(np.unique(np.concatenate(parts))) == len(x)
        sizes = np.array([len(p) for p in parts])
        assert sizes.max() - sizes.min() <= max(2, len(x) // 500)
    # equal coordinates never straddle a cut
    x = np.repeat(np.arange(10, dtype=np.float32), 100)
    parts = D.partition(x, D.slab_cuts(x, 4))
    for p in parts:
        assert set(np.unique(x[p])).isdisjoint(set(np.unique(np.delete(x, p)))) or len(p) == 0
