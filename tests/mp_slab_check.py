"""Run under torch.distributed.run with N processes (tests/test_gpu_multiprocess.py): every process is one rank of a slab
decomposition of a small dam break -- its own context, its own sph_step calls, the transport distributed.pick_transport chooses
(RCCL with one rank per GPU; the shared-memory transport when the ranks share a device) -- and rank 0 compares the gathered
result with a single context stepping the same scene."""
import os
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from adaptive_sph_amd import ffi, scene as sc  # noqa: E402
from adaptive_sph_amd.distributed import make_slab_context, pick_transport  # noqa: E402
from adaptive_sph_amd.workloads import dam_break_params  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    transport = pick_transport(world)
    dist.init_process_group("nccl" if transport == "rccl" else "gloo", rank=rank, world_size=world)
    lib = ffi.load_product()
    scn = sc.dam_break_small(128, 64, 1 / 64)
    pos, mass, vel = sc.init_particles(scn)
    vel = vel.copy()
    vel[:, 0] = 0.8                                  # particles cross the cuts
    planes = sc.boundary_planes(scn.boundary)
    p = dam_break_params().to_ffi()
    steps = int(os.environ.get("MP_STEPS", "12"))
    ctx = make_slab_context(lib, pos, mass, vel, planes, rank, world, local, transport)
    its = []
    for _ in range(steps):
        st = ctx.step(p)
        its.append((float(st.dt), int(st.div_solver.iters), int(st.density_solver.iters)))
    mine = {f: ctx.download(f) for f in ("particle_id", "position", "velocity", "density", "neighbor_count")}
    stats = ctx.dist_get_stats()
    parts = [None] * world
    dist.all_gather_object(parts, (mine, its, stats))
    if rank == 0:
        single = ffi.Context(lib, len(mass), planes, device_id=local)
        single.upload(mass, pos, vel)
        ref_its = []
        for _ in range(steps):
            st = single.step(p)
            ref_its.append((float(st.dt), int(st.div_solver.iters), int(st.density_solver.iters)))
        n = len(mass)
        ids = np.concatenate([q[0]["particle_id"] for q in parts])
        assert np.array_equal(np.sort(ids), np.arange(n)), "particles lost or duplicated"
        for q in parts:
            assert q[1] == parts[0][1], "ranks disagree on dt / iteration counts"
        # MP_SOAK (scripts/mp_ipc_soak.sh: hundreds of free-running steps): slabs and single context are two correct evaluations of a chaotic
        # scene and part ways after ~75 steps -- the soak keeps the exact checks (nothing lost, the ranks agree, BIT FOR BIT the loopback group)
        soak = bool(os.environ.get("MP_SOAK"))
        assert soak or all(abs(a[0] - b[0]) <= 1e-5 * b[0] and abs(a[1] - b[1]) <= 1 and abs(a[2] - b[2]) <= 1 for a, b in zip(parts[0][1], ref_its)), (parts[0][1][:20], ref_its[:20])
        for f, tol in (() if soak else (("position", 1e-5), ("velocity", 1e-3), ("density", 1e-4))):
            got = np.zeros_like(single.download(f))
            for q in parts:
                got[q[0]["particle_id"]] = q[0][f]
            ref = single.download(f)
            err = np.abs(got.astype(np.float64) - ref).max() / np.abs(ref).max()
            assert err <= tol, (f, err)
        cnt = np.zeros(n, np.uint32)
        for q in parts:
            cnt[q[0]["particle_id"]] = q[0]["neighbor_count"]
        assert soak or (cnt != single.download("neighbor_count")).mean() < 1e-3
        assert all(q[2]["exchanges"] > 0 and q[2]["bytes_sent"] > 0 for q in parts), [q[2] for q in parts]
        assert min(len(q[0]["particle_id"]) for q in parts) > 0 and len({len(q[0]["particle_id"]) for q in parts}) > 1     # particles migrated
        # ... and BIT FOR BIT the loopback group's result (the same slabs as contexts of one process: the verification form of the
        # decomposition) -- whatever carried the messages, the ranks did the same arithmetic on the same particles in the same order
        from adaptive_sph_amd.distributed import make_loopback_group
        grp = make_loopback_group(lib, pos, mass, vel, planes, world, device_id=local)
        for _ in range(steps):
            ffi.group_step(grp, p)
        for r, (q, c) in enumerate(zip(parts, grp)):
            for f in ("particle_id", "position", "velocity", "density", "neighbor_count"):
                assert np.array_equal(q[0][f], c.download(f)), (r, f)
        for c in grp:
            c.close()
        print(f"MP_CHECK OK world={world} transport={transport} steps={steps}", flush=True)
    dist.barrier()
    ctx.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    try:
        main()
    except BaseException:   # noqa: BLE001  (a rank that dies while the others sit in a collective would hang the launch)
        import traceback
        traceback.print_exc()
        sys.stderr.flush()
        os._exit(1)
