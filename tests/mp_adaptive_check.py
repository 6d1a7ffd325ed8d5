"""Run under torch.distributed.run with 2 processes (tests/test_gpu_multiprocess.py): BASELINE configs[0] -- default-config.yaml +
default-scene.yaml, merging / sharing / splitting on -- as a slab decomposition with ONE PROCESS PER RANK: sph_step on every
rank, then distributed.rank_single_step_adaptivity (gather to rank 0 through the launcher, decisions + device-side apply there,
scatter back).  Rank 0 also runs the single-context adaptive simulation and compares like tests/test_gpu_adaptivity.py compares
the in-process slab group: the first adaptive step (inputs equal to 1e-7) takes the same splits, the counts stay within 2 %."""
import os
import sys
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from adaptive_sph_amd import adaptivity as A, distributed as D, ffi, scene as sc  # noqa: E402
from adaptive_sph_amd.simulation import init_fluid_sim  # noqa: E402
from adaptive_sph_amd.workloads import default_params  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    transport = D.pick_transport(world)
    dist.init_process_group("nccl" if transport == "rccl" else "gloo", rank=rank, world_size=world)
    lib = ffi.load_product()
    P = default_params()
    scn = sc.SceneConfig.from_yaml(str(REPO / "tests" / "golden" / "default-scene.yaml"))
    sp = A.SplitPatterns.load_from_file(REPO / "tests" / "golden" / "split-patterns.yaml")
    pos, mass, vel = sc.init_particles(scn)
    planes = sc.boundary_planes(scn.boundary, P.init_boundary_handler)
    ctx = D.make_slab_context(lib, pos, mass, vel, planes, rank, world, local, transport)
    on_slabs = os.environ.get("MP_ADAPTIVE_ON_SLABS") == "1"   # the slab form of the apply (the particles stay on their ranks) instead of the gather to rank 0
    ctx.set_split_patterns(sp.patterns)
    gather = D.GatherContext(lib, planes, local, sp) if rank == 0 and not on_slabs else None
    single = init_fluid_sim(P, scn, lib=lib, split_patterns=sp, n_capacity=120000, device_id=local) if rank == 0 else None
    p = P.to_ffi()
    m0 = float(mass.sum(dtype=np.float64))
    counts, events = [[], []], {"shares": 0, "merges": 0, "splits": 0}
    for s in range(12):
        st = ctx.step(p)
        if on_slabs:
            info = D.rank_single_step_adaptivity_on_slabs(ctx, P, float(st.dt), int(st.step_number))
        else:
            info = D.rank_single_step_adaptivity(ctx, gather, P, float(st.dt), int(st.step_number), capacity=120000)
            ctx.set_split_patterns(sp.patterns)
        n_mine = torch.tensor([ctx.n], dtype=torch.int64)
        dist.all_reduce(n_mine.cuda() if transport == "rccl" else n_mine)
        assert info["n_after"] > 0
        if rank == 0:
            single.single_step(P)
            counts[0].append(single.num_fluid_particles())
            counts[1].append(info["n_after"])
            for k in events:
                events[k] += info[k]
    mine = {f: ctx.download(f) for f in ("particle_id", "mass", "position")}
    parts = [None] * world
    dist.all_gather_object(parts, mine)
    if rank == 0:
        n = counts[1][-1]
        ids = np.concatenate([q["particle_id"] for q in parts])
        assert np.array_equal(np.sort(ids), np.arange(n)), "ids are not the global indices again"
        assert counts[0][0] == counts[1][0] > 1035, counts                      # step 1 splits, the same particles on both sides
        assert max(abs(a - b) for a, b in zip(*counts)) <= 0.02 * max(counts[0]), counts
        assert events["splits"] > 0 and events["merges"] + events["shares"] > 0, events
        m1 = float(sum(q["mass"].sum(dtype=np.float64) for q in parts))
        assert abs(m1 - m0) < 0.005 * 12
        x = np.concatenate([q["position"] for q in parts])
        assert np.isfinite(x).all() and np.abs(x).max() < 1.05
        print(f"MP_ADAPTIVE OK world={world} transport={transport} on_slabs={int(on_slabs)} counts={counts[1]}", flush=True)
        if gather is not None:
            gather.close()
        single.close()
    dist.barrier()
    ctx.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    try:
        main()
    except BaseException:   # noqa: BLE001
        import traceback
        traceback.print_exc()
        sys.stderr.flush()
        os._exit(1)
