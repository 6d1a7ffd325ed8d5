#!/usr/bin/env python3
"""Run under torch.distributed.run with W processes on ONE GPU (RCCL refuses that; the shared-memory and the peer-mapped transports do
not):   SPH_TRANSPORT=shm|ipc python -m torch.distributed.run --nproc-per-node 2 ... scripts/mp_transport_time.py [workload]
BASELINE configs[1] split into W x-slabs, the driver window (steps 5..24 from rest) and the steps behind it; rank 0 prints one markdown
row: ms per step, Jacobi iterations per step, exchanges per step, and ms per step per iteration."""
import os
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
from adaptive_sph_amd import ffi, scene as sc  # noqa: E402
from adaptive_sph_amd.distributed import make_slab_context, pick_transport  # noqa: E402
from adaptive_sph_amd.workloads import WORKLOADS  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    transport = pick_transport(world)
    dist.init_process_group("nccl" if transport == "rccl" else "gloo", rank=rank, world_size=world)
    wl = sys.argv[1] if len(sys.argv) > 1 else "dam_break_1m"
    scene_f, params_f, _ = WORKLOADS[wl]
    scn, P = scene_f(), params_f()
    pos, mass, vel = sc.init_particles(scn)
    lib = ffi.load_product()
    ctx = make_slab_context(lib, pos, mass, vel, sc.boundary_planes(scn.boundary, P.init_boundary_handler), rank, world, local, transport)
    p = P.to_ffi()

    def window(warm, steps):
        for _ in range(warm):
            ctx.step(p)
        ctx.dist_get_stats(reset=True)
        dist.barrier()
        t0 = time.perf_counter()
        its = []
        for _ in range(steps):
            st = ctx.step(p)
            its.append(int(st.div_solver.iters) + int(st.density_solver.iters) + 2)
        torch.cuda.synchronize()
        dist.barrier()
        el = time.perf_counter() - t0
        w = ctx.dist_get_stats()
        return el / steps * 1e3, float(np.mean(its)), w["exchanges"] / steps, w["host_waits"] / steps

    d = window(5, 20)
    s = window(0, 60)
    if rank == 0:
        print(f"| {wl} on {world} ranks of ONE GPU | {transport} | {d[0]:.3f} | {d[1]:.1f} | {d[2]:.1f} | {d[3]:.1f} | {s[0]:.3f} | {s[1]:.1f} |", flush=True)
    dist.barrier()
    ctx.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    try:
        main()
    except BaseException:  # noqa: BLE001
        import traceback
        traceback.print_exc()
        sys.stderr.flush()
        os._exit(1)
