"""The distributed flow with REAL processes: one rank per process under torch.distributed.run, each calling sph_step by itself.
On a box with fewer GPUs than ranks the ranks share a device, which RCCL refuses; the launcher glue (distributed.pick_transport)
then takes the library's shared-memory transport (sph_comm_init_shm) -- the same per-rank driver code, the same collectives,
staged through the host.  On a multi-GPU box the same tests run over RCCL."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
REPO = Path(__file__).resolve().parent.parent


def _env():
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["MASTER_ADDR"] = "127.0.0.1"
    return env


@pytest.mark.parametrize("world", [2, 3])
def test_ranks_as_processes_reproduce_the_single_context(world):
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
                        "--master-port", str(29650 + world), str(REPO / "tests" / "mp_slab_check.py")],
                       capture_output=True, text=True, timeout=600, env=_env(), cwd=str(REPO))
    assert r.returncode == 0 and f"MP_CHECK OK world={world}" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.parametrize("world", [2, 3])
def test_peer_mapped_push_transport_between_processes(world):
    """The same check with SPH_TRANSPORT=ipc: every rank exports a device inbox (hipIpcGetMemHandle), maps the others'
    (hipIpcOpenMemHandle) and from then on ghost values, migrant / ghost records and the Jacobi totals are PUSHED device to device by
    the sender's kernel and awaited by the receiver's (sph_comm_ipc_export / sph_comm_init_ipc) -- between processes that here share
    one GPU, over xGMI on a multi-GPU node.  Same results as the single context, like the shared-memory transport's run."""
    env = _env()
    env["SPH_TRANSPORT"] = "ipc"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
                        "--master-port", str(29670 + world), str(REPO / "tests" / "mp_slab_check.py")],
                       capture_output=True, text=True, timeout=600, env=env, cwd=str(REPO))
    assert r.returncode == 0 and f"MP_CHECK OK world={world}" in r.stdout and "transport=ipc" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.parametrize("on_slabs", [0, 1])
def test_adaptive_steps_with_one_process_per_rank(on_slabs):
    """single_step = sph_step + single_step_adaptivity on a slab decomposition whose ranks are PROCESSES, against the single context:
    on_slabs = 1 the slab form of the apply (distributed.rank_single_step_adaptivity_on_slabs: decisions on the root, merge_partner /
    merge_counter broadcast, every rank's sph_share / merge / split_particles collective through its own transport); 0 the gather /
    decide / apply / scatter of round 3 (rank_single_step_adaptivity)."""
    env = _env()
    env["MP_ADAPTIVE_ON_SLABS"] = str(on_slabs)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", str(29661 + on_slabs), str(REPO / "tests" / "mp_adaptive_check.py")],
                       capture_output=True, text=True, timeout=600, env=env, cwd=str(REPO))
    assert r.returncode == 0 and f"MP_ADAPTIVE OK world=2" in r.stdout and f"on_slabs={on_slabs}" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_bench_starts_its_own_ranks_and_prints_the_schema():
    """`python bench.py --gpus 2` without a launcher: bench.py re-executes itself under torch.distributed.run, the ranks share the box's
    GPU(s), rank 0 prints the one JSON line with the driver's schema; the forced one-rank distributed flow prints it too."""
    for cmd, extra_env in ((["--gpus", "2"], {}), (["--gpus", "1"], {"BENCH_FORCE_DIST": "1", "SPH_FORCE_SLAB_MODE": "1"})):
        env = _env()
        env.update(extra_env)
        r = subprocess.run([sys.executable, str(REPO / "bench.py")] + cmd + ["--steps", "4", "--warmup", "2", "--workload", "dam_break_64k", "--no-8m",
                                                                           "--profile-steps", "2", "--no-cpu-baseline"],
                           capture_output=True, text=True, timeout=900, env=env, cwd=str(REPO))
        assert r.returncode == 0, r.stderr[-4000:]
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, r.stdout[-2000:]
        d = json.loads(lines[0])
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
            assert k in d, k
        assert d["n_gpus"] == int(cmd[1]) and d["value"] > 0 and d["steps"] == 4 and d["comm_rank0"]["exchanges_per_step"] >= 0
        if cmd[1] == "2":
            assert d["comm_rank0"]["exchanges_per_step"] > 0 and d["comm_rank0"]["halo_bytes_sent_per_step"] > 0


def test_a_particle_thrown_out_of_the_predicted_grid_does_not_fault_the_build_queued_ahead():
    """Round 5 (found on configs[4]'s blocks set 50 x too close to the floor, scripts/gpu_fault_bisect.sh): an exploding IISPH solve threw
    coarse particles far out; the step that saw it returned its error -- but the NEXT step's cell sort had already been queued behind
    the tail on a PREDICTED grid (this step's bounding box + 3 cells), and in a multi-resolution scene its tile bounds (k_tile_hmax)
    indexed a tile beyond the table: a memory fault AFTER the error had been reported.  Deterministic trigger, in a process of its own
    (a fault kills the process): two particle sizes, cfl_factor so large that dt = max_dt, and one particle with 2e5 m/s -- it lands
    200 m outside the box in ONE step.  The build queued ahead must run harmlessly (clamped) and must not be adopted: the following steps
    are bit for bit those of SPH_AHEAD_BUILD=0 (or refused alike), and a fresh context steps normally afterwards.  (The library before
    the fix dies here with "Memory access fault by GPU node".)"""
    code = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np
from adaptive_sph_amd import ffi, scene as sc
from adaptive_sph_amd.workloads import dam_break_params
lib = ffi.load_product()
scn = sc.SceneConfig(sc.SceneBoundary("box", 4.0, 2.0),
                     [sc.SceneFluidBlock([-1.99, -0.99], [0.5, 0.4], 1.0 / 64, 0.93, [0.0, 0.0]),
                      sc.SceneFluidBlock([-1.48, -0.985], [0.5, 0.5], 1.0 / 16, 0.93, [0.0, 0.0])])
pos, mass, vel = sc.init_particles(scn)
top = int(np.argmax(pos[:, 1] + 1e-3 * pos[:, 0]))
vel2 = vel.copy()
vel2[top] = (3.0e4, 2.0e5)
P = dam_break_params(cfl_factor=1.0e9, max_dt=0.001, hybrid_dfsph_max_avg_density_error=0.0, hybrid_dfsph_max_avg_divergence_error=0.0, max_iters=3)
planes = sc.boundary_planes(scn.boundary)
out = {}
for ahead in ("1", "0"):
    os.environ["SPH_AHEAD_BUILD"] = ahead
    g = ffi.Context(lib, len(mass), planes)
    g.upload(mass, pos, vel2)
    st = g.step(P.to_ffi())
    assert st.dt == np.float32(0.001), st.dt
    assert g.download("position")[top, 1] > 100.0
    status = 0
    try:                                           # (a scene this wide may be refused -- "cell grid too large" -- or stepped: the same either way)
        for s in range(3):
            g.step(P.to_ffi())
    except ffi.SphError as e:
        status = e.status
    out[ahead] = (status, None if status else {f: g.download(f) for f in ("position", "velocity", "density", "neighbor_count")})
    g.close()
    h = ffi.Context(lib, len(mass), planes)     # (drains whatever the last step left queued)
    h.upload(mass, pos, vel)
    h.step(dam_break_params().to_ffi())
    h.close()
assert out["1"][0] == out["0"][0], (out["1"][0], out["0"][0])
if out["1"][1] is not None:
    for f in out["1"][1]:
        assert np.array_equal(out["1"][1][f], out["0"][1][f]), f
print("CLEAN", flush=True)
''' % str(REPO)
    from adaptive_sph_amd import build
    build.build_lab()   # (SPH_AHEAD_BUILD is a laboratory switch: both runs on libsph_lab.so, whose default IS the product's path)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=dict(os.environ, SPH_HIP_LIBRARY="libsph_lab.so"))
    assert r.returncode == 0 and "CLEAN" in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-2500:])
