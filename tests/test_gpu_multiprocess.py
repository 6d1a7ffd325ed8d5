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
