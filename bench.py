#!/usr/bin/env python3
"""bench.py -- particle-steps/sec of the MI355X-native SPH step (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload dam_break_1m]

A "step" is one `sph_step` (= single_step_without_adaptivity, simulation.rs:1980-2730) over the whole
particle set, inputs resident in HBM.  Every N runs BASELINE.json configs[1] (2D dam-break, 1 048 576
uniform-h particles, HybridDFSPH): for N>1 the driver launches one rank per GPU through
torch.distributed.run and the SAME particle set is split into x-slabs (one per rank, RCCL halo exchange) --
STRONG scaling, the experiment north_star names ("N=1M at 1, 2, 4 and 8 GPUs").  north_star's second target,
>= 6x from 1 to 8 GPUs at N=8M, is measured in the same invocation on configs[3] (8 388 608 particles) and
reported under "strong_8m" at every N (at N=1: the whole scene on one GPU).  --scaling weak keeps last
round's ~1M-particles-per-GPU series as a side experiment.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline      dominant neighbour sweep by time share: algorithmic bytes / HIP-event launch time vs 8 TB/s HBM
                (HIP events of an instrumented pass right after the timed region -- see the comment in main())
  roofline_density  the same for the density kernel (north_star's named target kernel)
  kernels       per-kernel HIP-event breakdown of a profiled pass over the same workload
  cpu_baseline  the CPU oracle ("port" of the reference algorithm, OpenMP on the host cores) timed on a
                bounded sample of the same workload -- a reported baseline, never the measured path.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; 6.29 TB/s measured copy)

# algorithmic HBM bytes per particle per launch (DESIGN.md "Kernels")
ALGO_BYTES = {
    "density": 20, "aii_constfield": 40, "non_pressure_accel": 36, "source_term": 44,
    "aii_nonpressure": 56,   # the two above in one sweep: x, y, m, h, rho, lambda terms, v read once; a_ii, constant_field, v' written
    "pressure_accel": 40, "jacobi_update": 60,
    "solver_tail": 40,       # the integrate map behind the last pressure-acceleration sweep of a solve: x, y, vx, vy, a^p -> x, y, vx, vy
}


# VALU instructions per wave of the sweeps (rocprofv3 --pmc SQ_INSTS_VALU / SQ_WAVES, profiles/r5q_sq_counters.txt; re-collected on round 6's
# tree, profiles/r6p_sq_counters.txt: the same figures -- no sweep of the headline changed)
# and the shader clock those launches ran at (SQ_BUSY_CYCLES / 32 / duration, profiles/r4_sq_counters.txt): what the `valu_issue` object
# of a roofline entry is priced with
VALU_ISSUE = {"density": 1527, "aii_nonpressure": 985, "source_term": 440, "pressure_accel": 306, "jacobi_update": 394}
# Where a wave of the sweep spends its life (profiles/r5q_sq_counters.txt, separate --pmc passes over the driver window):
# SQ_WAIT_ANY (parked on s_waitcnt: its gathers), SQ_WAIT_INST_ANY (ready, waiting for an issue slot), SQ_ACTIVE_INST_ANY (issuing) as
# shares of SQ_WAVE_CYCLES; the mean life of a wave = SQ_WAVE_CYCLES x 4 / waves (16 384 waves on 1 024 SIMDs at <= 8 per SIMD: a sweep is two
# such lives end to end); L1 -> L2 read latency = TCP_TCC_READ_REQ_LATENCY / TCP_TCC_READ_REQ; L2 hit rate = TCC_HIT / TCC_REQ.
WAVE_STATE = {
    "density":         {"parked_on_memory": 0.55, "waiting_to_issue": 0.17, "issuing": 0.28, "wave_life_clocks": 36700, "l1_to_l2_read_latency_clocks": 230, "l2_hit_rate": 0.60},
    "aii_nonpressure": {"parked_on_memory": 0.26, "waiting_to_issue": 0.51, "issuing": 0.23, "wave_life_clocks": 24000, "l1_to_l2_read_latency_clocks": 252, "l2_hit_rate": 0.71},
    "source_term":     {"parked_on_memory": 0.44, "waiting_to_issue": 0.39, "issuing": 0.17, "wave_life_clocks": 14700, "l1_to_l2_read_latency_clocks": 481, "l2_hit_rate": 0.52},
    "pressure_accel":  {"parked_on_memory": 0.43, "waiting_to_issue": 0.44, "issuing": 0.13, "wave_life_clocks": 13700, "l1_to_l2_read_latency_clocks": 426, "l2_hit_rate": 0.56},
    "jacobi_update":   {"parked_on_memory": 0.45, "waiting_to_issue": 0.40, "issuing": 0.15, "wave_life_clocks": 14700, "l1_to_l2_read_latency_clocks": 468, "l2_hit_rate": 0.49},
}
SHADER_CLOCK_HZ = 2.06e9
# rocprofv3's duration of the profiler's calibration kernel (one wave spinning 10 us of the device clock): 10 us + the launch / exit
# of a one-wave dispatch, measured once against a kernel trace (profiles/r4_event_calibration.md: 5.33 / 20.44 / 40.43 / 100.47 for
# 5 / 20 / 40 / 100 us).  Since round 5 the sweeps are timed by their dispatches' own start / end timestamps (Profiler mode 4:
# hipExtLaunchKernelGGL's event pair -- the completion signal's timestamps, what rocprofv3 reads), so this constant only checks that
# mode (the calibration kernel's dispatch must read it) and prices the marker brackets of the kernels that are not sweeps.
SPIN10_ROCPROF_US = 10.44
# the guide's measured streaming rate (MI355X_MICROARCH.md: 6.29 TB/s float4 copy): what `bound` compares a sweep's real traffic with,
# next to this run's own copy kernel
GUIDE_COPY_GBS = 6290.0
SWEEP_KERNELS = ("density", "aii_constfield", "non_pressure_accel", "aii_nonpressure", "source_term", "pressure_accel", "jacobi_update")

# rocprofv3 kernel names of the sweeps (profiles/*_kernel_summary.json keys)
PMC_NAMES = {"density": "OpDensity[build]", "aii_constfield": "OpAiiConst", "non_pressure_accel": "OpNonPressure",
             "source_term": "OpSource", "pressure_accel": "OpPressureAccelU", "jacobi_update": "OpJacobiU"}


def newest_summary(pattern, regex):
    """The newest committed profiles/ summary of a family: by PARSED round number, then suffix ('r10' after 'r9', 'r5q' after 'r5p';
    a lexicographic sort puts r10 in front of r5q -- advisor r5)."""
    import re
    best = None
    for f in (REPO / "profiles").glob(pattern):
        m = re.fullmatch(regex, f.name)
        if m:
            key = (int(m.group(1)), m.group(2) or "")
            if best is None or key > best[0]:
                best = (key, f)
    return best[1] if best else None


def committed_pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the newest committed rocprofv3 --pmc summary (FETCH_SIZE and
    WRITE_SIZE are collected in their own passes, outside bench.py: scripts/summarize_profile.py)."""
    # profiles/<round>_kernel_summary.json = configs[1] (the other configs carry their name: r2a_dam_break_8m_kernel_summary.json)
    f = newest_summary("*_kernel_summary.json", r"r(\d+)([a-z]?)_kernel_summary\.json")
    if f is None or kernel not in PMC_NAMES:
        return None, None
    try:
        e = json.load(open(f)).get(PMC_NAMES[kernel], {})
        return e.get("hbm_traffic_bytes_per_launch"), f.name
    except Exception:  # noqa: BLE001
        return None, None


def committed_8m(kernel):
    """(HBM bytes per launch, rocprofv3 average us, file) of `kernel` on configs[3] (8.4 M particles: the step's arrays no longer sit in the
    256 MB Infinity Cache, so FETCH_SIZE / WRITE_SIZE are HBM traffic there) from the newest committed summary of that config."""
    f = newest_summary("*_dam_break_8m_kernel_summary.json", r"r(\d+)([a-z]?)_dam_break_8m_kernel_summary\.json")
    if f is None or kernel not in PMC_NAMES:
        return None, None, None
    try:
        e = json.load(open(f)).get(PMC_NAMES[kernel], {})
        return e.get("hbm_traffic_bytes_per_launch"), e.get("avg_us_working", e.get("median_us_working")), f.name
    except Exception:  # noqa: BLE001
        return None, None, None


def committed_pmc_avg(kernel):
    """rocprofv3's average duration (us) of `kernel`'s working launches in the newest committed summary of configs[1] (the same
    command under rocprofv3 --kernel-trace --stats): what `avg_us` must agree with."""
    f = newest_summary("*_kernel_summary.json", r"r(\d+)([a-z]?)_kernel_summary\.json")
    if f is None or kernel not in PMC_NAMES:
        return None
    try:
        e = json.load(open(f)).get(PMC_NAMES[kernel], {})
        return e.get("avg_us_working", e.get("median_us_working"))
    except Exception:  # noqa: BLE001
        return None


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--profile-steps", type=int, default=-1, help="0: skip the instrumented pass (it repeats the timed window: warm-up + steps)")
    ap.add_argument("--no-extra", action="store_true", help="skip the short runs of the other single-GPU BASELINE configs")
    ap.add_argument("--no-8m", action="store_true", help="skip the strong-scaling leg on configs[3] (8.4M particles)")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong")
    return ap.parse_args()


def short_run(plib, name, steps=40, warmup=10, math_policy=None, list_forms=False, **param_overrides):
    """A short untimed-warm-up + timed run of another BASELINE config on the same GPU (reported under "other_configs";
    never part of `value`).  `math_policy` = "exact": the same run under sph_set_math_policy(SPH_MATH_EXACT)."""
    from adaptive_sph_amd import ffi, scene as sc
    from adaptive_sph_amd.workloads import WORKLOADS
    scene_f, params_f, desc = WORKLOADS[name]
    scn, P = scene_f(), params_f(**param_overrides)
    pos, mass, vel = sc.init_particles(scn)
    ctx = ffi.Context(plib, len(mass), sc.boundary_planes(scn.boundary, P.init_boundary_handler))
    if math_policy is not None:
        ctx.set_math_policy(math_policy)
    ctx.upload(mass, pos, vel)
    p = P.to_ffi()
    for _ in range(warmup):
        ctx.step(p)
    t0 = time.perf_counter()
    its = []
    for _ in range(steps):
        st = ctx.step(p)
        its.append((int(st.div_solver.iters) + 1 if P.pressure_solver_method in ("HybridDFSPH", "OnlyDivergence") else 0,
                    int(st.density_solver.iters) + 1 if P.pressure_solver_method != "OnlyDivergence" else 0))
    dt = time.perf_counter() - t0
    forms = ctx.profile_list_forms() if list_forms else None
    ctx.close()
    out = {"workload": f"{name}: {desc}", "overrides": param_overrides, "particles": len(mass), "steps": steps, "warmup": warmup,
           "ms_per_step": dt * 1e3 / steps, "particle_steps_per_s": len(mass) * steps / dt,
           "mean_div_iterations": float(np.mean([a for a, _ in its])), "mean_density_iterations": float(np.mean([b for _, b in its]))}
    if math_policy is not None:
        out["math_policy"] = math_policy
    if forms is not None:
        out["list_forms"] = {k: int(v) for k, v in forms.items()}
    return out


def adaptive_steps(plib, name, steps=2, warmup=2, **param_overrides):
    """single_step WITH adaptivity on one context (reported under "other_configs", never part of `value`): the step path, then sharing +
    merging (even steps) or splitting (odd steps) -- the partner decisions on the host (the reference's sequential loops, compiled:
    sph_host_find_partners), the particle data on the device (adaptivity.AdaptivityDriver).  What VERDICT r4 weak 10 asked to see: where
    the adaptive half of a step goes (device -> host of the lists and fields, the host's searches, the apply calls)."""
    from adaptive_sph_amd import ffi, scene as sc
    from adaptive_sph_amd.adaptivity import AdaptivityDriver, SplitPatterns
    from adaptive_sph_amd.workloads import WORKLOADS
    pat = REPO / "tests" / "golden" / "split-patterns.yaml"
    if not pat.exists():
        return None
    scene_f, params_f, desc = WORKLOADS[name]
    scn, P = scene_f(), params_f(**param_overrides)
    pos, mass, vel = sc.init_particles(scn)
    ctx = ffi.Context(plib, 2 * len(mass), sc.boundary_planes(scn.boundary, P.init_boundary_handler))   # (splitting appends children)
    try:
        ctx.upload(mass, pos, vel)
        drv = AdaptivityDriver(ctx, SplitPatterns.load_from_file(pat))
        p = P.to_ffi()
        for _ in range(warmup):
            ctx.step(p)
        ctx.download_neighbors(drv.host)   # one untimed export: the library's device-side CSR buffers exist from here on, as a host's vectors do (steady state of an adaptive run)
        ev = {"shares": 0, "merges": 0, "splits": 0}
        brk, t_step = {}, 0.0
        t0 = time.perf_counter()
        for _ in range(steps):
            t1 = time.perf_counter()
            st = ctx.step(P.to_ffi())
            t_step += time.perf_counter() - t1
            info = drv.single_step_adaptivity(P, float(st.dt), int(st.step_number))
            for k in ev:
                ev[k] += info[k]
            for k, v in info["seconds"].items():
                brk[k] = brk.get(k, 0.0) + v / steps
        dt = time.perf_counter() - t0
        return {"workload": f"{name} WITH adaptivity: {desc}", "overrides": param_overrides, "particles": len(mass), "particles_after": int(ctx.n),
                "steps": steps, "warmup": warmup, "ms_per_step": dt * 1e3 / steps, "step_path_ms_per_step": t_step * 1e3 / steps, "events": ev,
                "adaptivity_breakdown_s_per_step": brk,
                "note": "download = device -> host of the neighbour lists (CSR), of the five fields a decision reads and of the masses the conservation check sums "
                        "(into persistent host buffers since round 6: profiles/r6_export_time.txt); host_decide = the reference's sequential partner searches, compiled, "
                        "on ONE host core; apply = classify + share / merge / split on the device; mass_check = the f64 sums of single_step_adaptivity's assertion (numpy)"}
    finally:
        ctx.close()


def cpu_baseline(scene, params, budget_s: float):
    """The oracle (kind "port") on the host cores, bounded sample of the SAME workload."""
    from adaptive_sph_amd import ffi, scene as sc
    from tests.oracle_harness import load_oracle
    olib = load_oracle()
    pos, mass, vel = sc.init_particles(scene)
    planes = sc.boundary_planes(scene.boundary)
    ctx = ffi.Context(olib, len(mass), planes)
    ctx.upload(mass, pos, vel)
    p = params.to_ffi()
    cores = int(olib.lib.oracle_num_threads())
    ctx.step(p)  # untimed first step (page faults, list allocation)
    t0 = time.perf_counter()
    steps = 0
    while steps < 50:
        ctx.step(p)
        steps += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    n = len(mass)
    ctx.close()
    return {"value": n * steps / dt, "unit": "particle-steps/s", "cores": cores, "kind": "port",
            "sample": f"steps 1..{steps} of the same {n}-particle scene from rest ({dt:.1f} s of CPU time, OpenMP oracle)"}


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # RCCL prints a version banner to stdout when a communicator is created (torch's and the library's): keep stdout for the ONE
    # JSON line -- everything else that writes to fd 1 goes to stderr until the line is printed
    real_stdout = os.dup(1)
    sys.stdout.flush()
    os.dup2(2, 1)

    def emit(line):
        import ctypes
        sys.stdout.flush()
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:  # noqa: BLE001
            pass
        os.write(real_stdout, (line + "\n").encode())

    t_start = time.perf_counter()
    # ---- watchdog: a collective that some rank never enters hangs every other rank inside the library or inside
    # torch.distributed with nothing on stdout.  A thread on every rank watches the time since the last progress mark; past the bar
    # (BENCH_WATCHDOG_S, default 420 s per leg -- the longest leg, the CPU baseline, takes ~30 s) it prints ONE JSON line naming the leg
    # and the rank to the real stdout and ends the process (torch.distributed.run then tears the other ranks down).
    import threading
    wd = {"leg": "start", "t": time.perf_counter(), "off": False}
    wd_limit = float(os.environ.get("BENCH_WATCHDOG_S", "420"))

    def watchdog():
        while not wd["off"]:
            time.sleep(1.0)
            idle = time.perf_counter() - wd["t"]
            if idle > wd_limit and not wd["off"]:
                line = {"metric": "particle-steps/sec (whole node), 2D dam-break N=1M DFSPH; 1/2/4/8 GPUs", "value": None, "unit": "particle-steps/s",
                        "n_gpus": world, "error": f"watchdog: rank {rank} made no progress for {idle:.0f} s in leg '{wd['leg']}' (a collective some rank never entered?)",
                        "leg": wd["leg"], "rank": rank, "seconds_since_start": time.perf_counter() - t_start}
                os.write(real_stdout, (json.dumps(line) + "\n").encode())
                sys.stderr.write(f"[bench] {line['error']}\n")
                sys.stderr.flush()
                os._exit(3)

    threading.Thread(target=watchdog, daemon=True).start()

    def leg(name):
        """progress marks on stderr: which leg of the run a failure (or the driver's clock) belongs to; resets the watchdog"""
        wd["leg"], wd["t"] = name, time.perf_counter()
        if rank == 0:
            print(f"[bench +{time.perf_counter() - t_start:6.1f}s] {name}", file=sys.stderr, flush=True)

    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started by itself (`python bench.py --gpus N`): become the launcher -- one rank per GPU through torch.distributed.run on
        # this node, rendezvous on 127.0.0.1 and a free port; the ranks are this same script with the same flags
        import socket
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        os.dup2(real_stdout, 1)
        os.execv(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
                                  "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:])
    if args.gpus != world:
        raise SystemExit(f"bench.py --gpus {args.gpus} was launched with WORLD_SIZE={world}: one rank per GPU")
    # where the time in front of the timed region goes (21 s there): on a fresh box `import torch` pages in ~2 GB of
    # libraries (10-60 s the first time, 1.5 s later); the library load, the scene (1M particles in numpy) and the upload are < 1 s together
    leg("start-up: import torch (first import on a fresh box pages the libraries in)")
    import torch
    import torch.distributed as dist
    leg("start-up: build check, library load, scene")
    if torch.cuda.is_available():
        local_rank %= max(torch.cuda.device_count(), 1)   # more ranks than devices: only for functional checks on a small box
    from adaptive_sph_amd import build, ffi, scene as sc
    from adaptive_sph_amd.workloads import WORKLOADS

    if rank == 0:
        build.build_hip()
    distributed = world > 1 or bool(os.environ.get('BENCH_FORCE_DIST'))   # the env knob exercises the launcher glue with one rank
    transport = None
    if distributed:
        from adaptive_sph_amd.distributed import pick_transport
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:   # (the forced one-rank distributed flow, started without a launcher)
            import socket
            with socket.socket() as so:
                so.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(so.getsockname()[1])
        torch.cuda.set_device(local_rank)
        # more ranks than devices (a functional run on a small box): RCCL refuses two ranks on one GPU -- the launcher's own process
        # group then runs over gloo and the library over its shared-memory transport; the line says so in config.parallelism
        transport = pick_transport(world)
        if transport == "rccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        dist.barrier()
    plib = ffi.load_product()

    # strong scaling: the metric's workload (configs[1], 1M particles) at every N.  (--scaling weak: ~1M particles per GPU,
    # configs[1] at N=1 ... configs[3] (8M) at N=8 -- a side experiment.)
    weak_wl = {1: "dam_break_1m", 2: "dam_break_2m", 4: "dam_break_4m"}.get(world, "dam_break_8m")
    wl = args.workload or (weak_wl if args.scaling == "weak" else "dam_break_1m")
    scene_f, params_f, desc = WORKLOADS[wl]
    scene, params = scene_f(), params_f()

    def barrier():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
            torch.cuda.synchronize()

    def make_context(scn, P, tname=None):
        pos, mass, vel = sc.init_particles(scn)
        planes = sc.boundary_planes(scn.boundary, P.init_boundary_handler)
        if distributed:
            from adaptive_sph_amd.distributed import make_slab_context
            c = make_slab_context(plib, pos, mass, vel, planes, rank, world, local_rank, tname or transport)
        else:
            c = ffi.Context(plib, len(mass), planes, device_id=local_rank)
            c.upload(mass, pos, vel)
        return c, len(mass)

    def timed_run(c, p, warmup, steps):
        """W untimed warm-up steps, then EXACTLY K steps between barriers; the MAX over ranks of the elapsed time."""
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        t_w0 = time.perf_counter()
        for _ in range(warmup):
            c.step(p)
        warm = time.perf_counter() - t_w0
        c.dist_get_stats(reset=True)
        div_it, dens_it = [], []
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            st = c.step(p)
            div_it.append(int(st.div_solver.iters) + 1)
            dens_it.append(int(st.density_solver.iters) + 1)
        barrier()
        el = time.perf_counter() - t0
        if distributed:
            tt = torch.tensor([el], device="cuda" if transport == "rccl" else "cpu", dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        return el, warm, div_it, dens_it, c.dist_get_stats()

    def comm_record(stats, steps):
        """what this rank put on the wire per step (sph_dist_get_stats): latency-bound messages, so count and size both matter"""
        k = max(steps, 1)
        return {"exchanges_per_step": stats["exchanges"] / k, "allreduces_per_step": stats["allreduces"] / k,
                "halo_bytes_sent_per_step": stats["bytes_sent"] / k, "halo_bytes_received_per_step": stats["bytes_received"] / k,
                "host_waits_per_step": stats["host_waits"] / k, "owned_particles": stats["n_owned"],
                "ghost_particles": stats["n_ghost"], "halo_particles": stats["n_halo"], "rank": rank,
                # what the LIBRARY ran over and how many ranks it sees there (RCCL: ncclCommCount of its communicator)
                "transport": stats.get("transport"), "comm_ranks": stats.get("comm_ranks")}

    ctx, n_total = make_context(scene, params)
    p = params.to_ffi()

    # ---- timed region: warm-up = steps 0..W-1 from rest (SURVEY.md section 8d also wants them reported: the rest lattice;
    # timed on the side, never part of `value`), then exactly K steps, uninstrumented.  (Recording ANY timing-enabled HIP
    # event switches the ROCm queue into per-dispatch profiling for the rest of the process: +~8 us per launch, +0.38 ms per
    # step here -- measured -- so the per-kernel HIP-event timings below come from an instrumented pass that continues the same
    # workload right after the timed region; `value` is never taken from it.)
    leg(f"timed region: {wl}, {args.warmup} + {args.steps} steps")
    elapsed, warmup_elapsed, div_iters, dens_iters, comm_stats = timed_run(ctx, p, args.warmup, args.steps)

    n_local = ctx.n

    # ---- instrumented pass: THE SAME WINDOW again -- a fresh context, the same upload, the same W warm-up steps unprofiled, then the
    # same K steps with HIP events (timing-enabled markers on the library's own stream) around every kernel.  The step is
    # deterministic, so these are the launches of the timed region (same kernels, same iteration counts: recorded below), and a
    # rocprofv3 --kernel-trace --stats of `bench.py --profile-steps 0` with the same flags (profiles/r4*_kernel_summary.*) sees exactly
    # them.  A marker pair reads MORE than the duration rocprofv3 reports for the kernel it brackets (the two marker packets are
    # processed inside the bracket): that excess is measured live, in the same pass -- the profiler launches one kernel of KNOWN
    # duration per step behind the density sweep ("calibration_spin10": one wave spinning 10 us of the device's 100 MHz clock, which
    # rocprofv3 reports as 10.44 us, profiles/r4_event_calibration.md), bracketed like every other kernel -- and subtracted.
    prof_all, prof_work, ev_overhead_us, marker_excess_us = {}, {}, 0.0, 0.0
    prof_window = None
    if args.profile_steps != 0:
        leg("instrumented repeat of the timed window")
        ctx.close()
        ctx, _ = make_context(scene, params)
        k_prof = args.steps if not distributed else min(args.steps, 5)   # (the per-kernel numbers that are judged come from the N = 1 run)
        for _ in range(args.warmup):
            ctx.step(p)
        ctx.profile_reset()
        ctx.profile_enable(4)   # sweeps: the dispatch's own timestamps; everything else: marker brackets
        di2, de2 = [], []
        for _ in range(k_prof):
            st = ctx.step(p)
            di2.append(int(st.div_solver.iters) + 1)
            de2.append(int(st.density_solver.iters) + 1)
        prof_all = ctx.profile_get()
        prof_work = ctx.profile_get_working()     # without the launches that return at once behind a stop decision
        ev_overhead_us = ctx.profile_event_overhead_us()
        ctx.profile_enable(0)
        cal = prof_all.pop("calibration_spin10", None)
        prof_work.pop("calibration_spin10", None)
        cal_d = prof_all.pop("calibration_spin10_dispatch", None)
        prof_work.pop("calibration_spin10_dispatch", None)
        if cal and cal[0]:
            marker_excess_us = max(0.0, cal[1] * 1e3 / cal[0] - SPIN10_ROCPROF_US)
        prof_window = {"steps": k_prof, "first_step": args.warmup, "same_iteration_counts_as_timed_region": (di2 == div_iters[:k_prof] and de2 == dens_iters[:k_prof]),
                       "marker_excess_us": marker_excess_us, "calibration_launches": cal[0] if cal else 0,
                       # the calibration kernel timed like the sweeps are (its dispatch's own timestamps): rocprofv3 reports 10.44 us for it
                       "calibration_spin10_dispatch_us": (cal_d[1] * 1e3 / cal_d[0]) if cal_d and cal_d[0] else None,
                       "calibration_spin10_rocprofv3_us": SPIN10_ROCPROF_US}
        args.profile_steps = k_prof
    leg("copy kernel")
    copy_gbs = ctx.profile_copy_bandwidth_gbs(1 << 30) if rank == 0 else 0.0   # achievable HBM rate of this device, same run
    ctx.close()

    # ---- every transport this launch can run, side by side: the headline and configs[3] again over the peer-mapped
    # push transport (hipIpc inboxes + device flags, no collective-library launch per exchange) when the primary one is RCCL (or the
    # shared-memory transport on a box with fewer GPUs than ranks) -- same scene, same K / W, same protocol; per transport: ms/step, what
    # one exchange costs in the step's queue (HIP-event brackets of a short instrumented run), the library's own view of the group.
    # `value` stays the primary transport's.  A transport that cannot be set up is recorded as such on every rank (make_slab_context
    # agrees on failures through the launcher's process group), it does not end the run.
    EXCHANGE_KERNELS = ("rccl_sendrecv", "rccl_allreduce", "ipc_push", "ipc_wait", "ipc_pack_push", "ipc_wait_unpack", "ghost_pack", "ghost_pack_totals", "ghost_unpack", "slab_refresh")
    transports = None

    def transport_run(tname, scn, P, label):
        rec = {"transport_requested": tname}
        c = None
        try:
            leg(f"{label} over {tname}")
            c, n_ = make_context(scn, P, tname)
            el_, _, di_, de_, st_ = timed_run(c, P.to_ffi(), args.warmup, args.steps)
            rec.update({"ms_per_step": el_ * 1e3 / args.steps, "value": n_ * args.steps / el_, "unit": "particle-steps/s", "particles": n_,
                        "mean_div_iterations": float(np.mean(di_)), "mean_density_iterations": float(np.mean(de_)), "comm_rank0": comm_record(st_, args.steps)})
            leg(f"{label} over {tname}: exchange cost")
            c.profile_reset()
            c.profile_enable(1)
            k_ = 5
            it_ = 0
            for _ in range(k_):
                s_ = c.step(P.to_ffi())
                it_ += int(s_.div_solver.iters) + int(s_.density_solver.iters)
            pa = c.profile_get()
            c.profile_enable(0)
            rec["exchange_rank0"] = {"steps": k_, "jacobi_iterations": it_,
                                     "us_per_launch": {k: pa[k][1] * 1e3 / max(pa[k][0], 1) for k in EXCHANGE_KERNELS if k in pa},
                                     "launches_per_step": {k: pa[k][0] / k_ for k in EXCHANGE_KERNELS if k in pa},
                                     "us_per_jacobi_iteration": sum(pa[k][1] * 1e3 for k in ("rccl_sendrecv", "rccl_allreduce", "ipc_push", "ipc_wait", "ipc_pack_push", "ipc_wait_unpack") if k in pa) / max(it_, 1),
                                     "note": "HIP-event brackets on rank 0's stream (they include the wait for the neighbour's message)"}
        except (ffi.SphError, RuntimeError) as e:
            rec["failed"] = str(e)[:300]
        finally:
            if c is not None:
                c.close()
        return rec

    other_transports = [t for t in (["ipc"] if distributed and not os.environ.get("BENCH_ONE_TRANSPORT") else []) if t != transport]
    if distributed and wl == "dam_break_1m":
        transports = {transport: {"ms_per_step": elapsed * 1e3 / args.steps, "value": n_total * args.steps / elapsed, "unit": "particle-steps/s",
                                  "comm_rank0": comm_record(comm_stats, args.steps), "primary": True}}
        for t in other_transports:
            transports[t] = transport_run(t, scene, params, "headline")

    # ---- north_star's second strong-scaling target: configs[3], 8.4M particles, same K / W, same protocol -------------------
    strong_8m = None
    if not args.no_8m and wl == "dam_break_1m":
        s8, p8f, d8 = WORKLOADS["dam_break_8m"]
        P8 = p8f()

        def run_8m(overlap, name="dam_break_8m"):
            """overlap: None = the library's own rule (sweep A split around the iteration's communication on slabs of >= 786k particles
            with neighbours), "0" / "1" = forced (SPH_OVERLAP, read once at sph_create: every run below creates its own context; same results either way)."""
            if overlap is None:
                os.environ.pop("SPH_OVERLAP", None)
            else:
                os.environ["SPH_OVERLAP"] = overlap
            sN, pNf, dN = WORKLOADS[name]
            PN = pNf()
            c8, n8 = make_context(sN(), PN)
            el8, _, di8, de8, st8 = timed_run(c8, PN.to_ffi(), args.warmup, args.steps)
            c8.close()
            os.environ.pop("SPH_OVERLAP", None)
            return {"workload": f"{name}: {dN}", "particles": n8, "value": n8 * args.steps / el8, "unit": "particle-steps/s",
                    "ms_per_step": el8 * 1e3 / args.steps, "steps": args.steps, "warmup": args.warmup, "n_gpus": world, "scaling": "strong",
                    "mean_div_iterations": float(np.mean(di8)), "mean_density_iterations": float(np.mean(de8)),
                    "comm_rank0": comm_record(st8, args.steps) if distributed else None}

        leg("configs[3]: dam_break_8m")
        strong_8m = run_8m(None)
        if distributed:   # the same run with sweep A forced into one launch / forced split: what the overlap is worth on this node
            strong_8m["sweep_a"] = "library rule (split on slabs of >= 786432 particles)"
            strong_8m["sweep_a_one_launch"] = {k: v for k, v in run_8m("0").items() if k in ("value", "ms_per_step", "comm_rank0")}
            strong_8m["sweep_a_split"] = {k: v for k, v in run_8m("1").items() if k in ("value", "ms_per_step", "comm_rank0")}
            strong_8m["transports"] = {transport: {"ms_per_step": strong_8m["ms_per_step"], "value": strong_8m["value"], "primary": True}}
            for t in other_transports:
                strong_8m["transports"][t] = transport_run(t, s8(), P8, "configs[3]")

    # ---- SURVEY section 8d's configs[3] geometry (2896 x 2896 at spacing 1/2048, box 4 x 2: ONE tall column, 2896-row halos -- the harder
    # strong-scaling case) with max_dt 0.00025, the largest of the three probed values at which it does not blow up
    # (profiles/r5_config3_divergence.md); same K / W, same protocol, at every N
    strong_8m_spec = None
    if not args.no_8m and wl == "dam_break_1m":
        leg("SURVEY's configs[3] geometry: dam_break_8m_spec")
        strong_8m_spec = run_8m(None, "dam_break_8m_spec")
        strong_8m_spec["max_dt"] = 0.00025

    # ---- forced-count legs: the headline's window is chaotic -- its iteration counts change with any rounding
    # and with the rank count -- so a ratio of two `value`s mixes communication with the counts.  Here the counts are FORCED (tolerances
    # 0, max_iters = K: every solve runs exactly K iterations) at two values of K on the same steps from rest; the difference divided by
    # the iterations is what ONE Jacobi iteration costs end to end (both sweeps, and on slabs: pack, exchange, unpack, totals) -- per
    # transport, comparable across N.  Never part of `value`.
    forced_count = None
    if wl == "dam_break_1m":
        FORCED = dict(hybrid_dfsph_max_avg_density_error=0.0, hybrid_dfsph_max_avg_divergence_error=0.0)
        K_LO, K_HI, W_F, S_F = 6, 18, 2, 6

        def forced_leg(name, tname):
            sF, pFf, dF = WORKLOADS[name]
            rec = {}
            for K in (K_LO, K_HI):
                c = None
                try:
                    PF = pFf(max_iters=K, **FORCED)
                    c, nF = make_context(sF(), PF, tname)
                    elF, _, diF, deF, _ = timed_run(c, PF.to_ffi(), W_F, S_F)
                    rec[K] = (elF * 1e3 / S_F, float(np.mean(diF)) - 1 + float(np.mean(deF)) - 1, nF)
                except (ffi.SphError, RuntimeError) as e:
                    return {"failed": str(e)[:300]}
                finally:
                    if c is not None:
                        c.close()
            (t_lo, it_lo, nF), (t_hi, it_hi, _) = rec[K_LO], rec[K_HI]
            return {"particles": nF, "steps": S_F, "warmup": W_F, "ms_per_step": {f"max_iters={K_LO}": t_lo, f"max_iters={K_HI}": t_hi},
                    "jacobi_iterations_per_step": {f"max_iters={K_LO}": it_lo, f"max_iters={K_HI}": it_hi},
                    "us_per_jacobi_iteration": (t_hi - t_lo) * 1e3 / max(it_hi - it_lo, 1e-9),
                    "ms_per_step_without_iterations": t_lo - (t_hi - t_lo) / max(it_hi - it_lo, 1e-9) * it_lo}

        forced_count = {"note": f"tolerances 0, max_iters = {K_LO} / {K_HI}: exact iteration counts; us_per_jacobi_iteration = the difference / the iterations "
                                "(sweeps A + B and, on slabs, one exchange + totals); steps 2..7 from rest; n_gpus as the line's"}
        for name in ["dam_break_1m"] + ([] if args.no_8m else ["dam_break_8m_spec"]):
            forced_count[name] = {}
            for t in ([transport] + other_transports) if distributed else ["single context"]:
                leg(f"forced counts: {name} over {t}")
                forced_count[name][t] = forced_leg(name, t if distributed else None)

    # ---- BASELINE configs[4] on the same ranks: the ratio-stress scene (4 004 343 particles at 50:1 radii, IISPH, Sdf2D box, EmptyAngle
    # level estimation) -- its step path, and a few calls of single_step WITH sharing / merging / splitting, the adaptive half in its
    # slab form (distributed.rank_single_step_adaptivity_on_slabs: the decisions on rank 0's host over the gathered 21 B per particle
    # and the lists, merge_partner / merge_counter broadcast, every rank applies to its own slab; the particles never leave their GPU).
    # The ghost layer of a cut is as wide as the largest smoothing length NEAR it asks for, so the fine block's slices stand on 8 ranks.
    config4 = None
    if distributed and not args.no_8m and wl == "dam_break_1m":
        from adaptive_sph_amd.distributed import rank_single_step_adaptivity_on_slabs
        s4, p4f, d4 = WORKLOADS["ratio_stress_4m"]
        r_fine = float(np.sqrt(np.float32(0.0004385) ** 2 * 0.93 / np.pi))
        P4 = p4f(level_estimation_method="EmptyAngle", merging=True, sharing=True, splitting=True, particle_radius_fine=r_fine,
                 particle_radius_base=50 * r_fine, maximum_surface_distance=0.3)
        config4 = {"workload": f"ratio_stress_4m: {d4}, EmptyAngle level estimation", "n_gpus": world}
        leg("configs[4]: ratio_stress_4m on the slabs")
        c4 = None
        try:
            scn4 = s4()
            c4, n4 = make_context(scn4, P4)
            k4 = max(2, min(args.steps, 10))
            el4, _, _, de4, st4 = timed_run(c4, P4.to_ffi(), 2, k4)
            config4.update({"particles": n4, "steps": k4, "ms_per_step": el4 * 1e3 / k4, "particle_steps_per_s": n4 * k4 / el4,
                            "mean_density_iterations": float(np.mean(de4)), "comm_rank0": comm_record(st4, k4)})
            from adaptive_sph_amd.adaptivity import SplitPatterns
            pat = REPO / "tests" / "golden" / "split-patterns.yaml"
            sp4 = SplitPatterns.load_from_file(pat) if pat.exists() else None
            if sp4 is not None:
                c4.set_split_patterns(sp4.patterns)
                barrier()
                t0 = time.perf_counter()
                ev = {"shares": 0, "merges": 0, "splits": 0}
                brk = {}
                for _ in range(2):
                    st = c4.step(P4.to_ffi())
                    info = rank_single_step_adaptivity_on_slabs(c4, P4, float(st.dt), int(st.step_number))
                    for k in ev:
                        ev[k] += info[k]
                    for k, v in info["seconds"].items():
                        brk[k] = brk.get(k, 0.0) + v / 2
                barrier()
                config4["single_step_with_adaptivity"] = {"steps": 2, "ms_per_step": (time.perf_counter() - t0) * 1e3 / 2, "events": ev,
                                                          "particles_after": info["n_after"], "breakdown_s_per_step_rank0": brk,
                                                          "note": "rank 0's clock; decisions on rank 0's host (download = device -> host of lists and fields, gather_broadcast = the launcher's pickled gather / broadcast, host_decide = the sequential partner searches), apply on the slabs"}
        except (ffi.SphError, RuntimeError) as e:   # a refusal taken on all-reduced values, or a failure of the adaptive step's root that
            config4["refused"] = str(e)[:300]         # rank_single_step_adaptivity re-raises on every rank: every rank is here
        finally:
            if c4 is not None:
                c4.close()

    if rank != 0:
        wd["leg"], wd["t"] = "waiting for rank 0's line", time.perf_counter()
        if distributed:
            dist.barrier()
            dist.destroy_process_group()
        wd["off"] = True
        return

    # The instrumented repeat runs every sweep ~0.6 us SHORTER than the timed region does: measured under ONE rocprofv3 kernel trace of
    # both passes (scripts/kt_two_passes.py, profiles/r5_two_passes.txt: OpJacobiU 17.75 us in the timed pass, 17.16 in the instrumented
    # one; OpPressureAccelU 15.55 / 14.44) -- an instrumented queue leaves a gap behind every launch, an uninstrumented one starts the
    # next dispatch's clock where the last one ended, so there the launch boundary is INSIDE the durations (rocprofv3's durations of the
    # timed pass add up to its wall clock, 99 % busy).  The line therefore adds that boundary back: the timed region's wall clock per
    # step minus the instrumented durations of the step's launches, spread over the launches -- one figure per run, printed -- so that
    # a kernel's `avg_us` is what `rocprofv3 --kernel-trace --stats -- bench.py --profile-steps 0` reports for it (within 2 %), and
    # the durations add up to the step.
    launch_boundary_us = 0.0
    if prof_all and not distributed:
        n_launch = sum(v[0] for v in prof_all.values())
        t_instr = sum((v[1] * 1e3 - (0.0 if k in SWEEP_KERNELS else marker_excess_us * v[0])) for k, v in prof_all.items())
        wall_us = elapsed * 1e6 / args.steps * max(args.profile_steps, 1)
        if n_launch:
            launch_boundary_us = max(0.0, (wall_us - t_instr) / n_launch)

    def roof(name, launches, total_ms):
        """Average duration of the kernel's launches that did work over the instrumented repeat of the timed window.  Sweeps: the
        dispatch's own start / end timestamps (Profiler mode 4) -- the quantity rocprofv3 --kernel-trace --stats reports as that
        kernel's duration (profiles/r5*_kernel_summary.*: `avg_us_working`), no correction.  Other kernels: marker brackets minus
        the marker excess calibrated in the same pass."""
        if not launches or name not in ALGO_BYTES:
            return None
        raw_us = total_ms * 1e3 / launches
        own_ts = name in SWEEP_KERNELS
        # + the launch boundary of the UNINSTRUMENTED queue (see `launch_boundary_us` below): what rocprofv3 sees of the timed region
        avg_s = ((raw_us if own_ts else raw_us - marker_excess_us) + launch_boundary_us) * 1e-6
        achieved = ALGO_BYTES[name] * n_local / avg_s / 1e9
        traffic, src = committed_pmc_traffic(name) if wl == "dam_break_1m" and not distributed else (None, None)
        issue = VALU_ISSUE.get(name)
        traffic_gbs = (traffic / avg_s / 1e9) if traffic else None
        # What bounds the launch.  Decided where the counters mean HBM: on configs[3] (8.4 M particles; at 1 M the step's ~240 MB sit
        # inside the 256 MB Infinity Cache and FETCH_SIZE counts its hits) -- committed traffic per launch / committed rocprofv3
        # duration of the same kernel there, against the streaming rate of this device (this run's copy kernel, and the guide's
        # 6.29 TB/s).  "hbm" from 85 % of it; else the sweep waits on its gathers' latency and on VALU issue, and the line says so.
        t8, us8, src8 = committed_8m(name) if not distributed else (None, None, None)
        gbs8 = (t8 / (us8 * 1e-6) / 1e9) if (t8 and us8) else None
        copy_ref = max(copy_gbs, GUIDE_COPY_GBS) if copy_gbs else GUIDE_COPY_GBS
        hbm_bound = bool(gbs8 and gbs8 >= 0.85 * copy_ref)
        r = {"kernel": name, "bound": "hbm" if (hbm_bound or not issue) else "latency / valu-issue", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
             "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": src, "traffic_GBs": traffic_gbs,
             "bound_evidence": {"config": "dam_break_8m (out of the Infinity Cache)", "traffic_bytes_per_launch": t8, "rocprofv3_avg_us": us8,
                                "traffic_GBs": gbs8, "source": src8, "copy_kernel_GBs": copy_gbs, "guide_copy_GBs": GUIDE_COPY_GBS,
                                "frac_of_streaming_rate": (gbs8 / copy_ref) if gbs8 else None, "rule": "hbm when >= 0.85",
                                "measured_in_this_run": False},   # committed profiles of the shipped tree (dropped for another build, below)
             "avg_us": avg_s * 1e6, "avg_us_instrumented": raw_us if own_ts else raw_us - marker_excess_us, "launch_boundary_us": launch_boundary_us,
             # (advisor r5) the same fraction on the dispatch's own timestamps alone -- MEASURED in this run; `launch_boundary_us` is DERIVED
             # ((wall clock of the timed region - instrumented durations) / launches: host gaps of the timed region would land in it too)
             "frac_dispatch_only": ALGO_BYTES[name] * n_local / ((raw_us if own_ts else raw_us - marker_excess_us) * 1e-6) / 1e9 / HBM_PEAK_GBS,
             "timed_by": ("dispatch timestamps (hipExtLaunchKernelGGL event pair)" if own_ts else "marker bracket - marker excess") + " + launch boundary",
             "marker_excess_us": None if own_ts else marker_excess_us,
             "rocprofv3_avg_us_committed": committed_pmc_avg(name) if wl == "dam_break_1m" and not distributed else None,
             "launches_timed": launches, "window": prof_window,
             "copy_kernel_GBs": copy_gbs, "frac_of_copy_kernel": (achieved / copy_gbs) if copy_gbs else None,
             "algorithmic_bytes_per_particle": ALGO_BYTES[name], "algorithmic_bytes_per_launch": ALGO_BYTES[name] * n_local}
        if r["rocprofv3_avg_us_committed"]:
            r["avg_us_vs_rocprofv3_committed"] = r["avg_us"] / r["rocprofv3_avg_us_committed"]
        if issue:
            # the sweep's VALU instruction issue by the SQ counters (profiles/r4_sq_counters.txt), priced with the measured
            # instruction classes (2 / 4 / 8 clocks per wave64 instruction, ~3 on the sweeps' mix; profiles/r3_valu_issue.md).
            # 16384 waves of 64 lanes per 2^20 particles on 1024 SIMDs; the launch's clocks = avg_us x the shader clock under load.
            waves_per_simd = (n_local / 64.0) / 1024.0
            clocks = avg_s * SHADER_CLOCK_HZ
            r["valu_issue"] = {"valu_instructions_per_wave": issue, "clocks_per_instruction": 3.0, "shader_clock_GHz": SHADER_CLOCK_HZ / 1e9,
                               "frac": issue * waves_per_simd * 3.0 / clocks, "source": "profiles/r6p_sq_counters.txt = r5q_sq_counters.txt (SQ_INSTS_VALU / SQ_WAVES), profiles/r3_valu_issue.md"}
            r["valu_issue"]["measured_in_this_run"] = False
        if name in WAVE_STATE:
            r["wave_state"] = dict(WAVE_STATE[name], measured_in_this_run=False, source="profiles/r5q_sq_counters.txt, re-collected as r6p_sq_counters.txt (SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES; TCP_TCC_READ_REQ_LATENCY / TCP_TCC_READ_REQ; TCC_HIT / TCC_REQ)")
        if os.environ.get("SPH_HIP_LIBRARY"):   # another build than the shipped one: the committed counters are not its counters
            for key in ("bound_evidence", "valu_issue", "wave_state", "traffic", "traffic_source", "traffic_GBs", "rocprofv3_avg_us_committed", "avg_us_vs_rocprofv3_committed"):
                r.pop(key, None)
            r["bound"] = "not decided (SPH_HIP_LIBRARY: the committed counters belong to the shipped build)"
        return r

    total_prof_ms = sum(v[1] for v in prof_all.values()) or 1.0
    kernels = []
    for name, (launches, total_ms) in sorted(prof_all.items(), key=lambda kv: -kv[1][1]):
        own_ts = name in SWEEP_KERNELS
        k = {"name": name, "launches_per_step": launches / max(args.profile_steps, 1),
             "avg_us": total_ms * 1e3 / max(launches, 1) - (0.0 if own_ts else marker_excess_us) + launch_boundary_us,
             "avg_us_instrumented": total_ms * 1e3 / max(launches, 1) - (0.0 if own_ts else marker_excess_us), "timed_by": "dispatch" if own_ts else "bracket - excess",
             "time_share": total_ms / total_prof_ms}
        r = roof(name, *prof_work.get(name, (launches, total_ms)))
        if r:
            k["achieved_GBs"] = r["achieved"]
            k["frac_hbm_peak"] = r["frac"]
        kernels.append(k)
    # the dominant kernel: largest total time of the launches that did WORK (speculative launches behind the stop decision are
    # event overhead, not sweeps)
    dominant = max((n for n in prof_work if n in ALGO_BYTES), key=lambda n: prof_work[n][1], default=None)
    roofline = roof(dominant, *prof_work[dominant]) if dominant else None
    roofline_density = roof("density", *prof_work["density"]) if "density" in prof_work else None
    timing_note = (f"HIP events on the library's stream, over an instrumented REPEAT of the timed window (fresh context, same {args.warmup} warm-up "
                   f"steps, same {args.profile_steps} steps; events perturb dispatch, so the timed region itself is uninstrumented); a sweep's avg_us = mean over "
                   f"its working launches of the dispatch's own start-to-end timestamps (the event pair hipExtLaunchKernelGGL attaches to the launch: "
                   f"the quantity rocprofv3 reports -- plus the launch boundary of the uninstrumented queue, {launch_boundary_us:.2f} us per launch = (the timed region's wall "
                   f"clock - the instrumented durations) / launches, so that the durations add up to the step as rocprofv3's do; the 10 us calibration kernel reads "
                   f"{(prof_window or {}).get('calibration_spin10_dispatch_us') or float('nan'):.2f} us that way, rocprofv3 {SPIN10_ROCPROF_US}); dominant kernel = "
                   f"largest total over that window; `traffic` is not measured in this run: it is the FETCH_SIZE x2 + WRITE_SIZE figure of the rocprofv3 "
                   f"--pmc passes summarised in `traffic_source`; `bound` is decided on configs[3]'s committed traffic and duration (out of the Infinity "
                   f"Cache) against the streaming rate")
    for r in (roofline, roofline_density):
        if r:
            r["timing"] = timing_note
    if roofline_density:
        # the 20 B charge is the harshest reading of a launch that fuses a3's predicate, a5's boundary lambda and a7's
        # density (SURVEY 8d's whole-step ledger: density 20 + boundary-lambda 24 B); both figures, and the roof the sweep actually runs
        # against -- VALU issue (0.40 of the HBM peak on 20 B is beyond it for any form that evaluates the predicate)
        fused_b = 44
        roofline_density["fused_ledger"] = {"algorithmic_bytes_per_particle": fused_b, "what": "a3 neighbour predicate + a5 boundary lambda (24 B) + a7 density (20 B): what this ONE launch does",
                                            "achieved": roofline_density["achieved"] * fused_b / 20.0, "frac": roofline_density["frac"] * fused_b / 20.0}
        vi = roofline_density.get("valu_issue") or {}
        roofline_density["operative_roof"] = {"roof": "valu-issue", "frac": vi.get("frac"),
                                              "note": "VALU instructions per wave x 3 clocks / the launch's shader clocks (committed SQ counters); the north_star target "
                                                      "(0.40 of 8 TB/s on 20 B = 6.6 us) allows ~280 VALU instructions per wave, the exact predicate over 42 candidates alone is ~340"}

    out = {
        "metric": "particle-steps/sec (whole node), 2D dam-break N=1M DFSPH; 1/2/4/8 GPUs",
        "value": n_total * args.steps / elapsed,
        "unit": "particle-steps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed * 1e3 / args.steps,
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"{wl}: {desc}", "particles": n_total, "solver": params.pressure_solver_method,
                   "mean_div_iterations": float(np.mean(div_iters)), "mean_density_iterations": float(np.mean(dens_iters)),
                   "parallelism": (f"x-slabs x{world}" + ("" if transport == "rccl" else f", {transport} transport (more ranks than GPUs: functional run)")) if distributed else "single GPU"},
        "roofline": roofline,
        "roofline_density": roofline_density,
        "kernels": kernels,
        "strong_8m": strong_8m,
        "strong_8m_spec": strong_8m_spec,
        "transports": transports,
        "forced_count": forced_count,
        "config4_ratio_stress_4m": config4,
        "comm_rank0": comm_record(comm_stats, args.steps) if distributed else {"host_waits_per_step": comm_stats["host_waits"] / max(args.steps, 1)},
    }
    if distributed:   # time inside RCCL per step, from the instrumented pass (HIP events around the grouped send/recv and the all-reduces)
        out["comm_rank0"]["rccl_us_per_step_instrumented"] = {
            k: prof_all[k][1] * 1e3 / max(args.profile_steps, 1) for k in ("rccl_sendrecv", "rccl_allreduce") if k in prof_all}
    # whole-step algorithmic bytes B_step = 388 + 100 (n_div + n_dens) B per particle-step with the
    # iteration counts of the timed steps (sweeps only; the neighbour build is listed apart), and the steps from rest
    b_step = 388.0 + 100.0 * (float(np.mean(div_iters)) + float(np.mean(dens_iters)))
    step_gbs = b_step * n_total * args.steps / elapsed / 1e9
    # iteration 0 of each solve is evaluated in closed form inside the source-term sweep (p = 0 => a^p = 0, Ap = 0): its two
    # sweeps never run, so the figure WITHOUT their 100 B each is what the launched kernels were asked to move
    n_solves = 2 if params.pressure_solver_method == "HybridDFSPH" else 1
    step_gbs_launched = (b_step - 100.0 * n_solves) * n_total * args.steps / elapsed / 1e9
    out["whole_step"] = {
        "algorithmic_bytes_per_particle_step": b_step,
        "achieved_GBs": step_gbs, "frac_hbm_peak": step_gbs / (HBM_PEAK_GBS * world),
        "without_closed_form_iteration_0": {"algorithmic_bytes_per_particle_step": b_step - 100.0 * n_solves, "achieved_GBs": step_gbs_launched,
                                            "frac_hbm_peak": step_gbs_launched / (HBM_PEAK_GBS * world)},
        # the build queued ahead as a merge (sph_sort.hip: incremental_cell_sort_reorder): classification in the integrating tail
        # (old cell 4r, new key 4w, mover flag 1w), place + reorder (key 4r, flag 1r, old cell 4r; record, velocity, id, level state
        # 40r + 40w; sorted key 4w, new cell 4w); the per-cell count and scan touch the cell tables, not the particles.
        # (the radix form it replaces: cell keys 16r + 8w, two passes of 4r + 8r + 8w, reorder 40r + 40w, cell-range table 4r = 148)
        # (slab ranks merge at the step's start, with a classification pass of its own and a stored permutation: record 16r, old cell 4r,
        #  key 4w, flag 1w; place: key 4r, flag 1r, old cell 4r, sorted key 4w, permutation 4w; gather reorder 4r + 40r + 40w + new cell 4w)
        "neighbour_build_algorithmic_bytes_per_particle": (25 + 17 + 88) if distributed else (9 + 9 + 80 + 8),
        "steps_from_rest": {"steps": args.warmup, "ms_per_step": warmup_elapsed * 1e3 / max(args.warmup, 1),
                            "note": "steps 0..warmup-1 (rest lattice, first launches included); rank 0's clock, not part of `value`"},
    }
    if not args.no_extra and not distributed and wl == "dam_break_1m":
        out["other_configs"] = []
        leg("other configs: dam_break_1m_adaptive")
        out["other_configs"].append(short_run(plib, "dam_break_1m_adaptive"))                                    # configs[2]: 4:1 radius ratio
        leg("other configs: dam_break_1m_adaptive_contact")
        out["other_configs"].append(short_run(plib, "dam_break_1m_adaptive_colliding", steps=40, warmup=20, list_forms=True))   # ... with the two resolutions in contact
        out["other_configs"][-1]["state"] = ("configs[2]'s blocks 1.5 coarse spacings apart: the fine column's collapse drives it into the coarse block from step ~13 on, "
                                             "steps 20-59 run the symmetric (h_i + h_j) / 2 rule on the interface (`list_forms`: particles on explicit index lists / "
                                             "candidate walks; profiles/r6_config2_contact.md).  One coarse spacing apart -- mixed-h pairs from step 0, the parity test "
                                             "tests/test_gpu_configs.py::test_config2_columns_in_contact_at_full_size -- the recipe blows up at step 5 (density solve at "
                                             "max_iters, 1e26 m/s) on device and oracle alike, so it is no bench window; the leg above is BASELINE's placement, 2.0 apart: "
                                             "two uniform columns on the fine sorting grid")
        leg("other configs: headline under the EXACT math policy")
        ex = short_run(plib, "dam_break_1m", steps=args.steps, warmup=args.warmup, math_policy="exact")          # what "identical results" costs: the headline window, same K / W
        ex["state"] = ("the headline's window under sph_set_math_policy(SPH_MATH_EXACT): IEEE division / sqrt, no fma, the reference's operation order, "
                       "mask-word replays (no record sweeps, no offset lists, no build queued ahead) -- bit for bit the oracle in the device's visiting "
                       "order (tests/test_gpu_bitexact.py); the free-running iteration counts are those of its own arithmetic")
        out["other_configs"].append(ex)
        leg("other configs: ratio_stress_4m")
        out["other_configs"].append(short_run(plib, "ratio_stress_4m", steps=20, warmup=5))                      # configs[4]'s scene (50:1, 4M), no adaptivity
        out["other_configs"][-1]["state"] = "FREE FALL (the reference scene hangs both blocks 0.5 above the floor): one Jacobi iteration per step, not an IISPH number"
        leg("other configs: ratio_stress_4m_settled")
        out["other_configs"].append(short_run(plib, "ratio_stress_4m_settled", steps=2, warmup=6))               # ... standing on the floor: the two steps in which IISPH iterates
        out["other_configs"][-1]["state"] = ("blocks on the floor: steps 6-7, the solve iterating on positive pressures (8 and 43 iterations) -- the only such window "
                                             "this geometry has at 4M: step 9 ends in SPH_ERR_AP_NOT_FINITE on the device and the oracle alike (profiles/r5_config4_settled.md)")
        leg("other configs: dam_break_1m + EmptyAngle")
        out["other_configs"].append(short_run(plib, "dam_break_1m", steps=20, warmup=20, level_estimation_method="EmptyAngle",
                                              maximum_surface_distance=0.2, particle_radius_fine=0.0005, particle_radius_base=0.002))   # + level estimation
        leg("other configs: ratio_stress_4m with adaptivity")
        r_fine = float(np.sqrt(np.float32(0.0004385) ** 2 * 0.93 / np.pi))
        try:   # configs[4] as BASELINE.json words it: the 50:1 scene WITH its adaptivity (2 steps: one merging, one splitting pass)
            ad = adaptive_steps(plib, "ratio_stress_4m", steps=2, warmup=2, level_estimation_method="EmptyAngle", merging=True, sharing=True, splitting=True,
                                particle_radius_fine=r_fine, particle_radius_base=50 * r_fine, maximum_surface_distance=0.3)
            if ad is not None:
                out["other_configs"].append(ad)
        except Exception as e:   # (a refusal of the adaptive step is a result, not a reason to lose the line)
            out["other_configs"].append({"workload": "ratio_stress_4m WITH adaptivity", "refused": str(e)[:300]})
    if not args.no_cpu_baseline and not distributed:
        leg("cpu baseline (oracle)")
        out["cpu_baseline"] = cpu_baseline(scene, params, args.cpu_seconds)
    wd["off"] = True
    emit(json.dumps(out))
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    try:
        main()
    except BaseException:  # noqa: BLE001
        # a rank that dies while the others sit in a collective would hang the whole launch: report and leave at once
        # (torch.distributed.run then tears the other ranks down)
        import traceback
        traceback.print_exc()
        sys.stderr.flush()
        os._exit(1)
