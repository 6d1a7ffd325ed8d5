"""`python -m adaptive_sph_amd run SIMULATION_CONFIG SCENE_CONFIG [...]` -- the reference's `run` subcommand without its
window (platform/desktop/main_loop.rs:36-82, 105-181, 346-350), on the HIP library.

Same arguments, same YAML formats, same override rule (`-c FILE`: every key of FILE must already exist in the
simulation config, main_loop.rs:113-126), same statistics text (`-p`, `-w PATH`; simulation.rs:3279-3359).  A config that
enables merging / sharing / splitting runs single_step (the step + single_step_adaptivity, simulation.rs:1973-1978): the
partner decisions on the host (adaptivity.py), the particle data on the device; the split patterns come from
`./split-patterns.yaml` like in the reference (main_loop.rs:200-203) or from `--split-patterns`.
`--without-adaptivity` steps with single_step_without_adaptivity only.
"""
from __future__ import annotations

import argparse
import sys
import time
from typing import Optional, Sequence

from . import ffi
from .scene import SceneConfig
from .simulation import init_fluid_sim, init_simulation_params
from .simulation_parameters import SimulationParams, load_yaml_mapping


def build_parser() -> argparse.ArgumentParser:
    ap = argparse.ArgumentParser(prog="adaptive_sph_amd", description="2-D adaptive SPH step on MI355X (kaegi/adaptive-sph hot path)")
    sub = ap.add_subparsers(dest="command", required=True)
    run = sub.add_parser("run", help="Run simulation with given config")
    run.add_argument("SIMULATION_CONFIG", help="Sets the simulation paramaters")
    run.add_argument("SCENE_CONFIG", help="Scene setup")
    run.add_argument("-s", "--max-seconds", type=float, default=None, help="Stop simulation after the given amount of seconds")
    run.add_argument("-c", "--overwrite-config-file", default=None, help="Overwrite config")
    run.add_argument("-p", "--statistics-enabled", action="store_true", help="Track performance of individual steps")
    run.add_argument("-w", "--statistics-path", default=None, help="Where to write statistics to")
    # not in the reference: the window's close button has no headless equivalent
    run.add_argument("--max-steps", type=int, default=None, help="Stop after this many steps (headless replacement of closing the window)")
    run.add_argument("--without-adaptivity", action="store_true",
                     help="step with single_step_without_adaptivity even if the config enables merging/sharing/splitting")
    run.add_argument("--split-patterns", default=None,
                     help="SplitPatterns file (load_split_patterns_from_file); default ./split-patterns.yaml, the path the reference reads (main_loop.rs:200-203)")
    run.add_argument("--capacity-factor", type=float, default=4.0, help="device capacity = this x the initial particle count (splitting adds particles)")
    run.add_argument("--device", type=int, default=0)
    # the reference's VtkExporter is compiled in but switched off (main_loop.rs:253 `export_vtk_data = false`); same writer here
    run.add_argument("--vtk", default=None, metavar="FOLDER", help="write FOLDER/my-sph-NNNNN.vtk + my-sph.vtk.series (one snapshot per step)")
    run.add_argument("--vtk-every", type=int, default=1, help="snapshot every N-th step")
    return ap


def run(args, lib: Optional[ffi.SphLibrary] = None, out=sys.stdout) -> int:
    overrides = load_yaml_mapping(args.overwrite_config_file) if args.overwrite_config_file else None
    params = SimulationParams.from_yaml(args.SIMULATION_CONFIG, overrides)
    print(params, file=out)
    scene = SceneConfig.from_yaml(args.SCENE_CONFIG)
    print(scene, file=out)
    if args.max_seconds is None and args.max_steps is None:
        raise SystemExit("headless run: give --max-seconds and/or --max-steps (there is no window to close)")
    adaptive = (params.merging or params.sharing or params.splitting) and not args.without_adaptivity
    params = init_simulation_params(params, scene)
    counters = bool(args.statistics_enabled or args.statistics_path)
    split_patterns, capacity = None, None
    if adaptive:
        from .adaptivity import SplitPatterns
        from .scene import init_particles
        if params.splitting:
            # main_loop.rs:200-203 reads ./split-patterns.yaml and panics without it; so does this
            from pathlib import Path
            path = Path(args.split_patterns if args.split_patterns is not None else "./split-patterns.yaml")
            split_patterns = SplitPatterns.load_from_file(path)
        capacity = int(len(init_particles(scene)[1]) * args.capacity_factor) + 1024
    sim = init_fluid_sim(params, scene, counters_enabled=counters, lib=lib, device_id=args.device, split_patterns=split_patterns,
                         n_capacity=capacity)
    p = params.to_ffi()
    vtk = None
    if getattr(args, "vtk", None):
        from .scene import boundary_planes
        from .vtk_exporter import VtkExporter
        vtk = VtkExporter(args.vtk, "my-sph")          # main_loop.rs:256
        vtk_planes = boundary_planes(scene.boundary, params.init_boundary_handler)
    steps, t0 = 0, time.perf_counter()
    while (args.max_seconds is None or sim.time < args.max_seconds) and (args.max_steps is None or steps < args.max_steps):
        if adaptive:
            sim.single_step(params)
        else:
            sim.single_step_without_adaptivity(p)
        steps += 1
        if vtk is not None and steps % max(args.vtk_every, 1) == 0:
            vtk.add_snapshot(sim.time, sim, vtk_planes)       # main_loop.rs:302-309
    if vtk is not None:
        vtk.close()
    wall = time.perf_counter() - t0
    print(f"{steps} steps, simulated time {sim.time:.6f} s, {sim.num_fluid_particles()} particles, "
          f"{sim.num_fluid_particles() * steps / max(wall, 1e-9) / 1e6:.2f} M particle-steps/s", file=out)
    if counters:
        text = sim.write_statistics()
        if args.statistics_path:
            with open(args.statistics_path, "w") as fh:
                fh.write(text)
        else:
            print(text, file=out)
    sim.close()
    return steps


def main(argv: Optional[Sequence[str]] = None) -> int:
    args = build_parser().parse_args(argv)
    if args.command == "run":
        run(args)
    return 0


if __name__ == "__main__":
    sys.exit(main())
