"""SURVEY.md section 8f-4 on the device: the reference's `run ... -p --vtk` (platform/desktop/main_loop.rs:36-82, 105-181; statistics
simulation.rs:137-189, 3279-3359; VtkExporter platform/desktop/vtk_exporter.rs:31-367) driven through the HIP library -- the
per-phase counters are the library's own event times, the snapshots are downloads of device-resident fields."""
import io
from pathlib import Path

import numpy as np
import pytest

from adaptive_sph_amd import adaptivity as A, scene as sc
from adaptive_sph_amd.__main__ import build_parser, run
from adaptive_sph_amd.simulation import init_fluid_sim
from adaptive_sph_amd.simulation_parameters import SimulationParams
from adaptive_sph_amd.vtk_exporter import VtkExporter
from tests.test_host_mirror import _read_legacy_vtk

pytestmark = pytest.mark.gpu
REPO = Path(__file__).resolve().parent.parent
CFG = str(REPO / "tests" / "golden" / "default-config.yaml")
SCN = str(REPO / "tests" / "golden" / "default-scene.yaml")
PATTERNS = str(REPO / "tests" / "golden" / "split-patterns.yaml")

# the ids the reference's counters carry (simulation.rs:1993, 2023-2069, 2517-2545, 2578-2620, 2690-2722, 2733-2795; 1990-1991, 2202)
PCOUNTER_IDS = ["adaptivity", "density-solver", "div-solver", "level-estimation", "neighborhood", "simulation-step"]
VCOUNTER_IDS = ["density-iterations", "div-iterations", "dt", "particle-count"]


def test_run_subcommand_with_statistics_and_vtk_on_the_device(product_lib, tmp_path):
    """BASELINE configs[0] as the reference runs it -- default-config.yaml + default-scene.yaml, merging / sharing / splitting on --
    through `python -m adaptive_sph_amd run ... -p -w FILE --vtk DIR` on the HIP library."""
    stat = tmp_path / "run.stat"
    out = io.StringIO()
    args = build_parser().parse_args(["run", CFG, SCN, "--max-steps", "6", "-p", "-w", str(stat), "--split-patterns", PATTERNS,
                                      "--capacity-factor", "40", "--vtk", str(tmp_path / "vtk"), "--vtk-every", "3"])
    assert run(args, lib=product_lib, out=out) == 6
    text = stat.read_text()
    # write_statistics' layout (simulation.rs:3279-3359): LaTeX row, blank, simulation-time, blank, pcounters sorted, blank, vcounters sorted
    blocks = text.strip("\n").split("\n\n")
    assert len(blocks) == 4 and blocks[0].endswith("& - \\\\") and blocks[1].startswith("simulation-time: ") and blocks[1].endswith("ms")
    assert [l.split(":")[0] for l in blocks[2].splitlines()] == PCOUNTER_IDS
    assert [l.split(":")[0] for l in blocks[3].splitlines()] == VCOUNTER_IDS
    avg = {l.split(":")[0]: float(l.split("avg:")[1].rstrip("ms")) for l in blocks[2].splitlines()}
    assert all(v > 0 for v in avg.values()), avg
    # the phases are parts of the step (the adaptivity time belongs to the step's own sample: end_add_to_last, :2795)
    assert avg["neighborhood"] + avg["div-solver"] + avg["density-solver"] <= avg["simulation-step"]
    assert avg["adaptivity"] < avg["simulation-step"]
    assert sorted(f.name for f in (tmp_path / "vtk").iterdir()) == ["my-sph-00001.vtk", "my-sph-00002.vtk", "my-sph.vtk.series"]
    assert "6 steps" in out.getvalue()


def test_phase_times_and_snapshot_fields_come_from_the_device(product_lib, tmp_path):
    """The per-phase times of sph_step_stats (HIP events of the library's stream) are positive and nest inside the step; a VTK
    snapshot read back with an independent legacy-VTK reader holds exactly what sph_download returns for every exported field."""
    P = SimulationParams.from_yaml(CFG, None)
    scn = sc.SceneConfig.from_yaml(SCN)
    sim = init_fluid_sim(P, scn, counters_enabled=True, lib=product_lib, split_patterns=A.SplitPatterns.load_from_file(PATTERNS), n_capacity=60000)
    planes = sc.boundary_planes(scn.boundary, P.init_boundary_handler)
    for s in range(4):
        sim.single_step_without_adaptivity(P)
        st = sim.last_stats
        assert st.ms_simulation_step > 0
        for name in ("ms_neighborhood", "ms_level_estimation", "ms_div_solver", "ms_density_solver"):
            v = getattr(st, name)
            assert 0 < v <= st.ms_simulation_step, (s, name, v, st.ms_simulation_step)
        assert st.ms_neighborhood + st.ms_div_solver + st.ms_density_solver <= st.ms_simulation_step * 1.001
    assert sorted(sim.pcounters) == [i for i in PCOUNTER_IDS if i != "adaptivity"]
    # (the iteration counters only take steps whose solve iterated, simulation.rs:2542, 2617: `if div_iters > 0`)
    assert {"dt", "particle-count"} <= set(sim.vcounters) <= set(VCOUNTER_IDS)
    sim.single_step_adaptivity(P, float(sim.last_stats.dt))          # share + merge (step 4 is even): the vector is edited on the device
    assert "adaptivity" in sim.pcounters and len(sim.pcounters["simulation-step"].values) == 4     # joined the last sample, no new one
    sim.single_step_without_adaptivity(P)
    with VtkExporter(tmp_path, "snap") as vtk:
        vtk.add_snapshot(sim.time, sim, planes)
    v = _read_legacy_vtk(tmp_path / "snap-00001.vtk")
    n = sim.num_fluid_particles()
    p = sim.particles
    assert v["points"].shape[0] == n == v["n_point_data"] and np.array_equal(v["points"][:, :2], p.position) and not v["points"][:, 2].any()
    assert np.array_equal(v["verts"], np.stack([np.ones(n, np.int32), np.arange(n, dtype=np.int32)], 1))
    arrays = {name: (ty, comps, a) for name, ty, comps, a in v["arrays"]}
    assert list(arrays) == ["density", "density_error", "density_error2", "pressure", "mass", "aii", "h", "ppe_source_term", "distances", "lambda",
                            "velocity", "pressure_accel", "flag_is_fluid_surface", "flag_neighborhood_reduced"]
    for name, field in (("density", "density"), ("density_error", "density_error"), ("pressure", "pressure"), ("mass", "mass"), ("aii", "aii"),
                        ("h", "h2"), ("ppe_source_term", "ppe_source_term"), ("lambda", "lambda_sum")):
        ty, comps, a = arrays[name]
        assert (ty, comps) == ("float", 1) and np.array_equal(a[:, 0], getattr(p, field), equal_nan=True), name
    for name in ("velocity", "pressure_accel"):
        ty, comps, a = arrays[name]
        assert (ty, comps) == ("float", 3) and np.array_equal(a[:, :2], getattr(p, name)) and not a[:, 2].any(), name
    for name in ("flag_is_fluid_surface", "flag_neighborhood_reduced"):
        ty, comps, a = arrays[name]
        assert (ty, comps) == ("unsigned_char", 1) and np.array_equal(a[:, 0], getattr(p, name)), name
    assert arrays["flag_is_fluid_surface"][2].sum() > 0 and np.abs(arrays["pressure_accel"][2]).max() > 0
    # distances: distance_to_boundary of the four planes (boundary_winchenbach2020.rs:320-325), recomputed here in f64
    x = p.position.astype(np.float64)
    d = np.min([x[:, 0] * dx + x[:, 1] * dy + delta for dx, dy, delta in planes], axis=0)
    assert np.abs(arrays["distances"][2][:, 0] - d).max() < 1e-6
    sim.close()
