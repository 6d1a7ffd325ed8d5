"""Host-side mirror of the reference's config/scene interface (no GPU)."""
import numpy as np
import pytest
import yaml

from adaptive_sph_amd import scene as sc
from adaptive_sph_amd.simulation_parameters import SimulationParams, apply_overrides, load_yaml_mapping
from adaptive_sph_amd.workloads import DEFAULT_CONFIG, dam_break_params, default_params
from tests.oracle_harness import REPO

CFG = str(REPO / "tests" / "golden" / "default-config.yaml")


def test_default_config_yaml_roundtrip():
    p = SimulationParams.from_yaml(CFG)
    assert p.pressure_solver_method == "HybridDFSPH" and p.level_estimation_method == "EmptyAngle"
    assert p.pull_fluid_to is None and p.fill_stash_with is None
    assert p == default_params()
    assert load_yaml_mapping(CFG) == {k: v for k, v in DEFAULT_CONFIG.items()}


def test_override_must_hit_existing_key():
    m = load_yaml_mapping(CFG)
    apply_overrides(m, {"max_dt": 0.002})
    assert m["max_dt"] == 0.002
    with pytest.raises(KeyError, match="not able to find attribute"):
        apply_overrides(m, {"no_such_key": 1})


def test_missing_mandatory_field_is_an_error():
    m = load_yaml_mapping(CFG)
    del m["jacobi_omega"]
    with pytest.raises(KeyError, match="jacobi_omega"):
        SimulationParams.from_mapping(m)
    m = load_yaml_mapping(CFG)
    m["viscosity_type"] = "Nope"
    with pytest.raises(ValueError):
        SimulationParams.from_mapping(m)


def test_level_estimation_none_spelling():
    # `level_estimation_method: None` in the media recipes is the enum variant None, not a YAML null
    p = SimulationParams.from_yaml(CFG, {"level_estimation_method": "None"})
    assert p.to_ffi().level_estimation_method == 0


def test_to_ffi_values():
    f = dam_break_params().to_ffi()
    assert f.pressure_solver_method == 2 and f.viscosity_type == 1 and f.boundary_penalty_term == 2
    assert f.max_iters == 200 and abs(f.max_dt - 0.002) < 1e-9 and f.hybrid_dfsph_factor == 20000000.0
    assert f.has_pull_fluid_to == 0 and f.operator_discretization == 0 and f.support_length_estimation == 4


def test_add_fluid_block_layout():
    b = sc.SceneFluidBlock([0.4, -0.5], [0.55, 1.4], 0.03, 0.93, [0.5, -1.0])
    pos, mass, vel = sc.add_fluid_block(b)
    assert pos.shape == (18 * 46, 2)
    f32 = np.float32
    # x outer, y inner; idx * spacing + min in f32
    assert pos[0, 0] == f32(0.4) and pos[0, 1] == f32(-0.5)
    assert pos[1, 0] == f32(0.4) and pos[1, 1] == f32(1) * f32(0.03) + f32(-0.5)
    assert pos[46, 0] == f32(1) * f32(0.03) + f32(0.4)
    assert np.all(mass == f32(0.03) * f32(0.03) * f32(0.93) * f32(1.0))
    assert np.all(vel == np.array([0.5, -1.0], np.float32))


def test_scene_yaml():
    scn = sc.SceneConfig.from_yaml(str(REPO / "tests" / "golden" / "default-scene.yaml"))
    assert scn.boundary.type == "box" and scn.boundary.width == 2 and len(scn.blocks) == 2
    poly = sc.boundary_planes(scn.boundary, "AnalyticUnderestimate")   # Sdf2D::new_boundary_box (sdf2d.rs:167-179)
    assert poly.points == [(-1.0, -1.0), (1.0, -1.0), (1.0, 1.0), (-1.0, 1.0)]
    with pytest.raises(NotImplementedError):
        sc.boundary_planes(scn.boundary, "Particles")
    assert sc.boundary_planes(scn.boundary, "NoBoundary") == []


def test_run_subcommand_headless(tmp_path):
    """`run SIMULATION_CONFIG SCENE_CONFIG -s T -c OVERRIDES -p -w PATH` (main_loop.rs:36-82, 105-181, 346-350) on a
    library with the step ABI (here the CPU oracle, so the host logic is covered without a GPU)."""
    import io
    from adaptive_sph_amd.__main__ import build_parser, run
    from tests.oracle_harness import load_oracle
    cfg = str(REPO / "tests" / "golden" / "default-config.yaml")
    scn = str(REPO / "tests" / "golden" / "default-scene.yaml")
    ov = tmp_path / "ov.yaml"
    ov.write_text("max_dt: 0.001\nviscosity: 0.002\n")
    stat = tmp_path / "run.stat"
    args = build_parser().parse_args(["run", cfg, scn, "-s", "0.0035", "-c", str(ov), "-p", "-w", str(stat), "--without-adaptivity",
                                      "--vtk", str(tmp_path / "vtk"), "--vtk-every", "2"])
    out = io.StringIO()
    steps = run(args, lib=load_oracle(), out=out)
    assert steps == 4                                   # dt = max_dt = 0.001 -> time passes 0.0035 after 4 steps
    assert sorted(f.name for f in (tmp_path / "vtk").iterdir()) == ["my-sph-00001.vtk", "my-sph-00002.vtk", "my-sph.vtk.series"]
    assert (tmp_path / "vtk" / "my-sph-00002.vtk").read_bytes().count(b"SCALARS ") == 14
    text = stat.read_text()
    assert "simulation-time:" in text and "dt: min:" in text and "particle-count:" in text
    assert "max_dt=0.001" in out.getvalue().replace(" ", "") or "max_dt: 0.001" in out.getvalue() or "0.001" in out.getvalue()
    # with adaptivity (the config enables merging / sharing / splitting): single_step = the step + single_step_adaptivity; the
    # split patterns are read like the reference reads ./split-patterns.yaml -- a missing file is its unwrap() panic
    with pytest.raises(FileNotFoundError):
        run(build_parser().parse_args(["run", cfg, scn, "-s", "0.001", "--split-patterns", str(tmp_path / "nope.yaml")]), lib=load_oracle(), out=io.StringIO())
    out2 = io.StringIO()
    steps2 = run(build_parser().parse_args(["run", cfg, scn, "--max-steps", "3", "--split-patterns", str(REPO / "tests" / "golden" / "split-patterns.yaml"),
                                            "--capacity-factor", "40"]), lib=load_oracle(), out=out2)
    assert steps2 == 3
    n_final = int(out2.getvalue().split(" particles")[0].split()[-1])
    assert n_final > 1035                                # splitting at step 1 (odd step numbers, simulation.rs:2759-2785)
    # an unknown override key is the reference's panic
    bad = tmp_path / "bad.yaml"
    bad.write_text("no_such_key: 1\n")
    with pytest.raises(KeyError):
        run(build_parser().parse_args(["run", cfg, scn, "-s", "0.001", "-c", str(bad), "--without-adaptivity"]),
            lib=load_oracle(), out=io.StringIO())


def test_vtk_exporter_layout_and_boundary_distances(tmp_path):
    """VtkExporter (platform/desktop/vtk_exporter.rs:31-367): legacy VTK 4.2 BINARY big-endian POLYDATA, point data in the
    reference's order, Sdf2D edges as LINES with two padded points each, and the .vtk.series index.  No GPU: a stand-in for
    the simulation object supplies the arrays."""
    import json
    import numpy as np
    from adaptive_sph_amd import scene as sc
    from adaptive_sph_amd.vtk_exporter import VtkExporter, _plane_distance, _polygon_distance

    rng = np.random.default_rng(0)
    n = 7
    fields = {k: rng.normal(size=n).astype(np.float32) for k in
              ("density", "density_error", "pressure", "mass", "aii", "h2", "ppe_source_term", "lambda_sum")}
    fields["position"] = (rng.uniform(-0.9, 0.9, (n, 2)) * [2.0, 1.0]).astype(np.float32)
    fields["velocity"] = rng.normal(size=(n, 2)).astype(np.float32)
    fields["pressure_accel"] = rng.normal(size=(n, 2)).astype(np.float32)
    fields["flag_is_fluid_surface"] = np.array([0, 1, 0, 1, 1, 0, 0], np.uint8)
    fields["flag_neighborhood_reduced"] = np.zeros(n, np.uint8)

    class Sim:
        particles = type("P", (), fields)()

    boundary = sc.SceneBoundary("box", 4.0, 2.0)
    poly = sc.boundary_planes(boundary, "AnalyticUnderestimate")
    planes = sc.boundary_planes(boundary, "AnalyticOverestimate")
    with VtkExporter(tmp_path, "my-sph") as ex:
        ex.add_snapshot(0.0, Sim, poly)
        ex.add_snapshot(0.5, Sim, planes)
    series = json.loads((tmp_path / "my-sph.vtk.series").read_text())
    assert series["file-series-version"] == "1.0"
    assert [f["name"] for f in series["files"]] == ["my-sph-00001.vtk", "my-sph-00002.vtk"] and series["files"][1]["time"] == 0.5

    raw = (tmp_path / "my-sph-00001.vtk").read_bytes()
    assert raw.startswith(b"# vtk DataFile Version 4.2\nSPH Particles 1.0\nBINARY\nDATASET POLYDATA\nPOINTS 15 float\n")   # 7 + 2 * 4 edges
    off = raw.index(b"POINTS 15 float\n") + len(b"POINTS 15 float\n")
    pts = np.frombuffer(raw[off:off + 15 * 12], ">f4").reshape(15, 3)
    assert np.array_equal(pts[:n, :2], fields["position"]) and np.all(pts[:, 2] == 0)
    assert np.allclose(pts[n:, :2], [[-2, -1], [2, -1], [2, -1], [2, 1], [2, 1], [-2, 1], [-2, 1], [-2, -1]])
    assert b"\nVERTICES 7 14\n" in raw and b"\nLINES 4 12\n" in raw and b"\nPOINT_DATA 15\n" in raw
    names = [ln.split()[1].decode() for ln in raw.split(b"\n") if ln.startswith(b"SCALARS ")]
    assert names == ["density", "density_error", "density_error2", "pressure", "mass", "aii", "h", "ppe_source_term", "distances", "lambda",
                     "velocity", "pressure_accel", "flag_is_fluid_surface", "flag_neighborhood_reduced"]
    off = raw.index(b"SCALARS pressure float 1\nLOOKUP_TABLE default\n") + len(b"SCALARS pressure float 1\nLOOKUP_TABLE default\n")
    assert np.array_equal(np.frombuffer(raw[off:off + 15 * 4], ">f4"), np.concatenate([fields["pressure"], np.zeros(8, np.float32)]))
    assert b"LINES" not in (tmp_path / "my-sph-00002.vtk").read_bytes()

    # inside the box both boundary descriptions measure the distance to the nearest wall, positive on the air side
    x = fields["position"]
    inside = (np.abs(x[:, 0]) < 2) & (np.abs(x[:, 1]) < 1)
    want = np.minimum(2 - np.abs(x[:, 0]), 1 - np.abs(x[:, 1]))
    assert inside.all()
    assert np.allclose(_plane_distance(planes, x), want, atol=1e-6) and np.allclose(_polygon_distance(poly.points, x), want, atol=1e-6)
    out = np.array([[2.5, 0.0], [3.0, 2.0], [-2.2, -1.3]], np.float32)
    assert np.allclose(_polygon_distance(poly.points, out), [-0.5, -np.hypot(1, 1), -np.hypot(0.2, 0.3)], atol=1e-6)


def test_grid_search_is_the_uniform_builds_only():
    """neighborhood_search.rs:334-342: `Grid` asserts PARTICLE_SIZES == Uniform; the adaptive build (the one replaced) panics."""
    from adaptive_sph_amd.workloads import dam_break_params
    with pytest.raises(ValueError):
        dam_break_params(neighborhood_search_algorithm="Grid").to_ffi()
