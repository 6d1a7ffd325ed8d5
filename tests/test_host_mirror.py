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


def test_write_statistics_known_answer():
    """write_statistics (simulation.rs:3279-3359) number for number.  The expected text is what the reference's format strings
    give for these counters: `{:.2}` / `{:.02}` in the LaTeX row, Rust's `{}` elsewhere (shortest round-trip digits of the f32 /
    f64, no exponent, no trailing `.0`), Counter<FT>::avg as a sequential f32 sum (:96-106), Counter<Duration>::avg in whole
    nanoseconds (:127-129), labels sorted."""
    from adaptive_sph_amd.simulation import FluidSimulation, _Counter, _PCounter, rust_display
    assert rust_display(1035.0, True) == "1035" and rust_display(0.1, True) == "0.1" and rust_display(np.float32(1) / np.float32(3), True) == "0.33333334"
    assert rust_display(1e-7) == "0.0000001" and rust_display(2.5) == "2.5" and rust_display(float("nan")) == "NaN" and rust_display(1234.5 * 1000.) == "1234500"
    sim = FluidSimulation.__new__(FluidSimulation)
    sim.pcounters, sim.vcounters = {}, {}
    for ms in (1.5, 2.25, 0.75):                       # Durations of 1 500 000, 2 250 000 and 750 000 ns
        sim.pcounters.setdefault("simulation-step", _PCounter()).add_value(ms)
    for ms in (0.5, 0.25):
        sim.pcounters.setdefault("neighborhood-search", _PCounter()).add_value(ms)
    for n in (1035, 1035, 1036):
        sim.vcounters.setdefault("particle-count", _Counter()).add_value(n)
    for it in (3, 4, 3):
        sim.vcounters.setdefault("div-iterations", _Counter()).add_value(it)
    for it in (5, 5, 6):
        sim.vcounters.setdefault("density-iterations", _Counter()).add_value(it)
    for dt in (0.5, 0.25, 0.125):                      # exact in f32: the average is 0.875 / 3 rounded to f32
        sim.vcounters.setdefault("dt", _Counter()).add_value(dt)
    expected = (
        "$0.00\\si{\\second}$ & 1035 & 3.33 & 5.33 & - \\\\\n"
        "\n"
        "simulation-time: 4.5ms\n"
        "\n"
        "neighborhood-search: avg:0.375ms\n"
        "simulation-step: avg:1.5ms\n"
        "\n"
        "density-iterations: min:5 max:6 avg:5.3333335\n"
        "div-iterations: min:3 max:4 avg:3.3333333\n"
        "dt: min:0.125 max:0.5 avg:0.29166666\n"
        "particle-count: min:1035 max:1036 avg:1035.3334\n")
    assert sim.write_statistics() == expected


def _read_legacy_vtk(path):
    """An independent reader of the legacy VTK POLYDATA format (file-formats.pdf of VTK 4.2: header, POINTS, VERTICES, LINES,
    POINT_DATA with SCALARS arrays; BINARY sections are big-endian), used to pin what VtkExporter writes."""
    raw = open(path, "rb").read()
    pos = 0

    def line():
        nonlocal pos
        while raw[pos:pos + 1] == b"\n":
            pos += 1
        end = raw.index(b"\n", pos)
        out = raw[pos:end].decode()
        pos = end + 1
        return out

    def take(dtype, count):
        nonlocal pos
        a = np.frombuffer(raw, dtype, count, pos)
        pos += a.nbytes
        return a

    out = {"version": line(), "title": line(), "encoding": line(), "dataset": line(), "arrays": []}
    kw, n, ty = line().split()
    assert kw == "POINTS" and ty == "float"
    out["points"] = take(">f4", 3 * int(n)).reshape(-1, 3)
    kw, nc, sz = line().split()
    assert kw == "VERTICES"
    out["verts"] = take(">i4", int(sz)).reshape(int(nc), -1)
    nxt = line()
    if nxt.startswith("LINES"):
        _, nc, sz = nxt.split()
        out["lines"] = take(">i4", int(sz)).reshape(int(nc), -1)
        nxt = line()
    kw, npd = nxt.split()
    assert kw == "POINT_DATA"
    out["n_point_data"] = int(npd)
    while pos < len(raw) and raw[pos:].strip():
        kw, name, ty, comps = line().split()
        assert kw == "SCALARS" and line() == "LOOKUP_TABLE default"
        dt = {"float": ">f4", "unsigned_char": "u1"}[ty]
        out["arrays"].append((name, ty, int(comps), take(dt, int(npd) * int(comps)).reshape(int(npd), int(comps))))
    return out


def test_vtk_snapshot_read_back(tmp_path):
    """What write_vtk_file / write_vtk_file2 (vtk_exporter.rs:82-152, 256-367) put into a snapshot: version 4.2, title, BINARY
    big-endian POLYDATA, particles as VERTICES (`1 i`), Sdf2D edges as LINES over two extra points each, point data in the
    reference's order, names, component counts and types with zeros on the line points."""
    from adaptive_sph_amd.vtk_exporter import write_vtk_file2
    rng = np.random.default_rng(3)
    n = 37
    pos = rng.random((n, 2), np.float32)
    ft = [(k, rng.random(n, np.float32)) for k in ("density", "density_error", "density_error2", "pressure", "mass", "aii", "h", "ppe_source_term", "distances", "lambda")]
    vec = [(k, rng.random((n, 2), np.float32)) for k in ("velocity", "pressure_accel")]
    u8 = [(k, (rng.random(n) < 0.3).astype(np.uint8)) for k in ("flag_is_fluid_surface", "flag_neighborhood_reduced")]
    lines = [((0.0, 0.0), (1.0, 0.0)), ((1.0, 0.0), (1.0, 1.0)), ((1.0, 1.0), (0.0, 0.0))]
    path = tmp_path / "snap.vtk"
    write_vtk_file2(path, pos, ft, vec, u8, lines)
    v = _read_legacy_vtk(path)
    assert (v["version"], v["title"], v["encoding"], v["dataset"]) == ("# vtk DataFile Version 4.2", "SPH Particles 1.0", "BINARY", "DATASET POLYDATA")
    assert v["points"].shape == (n + 6, 3) and np.array_equal(v["points"][:n, :2], pos) and not v["points"][:, 2].any()
    assert np.array_equal(v["points"][n:, :2].reshape(3, 2, 2), np.array(lines, np.float32))
    assert np.array_equal(v["verts"], np.stack([np.ones(n, np.int32), np.arange(n, dtype=np.int32)], 1))
    assert np.array_equal(v["lines"], np.array([[2, n + 2 * k, n + 2 * k + 1] for k in range(3)]))
    assert v["n_point_data"] == n + 6
    assert [(a[0], a[1], a[2]) for a in v["arrays"]] == [(k, "float", 1) for k, _ in ft] + [(k, "float", 3) for k, _ in vec] + [(k, "unsigned_char", 1) for k, _ in u8]
    for (name, _, comps, got), (_, want) in zip(v["arrays"], ft + vec + u8):
        w = np.asarray(want).reshape(n, -1)
        assert np.array_equal(got[:n, :w.shape[1]], w), name
        assert not got[n:].any() and not got[:, w.shape[1]:].any(), name       # dummy data on the line points, z = 0


def test_assemble_lists_of_the_ranks_in_global_index_order():
    """distributed.assemble_lists: the CSR exports of the ranks (rows in each rank's own order, global ids as indices) become one
    CSR whose row i is the list of particle i -- what the sequential partner searches of an adaptive slab step iterate."""
    from adaptive_sph_amd.distributed import assemble_lists
    rng = np.random.default_rng(11)
    n = 500
    truth = [rng.choice(n, size=int(rng.integers(0, 9)), replace=False).astype(np.uint32) for _ in range(n)]
    owner = rng.integers(0, 3, n)
    ids, lists = [], []
    for r in range(3):
        mine = rng.permutation(np.nonzero(owner == r)[0])          # a rank's rows are in ITS order, not ascending ids
        off = np.zeros(len(mine) + 1, np.uint32)
        off[1:] = np.cumsum([len(truth[i]) for i in mine])
        ids.append(mine.astype(np.uint32))
        lists.append((off, np.concatenate([truth[i] for i in mine]) if len(mine) else np.zeros(0, np.uint32)))
    off_g, idx_g = assemble_lists(ids, lists, n)
    assert off_g[0] == 0 and off_g[-1] == sum(len(t) for t in truth)
    for i in range(n):
        assert np.array_equal(idx_g[off_g[i]:off_g[i + 1]], truth[i]), i


def test_host_buffers_are_reused_and_grow():
    """ffi.HostBuffers (round 6): an export lands in the same, already touched host memory every step; a request that does not fit gets a
    larger buffer; views of different keys never alias."""
    from adaptive_sph_amd import ffi
    hb = ffi.HostBuffers()
    a = hb.view("field:mass", np.float32, 1000)
    a[:] = 7.0
    b = hb.view("field:mass", np.float32, 900)
    assert b.ctypes.data == a.ctypes.data and b.shape == (900,) and (b == 7.0).all()          # the same memory, nothing reallocated
    c = hb.view("csr:indices", np.uint32, 1000)
    assert c.ctypes.data != a.ctypes.data and (c == 0).all()                                 # another key, zero-filled (= touched)
    assert hb.capacity("field:mass", np.float32) >= 1000
    big = hb.view("field:mass", np.float32, 100000)                                          # does not fit: a new, larger buffer
    assert big.shape == (100000,) and hb.capacity("field:mass", np.float32) >= 100000
    hb2 = ffi.HostBuffers()
    hb2.reserve(1234)
    assert hb2.capacity("csr:indices", np.uint32) >= 16 * 1234 and hb2.capacity("field:position", np.float32) >= 2 * 1234
    assert hb2.capacity("merge_counter", np.uint16) >= 1234


def test_bench_picks_the_newest_committed_summary_by_parsed_round(tmp_path, monkeypatch):
    """bench.py's `bound` / `traffic` come from the newest committed profiles/ summary: 'r10' is newer than 'r5q' although it sorts in
    front of it (advisor r5), 'r6p' newer than 'r6', a suffix newer than none."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_for_test", REPO / "bench.py")
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    (tmp_path / "profiles").mkdir()
    for name in ("r5q_kernel_summary.json", "r10_kernel_summary.json", "r6_kernel_summary.json", "r6p_kernel_summary.json", "r9z_dam_break_8m_kernel_summary.json",
                 "r2a_level_1m_kernel_summary.json"):
        (tmp_path / "profiles" / name).write_text("{}")
    monkeypatch.setattr(bench, "REPO", tmp_path)
    assert bench.newest_summary("*_kernel_summary.json", r"r(\d+)([a-z]?)_kernel_summary\.json").name == "r10_kernel_summary.json"
    (tmp_path / "profiles" / "r10_kernel_summary.json").unlink()
    assert bench.newest_summary("*_kernel_summary.json", r"r(\d+)([a-z]?)_kernel_summary\.json").name == "r6p_kernel_summary.json"
    assert bench.newest_summary("*_dam_break_8m_kernel_summary.json", r"r(\d+)([a-z]?)_dam_break_8m_kernel_summary\.json").name == "r9z_dam_break_8m_kernel_summary.json"
    assert bench.newest_summary("*_nothing.json", r"r(\d+)([a-z]?)_nothing\.json") is None
