"""Pin the CPU oracle against every known-answer test the reference holds for this path
(SURVEY.md section 8c).  No GPU.

Reference tests restated here:
  sph_kernels.rs:88-114    cubic_kernel_2d_integration_test      (integral == 1 within 1.00001)
  sph_kernels.rs:116-163   cubic_kernel_2d_derivative_test       (|analytic - central diff| < 1e-3)
  sph_kernels.rs:214-227   test_radius_and_sphere_volume_conversion (round trip within 1e-6)
  plane_numerics.rs:180-203  test_dlambda2_specific_values       (12 Maxima values, tol 1e-8)
  plane_numerics.rs:205-224  test_dlambda2_finite_diffs          (600 001 points, tol 1e-7)
  plane_numerics.rs:226-249  test_lambda2_specific_values        (11 Maxima values, tol 1e-8)
  plane_numerics.rs:251-299  test_lambda2_integrations           (5 h x 101 d, tol 1e-5)
plus the scene-initialiser particle counts derived in SURVEY.md (f32 floor rule).
"""
import ctypes as C

import numpy as np
import pytest

from adaptive_sph_amd import scene as sc
from tests.oracle_harness import oracle_kernel_deriv, oracle_scene_block, oracle_box

# (d, value) pairs are DATA quoted from the reference's tests (Maxima evaluations)
DLAMBDA2_VALUES = [
    (1.0e-5, -1.364185225745495), (0.1, -1.291255734976317), (0.2, -1.09590958428671),
    (0.3, -0.8294373145386852), (0.475, -0.3694455226951835), (0.49999999, -0.3172459084022253),
    (0.5, -0.3172458884798477), (0.6, -0.1553847490374719), (0.7, -0.06022919733948317),
    (0.8, -0.01536108745740005), (0.9, -0.001424092559566546), (0.9999999999, -1.37123132821062e-10),
]
LAMBDA2_VALUES = [
    (1.0e-5, 0.4999863581477375), (0.1, 0.3660454031974235), (0.2, 0.2458568798927798),
    (0.3, 0.1492433688434099), (0.475, 0.04601588929110174), (0.5, 0.03744216427059437),
    (0.6, 0.01442031051340694), (0.7, 0.00413432923941152), (0.8, 6.949615905699156e-4),
    (0.9, 3.190640160164168e-5), (1.0, 0.0),
]


def test_dlambda2_specific_values(oracle_lib):
    for d, want in DLAMBDA2_VALUES:
        assert abs(oracle_lib.lib.oracle_dlambda2(d) - want) <= 1e-8, d


def test_lambda2_specific_values(oracle_lib):
    for d, want in LAMBDA2_VALUES:
        assert abs(oracle_lib.lib.oracle_lambda2(d) - want) <= 1e-8, d


def test_lambda2_symmetry_and_limits(oracle_lib):
    L = oracle_lib.lib
    assert L.oracle_lambda2(0.0) == 0.5
    assert L.oracle_lambda2(1.5) == 0.0
    assert L.oracle_lambda2(-1.5) == 1.0
    for d in (0.05, 0.3, 0.77):
        assert abs(L.oracle_lambda2(d) + L.oracle_lambda2(-d) - 1.0) < 1e-15
        assert L.oracle_dlambda2(d) == L.oracle_dlambda2(-d)


def test_dlambda2_finite_diffs(oracle_lib):
    L = oracle_lib.lib
    steps, eps, tol = 300000, 0.0000001, 0.0000001
    worst = 0.0
    for i in range(-steps, steps + 1):
        x = i / steps
        num = (L.oracle_lambda2(x + eps) - L.oracle_lambda2(x - eps)) / (2.0 * eps)
        worst = max(worst, abs(num - L.oracle_dlambda2(x)))
    assert worst <= tol


def test_lambda2_integrations(oracle_lib):
    L = oracle_lib.lib
    L.oracle_test_lambda2_integral.restype = C.c_double
    L.oracle_test_lambda2_integral.argtypes = [C.c_double, C.c_double]
    for h in (1.0, 0.0001, 0.05, 2.0, 10.0):
        for step in range(50, -51, -1):
            d = (step / 40.0) * h
            numeric = L.oracle_test_lambda2_integral(h, d)
            analytic = L.oracle_lambda2(d / (2.0 * h))
            assert abs(analytic - numeric) <= 0.00001, (h, d)


def test_cubic_kernel_2d_integration(oracle_lib):
    L = oracle_lib.lib
    L.oracle_test_kernel_integral.restype = C.c_float
    L.oracle_test_kernel_integral.argtypes = [C.c_float, C.c_int]
    integral = L.oracle_test_kernel_integral(5.0, 200)
    assert 1.0 / 1.00001 <= integral <= 1.00001


def test_cubic_kernel_2d_derivative(oracle_lib):
    L = oracle_lib.lib
    f32 = np.float32
    h = f32(5.0)
    sr = f32(2.0) * h
    n = 100
    diff = sr * f32(1e-2)
    half = diff * f32(0.5)
    off = f32(2.0) * sr / f32(n)
    W = lambda x, y: L.oracle_cubic_kernel_2d(float(np.sqrt(f32(x) * f32(x) + f32(y) * f32(y))), float(h))
    for y in range(0, n + 1, 3):       # every third probe of the reference's 101x101 grid
        for x in range(0, n + 1, 3):
            px = (f32(x) + f32(0.5)) * off - sr
            py = (f32(y) + f32(0.5)) * off - sr
            gx, gy = oracle_kernel_deriv(oracle_lib, float(px), float(py), float(h))
            ax = (W(px + half, py) - W(px - half, py)) / float(diff)
            ay = (W(px, py + half) - W(px, py - half)) / float(diff)
            assert abs(gx - ax) < 0.001 and abs(gy - ay) < 0.001


def test_radius_and_sphere_volume_conversion(oracle_lib):
    L = oracle_lib.lib
    for x in (0.1, 0.5, 1.0, 100.0):
        x2 = L.oracle_radius_to_sphere_volume(L.oracle_sphere_volume_to_radius(x))
        assert abs(x - x2) <= 0.000001 * max(1.0, x)   # f32: the reference's 1e-6 is absolute at x<=1


def test_lut_endpoints_and_lerp(oracle_lib):
    from adaptive_sph_amd import ffi
    ctx = ffi.Context(oracle_lib, 4, [(1.0, 0.0, 1.0)])
    lam = np.empty(10001, np.float32)
    dlam = np.empty(10001, np.float32)
    oracle_lib.lib.oracle_lambda_luts(ctx.handle, lam.ctypes.data, dlam.ctypes.data)
    assert lam[0] == np.float32(1.0) and lam[10000] == np.float32(0.0) and lam[5000] == np.float32(0.5)
    assert np.all(np.diff(lam.astype(np.float64)) <= 0)          # lambda is monotone decreasing in d
    assert np.all(dlam <= 0)
    # LookupTable1D::get lerps between neighbouring samples
    x = np.float32(-0.33337)
    got = oracle_lib.lib.oracle_lut_get(ctx.handle, 0, float(x))
    assert abs(got - oracle_lib.lib.oracle_lambda2(float(x))) < 1e-6
    ctx.close()


# ---- scene initialiser: counts pinned by the reference's f32 floor rule -------

def test_scene_counts_default_scene(oracle_lib):
    from tests.oracle_harness import REPO
    scene = sc.SceneConfig.from_yaml(str(REPO / "tests" / "golden" / "default-scene.yaml"))
    dims = [sc.block_dims(b) for b in scene.blocks]
    assert dims == [(18, 46), (9, 23)]
    pos, mass, vel = sc.init_particles(scene)
    assert pos.shape[0] == 1035
    # the oracle's own C restatement of add_fluid_block agrees bit for bit with the host mirror
    for b in scene.blocks:
        p_o, m_o, v_o = oracle_scene_block(oracle_lib, b)
        p_h, m_h, v_h = sc.add_fluid_block(b)
        assert np.array_equal(p_o, p_h) and np.array_equal(m_o, m_h) and np.array_equal(v_o, v_h)


def test_scene_counts_baseline_configs():
    assert sc.block_dims(sc.dam_break_1m().blocks[0]) == (1024, 1024)
    a = sc.dam_break_1m_adaptive()
    assert [sc.block_dims(b) for b in a.blocks] == [(1024, 920), (230, 256)]
    assert sc.block_dims(sc.dam_break_8m().blocks[0]) == (8192, 1024)
    # media/motivation-scene2.yaml geometry: 150 x 224 (f32 floor of 1.8/0.008 is 224)
    blk = sc.SceneFluidBlock([-0.95, -0.9], [1.2, 1.8], 0.008, 0.93, [0, 0])
    assert sc.block_dims(blk) == (150, 224)


def test_boundary_box_planes(oracle_lib):
    planes = sc.boundary_planes(sc.SceneBoundary("box", 4.0, 2.0))
    assert planes == oracle_box(oracle_lib, 4.0, 2.0)
    assert planes == [(1.0, 0.0, 2.0), (-1.0, 0.0, 2.0), (0.0, 1.0, 1.0), (0.0, -1.0, 1.0)]
