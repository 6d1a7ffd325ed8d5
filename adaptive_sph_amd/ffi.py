"""ctypes binding of include/sph_ffi.h.

The product library is ``adaptive_sph_amd/csrc/libsph_hip.so`` (symbol prefix ``sph_``).  The binding
is prefix-parametrised only so that the test-suite can drive the CPU oracle (``oracle_`` prefix,
same signatures) through the identical harness; nothing in this package loads anything under
``oracle/`` -- `load_product()` fails loudly when the HIP library is missing.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

PKG_DIR = Path(__file__).resolve().parent
# SPH_HIP_LIBRARY: another build of the SAME sources next to the product library (scripts/variants: compile-time switches,
# prebuilt where hipcc is, timed on the GPU box) -- a file name inside csrc/, never a path to anything else
PRODUCT_LIB = PKG_DIR / "csrc" / os.path.basename(os.environ.get("SPH_HIP_LIBRARY", "libsph_hip.so"))

# ---- enums (include/sph_ffi.h; names follow simulation_parameters.rs) ---------------------------
VISCOSITY_TYPE = {"WCSPH": 0, "ApproxLaplace": 1, "XSPH": 2}
LEVEL_ESTIMATION_METHOD = {"None": 0, "CenterDiff": 1, "EmptyAngle": 2}
SUPPORT_LENGTH_ESTIMATION = {
    "FromDistribution": 0, "FromDistributionClamped1": 1, "FromDistributionClamped2": 2,
    "FromDistribution2": 3, "FromMass": 4,
}
PRESSURE_SOLVER_METHOD = {"IISPH": 0, "IISPH2": 1, "HybridDFSPH": 2, "OnlyDivergence": 3}
HYBRID_DFSPH_DENSITY_SOURCE_TERM = {"DensityAndDivergence": 0, "OnlyDensity": 1}
BOUNDARY_PENALTY_TERM = {"None": 0, "Linear": 1, "Quadratic1": 2, "Quadratic2": 3}
OPERATOR_DISCRETIZATION = {"ConsistentSimpleGradient": 0, "ConsistentSymmetricGradient": 1, "Winchenbach2020": 2}
FILL_STASH_WITH = {None: 0, "SurfaceDistanceFirstIteration": 1, "SurfaceDistanceMiddle": 2}
SIZING_FUNCTION = {"Radius2": 0, "Radius": 1, "Mass": 2}

# field id -> (numpy dtype, components)
FIELDS = {
    "mass": (0, np.float32, 1), "position": (1, np.float32, 2), "velocity": (2, np.float32, 2),
    "pressure_accel": (3, np.float32, 2), "density": (4, np.float32, 1), "ppe_source_term": (5, np.float32, 1),
    "pressure": (6, np.float32, 1), "aii": (7, np.float32, 1), "density_error": (8, np.float32, 1),
    "h2": (9, np.float32, 1), "h2_next": (10, np.float32, 1), "constant_field": (11, np.float32, 1),
    "neighbor_count": (12, np.uint32, 1), "level_estimation": (13, np.float32, 1), "level_old": (14, np.float32, 1),
    "stash": (15, np.float32, 1), "flag_is_fluid_surface": (16, np.uint8, 1),
    "flag_insufficient_neighs": (17, np.uint8, 1), "particle_size_class": (18, np.uint8, 1),
    "lambda_sum": (19, np.float32, 1), "lambda_grad_sum": (20, np.float32, 2), "cell_index": (21, np.uint32, 1),
    "particle_id": (22, np.uint32, 1), "flag_neighborhood_reduced": (23, np.uint8, 1),
}

STATUS_NAMES = {
    0: "SPH_OK", 1: "SPH_ERR_INVALID_ARGUMENT", 2: "SPH_ERR_DEVICE", 3: "SPH_ERR_CAPACITY", 4: "SPH_ERR_NO_BOUNDARY",
    10: "SPH_ERR_DENSITY_NOT_FINITE", 11: "SPH_ERR_DENSITY_TOO_SMALL", 12: "SPH_ERR_AII_NOT_FINITE",
    13: "SPH_ERR_AII_NEGATIVE", 14: "SPH_ERR_AP_NOT_FINITE", 15: "SPH_ERR_PRESSURE_NOT_FINITE",
    16: "SPH_ERR_TOO_MANY_NEIGHBORS", 17: "SPH_ERR_VELOCITY_NOT_FINITE", 18: "SPH_ERR_POSITION_NOT_FINITE",
    19: "SPH_ERR_VISCOSITY_NOT_FINITE", 20: "SPH_ERR_XSPH_TODO", 21: "SPH_ERR_CHECK_NEIGHBORHOOD",
    22: "SPH_ERR_CHECK_AII", 23: "SPH_ERR_LEVEL_WEIGHT", 24: "SPH_ERR_VOLUME_ESTIMATE", 25: "SPH_ERR_CONSTRAIN_NOT_SMALLER", 26: "SPH_ERR_CONSTRAIN_NEGATIVE", 27: "SPH_ERR_NO_SPLIT_PATTERN", 30: "SPH_ERR_UNSUPPORTED", 31: "SPH_ERR_POISONED",
}


class SphParams(C.Structure):
    _fields_ = [
        ("rest_density", C.c_float), ("cfl_factor", C.c_float), ("max_dt", C.c_float), ("viscosity", C.c_float),
        ("viscosity_type", C.c_int32), ("gravity", C.c_float), ("jacobi_omega", C.c_float),
        ("level_estimation_method", C.c_int32), ("maximum_range", C.c_float),
        ("support_length_estimation", C.c_int32), ("sdf_gradient_eps", C.c_float),
        ("has_pull_fluid_to", C.c_int32), ("pull_fluid_to", C.c_float * 3),
        ("maximum_surface_distance", C.c_float), ("boundary_is_fluid_surface", C.c_int32),
        ("use_extended_range_for_level_estimation", C.c_int32), ("level_estimation_after_advection", C.c_int32),
        ("level_estimation_range", C.c_float), ("pressure_solver_method", C.c_int32),
        ("iisph_max_avg_density_error", C.c_float), ("hybrid_dfsph_factor", C.c_float),
        ("hybrid_dfsph_max_avg_density_error", C.c_float), ("hybrid_dfsph_max_avg_divergence_error", C.c_float),
        ("hybrid_dfsph_density_source_term", C.c_int32),
        ("hybrid_dfsph_non_pressure_accel_before_divergence_free", C.c_int32),
        ("boundary_penalty_term", C.c_int32), ("operator_discretization", C.c_int32), ("max_iters", C.c_uint32),
        ("check_neighborhood", C.c_int32), ("check_aii", C.c_int32), ("constrain_neighborhood_count", C.c_int32),
        ("fill_stash_with", C.c_int32), ("sizing_function", C.c_int32), ("particle_radius_fine", C.c_float),
        ("particle_radius_base", C.c_float),
    ]


class SphPlane(C.Structure):
    _fields_ = [("dir_x", C.c_float), ("dir_y", C.c_float), ("delta", C.c_float)]


class SphSolverStats(C.Structure):
    _fields_ = [
        ("iters", C.c_uint32), ("converged", C.c_int32), ("normal_count", C.c_uint32), ("singular_count", C.c_uint32),
        ("negative_count", C.c_uint32), ("avg_error", C.c_float), ("max_error", C.c_float),
    ]


class SphStepStats(C.Structure):
    _fields_ = [
        ("dt", C.c_float), ("time", C.c_float), ("step_number", C.c_uint64), ("n_particles", C.c_uint64),
        ("div_solver", SphSolverStats), ("density_solver", SphSolverStats),
        ("ms_simulation_step", C.c_double), ("ms_neighborhood", C.c_double), ("ms_level_estimation", C.c_double),
        ("ms_div_solver", C.c_double), ("ms_density_solver", C.c_double),
    ]


class SphGridInfo(C.Structure):
    _fields_ = [("cell_size", C.c_float), ("cells_min_x", C.c_int32), ("cells_min_y", C.c_int32),
                ("size_x", C.c_int32), ("size_y", C.c_int32)]


class SphEditOp(C.Structure):
    _fields_ = [("kind", C.c_int32), ("a", C.c_uint32), ("b", C.c_uint32), ("fields", C.c_uint32), ("mass", C.c_float),
                ("position", C.c_float * 2), ("velocity", C.c_float * 2), ("h2", C.c_float), ("h2_next", C.c_float),
                ("level_estimation", C.c_float), ("level_old", C.c_float)]


EDIT_SET, EDIT_SWAP, EDIT_TRUNCATE, EDIT_EXTEND = 0, 1, 2, 3
EDIT_FIELD_BITS = {"mass": 1, "position": 2, "velocity": 4, "h2": 8, "h2_next": 16, "level_estimation": 32, "level_old": 64}


class SphKernelTime(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_uint64), ("total_ms", C.c_double),
                ("working_launches", C.c_uint64), ("working_ms", C.c_double)]


class SphAdaptParams(C.Structure):
    _fields_ = [("dt", C.c_float), ("max_mass_transfer_sharing", C.c_float), ("minimum_share_partners", C.c_uint32),
                ("minimum_merge_partners", C.c_uint32), ("fail_on_missing_split_pattern", C.c_int32),
                ("max_share_distance", C.c_float), ("max_merge_distance", C.c_float),
                ("allow_share_with_optimal_particle", C.c_int32), ("allow_share_with_too_small_particle", C.c_int32),
                ("allow_merge_with_optimal_particle", C.c_int32), ("allow_merge_on_size_difference", C.c_int32)]


MERGE_PARTNER_AVAILABLE = 0xFFFFFFFF   # adaptivity/mod.rs:29
MERGE_PARTNER_DELETE = 0xFFFFFFFE      # adaptivity/mod.rs:30


class SphDistStats(C.Structure):
    _fields_ = [("steps", C.c_uint64), ("exchanges", C.c_uint64), ("bytes_sent", C.c_uint64), ("bytes_received", C.c_uint64),
                ("allreduces", C.c_uint64), ("host_waits", C.c_uint64), ("n_owned", C.c_uint64), ("n_halo", C.c_uint32 * 2),
                ("n_ghost", C.c_uint32 * 2), ("transport", C.c_uint32), ("comm_ranks", C.c_uint32)]


TRANSPORT_NAMES = {0: "none", 1: "loopback", 2: "rccl", 3: "threads", 4: "shm", 5: "ipc"}


class SphListForms(C.Structure):
    _fields_ = [("n_lists", C.c_uint64), ("n_mask", C.c_uint64), ("n_index", C.c_uint64), ("n_walk", C.c_uint64), ("n_wall", C.c_uint64)]


class SphError(RuntimeError):
    """Non-zero status from the library == a panic!/assert! of the reference step."""

    def __init__(self, status: int, message: str):
        self.status = status
        super().__init__(f"{STATUS_NAMES.get(status, status)}: {message}")


# every symbol include/sph_ffi.h declares (checked by tests/test_abi.py)
MATH_POLICIES = {"fast": 0, "exact": 1}   # enum sph_math_policy

ABI_SYMBOLS = [
    "create", "destroy", "upload", "upload_field", "download", "download_neighbors", "num_particles", "time",
    "set_time", "step", "classify", "share_particles", "merge_particles", "set_split_patterns", "split_particles", "host_find_partners", "last_error", "grid", "set_boundary_polygon", "set_math_policy", "get_math_policy", "apply_edits", "profile_enable", "profile_reset", "profile_get", "profile_event_overhead", "profile_dispatch_bracket", "profile_copy_bandwidth", "profile_list_forms", "set_sweep_variant",
    "dist_configure", "dist_set_rebalance", "dist_get_cuts", "dist_get_stats", "comm_unique_id", "comm_init", "comm_init_shm", "comm_ipc_export", "comm_init_ipc", "group_step", "group_adapt", "thread_group_create", "thread_group_destroy", "comm_init_threads",
]


class HostBuffers:
    """Persistent host arrays for repeated device -> host exports (Context.download / download_neighbors): allocated once, TOUCHED once,
    grown when a request does not fit.  What a Rust host has for free -- its ParticleVec and NeighborhoodCache vectors live across steps."""

    def __init__(self):
        self._bufs = {}

    def capacity(self, key, dtype) -> int:
        b = self._bufs.get(key)
        return 0 if b is None else b.nbytes // np.dtype(dtype).itemsize

    def view(self, key, dtype, count: int) -> np.ndarray:
        need = int(count) * np.dtype(dtype).itemsize
        b = self._bufs.get(key)
        if b is None or b.nbytes < need:
            b = np.empty(need + need // 8 + 4096, dtype=np.uint8)
            b.fill(0)   # every page is touched HERE, not inside a copy (np.zeros would hand out untouched calloc pages)
            self._bufs[key] = b
        return b[:need].view(dtype)

    def reserve(self, n: int, neighbours_per_particle: int = 16):
        """Touch the buffers an adaptive step of `n` particles exports into (the five decision fields, the CSR lists, the partner arrays)
        ahead of the first step -- what a host whose vectors exist from the start has anyway."""
        for name in ("particle_size_class", "mass", "level_estimation", "position", "h2"):
            fid, dt, w = FIELDS[name]
            self.view("field:" + name, dt, n * w)
        self.view("csr:offsets", np.uint32, n + 1)
        self.view("csr:indices", np.uint32, neighbours_per_particle * n)
        self.view("merge_partner", np.uint32, n)
        self.view("merge_counter", np.uint16, n)


class SphLibrary:
    """A loaded implementation of include/sph_ffi.h."""

    def __init__(self, path: os.PathLike, prefix: str = "sph_", global_symbols: bool = None):
        path = Path(path)
        if not path.exists():
            raise FileNotFoundError(
                f"{path} is missing: the HIP library must be built first "
                f"(python -c 'import __graft_entry__ as g; g.build()' or adaptive_sph_amd.build.build_hip())")
        self.path = path
        self.prefix = prefix
        if global_symbols is None:
            global_symbols = prefix == "sph_"
        self.lib = C.CDLL(str(path), mode=C.RTLD_GLOBAL if global_symbols else C.RTLD_LOCAL)
        L, P = self.lib, prefix
        vp, u64, i32 = C.c_void_p, C.c_uint64, C.c_int

        def sig(name, res, args, required=True):
            try:
                fn = getattr(L, P + name)
            except AttributeError:
                if required:
                    raise
                return None
            fn.restype, fn.argtypes = res, args
            return fn

        self.create = sig("create", i32, [u64, i32, C.POINTER(SphPlane), i32, C.POINTER(vp)])
        self.destroy = sig("destroy", None, [vp])
        self.set_boundary_polygon = sig("set_boundary_polygon", i32, [vp, C.POINTER(C.c_float), i32])
        self.set_math_policy = sig("set_math_policy", i32, [vp, i32])
        self.get_math_policy = sig("get_math_policy", i32, [vp])
        self.upload = sig("upload", i32, [vp, u64, vp, vp, vp])
        self.upload_field = sig("upload_field", i32, [vp, i32, vp, u64])
        self.apply_edits = sig("apply_edits", i32, [vp, C.POINTER(SphEditOp), u64])
        self.download = sig("download", i32, [vp, i32, vp, u64])
        self.download_neighbors = sig("download_neighbors", i32, [vp, vp, vp, u64, C.POINTER(u64)])
        self.num_particles = sig("num_particles", u64, [vp])
        self.time = sig("time", C.c_float, [vp])
        self.set_time = sig("set_time", i32, [vp, C.c_float, u64])
        self.step = sig("step", i32, [vp, C.POINTER(SphParams), C.POINTER(SphStepStats)])
        self.classify = sig("classify", i32, [vp, C.POINTER(SphParams)])
        ap = C.POINTER(SphAdaptParams)
        self.share_particles = sig("share_particles", i32, [vp, C.POINTER(SphParams), ap, vp, vp])
        self.merge_particles = sig("merge_particles", i32, [vp, C.POINTER(SphParams), ap, vp, vp])
        self.set_split_patterns = sig("set_split_patterns", i32, [vp, C.c_uint32, vp])
        self.split_particles = sig("split_particles", i32, [vp, C.POINTER(SphParams), ap])
        self.host_find_partners = sig("host_find_partners", i32, [i32, u64, vp, vp, vp, vp, vp, vp, vp, C.POINTER(SphParams), ap, vp, vp, C.POINTER(u64)],
                                      required=False)
        self.last_error = sig("last_error", C.c_char_p, [vp])
        self.grid = sig("grid", i32, [vp, C.POINTER(SphGridInfo)])
        # product-only entry points (the oracle has no device, profiler or communicator)
        self.profile_enable = sig("profile_enable", i32, [vp, i32], required=False)
        self.profile_reset = sig("profile_reset", i32, [vp], required=False)
        self.profile_get = sig("profile_get", i32, [vp, C.POINTER(SphKernelTime), i32, C.POINTER(i32)], required=False)
        self.profile_event_overhead = sig("profile_event_overhead", i32, [vp, C.POINTER(C.c_double)], required=False)
        self.profile_dispatch_bracket = sig("profile_dispatch_bracket", i32, [vp, C.c_uint32, i32, C.POINTER(C.c_double)], required=False)
        self.profile_copy_bandwidth = sig("profile_copy_bandwidth", i32, [vp, u64, C.POINTER(C.c_double)], required=False)
        self.profile_list_forms = sig("profile_list_forms", i32, [vp, C.POINTER(SphListForms)], required=False)
        self.set_sweep_variant = sig("set_sweep_variant", i32, [i32], required=False)
        self.group_adapt = sig("group_adapt", i32, [vp, i32, i32, C.POINTER(SphParams), ap, vp, vp], required=False)
        self.comm_unique_id = sig("comm_unique_id", i32, [C.POINTER(C.c_uint8)], required=False)
        self.comm_init = sig("comm_init", i32, [vp, C.POINTER(C.c_uint8), i32, i32], required=False)
        self.comm_init_shm = sig("comm_init_shm", i32, [vp, C.c_char_p, i32, i32, u64, i32], required=False)
        self.comm_ipc_export = sig("comm_ipc_export", i32, [vp, u64, C.POINTER(C.c_uint8)], required=False)
        self.comm_init_ipc = sig("comm_init_ipc", i32, [vp, C.POINTER(C.c_uint8), i32], required=False)
        self.dist_configure = sig("dist_configure", i32, [vp, i32, i32, C.c_float, C.c_float], required=False)
        self.group_step = sig("group_step", i32, [C.POINTER(vp), i32, C.POINTER(SphParams), C.POINTER(SphStepStats)], required=False)
        self.thread_group_create = sig("thread_group_create", i32, [i32, C.POINTER(vp)], required=False)
        self.thread_group_destroy = sig("thread_group_destroy", None, [vp], required=False)
        self.comm_init_threads = sig("comm_init_threads", i32, [vp, vp, i32, i32], required=False)
        self.dist_set_rebalance = sig("dist_set_rebalance", i32, [vp, i32], required=False)
        self.dist_get_cuts = sig("dist_get_cuts", i32, [vp, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_uint32)], required=False)
        self.dist_get_stats = sig("dist_get_stats", i32, [vp, C.POINTER(SphDistStats), i32], required=False)


_PRODUCT = None


def load_product() -> SphLibrary:
    """The HIP library.  No fallback: a missing/unbuildable extension is an error."""
    global _PRODUCT
    if _PRODUCT is None:
        # torch bundles its own libamdhip64 / libhsa-runtime64 / librccl with the SAME sonames as
        # /opt/rocm's.  Whichever set is loaded first serves the whole process, and a mixed set aborts at
        # exit -- so pin the order: torch's runtime first, then this library binds to it.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        _PRODUCT = SphLibrary(PRODUCT_LIB, "sph_")
    return _PRODUCT


def _as_f32(a, shape=None) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.float32)
    if shape is not None:
        a = a.reshape(shape)
    return a


class Context:
    """One simulation context (``FluidSimulation`` state container on the library side)."""

    def __init__(self, lib: SphLibrary, n_capacity: int, planes, device_id: int = 0):
        """`planes`: the (dir_x, dir_y, delta) planes of an AnalyticOverestimate box, or a scene.BoundaryPolygon (the single
        Sdf2D of AnalyticUnderestimate)."""
        self.lib = lib
        self.is_slab = False
        polygon = getattr(planes, "points", None)
        planes = [] if polygon is not None else list(planes)
        arr = (SphPlane * max(1, len(planes)))()
        for k, (dx, dy, delta) in enumerate(planes):
            arr[k] = SphPlane(dx, dy, delta)
        h = C.c_void_p()
        rc = lib.create(int(n_capacity), int(device_id), arr, len(planes), C.byref(h))
        if rc != 0:
            raise SphError(rc, "create failed")
        self.handle = h
        self.capacity = int(n_capacity)
        if polygon is not None:
            flat = [float(v) for pt in polygon for v in pt]
            self._check(lib.set_boundary_polygon(h, (C.c_float * len(flat))(*flat), len(polygon)))

    def close(self):
        if getattr(self, "handle", None):
            self.lib.destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int):
        if rc != 0:
            msg = self.lib.last_error(self.handle)
            raise SphError(rc, msg.decode(errors="replace") if msg else "")

    @property
    def n(self) -> int:
        return int(self.lib.num_particles(self.handle))

    @property
    def time(self) -> float:
        return float(self.lib.time(self.handle))

    def set_math_policy(self, policy):
        """sph_set_math_policy: "fast" (default) or "exact" (the reference's IEEE operations in its order; include/sph_ffi.h)."""
        self._check(self.lib.set_math_policy(self.handle, MATH_POLICIES[policy] if isinstance(policy, str) else int(policy)))

    def math_policy(self) -> str:
        return {v: k for k, v in MATH_POLICIES.items()}[int(self.lib.get_math_policy(self.handle))]

    def set_time(self, t: float, step_number: int = 0):
        self._check(self.lib.set_time(self.handle, float(t), int(step_number)))

    def upload(self, mass, position, velocity):
        mass = _as_f32(mass)
        n = mass.shape[0]
        position = _as_f32(position, (n, 2))
        velocity = _as_f32(velocity, (n, 2))
        self._check(self.lib.upload(self.handle, n, mass.ctypes.data, position.ctypes.data, velocity.ctypes.data))

    def upload_field(self, name: str, values):
        fid, dt, w = FIELDS[name]
        a = np.ascontiguousarray(values, dtype=dt)
        self._check(self.lib.upload_field(self.handle, fid, a.ctypes.data, a.nbytes))

    def apply_edits(self, ops) -> None:
        """Sparse edits between steps (sph_ffi.h): ops = [("set", i, {field: value, ...}) | ("swap", i, j) | ("truncate", n) |
        ("extend", k)], applied in order with Vec semantics in host index space."""
        arr = (SphEditOp * max(1, len(ops)))()
        for k, op in enumerate(ops):
            e = arr[k]
            if op[0] == "set":
                e.kind, e.a = EDIT_SET, int(op[1])
                for name, val in op[2].items():
                    e.fields |= EDIT_FIELD_BITS[name]
                    if name in ("position", "velocity"):
                        getattr(e, name)[0], getattr(e, name)[1] = float(val[0]), float(val[1])
                    else:
                        setattr(e, name, float(val))
            elif op[0] == "swap":
                e.kind, e.a, e.b = EDIT_SWAP, int(op[1]), int(op[2])
            elif op[0] == "truncate":
                e.kind, e.a = EDIT_TRUNCATE, int(op[1])
            elif op[0] == "extend":
                e.kind, e.a = EDIT_EXTEND, int(op[1])
            else:
                raise ValueError(op[0])
        self._check(self.lib.apply_edits(self.handle, arr, len(ops)))

    def download(self, name: str, host: "HostBuffers" = None) -> np.ndarray:
        """`host`: persistent host buffers to download into (the returned array is a VIEW, overwritten by the next download of the same
        field into the same buffers) -- a fresh numpy array costs a page fault per 4 KB the copy touches: 7.5 ms for configs[4]'s five
        decision fields against 1.75 ms into touched memory (profiles/r6_export_time.txt)."""
        fid, dt, w = FIELDS[name]
        n = self.n
        if host is None:
            out = np.empty((n, w) if w > 1 else (n,), dtype=dt)
        else:
            out = host.view("field:" + name, dt, n * w).reshape((n, w) if w > 1 else (n,))
        self._check(self.lib.download(self.handle, fid, out.ctypes.data, out.nbytes))
        return out

    def download_neighbors(self, host: "HostBuffers" = None):
        """CSR (offsets[n+1], indices) of the current neighbour lists, host particle order.  `host`: as in download() -- then ONE library
        call fills persistent buffers (a second one only when the lists outgrew them); without it the sizing call comes first."""
        n = self.n
        total = C.c_uint64(0)
        if host is None:
            offsets = np.empty(n + 1, dtype=np.uint32)
            self._check(self.lib.download_neighbors(self.handle, offsets.ctypes.data, None, 0, C.byref(total)))
            indices = np.empty(int(total.value), dtype=np.uint32)
            self._check(self.lib.download_neighbors(self.handle, offsets.ctypes.data, indices.ctypes.data,
                                                    indices.size, C.byref(total)))
            return offsets, indices
        offsets = host.view("csr:offsets", np.uint32, n + 1)
        cap = max(host.capacity("csr:indices", np.uint32), 16 * n)
        indices = host.view("csr:indices", np.uint32, cap)
        rc = self.lib.download_neighbors(self.handle, offsets.ctypes.data, indices.ctypes.data, indices.size, C.byref(total))
        if rc != 0 and int(total.value) > indices.size:   # the lists outgrew the buffer: the total is known now
            indices = host.view("csr:indices", np.uint32, int(total.value) + int(total.value) // 8)
            rc = self.lib.download_neighbors(self.handle, offsets.ctypes.data, indices.ctypes.data, indices.size, C.byref(total))
        self._check(rc)
        return offsets, indices[: int(total.value)]

    def step(self, params: SphParams) -> SphStepStats:
        st = SphStepStats()
        self._check(self.lib.step(self.handle, C.byref(params), C.byref(st)))
        return st

    def classify(self, params: SphParams) -> None:
        """classify_particles (adaptivity/mod.rs:50-59): the host's call, never part of the step."""
        self._check(self.lib.classify(self.handle, C.byref(params)))

    # ---- adaptivity data path: decisions on the host, data on the device (sph_ffi.h) ----
    def _partner_arrays(self, merge_partner, merge_counter):
        mp = np.ascontiguousarray(merge_partner, dtype=np.uint32)
        mc = np.ascontiguousarray(merge_counter, dtype=np.uint16)
        # (a slab context takes the arrays of the WHOLE vector, indexed by global particle id: its own count says nothing about their length)
        if mp.shape != mc.shape or mp.ndim != 1 or (not self.is_slab and mp.shape != (self.n,)):
            raise ValueError("merge_partner / merge_counter must have one entry per particle")
        return mp, mc

    def share_particles(self, params: SphParams, ap: "SphAdaptParams", merge_partner, merge_counter) -> None:
        mp, mc = self._partner_arrays(merge_partner, merge_counter)
        self._check(self.lib.share_particles(self.handle, C.byref(params), C.byref(ap), mp.ctypes.data, mc.ctypes.data))

    def merge_particles(self, params: SphParams, ap: "SphAdaptParams", merge_partner, merge_counter) -> None:
        mp, mc = self._partner_arrays(merge_partner, merge_counter)
        self._check(self.lib.merge_particles(self.handle, C.byref(params), C.byref(ap), mp.ctypes.data, mc.ctypes.data))

    def set_split_patterns(self, patterns) -> None:
        """patterns[k] = (k + 2, 2) array of child offsets pos_s (SplitPatterns, splitting.rs:84-120)."""
        for k, q in enumerate(patterns):
            if np.asarray(q).shape != (k + 2, 2):
                raise ValueError(f"assertion failed: sp.pos_s.len() == i + 2 (pattern {k})")
        flat = np.ascontiguousarray(np.concatenate([np.asarray(q, np.float32).reshape(-1, 2) for q in patterns]) if patterns
                                    else np.zeros((0, 2)), dtype=np.float32)
        self._check(self.lib.set_split_patterns(self.handle, len(patterns), flat.ctypes.data))

    def split_particles(self, params: SphParams, ap: "SphAdaptParams") -> None:
        self._check(self.lib.split_particles(self.handle, C.byref(params), C.byref(ap)))

    def grid(self) -> SphGridInfo:
        g = SphGridInfo()
        self._check(self.lib.grid(self.handle, C.byref(g)))
        return g

    # ---- measurement hooks (product only) ----
    def profile_enable(self, mode=True):
        """0 / False off; 1 / True marker events around every kernel; 3 the sweeps only, on the device's own clock (sph_ffi.h)"""
        self._check(self.lib.profile_enable(self.handle, int(mode)))

    def profile_reset(self):
        self._check(self.lib.profile_reset(self.handle))

    def profile_get(self) -> dict:
        cap = 64
        arr = (SphKernelTime * cap)()
        n = C.c_int(0)
        self._check(self.lib.profile_get(self.handle, arr, cap, C.byref(n)))
        return {arr[i].name.decode(): (int(arr[i].launches), float(arr[i].total_ms)) for i in range(n.value)}

    def profile_get_working(self) -> dict:
        """name -> (launches that did work, their total ms): speculative launches behind a stop decision excluded."""
        cap = 64
        arr = (SphKernelTime * cap)()
        n = C.c_int(0)
        self._check(self.lib.profile_get(self.handle, arr, cap, C.byref(n)))
        return {arr[i].name.decode(): (int(arr[i].working_launches), float(arr[i].working_ms)) for i in range(n.value)}

    def profile_copy_bandwidth_gbs(self, nbytes: int = 1 << 30) -> float:
        v = C.c_double(0.0)
        self._check(self.lib.profile_copy_bandwidth(self.handle, int(nbytes), C.byref(v)))
        return float(v.value)

    def profile_list_forms(self) -> dict:
        """How the last step recorded its neighbour lists: {"n_lists", "n_mask", "n_index", "n_walk", "n_wall"} (sph_list_forms)."""
        st = SphListForms()
        self._check(self.lib.profile_list_forms(self.handle, C.byref(st)))
        return {k: int(getattr(st, k)) for k, _ in SphListForms._fields_}

    def profile_event_overhead_us(self) -> float:
        v = C.c_double(0.0)
        self._check(self.lib.profile_event_overhead(self.handle, C.byref(v)))
        return float(v.value)

    def profile_dispatch_bracket_us(self, spin_us: int = 20, reps: int = 50) -> float:
        """mean dispatch-event bracket (microseconds) around a one-wave kernel that spins `spin_us` of the device clock"""
        v = C.c_double(0.0)
        self._check(self.lib.profile_dispatch_bracket(self.handle, int(spin_us), int(reps), C.byref(v)))
        return float(v.value)

    def dist_configure(self, rank: int, n_ranks: int, cut_lo: float, cut_hi: float):
        self._check(self.lib.dist_configure(self.handle, int(rank), int(n_ranks), float(cut_lo), float(cut_hi)))
        self.is_slab = n_ranks > 1

    def dist_set_rebalance(self, every_n_steps: int):
        self._check(self.lib.dist_set_rebalance(self.handle, int(every_n_steps)))

    def dist_get_cuts(self):
        """-> (cut_lo, cut_hi, number of times the cuts moved)"""
        lo, hi, k = C.c_float(), C.c_float(), C.c_uint32()
        self._check(self.lib.dist_get_cuts(self.handle, C.byref(lo), C.byref(hi), C.byref(k)))
        return lo.value, hi.value, k.value

    def dist_get_stats(self, reset: bool = False) -> dict:
        """Communication counters of this rank since the last reset (sph_dist_stats)."""
        st = SphDistStats()
        self._check(self.lib.dist_get_stats(self.handle, C.byref(st), 1 if reset else 0))
        return {"steps": int(st.steps), "exchanges": int(st.exchanges), "bytes_sent": int(st.bytes_sent), "bytes_received": int(st.bytes_received),
                "allreduces": int(st.allreduces), "host_waits": int(st.host_waits), "n_owned": int(st.n_owned),
                "n_halo": [int(st.n_halo[0]), int(st.n_halo[1])], "n_ghost": [int(st.n_ghost[0]), int(st.n_ghost[1])],
                "transport": TRANSPORT_NAMES.get(int(st.transport), str(int(st.transport))), "comm_ranks": int(st.comm_ranks)}

    def comm_init_threads(self, group, rank: int, n_ranks: int):
        """Thread transport (sph_ffi.h): `group` from SphLibrary.thread_group_create; every rank then steps on a thread of its own."""
        self._check(self.lib.comm_init_threads(self.handle, group, int(rank), int(n_ranks)))

    def comm_init_shm(self, name: str, rank: int, n_ranks: int, bytes_per_side: int, create: bool):
        """Shared-memory transport between processes of one node (sph_ffi.h): rank 0 creates, the others map afterwards."""
        self._check(self.lib.comm_init_shm(self.handle, name.encode(), int(rank), int(n_ranks), int(bytes_per_side), 1 if create else 0))

    def comm_ipc_export(self, bytes_per_side: int) -> bytes:
        """Peer-mapped push transport (sph_ffi.h): allocate this rank's box, return its 64-byte IPC handle."""
        buf = (C.c_uint8 * 64)()
        self._check(self.lib.comm_ipc_export(self.handle, int(bytes_per_side), buf))
        return bytes(buf)

    def comm_init_ipc(self, handles: bytes, n_ranks: int):
        buf = (C.c_uint8 * len(handles)).from_buffer_copy(handles)
        self._check(self.lib.comm_init_ipc(self.handle, buf, int(n_ranks)))

    def comm_init(self, unique_id: bytes, rank: int, n_ranks: int):
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        self._check(self.lib.comm_init(self.handle, buf, int(rank), int(n_ranks)))


def group_step(contexts, params: SphParams):
    """Step k slab contexts of this process as ranks 0..k-1 (loopback transport); returns their stats."""
    lib = contexts[0].lib
    k = len(contexts)
    handles = (C.c_void_p * k)(*[c.handle for c in contexts])
    stats = (SphStepStats * k)()
    rc = lib.group_step(handles, k, C.byref(params), stats)
    if rc != 0:
        msgs = [c.lib.last_error(c.handle) for c in contexts]
        raise SphError(rc, " | ".join(m.decode(errors="replace") for m in msgs if m))
    return list(stats)


def group_adapt(contexts, op: str, params: SphParams, ap, merge_partner=None, merge_counter=None):
    """share / merge / split on the k slab contexts of this process (sph_group_adapt): `merge_partner` / `merge_counter` are the arrays
    of the WHOLE vector, indexed by global particle id.  The contexts' particle counts change with merge and split."""
    lib = contexts[0].lib
    k = len(contexts)
    handles = (C.c_void_p * k)(*[c.handle for c in contexts])
    code = {"share": 0, "merge": 1, "split": 2}[op]
    mp = np.ascontiguousarray(merge_partner, np.uint32) if merge_partner is not None else None
    mc = np.ascontiguousarray(merge_counter, np.uint16) if merge_counter is not None else None
    rc = lib.group_adapt(handles, k, code, C.byref(params), C.byref(ap), mp.ctypes.data if mp is not None else None, mc.ctypes.data if mc is not None else None)
    if rc != 0:
        msgs = [c.lib.last_error(c.handle) for c in contexts]
        raise SphError(rc, " | ".join(m.decode(errors="replace") for m in msgs if m))
