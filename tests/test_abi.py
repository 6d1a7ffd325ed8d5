"""The C-ABI library loads and exports every symbol include/sph_ffi.h declares (no GPU, no compute)."""
import ctypes as C
import re
from pathlib import Path

from adaptive_sph_amd import ffi

REPO = Path(__file__).resolve().parent.parent


def _declared_functions():
    text = (REPO / "include" / "sph_ffi.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sph_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_exported(product_lib):
    names = _declared_functions()
    assert len(names) >= 17
    for nm in names:
        assert hasattr(product_lib.lib, nm), f"libsph_hip.so does not export {nm}"
    assert sorted("sph_" + s for s in ffi.ABI_SYMBOLS) == names


def test_oracle_exports_same_surface(oracle_lib):
    for s in ffi.ABI_SYMBOLS:
        if s.startswith(("profile_", "comm_", "dist_", "group_", "set_sweep_", "host_", "thread_")):   # device / measurement entry points
            continue
        assert hasattr(oracle_lib.lib, "oracle_" + s)


def test_struct_layouts_match_header():
    # sizes implied by the header (natural alignment): guards the ctypes mirror against drift
    assert C.sizeof(ffi.SphParams) == 37 * 4
    assert C.sizeof(ffi.SphPlane) == 12
    assert C.sizeof(ffi.SphSolverStats) == 28
    assert C.sizeof(ffi.SphStepStats) == 8 + 8 + 8 + 28 + 28 + 5 * 8
    assert C.sizeof(ffi.SphGridInfo) == 20
    assert C.sizeof(ffi.SphKernelTime) == 80
    assert C.sizeof(ffi.SphEditOp) == 52
    assert C.sizeof(ffi.SphAdaptParams) == 11 * 4
    assert C.sizeof(ffi.SphDistStats) == 7 * 8 + 6 * 4


def test_no_gpu_create_fails_loudly(product_lib, gpu_available):
    """Without a device the product must report an error, never fall back to a CPU path."""
    if gpu_available:
        return
    h = C.c_void_p()
    arr = (ffi.SphPlane * 1)(ffi.SphPlane(1.0, 0.0, 1.0))
    rc = product_lib.create(16, 0, arr, 1, C.byref(h))
    assert rc == 2  # SPH_ERR_DEVICE


def test_product_package_does_not_reference_oracle():
    for f in (REPO / "adaptive_sph_amd").rglob("*"):
        if f.suffix in (".py", ".hip", ".h", ".hpp") and f.is_file():
            txt = f.read_text()
            assert "liboracle" not in txt and "oracle_harness" not in txt, f


def test_rust_shim_mirrors_the_header():
    """rust_shim/src/lib.rs cannot be compiled here (no cargo): keep its #[repr(C)] structs, field ids and extern block in
    step with include/sph_ffi.h textually."""
    import re
    root = Path(__file__).resolve().parent.parent
    header = (root / "include" / "sph_ffi.h").read_text()
    rust = (root / "rust_shim" / "src" / "lib.rs").read_text()

    def c_fields(struct):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (struct, struct), header, re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        names = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            names += [re.sub(r"\[.*\]", "", x.strip().split()[-1]) for x in decl.split(",")]
        return names

    def rust_fields(struct):
        body = re.search(r"pub struct %s \{(.*?)\n    \}" % struct, rust, re.S).group(1)
        return re.findall(r"pub (\w+):", body)

    for c_name, r_name in [("sph_params", "SphParams"), ("sph_plane", "SphPlane"), ("sph_solver_stats", "SphSolverStats"),
                           ("sph_step_stats", "SphStepStats"), ("sph_grid_info", "SphGridInfo"), ("sph_edit_op", "SphEditOp"),
                           ("sph_adapt_params", "SphAdaptParams")]:
        assert c_fields(c_name) == rust_fields(r_name), c_name
    # every SPH_F_* / enum constant the shim declares has the header's value
    for name, val in re.findall(r"pub const (SPH_\w+): (?:i32|u32|c_int) = (\d+);", rust):
        m = re.search(r"\b%s = (\d+)" % name, header)
        assert m and int(m.group(1)) == int(val), name
    # every function of the extern block is declared by the header
    for fn in re.findall(r"pub fn (sph_\w+)\(", rust):
        assert re.search(r"\b%s\(" % fn, header), fn
