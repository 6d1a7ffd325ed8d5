"""Build helpers: the HIP library (hipcc, gfx950, in-tree) -- product code; nothing here touches oracle/."""
from __future__ import annotations

import os
import shutil
import subprocess
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
CSRC = PKG_DIR / "csrc"
REPO = PKG_DIR.parent


def hipcc_path() -> str:
    p = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(p):
        raise RuntimeError("hipcc not found")
    return p


def _stale(target: Path, sources) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(Path(s).stat().st_mtime > t for s in sources)


def build_lab(force: bool = False, verbose: bool = False) -> Path:
    """The LABORATORY build of the same sources: -DSPH_LAB compiles in the ablation switches (sph_context.hpp: Options) that put older or
    alternative forms of a kernel / a queueing policy beside the product's.  The bit-identity tests load it (tests/conftest.py: lab_lib),
    scripts/variants and scripts/gpu_*.py select it with SPH_HIP_LIBRARY=libsph_lab.so.  Never the measured or shipped path."""
    return build_hip(force=force, verbose=verbose, out_name="libsph_lab.so", extra_flags=["-DSPH_LAB=1"])


def build_hip(force: bool = False, verbose: bool = False, out_name: str = "libsph_hip.so", extra_flags=()) -> Path:
    """hipcc --offload-arch=gfx950 -> adaptive_sph_amd/csrc/libsph_hip.so

    Each translation unit is compiled to an object of its own (in parallel; only the stale ones), then linked: a change to
    one kernel file costs that file's compile time, not the library's.  SPH_EXTRA_HIPCC_FLAGS (variants, scripts/variants)
    goes to every compile; objects are kept under csrc/build/<hash of the flags>."""
    import hashlib
    from concurrent.futures import ThreadPoolExecutor
    out = CSRC / out_name   # (variants are built HERE, next to the product library, and selected with SPH_HIP_LIBRARY: ffi.py)
    srcs = sorted(CSRC.glob("*.hip"))
    headers = sorted(CSRC.glob("*.h")) + sorted(CSRC.glob("*.hpp")) + [REPO / "include" / "sph_ffi.h"]
    extra = os.environ.get("SPH_EXTRA_HIPCC_FLAGS", "").split() + list(extra_flags)
    # -fno-slp-vectorize: the SLP vectoriser turns pairs of scalar f32 operations into v_pk_* instructions and pays for each with register
    # moves (73 v_mov in sweep B's hot path); without it the same arithmetic is 1-5 % faster per sweep (profiles/r5_variants.md section 3)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize", "-Wall",
             "-I", str(REPO / "include")] + extra
    tag = hashlib.sha1(" ".join(flags).encode()).hexdigest()[:10]
    objdir = CSRC / "build" / tag
    objdir.mkdir(parents=True, exist_ok=True)
    stamp = objdir / ("linked-" + out_name)   # which flag set that .so was linked from
    objs = [objdir / (s.stem + ".o") for s in srcs]
    todo = [(s, o) for s, o in zip(srcs, objs) if force or _stale(o, [s] + headers)]
    if not todo and out.exists() and stamp.exists() and not _stale(out, objs) and not _stale(stamp, [out]):
        return out

    def compile_one(so):
        s, o = so
        cmd = [hipcc_path()] + flags + ["-c", str(s), "-o", str(o)]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {s.name}:\n" + r.stdout + r.stderr)

    with ThreadPoolExecutor(max_workers=min(len(todo), os.cpu_count() or 4) or 1) as ex:
        list(ex.map(compile_one, todo))
    # -Bsymbolic: the library's references to its own global symbols bind to ITS definitions -- two builds of these sources in one process
    # (the product and the laboratory build, tests/conftest.py) must not interpose each other's functions
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,-Bsymbolic", "-o", str(out)] + [str(o) for o in objs] + ["-L/opt/rocm/lib", "-lrccl"]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stdout + r.stderr)
    for other in (CSRC / "build").glob("*/linked-" + out_name):
        other.unlink()
    stamp.write_text("")
    return out
