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


def build_hip(force: bool = False, verbose: bool = False) -> Path:
    """hipcc --offload-arch=gfx950 -> adaptive_sph_amd/csrc/libsph_hip.so"""
    out = CSRC / "libsph_hip.so"
    srcs = sorted(CSRC.glob("*.hip"))
    deps = srcs + sorted(CSRC.glob("*.h")) + sorted(CSRC.glob("*.hpp")) + [REPO / "include" / "sph_ffi.h"]
    if not force and not _stale(out, deps):
        return out
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-ffp-contract=off", "-fno-fast-math", "-fgpu-rdc=0" if False else "-Wall",
           "-I", str(REPO / "include"), "-o", str(out)] + os.environ.get("SPH_EXTRA_HIPCC_FLAGS", "").split() + [str(s) for s in srcs] + ["-L/opt/rocm/lib", "-lrccl"]
    cmd = [c for c in cmd if c]
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + r.stdout + r.stderr)
    return out
