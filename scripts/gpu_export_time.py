"""Where do the ~26 ms of configs[4]'s adaptivity export go?  The CSR lists and the five fields a partner search
reads, device -> host at 4 M particles: fresh numpy arrays (what adaptivity.py did through round 5) against persistent ones against
persistent ones registered with hipHostRegister (pinned).  usage: python scripts/gpu_export_time.py [workload=ratio_stress_4m]"""
import ctypes as C
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from adaptive_sph_amd import ffi, scene as sc  # noqa: E402
from adaptive_sph_amd.workloads import WORKLOADS  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "ratio_stress_4m"
    scene_f, params_f, _ = WORKLOADS[name]
    scn, P = scene_f(), params_f(level_estimation_method="EmptyAngle")
    pos, mass, vel = sc.init_particles(scn)
    lib = ffi.load_product()
    hip = C.CDLL("libamdhip64.so")
    ctx = ffi.Context(lib, len(mass), sc.boundary_planes(scn.boundary, P.init_boundary_handler))
    ctx.upload(mass, pos, vel)
    for _ in range(3):
        ctx.step(P.to_ffi())
    n = ctx.n

    def t(f, reps=3):
        best = 1e9
        for _ in range(reps):
            t0 = time.perf_counter()
            f()
            best = min(best, time.perf_counter() - t0)
        return best * 1e3

    print(f"{name}: {n} particles")
    print(f"download_neighbors as adaptivity.py calls it (two calls, fresh arrays): {t(ctx.download_neighbors):.2f} ms")
    off, idx = ctx.download_neighbors()
    tot = C.c_uint64(0)
    print(f"  ... {idx.nbytes / 1e6:.0f} MB of indices + {off.nbytes / 1e6:.0f} MB of offsets")
    print(f"  the sizing call alone (counts down, prefix on the host): {t(lambda: lib.download_neighbors(ctx.handle, off.ctypes.data, None, 0, C.byref(tot))):.2f} ms")
    print(f"  ONE call into persistent (touched) arrays: {t(lambda: lib.download_neighbors(ctx.handle, off.ctypes.data, idx.ctypes.data, idx.size, C.byref(tot))):.2f} ms")
    for a in (off, idx):
        rc = hip.hipHostRegister(C.c_void_p(a.ctypes.data), C.c_size_t(a.nbytes), C.c_uint(0))
        assert rc == 0, rc
    print(f"  ONE call into persistent REGISTERED (pinned) arrays: {t(lambda: lib.download_neighbors(ctx.handle, off.ctypes.data, idx.ctypes.data, idx.size, C.byref(tot))):.2f} ms")
    fields = ("particle_size_class", "mass", "level_estimation", "position", "h2")
    print(f"the five fields, fresh arrays: {t(lambda: [ctx.download(f) for f in fields]):.2f} ms")
    bufs = {f: ctx.download(f) for f in fields}

    def into():
        for f in fields:
            fid = ffi.FIELDS[f][0]
            lib.download(ctx.handle, fid, bufs[f].ctypes.data, bufs[f].nbytes)
    print(f"the five fields, persistent arrays: {t(into):.2f} ms")
    for a in bufs.values():
        assert hip.hipHostRegister(C.c_void_p(a.ctypes.data), C.c_size_t(a.nbytes), C.c_uint(0)) == 0
    print(f"the five fields, persistent REGISTERED arrays: {t(into):.2f} ms ({sum(a.nbytes for a in bufs.values()) / 1e6:.0f} MB)")
    t0 = time.perf_counter()
    big = np.empty(idx.size, np.uint32)
    rc = hip.hipHostRegister(C.c_void_p(big.ctypes.data), C.c_size_t(big.nbytes), C.c_uint(0))
    print(f"hipHostRegister of a fresh {big.nbytes / 1e6:.0f} MB array: {(time.perf_counter() - t0) * 1e3:.2f} ms (rc {rc})")


if __name__ == "__main__":
    main()
