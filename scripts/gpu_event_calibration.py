#!/usr/bin/env python3
"""What the dispatch-event bracket (Profiler mode 3, hipExtLaunchKernelGGL) reads for kernels of known duration, to be compared
with rocprofv3's durations of the SAME launches:   rocprofv3 --kernel-trace ... -- python scripts/gpu_event_calibration.py
prints the bracket means; the trace holds k_spin_calib's durations."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from adaptive_sph_amd import ffi, scene as sc   # noqa: E402

plib = ffi.load_product()
scn = sc.dam_break_small(64, 64, 1 / 64)
pos, mass, vel = sc.init_particles(scn)
c = ffi.Context(plib, len(mass), sc.boundary_planes(scn.boundary))
c.upload(mass, pos, vel)
for us in (0, 5, 20, 40, 100):
    print(f"spin {us:4d} us: bracket {c.profile_dispatch_bracket_us(us, 50):8.3f} us", flush=True)
c.close()
