#!/bin/bash
# A/B of several builds of the library on ONE box, alternating: scripts/ab_libs.sh <out prefix> <runs> <libA.so> <libB.so> [libC.so ...]
# (the libraries live in adaptive_sph_amd/csrc/, selected by SPH_HIP_LIBRARY: ffi.py); summary: per build ms/step, iterations, per-kernel us
P=$1; N=$2; shift 2
cd $GRAFT_REPO_ROOT
for i in $(seq 1 $N); do
  for L in "$@"; do
    SPH_HIP_LIBRARY=$L python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline --no-8m > gpurun_out/${P}_${L%.so}_$i.json 2> gpurun_out/${P}_${L%.so}_$i.err || echo "run $i of $L failed"
  done
done
python - "$P" "$@" <<PY
import glob, json, sys
P = sys.argv[1]
for lib in [l[:-3] for l in sys.argv[2:]]:
    rows = [json.load(open(f)) for f in sorted(glob.glob(f"gpurun_out/{P}_{lib}_*.json"))]
    print(lib, "ms/step", [round(r["ms_per_step"], 4) for r in rows], "iterations", [(round(r["config"]["mean_div_iterations"], 2), round(r["config"]["mean_density_iterations"], 2)) for r in rows])
    for name in ("jacobi_update", "pressure_accel", "source_term", "density", "aii_nonpressure", "solver_tail"):
        print("   ", name, [round(k.get("avg_us_instrumented", k["avg_us"]), 2) for r in rows for k in r["kernels"] if k["name"] == name])
PY
