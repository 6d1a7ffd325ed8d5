#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 5 120 python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
timeout -k 5 900 python bench.py --gpus 2 --steps 5 --warmup 2 --profile-steps 2 > gpurun_out/bench_gpus2.json 2> gpurun_out/bench_gpus2.err; echo "rc=$?"; tail -5 gpurun_out/bench_gpus2.err
python -c "
import json;d=json.load(open('gpurun_out/bench_gpus2.json'));print(d['value'], d['ms_per_step'], d['config']['parallelism']); print(json.dumps(d['config4_ratio_stress_4m'])[:1500]); print(json.dumps(d['strong_8m'])[:600])"
