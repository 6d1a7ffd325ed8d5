#!/bin/bash
# the Jacobi laboratory on the GPU box: index check first, then timings on the lattice and a jittered scene
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/lab
timeout -k 5 120 ./scripts/ubench/bin/jacobi_lab_check 512 0.15 2 > gpurun_out/lab/check.txt 2>&1; echo "check rc=$?"
tail -3 gpurun_out/lab/check.txt
timeout -k 5 150 ./scripts/ubench/bin/jacobi_lab 1024 0.0 50 > gpurun_out/lab/lattice.txt 2>&1; echo "lattice rc=$?"
timeout -k 5 150 ./scripts/ubench/bin/jacobi_lab 1024 0.15 50 > gpurun_out/lab/jitter.txt 2>&1; echo "jitter rc=$?"
cat gpurun_out/lab/lattice.txt gpurun_out/lab/jitter.txt
