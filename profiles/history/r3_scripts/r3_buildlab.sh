#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/lab
for j in 0.0 0.15 0.3; do timeout -k 5 120 ./scripts/ubench/bin/build_lab 1024 $j 50 2>&1 | tee -a gpurun_out/lab/build_lab.txt; done
