#!/bin/bash
mkdir -p gpurun_out
COPIES=${COPIES:-426} VARIANTS="${VARIANTS:-1}" timeout 400 python tools/s1_experiment.py > gpurun_out/s1_experiment_j.log 2>&1
tail -3 gpurun_out/s1_experiment_j.log | cut -c1-1200
