#!/bin/bash
mkdir -p gpurun_out
COPIES=426 VARIANTS="1 4" timeout 400 python tools/s1_experiment.py > gpurun_out/s1_experiment_j.log 2>&1
tail -5 gpurun_out/s1_experiment_j.log | cut -c1-400
