#!/bin/bash
# round-2 profiles: stage-1 bench trace + PMC, whole-parse trace + PMC, stage-1 timelines
timeout 900 tools/profile_round.sh r02c > gpurun_out/r02c_round.log 2>&1
timeout 900 tools/profile_parse_r2.sh r02c_parse > gpurun_out/r02c_parse.log 2>&1
COPIES=426 VARIANTS="1 4" timeout 300 python tools/s1_experiment.py > gpurun_out/s1_experiment_r02c.log 2>&1
ls gpurun_out/r02c gpurun_out/r02c_parse | head -40
tail -3 gpurun_out/r02c_round.log gpurun_out/r02c_parse.log
