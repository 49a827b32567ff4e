#!/bin/bash
# round-2 profiles: stage-1 bench trace + PMC, whole-parse trace + PMC, stage-1 timelines
timeout 900 tools/profile_round.sh r02b > gpurun_out/r02b_round.log 2>&1
timeout 900 tools/profile_parse_r2.sh r02b_parse > gpurun_out/r02b_parse.log 2>&1
COPIES=426 VARIANTS="1 4" timeout 300 python tools/s1_experiment.py > gpurun_out/s1_experiment_r02b.log 2>&1
ls gpurun_out/r02b gpurun_out/r02b_parse | head -40
tail -3 gpurun_out/r02b_round.log gpurun_out/r02b_parse.log
