#!/bin/bash
SJHIP_LIB=$PWD/build_ab/libsjhip_poll.so timeout 900 python -m pytest tests/test_gpu_stage1.py tests/test_gpu_parse.py tests/test_gpu_quirks.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -2
tools/gpu_ab_parse.sh 2>&1 | grep build_ab
