#!/bin/bash
# full GPU suite + bench line
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 > gpurun_out/n_tests.txt
cat gpurun_out/n_tests.txt
timeout 600 python bench.py > gpurun_out/n_bench.json 2> gpurun_out/n_bench.err
tail -c 600 gpurun_out/n_bench.err
head -c 1500 gpurun_out/n_bench.json
