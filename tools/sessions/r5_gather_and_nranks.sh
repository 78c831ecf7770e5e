#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_bench_nranks.py -m gpu -q > gpurun_out/nranks_tests.txt 2>&1; tail -4 gpurun_out/nranks_tests.txt
bash tools/sessions/r5_gather.sh
