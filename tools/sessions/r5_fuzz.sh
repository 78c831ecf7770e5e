#!/bin/bash
# round 5: the long fuzz sweeps on the final tree (posterior pass: 600 random plans against the host route; set kernel: 400
# random shapes against the oracle)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5j
timeout 1500 python tools/fuzz_posterior.py 0 600 > gpurun_out/r5j/fuzz_posterior.txt 2>&1; tail -4 gpurun_out/r5j/fuzz_posterior.txt
timeout 1500 python tools/fuzz_more.py > gpurun_out/r5j/fuzz_more.txt 2>&1; tail -3 gpurun_out/r5j/fuzz_more.txt
