#!/bin/bash
# round 6, final tree: large fuzz sweeps on fresh seeds
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06fz; mkdir -p $O
timeout 1500 python tools/fuzz_posterior.py 10000 11500 --oracle 2>&1 | grep -v amdgpu.ids > $O/posterior.txt; tail -4 $O/posterior.txt
timeout 1200 python tools/fuzz_more.py 424 1024 2>&1 | grep -v amdgpu.ids > $O/sets.txt; tail -12 $O/sets.txt
timeout 1500 python tools/fuzz_vl.py 260 460 2>&1 | grep -v amdgpu.ids > $O/vl.txt; tail -12 $O/vl.txt
timeout 1200 python tools/fuzz_prediction.py 200 500 2>&1 | grep -v amdgpu.ids > $O/pred.txt; tail -4 $O/pred.txt
