#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4x; mkdir -p $O
./tools/ubench/placement 1024 36 30 1 > $O/placement_1024_36_w1.txt 2>&1; tail -12 $O/placement_1024_36_w1.txt
./tools/ubench/placement 2048 36 30 1 > $O/placement_2048_36_w1.txt 2>&1; tail -8 $O/placement_2048_36_w1.txt
export GPV_LIB=$PWD/gpvecchia_amd/libgpvecchia_hip.so
