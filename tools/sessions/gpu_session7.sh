#!/bin/bash
# tuning: record-prefetch position of the set kernel inside mode S; WPC thresholds of the posterior levels
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s7
run() { # label, env..., lib
  echo "== $1" >> gpurun_out/s7/log.txt
  shift
  env "$@" timeout 600 python tools/kbench.py --configs 30x2 --sgv --child 2>&1 | grep KBENCH >> gpurun_out/s7/log.txt
}
for rep in 1 2; do
run "default" GPV_LIB=$PWD/gpvecchia_amd/libgpvecchia_hip.so
run "pfrec P/4" GPV_LIB=$PWD/gpvecchia_amd/libgpvecchia_hip_pf4.so
run "pfrec 1" GPV_LIB=$PWD/gpvecchia_amd/libgpvecchia_hip_pf1.so
done
run "wide16=1024" GPV_POST_WIDE16=1024
run "wide16=256" GPV_POST_WIDE16=256
run "wide16=128" GPV_POST_WIDE16=128
run "wide8=4096" GPV_POST_WIDE=4096
run "wide8=1024 wide16=1024" GPV_POST_WIDE=1024 GPV_POST_WIDE16=1024
run "wide8=8192" GPV_POST_WIDE=8192
run "default" GPV_LIB=$PWD/gpvecchia_amd/libgpvecchia_hip.so
cat gpurun_out/s7/log.txt
