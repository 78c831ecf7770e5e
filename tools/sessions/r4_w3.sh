#!/bin/bash
# round 4: three wavefronts per SIMD at P = 21 (tagged build _w3, 168 VGPRs, three workgroups per CU) against the default
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4p; mkdir -p $O
(rocm-smi --showclocks --showpower 2>&1 | head -30) > $O/box.txt
for rep in 1 2; do
for t in _k _w3; do
  export GPV_LIB=$PWD/gpvecchia_amd/libgpvecchia_hip$t.so
  echo "== lib$t"
  python tools/short_launch.py --m 20 --d 2 --sizes 25000,50000,100000,200000,400000 --iters 150 2>&1 | grep -v "amdgpu.ids\|Rank\|polyfit"
done
done 2>&1 | tee $O/ab.txt
export GPV_LIB=$PWD/gpvecchia_amd/libgpvecchia_hip_w3.so
timeout 600 python -m pytest tests/test_gpu_configs.py -q -x -k "C2" > $O/tests.log 2>&1; tail -3 $O/tests.log
