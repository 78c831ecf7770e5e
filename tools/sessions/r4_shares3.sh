#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4l; mkdir -p $O
export GPV_LIB=$PWD/gpvecchia_amd/libgpvecchia_hip_k.so
for sh in default 2,1 7,4 5,3 9,5 3,2; do
  if [ $sh = default ]; then unset GPV_SHARES; else export GPV_SHARES=$sh; fi
  echo "== shares $sh"
  python tools/short_launch.py --m 30 --d 2 --sizes 125000,1000000 --iters 150 2>&1 | grep -v "amdgpu.ids\|Rank\|polyfit\|^fit"
  python tools/short_launch.py --m 20 --d 2 --sizes 50000,100000 --iters 150 2>&1 | grep -v "amdgpu.ids\|Rank\|polyfit\|^fit"
  python tools/short_launch.py --m 30 --d 2 --sizes 125000,1000000 --nu 1.1 --iters 100 2>&1 | grep -v "amdgpu.ids\|Rank\|polyfit\|^fit"
done 2>&1 | tee $O/ab.txt
