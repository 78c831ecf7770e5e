#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4k; mkdir -p $O
export GPV_LIB=$PWD/gpvecchia_amd/libgpvecchia_hip_k.so
for cfg in "off:0" "2,1:1" "off:1" "2,1:0"; do
  sh=${cfg%%:*}; gm=${cfg#*:}
  if [ $sh = off ]; then export GPV_NO_UNEVEN=1; unset GPV_SHARES; else unset GPV_NO_UNEVEN; export GPV_SHARES=$sh; fi
  if [ $gm = 0 ]; then unset GPV_GRID_MULT; else export GPV_GRID_MULT=$gm; fi
  echo "== shares $sh grid_mult $gm"
  python tools/short_launch.py --m 30 --d 2 --sizes 500000,1000000 --iters 150 2>&1 | grep -v "amdgpu.ids\|Rank\|polyfit\|^fit"
  python tools/short_launch.py --m 20 --d 2 --sizes 400000,1000000 --iters 150 2>&1 | grep -v "amdgpu.ids\|Rank\|polyfit\|^fit"
  python tools/short_launch.py --m 30 --d 2 --sizes 1000000 --nu 1.1 --iters 100 2>&1 | grep -v "amdgpu.ids\|Rank\|polyfit\|^fit"
done 2>&1 | tee $O/ab.txt
export GPV_SHARES=3,2 GPV_GRID_MULT=1; unset GPV_NO_UNEVEN
echo "== shares 3,2 grid_mult 1"
python tools/short_launch.py --m 30 --d 2 --sizes 1000000 --nu 1.1 --iters 100 2>&1 | grep -v "amdgpu.ids\|Rank\|polyfit\|^fit"
