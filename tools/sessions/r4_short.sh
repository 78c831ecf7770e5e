#!/bin/bash
# round 4: anatomy of the short launch (fixed cost of the set kernel, workgroup placement)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4b; mkdir -p $O
./tools/ubench/placement 512 80 20 > $O/placement_512_80.txt 2>&1
./tools/ubench/placement 512 44 20 > $O/placement_512_44.txt 2>&1
./tools/ubench/placement 2048 80 20 > $O/placement_2048_80.txt 2>&1
head -60 $O/placement_512_80.txt; tail -8 $O/placement_512_80.txt; tail -5 $O/placement_512_44.txt; tail -5 $O/placement_2048_80.txt
for gm in 0 1 2 4; do
  if [ $gm = 0 ]; then unset GPV_GRID_MULT; else export GPV_GRID_MULT=$gm; fi
  python tools/short_launch.py --m 30 --d 2 2>&1 | tee -a $O/short_m30.txt
  python tools/short_launch.py --m 20 --d 2 --sizes 25000,50000,100000,200000,400000 2>&1 | tee -a $O/short_m20.txt
done
