#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s10
for mult in 4 1 2 3 8 4 1 2; do
echo "== mult $mult" >> gpurun_out/s10/log.txt
GPV_GRID_MULT=$mult timeout 600 python tools/kbench.py --configs 20x2,30x2 --n 100000 --child 2>&1 | grep KBENCH >> gpurun_out/s10/log.txt
done
for mult in 4 1 2 8; do
echo "== 1e6 mult $mult" >> gpurun_out/s10/log.txt
GPV_GRID_MULT=$mult timeout 600 python tools/kbench.py --configs 30x2,60x3 --child 2>&1 | grep KBENCH >> gpurun_out/s10/log.txt
done
cat gpurun_out/s10/log.txt
