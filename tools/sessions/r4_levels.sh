#!/bin/bash
# round 4: the level schedule's shape (columns per level) at n = 1e6, m = 30, maxmin + SGV
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4lev; mkdir -p $O
GPV_POST_DEBUG=1 python bench.py --mode S --steps 2 --warmup 1 --no-cpu-baseline --clock-warmup-s 0 > $O/bench.json 2> $O/levels.txt
grep "gpv post" $O/levels.txt | tail -100
