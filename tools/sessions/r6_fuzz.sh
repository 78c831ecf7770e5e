#!/bin/bash
# round 6: fuzz sweeps on fresh seeds (VL loop against the sparse oracle: new; posterior pass against host route + oracle; set kernel)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06fuzz; mkdir -p $O
timeout 1500 python tools/fuzz_vl.py 0 150 > $O/vl.txt 2>&1; tail -12 $O/vl.txt
timeout 1200 python tools/fuzz_posterior.py 9000 9400 --oracle > $O/posterior.txt 2>&1; tail -5 $O/posterior.txt
timeout 900 python tools/fuzz_more.py 9000 9300 > $O/sets.txt 2>&1; tail -5 $O/sets.txt
