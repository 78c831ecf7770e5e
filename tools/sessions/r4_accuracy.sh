#!/bin/bash
# round 4: which arithmetic shortcut of the set kernel costs accuracy on ill-conditioned blocks (tools/accuracy_probe.py)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c; mkdir -p $O
for t in "" _nopre _norcp3 _nofreeze _nosqrt; do
  GPV_LIB=$PWD/gpvecchia_amd/libgpvecchia_hip$t.so timeout 900 python tools/accuracy_probe.py > $O/acc$t.txt 2>&1
  cat $O/acc$t.txt
done
bash tools/sessions/r4_short.sh
