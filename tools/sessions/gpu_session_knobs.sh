#!/bin/bash
# the alternate routes behind the environment switches still pass the GPU suite
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/knobs
for kv in GPV_POST_NO_FUSE=1 GPV_NO_SEQ_HANDOFF=1 GPV_MATERN_TABLE_HOST=1 GPV_NO_GRAPH=1 GPV_POST_TOP=0; do
  env $kv timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/knobs/$kv.log 2>&1
  echo "$kv: $(tail -1 gpurun_out/knobs/$kv.log)"
done
