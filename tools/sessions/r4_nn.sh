#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4u; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_configs.py -q -x -k "grid_nn or C2_full" --durations=5 > $O/tests.log 2>&1; tail -12 $O/tests.log
for d in 2 3; do
  python tools/nnbench2.py --n 1000000 --d $d 2>&1 | grep -v amdgpu
  GPV_NN_BRUTE=1 python tools/nnbench2.py --n 1000000 --d $d 2>&1 | grep -v amdgpu
done | tee $O/nn.txt
python tools/nnbench2.py --n 4000000 --d 2 2>&1 | grep -v amdgpu | tee -a $O/nn.txt
