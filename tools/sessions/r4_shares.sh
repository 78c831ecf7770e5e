#!/bin/bash
# round 4: task shares by dispatch order: which ratio for which instantiation (GPV_SHARES / GPV_NO_UNEVEN, same library)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4j; mkdir -p $O
export GPV_LIB=$PWD/gpvecchia_amd/libgpvecchia_hip_k.so
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "fused_reduction or full_size or general_nu" > $O/tests.log 2>&1; tail -3 $O/tests.log
for sh in off 2,1 3,2 3,1; do
  if [ $sh = off ]; then export GPV_NO_UNEVEN=1; unset GPV_SHARES; else unset GPV_NO_UNEVEN; export GPV_SHARES=$sh; fi
  echo "== shares $sh"
  python tools/short_launch.py --m 30 --d 2 --sizes 31250,62500,125000,250000 --iters 150 2>&1 | grep -v "amdgpu.ids\|Rank\|polyfit"
  python tools/short_launch.py --m 20 --d 2 --sizes 25000,50000,100000,200000 --iters 150 2>&1 | grep -v "amdgpu.ids\|Rank\|polyfit"
  python tools/short_launch.py --m 30 --d 2 --sizes 62500,125000,250000 --nu 1.1 --iters 100 2>&1 | grep -v "amdgpu.ids\|Rank\|polyfit"
done 2>&1 | tee $O/ab.txt
