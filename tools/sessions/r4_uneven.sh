#!/bin/bash
# round 4: task shares by dispatch order (GPV_NO_UNEVEN=1 switches it off in the same library)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4i; mkdir -p $O
export GPV_LIB=$PWD/gpvecchia_amd/libgpvecchia_hip_trace.so
python tools/wave_timeline.py --m 30 --rows 125000 2>&1 | grep -v amdgpu.ids | tee $O/timeline_m30_125k.txt
python tools/wave_timeline.py --m 20 --n 100000 --rows 100000 2>&1 | grep -v amdgpu.ids | tee $O/timeline_m20_1e5.txt
export GPV_LIB=$PWD/gpvecchia_amd/libgpvecchia_hip_k.so
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "fused_reduction or full_size or general_nu" > $O/tests.log 2>&1; tail -3 $O/tests.log
for rep in 1 2; do
for e in 1 0; do
  if [ $e = 1 ]; then export GPV_NO_UNEVEN=1; else unset GPV_NO_UNEVEN; fi
  echo "== GPV_NO_UNEVEN=$e"
  python tools/short_launch.py --m 30 --d 2 --sizes 31250,62500,125000,250000 --iters 200 2>&1 | grep -v amdgpu.ids
  python tools/short_launch.py --m 20 --d 2 --sizes 25000,50000,100000,200000 --iters 200 2>&1 | grep -v amdgpu.ids
  python tools/short_launch.py --m 30 --d 2 --sizes 125000 --nu 1.1 --iters 100 2>&1 | grep -v amdgpu.ids
done
done 2>&1 | tee $O/ab.txt
