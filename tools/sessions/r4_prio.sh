#!/bin/bash
# round 4: per-task wave priority: timeline and A/B (tagged builds _k = with, _noprio = without; both row lengths 21, 31 only)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4h; mkdir -p $O
export GPV_LIB=$PWD/gpvecchia_amd/libgpvecchia_hip_trace.so
python tools/wave_timeline.py --m 30 --rows 125000 2>&1 | grep -v amdgpu.ids | tee $O/timeline_m30_125k.txt
python tools/wave_timeline.py --m 30 --rows 1000000 2>&1 | grep -v amdgpu.ids | tee $O/timeline_m30_1e6.txt
python tools/wave_timeline.py --m 20 --n 100000 --rows 100000 2>&1 | grep -v amdgpu.ids | tee $O/timeline_m20_1e5.txt
for rep in 1 2; do
for t in _noprio _k; do
  export GPV_LIB=$PWD/gpvecchia_amd/libgpvecchia_hip$t.so
  echo "== lib$t"
  python tools/short_launch.py --m 30 --d 2 --sizes 62500,125000,250000,1000000 --iters 200 2>&1 | grep -v amdgpu.ids
  python tools/short_launch.py --m 20 --d 2 --sizes 50000,100000,400000 --iters 200 2>&1 | grep -v amdgpu.ids
  python tools/short_launch.py --m 30 --d 2 --sizes 125000,1000000 --nu 1.1 --iters 100 2>&1 | grep -v amdgpu.ids
done
done 2>&1 | tee $O/ab.txt
