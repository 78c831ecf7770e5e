#!/bin/bash
# round 4: wave timeline of short launches; general-nu NOLIVE A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4g; mkdir -p $O
export GPV_LIB=$PWD/gpvecchia_amd/libgpvecchia_hip_trace.so
python tools/wave_timeline.py --m 30 --rows 125000 2>&1 | grep -v amdgpu.ids | tee $O/timeline_m30_125k.txt
python tools/wave_timeline.py --m 30 --rows 1000000 2>&1 | grep -v amdgpu.ids | tee $O/timeline_m30_1e6.txt
python tools/wave_timeline.py --m 20 --n 100000 --rows 100000 2>&1 | grep -v amdgpu.ids | tee $O/timeline_m20_1e5.txt
export GPV_LIB=$PWD/gpvecchia_amd/libgpvecchia_hip_k.so
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "general_nu or table or coincident" > $O/gen_tests.log 2>&1; tail -3 $O/gen_tests.log
unset GPV_LIB
for rep in 1 2; do
for t in "" _k; do
  for nu in 1.1 0.3; do
    GPV_LIB=$PWD/gpvecchia_amd/libgpvecchia_hip$t.so python bench.py --nu $nu --steps 40 --no-secondary --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('lib$t nu', j['config']['covparms'][2], 'evals/s %.1f' % j['value'], 'ms %.4f' % j['ms_per_step'], 'kernel %.4f' % j['roofline']['kernel_ms'], 'loglik %.9f' % j['config']['loglik'])"
  done
done
done
