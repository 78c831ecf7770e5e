#!/bin/bash
# round 4: prediction plans on the device; general-nu rounds A/B (default library against the tagged build _k)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4f; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_prediction.py -q -x > $O/pred.log 2>&1; tail -5 $O/pred.log
GPV_LIB=$PWD/gpvecchia_amd/libgpvecchia_hip_k.so timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "general_nu or table" > $O/gen_tests.log 2>&1; tail -5 $O/gen_tests.log
for rep in 1 2; do
for t in "" _k; do
  for nu in 1.1 0.3 2.2; do
    GPV_LIB=$PWD/gpvecchia_amd/libgpvecchia_hip$t.so python bench.py --nu $nu --steps 40 --no-secondary --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('lib$t nu', j['config']['covparms'][2], 'evals/s %.1f' % j['value'], 'ms %.4f' % j['ms_per_step'], 'kernel %.4f' % j['roofline']['kernel_ms'], 'loglik %.9f' % j['config']['loglik'])"
  done
done
done
