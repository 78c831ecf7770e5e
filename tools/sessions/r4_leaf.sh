#!/bin/bash
# round 4: leaf columns of the posterior schedule finished by the set kernel (tagged build _k; GPV_POST_NO_LEAF=1 switches it off)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4w; mkdir -p $O
export GPV_LIB=$PWD/gpvecchia_amd/libgpvecchia_hip_k.so
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -q -k "posterior or sgv or SGV or denom or dense_top or laplace or C5 or C2_full or vecchia_likelihood or mean" > $O/tests.log 2>&1; tail -5 $O/tests.log
for rep in 1 2; do
for e in 1 0; do
  if [ $e = 1 ]; then export GPV_POST_NO_LEAF=1; else unset GPV_POST_NO_LEAF; fi
  python bench.py --mode S --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('NO_LEAF=$e mode S', 'evals/s %.1f' % j['value'], 'ms %.4f' % j['ms_per_step'], 'set kernel %.4f' % j['roofline']['kernel_ms'], 'loglik %.12f' % j['config']['loglik'])"
done
done 2>&1 | tee $O/ab.txt
