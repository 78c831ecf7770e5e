#!/bin/bash
# round 5: the posterior fuzz against the oracle, long rows, the a-vector store left out of the fused deposit (mode S A/B)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5i; mkdir -p $O
timeout 1800 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_prediction.py tests/test_gpu_parity.py -m gpu -x -q -k "fuzz or random or long_rows or masked or sgv or SGV or posterior or denominator or laplace or zy" > $O/tests1.txt 2>&1
tail -6 $O/tests1.txt
for rep in 1 2 3; do
  for t in _base ""; do
    GPV_LIB=$GRAFT_REPO_ROOT/gpvecchia_amd/libgpvecchia_hip$t.so python bench.py --mode S --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('lib[$t]', 'evals/s %.1f' % j['value'], 'ms %.4f' % j['ms_per_step'], 'set kernel %.4f' % j['roofline']['kernel_ms'], 'loglik', j['config']['loglik'])"
  done
done
