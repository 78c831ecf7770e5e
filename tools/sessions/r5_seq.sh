#!/bin/bash
# round 5: posterior totals published by sequence number (gpv_plan_get_sums spins instead of sleeping in the stream wait):
# mode S and mode_S_mean A/B against libgpvecchia_hip_prev.so (neither this nor the record prefetch), full GPU suite
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5l; mkdir -p $O
for rep in 1 2 3 4; do
  for t in _prev ""; do
    GPV_LIB=$GRAFT_REPO_ROOT/gpvecchia_amd/libgpvecchia_hip$t.so python bench.py --mode S --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('lib[$t]', 'evals/s %.1f' % j['value'], 'ms %.4f' % j['ms_per_step'], 'set kernel %.4f' % j['roofline']['kernel_ms'], 'loglik', j['config']['loglik'])"
  done
done
timeout 3000 python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; tail -4 $O/gpu_tests.txt
