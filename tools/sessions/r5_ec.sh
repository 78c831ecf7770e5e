#!/bin/bash
# round 5: entries per lane and round of the posterior level kernels (GPV_POST_EC, compile time; 4 in the tree)
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for t in "" _ec2 _ec3 _ec6; do
    GPV_LIB=$GRAFT_REPO_ROOT/gpvecchia_amd/libgpvecchia_hip$t.so python bench.py --mode S --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('lib[$t]', 'evals/s %.1f' % j['value'], 'ms %.4f' % j['ms_per_step'], 'loglik', j['config']['loglik'])"
  done
done
