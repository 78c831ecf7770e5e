#!/bin/bash
# round 5: compiler scheduling options on the P = 31 set kernel TU (tagged libraries _v2 max-ilp strategy, _v4 metric bias 0,
# _v5 AMDGPU register-pressure trackers, _v6 no post-RA scheduler), headline mode L
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for t in "" _v2 _v4 _v5 _v6; do
    GPV_LIB=$GRAFT_REPO_ROOT/gpvecchia_amd/libgpvecchia_hip$t.so python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('lib[$t]', 'evals/s %.1f' % j['value'], 'kernel ms %.4f' % j['roofline']['kernel_ms'], 'loglik', j['config']['loglik'])"
  done
done
