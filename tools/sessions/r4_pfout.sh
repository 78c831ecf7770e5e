#!/bin/bash
# round 4: output addresses prefetched (tagged build _k) against the default library: mode S, mode U, mode L
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4r; mkdir -p $O
for rep in 1 2; do
for t in _nopf _k; do
  export GPV_LIB=$PWD/gpvecchia_amd/libgpvecchia_hip$t.so
  python bench.py --mode S --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('lib$t mode S', 'evals/s %.1f' % j['value'], 'ms %.4f' % j['ms_per_step'], 'set kernel %.4f' % j['roofline']['kernel_ms'], 'loglik %.9f' % j['config']['loglik'])"
  python bench.py --mode U --steps 20 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('lib$t mode U', 'evals/s %.1f' % j['value'], 'ms %.4f' % j['ms_per_step'], 'set kernel %.4f' % j['roofline']['kernel_ms'], 'loglik %.9f' % j['config']['loglik'])"
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('lib$t mode L', 'evals/s %.1f' % j['value'], 'ms %.4f' % j['ms_per_step'], 'set kernel %.4f' % j['roofline']['kernel_ms'], 'loglik %.9f' % j['config']['loglik'])"
done
done 2>&1 | tee $O/ab.txt
