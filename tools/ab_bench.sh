#!/bin/bash
# A/B of tagged library builds through bench.py: tools/ab_bench.sh "<bench args>" tagA tagB ...   (alternating, 2 rounds)
cd $GRAFT_REPO_ROOT
ARGS="$1"; shift
for rep in 1 2; do
  for t in "$@"; do
    GPV_LIB=$GRAFT_REPO_ROOT/gpvecchia_amd/libgpvecchia_hip$t.so python bench.py $ARGS --no-secondary --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$t', 'evals/s %.1f' % j['value'], 'ms %.4f' % j['ms_per_step'], 'kernel %.4f' % j['roofline']['kernel_ms'], 'loglik', j['config']['loglik'])"
  done
done
