#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/fuse
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/fuse/tests.log 2>&1
tail -5 gpurun_out/fuse/tests.log
for f in 0 1; do
  if [ $f = 1 ]; then export GPV_POST_NO_FUSE=1; else unset GPV_POST_NO_FUSE; fi
  for rep in 1 2; do
  python bench.py --mode S --steps 20 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('no_fuse=$f', 'evals/s %.1f' % j['value'], 'ms %.4f' % j['ms_per_step'], 'sets kernel %.4f' % j['roofline']['kernel_ms'], 'loglik', j['config']['loglik'])"
  done
done
unset GPV_POST_NO_FUSE
python bench.py --steps 20 --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('mode L', 'evals/s %.1f' % j['value'], 'ms %.4f' % j['ms_per_step'], 'kernel %.4f' % j['roofline']['kernel_ms'])"
