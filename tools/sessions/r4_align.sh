#!/bin/bash
# round 4: compact (B, R) blocks aligned to 1 / 4 / 8 entries (16 / 64 / 128 bytes): set kernel in mode S, whole evaluation
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for al in 1 4 8; do
  GPV_POST_ALIGN=$al python bench.py --mode S --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('ALIGN=$al', 'evals/s %.1f' % j['value'], 'ms %.4f' % j['ms_per_step'], 'set kernel %.4f' % j['roofline']['kernel_ms'])"
done; done
