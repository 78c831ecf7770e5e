#!/bin/bash
# round 4: thresholds between the level kernel forms (columns per level): 16 waves per column up to WIDE16, 8 up to WIDE
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for w in "512 2048" "256 2048" "1024 2048" "512 1024" "512 4096" "1024 4096" "2048 2048"; do
  set -- $w
  GPV_POST_WIDE16=$1 GPV_POST_WIDE=$2 python bench.py --mode S --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('WIDE16=$1 WIDE=$2', 'evals/s %.1f' % j['value'], 'ms %.4f' % j['ms_per_step'], 'set kernel %.4f' % j['roofline']['kernel_ms'], 'post %.4f' % (j['ms_per_step']-j['roofline']['kernel_ms']))"
done; done
