#!/bin/bash
# round 5: lanes per column of the early levels (developer build: a level takes 16 lanes per column up to a mean row list of
# GPV_POST_T16, 32 up to GPV_POST_T32), re-swept with the record prefetch in the kernels
cd $GRAFT_REPO_ROOT
run() { # tag, env
  env $2 GPV_LIB=$GRAFT_REPO_ROOT/gpvecchia_amd/libgpvecchia_hip_dev.so python bench.py --mode S --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', 'evals/s %.1f' % j['value'], 'ms %.4f' % j['ms_per_step'])"
}
for rep in 1 2; do
  run default X=1
  for a in 3.5 6 8; do for b in 8 10 14 20; do run "t16=$a,t32=$b" "GPV_POST_T16=$a GPV_POST_T32=$b"; done; done
done
