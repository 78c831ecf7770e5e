#!/bin/bash
# round 5: which levels get 8 / 16 wavefronts per column (developer build: GPV_POST_WIDE / GPV_POST_WIDE16), mode S
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5f; mkdir -p $O
run() { # tag, env
  env $2 GPV_LIB=$GRAFT_REPO_ROOT/gpvecchia_amd/libgpvecchia_hip_dev.so python bench.py --mode S --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', 'evals/s %.1f' % j['value'], 'ms %.4f' % j['ms_per_step'], 'set kernel %.4f' % j['roofline']['kernel_ms'], 'loglik', j['config']['loglik'])"
}
for rep in 1 2; do
  run default X=1
  for w in 4096 8192 16384 32768; do run wide8_$w GPV_POST_WIDE=$w; done
  run wide16_1024 GPV_POST_WIDE16=1024
  run wide16_2048_wide8_8192 "GPV_POST_WIDE16=2048 GPV_POST_WIDE=8192"
  run wide16_4096_wide8_16384 "GPV_POST_WIDE16=4096 GPV_POST_WIDE=16384"
done
