#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06u; mkdir -p $O
for rep in 1 2 3 4; do
  for lib in new nopin oldsweep; do
    if [ $lib = new ]; then unset GPV_LIB; else export GPV_LIB=$GRAFT_REPO_ROOT/gpvecchia_amd/libgpvecchia_hip_$lib.so; fi
    python bench.py --mode L --no-secondary --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python3 -c "
import sys, json
j = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$lib rep $rep mode L: evals/s %.1f kernel_ms %.4f loglik %.12f' % (j['value'], j['roofline']['kernel_ms'], j['config']['loglik']))" | tee -a $O/ab_nopin.txt
  done
done
