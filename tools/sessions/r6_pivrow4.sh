#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06pr; mkdir -p $O
for rep in 1 2; do
  for lib in base pivrow dpp4; do
    if [ $lib = base ]; then unset GPV_LIB; else export GPV_LIB=$GRAFT_REPO_ROOT/gpvecchia_amd/libgpvecchia_hip_$lib.so; fi
    for m in 7 3; do
    python bench.py --n 1000000 --m $m --no-secondary --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python3 -c "
import sys, json
j = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$lib rep $rep m=$m: evals/s %.2f kernel_ms %.4f loglik %.10f' % (j['value'], j['roofline']['kernel_ms'], j['config']['loglik']))" | tee -a $O/ab4.txt
    done
  done
done
