#!/bin/bash
# round 6: mode U and mode S of the headline workload, the sweep that reads the pivot row (shipped) against the old column read
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06u; mkdir -p $O
for rep in 1 2 3; do
  for lib in new oldsweep; do
    if [ $lib = new ]; then unset GPV_LIB; else export GPV_LIB=$GRAFT_REPO_ROOT/gpvecchia_amd/libgpvecchia_hip_$lib.so; fi
    for mode in L U S; do
    python bench.py --mode $mode --no-secondary --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python3 -c "
import sys, json
j = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$lib rep $rep mode $mode: evals/s %.1f kernel_ms %.4f' % (j['value'], j['roofline']['kernel_ms']))" | tee -a $O/ab.txt
    done
  done
done
