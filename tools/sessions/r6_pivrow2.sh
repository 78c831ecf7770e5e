#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06pr; mkdir -p $O
for lib in base pivrow; do
  if [ $lib = base ]; then unset GPV_LIB; else export GPV_LIB=$GRAFT_REPO_ROOT/gpvecchia_amd/libgpvecchia_hip_$lib.so; fi
  echo "=== $lib" | tee -a $O/accuracy2.txt
  for m in 7 10 13 30 60; do python tools/accuracy_mat_probe.py $m 2>&1 | grep -v amdgpu.ids | tee -a $O/accuracy2.txt; done
done
for rep in 1 2; do
  for lib in base pivrow; do
    if [ $lib = base ]; then unset GPV_LIB; else export GPV_LIB=$GRAFT_REPO_ROOT/gpvecchia_amd/libgpvecchia_hip_$lib.so; fi
    python bench.py --config C4 --no-secondary --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python3 -c "
import sys, json
j = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$lib rep $rep C4: evals/s %.2f kernel_ms %.4f' % (j['value'], j['roofline']['kernel_ms']))" | tee -a $O/ab2.txt
    python bench.py --n 1000000 --m 10 --no-secondary --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python3 -c "
import sys, json
j = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$lib rep $rep m=10: evals/s %.2f kernel_ms %.4f' % (j['value'], j['roofline']['kernel_ms']))" | tee -a $O/ab2.txt
  done
done
