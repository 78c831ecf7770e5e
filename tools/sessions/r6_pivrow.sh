#!/bin/bash
# round 6: the pivot row read from the pivot row itself (GPV_OPT_PIVROW) against the shipped column-by-symmetry read:
# backward errors, posterior means against extended precision, Newton iteration counts, and the headline kernel's time
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06pr; mkdir -p $O
for lib in base pivrow; do
  if [ $lib = base ]; then unset GPV_LIB; else export GPV_LIB=$GRAFT_REPO_ROOT/gpvecchia_amd/libgpvecchia_hip_$lib.so; fi
  echo "=== $lib" | tee -a $O/accuracy.txt
  python tools/accuracy_mat_probe.py 2>&1 | grep -v amdgpu.ids | tee -a $O/accuracy.txt
  python tools/accuracy_rows_probe.py 2>&1 | grep -v amdgpu.ids | tee -a $O/accuracy.txt
  python tools/accuracy_posterior_probe.py post:9298 vl:142 vl:102 vl:124 2>&1 | grep -v amdgpu.ids | tee -a $O/accuracy.txt
done
for rep in 1 2 3; do
  for lib in base pivrow; do
    if [ $lib = base ]; then unset GPV_LIB; else export GPV_LIB=$GRAFT_REPO_ROOT/gpvecchia_amd/libgpvecchia_hip_$lib.so; fi
    python bench.py --no-secondary --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python3 -c "
import sys, json
j = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$lib rep $rep: evals/s %.1f kernel_ms %.4f loglik %.12f' % (j['value'], j['roofline']['kernel_ms'], j['config']['loglik']))" | tee -a $O/ab.txt
  done
done
