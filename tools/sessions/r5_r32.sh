#!/bin/bash
# round 5: rounds of 32 columns (2 lanes per column) for the one-wavefront-per-column levels whose mean row list is >= 18:
# posterior tests, A/B against libgpvecchia_hip_prev.so, level times
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5n; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_posterior_oracle.py tests/test_gpu_parity.py tests/test_golden.py -m gpu -x -q -k "(posterior or sgv or top or random_plans or golden) and not C5" > $O/tests1.txt 2>&1
tail -3 $O/tests1.txt
for rep in 1 2 3 4; do
  for t in _prev ""; do
    GPV_LIB=$GRAFT_REPO_ROOT/gpvecchia_amd/libgpvecchia_hip$t.so python bench.py --mode S --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('lib[$t]', 'evals/s %.1f' % j['value'], 'ms %.4f' % j['ms_per_step'], 'set kernel %.4f' % j['roofline']['kernel_ms'], 'loglik', j['config']['loglik'])"
  done
done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for t in _prev ""; do
  rm -rf $O/trace$t
  GPV_LIB=$GRAFT_REPO_ROOT/gpvecchia_amd/libgpvecchia_hip$t.so timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace$t -- python3 bench.py --mode S --steps 10 --warmup 3 --no-cpu-baseline > $O/bench$t.json 2> $O/err$t.log
  python3 tools/sgv_levels.py $O/trace$t > $O/levels$t.txt 2>&1
  echo "== levels lib[$t]"; tail -3 $O/levels$t.txt
done
