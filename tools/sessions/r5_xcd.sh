#!/bin/bash
# round 5: XCD-aware column mapping of the posterior levels: tests, same-box A/B against the round's first library, level
# times, level counters (L2 hits / misses, FETCH_SIZE), drop-in timing, C2 geometry A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5d; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_bench_nranks.py tests/test_gpu_fuzz.py tests/test_gpu_posterior_oracle.py -m gpu -q -k "not C5 and not 1e6" > $O/tests1.txt 2>&1
tail -5 $O/tests1.txt
for rep in 1 2 3; do
  for t in _base ""; do
    GPV_LIB=$GRAFT_REPO_ROOT/gpvecchia_amd/libgpvecchia_hip$t.so python bench.py --mode S --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('lib[$t]', 'evals/s %.1f' % j['value'], 'ms %.4f' % j['ms_per_step'], 'set kernel %.4f' % j['roofline']['kernel_ms'], 'loglik', j['config']['loglik'])"
  done
done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for t in _base ""; do
  rm -rf $O/trace$t
  GPV_LIB=$GRAFT_REPO_ROOT/gpvecchia_amd/libgpvecchia_hip$t.so timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace$t -- python3 bench.py --mode S --steps 10 --warmup 3 --no-cpu-baseline > $O/bench$t.json 2> $O/err$t.log
  python3 tools/sgv_levels.py $O/trace$t > $O/levels$t.txt 2>&1
  echo "== levels lib[$t]"; tail -3 $O/levels$t.txt
done
for t in _base ""; do
  echo "== level counters lib[$t]"
  GPV_LIB=$GRAFT_REPO_ROOT/gpvecchia_amd/libgpvecchia_hip$t.so bash tools/sessions/pmc_post.sh $O/pmc$t 2>&1 | tail -6
done
cd $GRAFT_REPO_ROOT
GPV_TIMING=1 python3 tools/dropin_timing.py > $O/dropin.txt 2>&1; grep "call\|hash" $O/dropin.txt | tail -8
for n in 100000 125000; do
  python3 tools/kbench.py --n $n --configs 20x2 --iters 30 gpvecchia_amd/libgpvecchia_hip.so gpvecchia_amd/libgpvecchia_hip_c2lds.so gpvecchia_amd/libgpvecchia_hip.so gpvecchia_amd/libgpvecchia_hip_c2lds.so 2>&1 | tee -a $O/c2_ab.txt
done
