#!/bin/bash
# round 5: whole-pair stores in the posterior levels (A/B against the round's first library), block alignment 64 / 128 bytes
# with the developer build
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_posterior_oracle.py tests/test_gpu_prediction.py -m gpu -q -k "not C5 and not 1e6" > $O/tests1.txt 2>&1
tail -3 $O/tests1.txt
run() { # tag, lib suffix, env
  env $3 GPV_LIB=$GRAFT_REPO_ROOT/gpvecchia_amd/libgpvecchia_hip$2.so python bench.py --mode S --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', 'evals/s %.1f' % j['value'], 'ms %.4f' % j['ms_per_step'], 'set kernel %.4f' % j['roofline']['kernel_ms'], 'loglik', j['config']['loglik'])"
}
for rep in 1 2 3; do
  run base _base X=1
  run new "" X=1
  run dev_align64 _dev GPV_POST_ALIGN=4
  run dev_align128 _dev GPV_POST_ALIGN=8
done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for t in _base ""; do
  rm -rf $O/trace$t
  GPV_LIB=$GRAFT_REPO_ROOT/gpvecchia_amd/libgpvecchia_hip$t.so timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace$t -- python3 bench.py --mode S --steps 10 --warmup 3 --no-cpu-baseline > $O/bench$t.json 2> $O/err$t.log
  python3 tools/sgv_levels.py $O/trace$t > $O/levels$t.txt 2>&1
  echo "== levels lib[$t]"; tail -3 $O/levels$t.txt
done
GPV_POST_ALIGN=8 GPV_LIB=$GRAFT_REPO_ROOT/gpvecchia_amd/libgpvecchia_hip_dev.so timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_a128 -- python3 bench.py --mode S --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_a128.json 2> $O/err_a128.log
python3 tools/sgv_levels.py $O/trace_a128 > $O/levels_a128.txt 2>&1; echo "== levels align128"; tail -3 $O/levels_a128.txt
