#!/bin/bash
# round 5: pipelined rounds in the one-wavefront-per-column levels (developer build: GPV_POST_PIPE = mean row list from which a
# level takes the pipelined form; 1e9 = never), mode S, tests first
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5h; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_posterior_oracle.py tests/test_gpu_parity.py -m gpu -q -k "(posterior or sgv or top or fuzz) and not C5 and not 1e6" > $O/tests1.txt 2>&1
tail -3 $O/tests1.txt
run() { # tag, env
  env $2 GPV_LIB=$GRAFT_REPO_ROOT/gpvecchia_amd/libgpvecchia_hip_dev.so python bench.py --mode S --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', 'evals/s %.1f' % j['value'], 'ms %.4f' % j['ms_per_step'], 'set kernel %.4f' % j['roofline']['kernel_ms'], 'loglik', j['config']['loglik'])"
}
for rep in 1 2 3; do
  run pipe_off GPV_POST_PIPE=1e9
  run pipe_20 GPV_POST_PIPE=20
  run pipe_14 GPV_POST_PIPE=14
  run pipe_32 GPV_POST_PIPE=32
done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for p in 1e9 20; do
  rm -rf $O/trace_$p
  GPV_POST_PIPE=$p GPV_LIB=$GRAFT_REPO_ROOT/gpvecchia_amd/libgpvecchia_hip_dev.so timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$p -- python3 bench.py --mode S --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_$p.json 2> $O/err_$p.log
  python3 tools/sgv_levels.py $O/trace_$p > $O/levels_$p.txt 2>&1
  echo "== levels pipe from $p"; tail -3 $O/levels_$p.txt
done
