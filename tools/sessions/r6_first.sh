#!/bin/bash
# round 6, first contact: the new GPU tests (VL loop vs sparse oracle, 8 ranks on one GPU), then the whole suite, then the bench line
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06a; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_posterior_oracle.py tests/test_gpu_bench_nranks.py -m gpu -x -q -s --durations=8 -k "vecchia_laplace_loop or eight_ranks" > $O/new_tests.txt 2>&1
tail -15 $O/new_tests.txt
GPV_PARITY_LOG=$GRAFT_REPO_ROOT/$O/parity_counts.jsonl timeout 3000 python -m pytest tests -m gpu -x -q --durations=10 > $O/gpu_tests.txt 2>&1
tail -16 $O/gpu_tests.txt
( time python bench.py ) > $O/bench.json 2> $O/bench.err; tail -4 $O/bench.err
python3 - <<'PY'
import json
j = json.loads([l for l in open('gpurun_out/r06a/bench.json') if l.startswith('{')][-1])
print('value', j['value'], 'frac', j['roofline']['frac'], 'from_idle', j['config']['from_idle']['value'])
print('cpu_baseline', json.dumps(j.get('cpu_baseline'))[:1500])
print('C5', json.dumps(j['secondary'].get('C5_vl'))[:1200])
PY
