#!/bin/bash
# round 5, first contact: the new parity tests (posterior pass against the sparse oracle at 6e4 / 5e5 / 1e6; bench.py's
# two-rank flow on one GPU), then the default bench line with the mode S oracle leg
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5a
timeout 2400 python -m pytest tests/test_gpu_posterior_oracle.py tests/test_gpu_bench_nranks.py -m gpu -x -q --durations=10 > gpurun_out/r5a/tests.txt 2>&1
tail -25 gpurun_out/r5a/tests.txt
timeout 900 python bench.py > gpurun_out/r5a/bench.json 2> gpurun_out/r5a/bench.err
tail -c 1500 gpurun_out/r5a/bench.err
python3 - <<'PY'
import json
j = json.loads([l for l in open('gpurun_out/r5a/bench.json') if l.startswith('{')][-1])
print('value', j['value'], 'frac', j['roofline']['frac'])
print(json.dumps(j['secondary'].get('mode_S'), indent=1))
for k in ('C2', 'C4', 'dropin_U_D2H', 'mode_S_mean'):
    print(k, json.dumps(j['secondary'].get(k))[:400])
PY
