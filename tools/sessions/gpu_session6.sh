#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s6
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/s6/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/s6/pytest.log
grep -n "FAILED\|passed\|failed\|rc=\|^E  " gpurun_out/s6/pytest.log | tail -20
timeout 900 python bench.py > gpurun_out/s6/bench.json 2> gpurun_out/s6/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
j=json.loads(open('gpurun_out/s6/bench.json').read().strip().splitlines()[-1])
print('value',j['value'],'ms',j['ms_per_step'],'kernel',j['roofline']['kernel_ms'],'frac',j['roofline']['frac'])
for k,v in j['secondary'].items():
    print(k, {a:b for a,b in v.items() if a not in ('what',)})
PY
tail -3 gpurun_out/s6/bench.err
