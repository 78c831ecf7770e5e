#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s5
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/s5/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/s5/pytest.log
grep -n "FAILED\|passed\|failed\|rc=" gpurun_out/s5/pytest.log | tail -8
for i in 1 2; do
GPV_POST_ZST=0 timeout 600 python tools/kbench.py --configs 30x2 --sgv --child >> gpurun_out/s5/kbench_zst0.log 2>&1
timeout 600 python tools/kbench.py --configs 30x2 --sgv --child >> gpurun_out/s5/kbench_zst1.log 2>&1
done
grep KBENCH gpurun_out/s5/kbench_zst0.log gpurun_out/s5/kbench_zst1.log
timeout 900 python bench.py > gpurun_out/s5/bench.json 2> gpurun_out/s5/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
j=json.loads(open('gpurun_out/s5/bench.json').read().strip().splitlines()[-1])
print('value',j['value'],'ms',j['ms_per_step'],'kernel',j['roofline']['kernel_ms'],'frac',j['roofline']['frac'])
for k,v in j['secondary'].items():
    print(k, {a:b for a,b in v.items() if a not in ('what',)})
PY
tail -3 gpurun_out/s5/bench.err
