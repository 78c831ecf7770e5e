#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s3
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/s3/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/s3/pytest.log
grep -n "FAILED\|passed\|failed\|rc=" gpurun_out/s3/pytest.log | tail -8
for i in 1 2; do
timeout 600 python tools/kbench.py --configs 30x2 gpvecchia_amd/libgpvecchia_hip_base.so gpvecchia_amd/libgpvecchia_hip_nofix.so gpvecchia_amd/libgpvecchia_hip_nofreeze.so gpvecchia_amd/libgpvecchia_hip.so >> gpurun_out/s3/kbench.log 2>&1
done
timeout 600 python tools/kbench.py --configs 20x2,60x3,10x2 gpvecchia_amd/libgpvecchia_hip_base.so gpvecchia_amd/libgpvecchia_hip.so >> gpurun_out/s3/kbench.log 2>&1
cat gpurun_out/s3/kbench.log
timeout 900 python bench.py > gpurun_out/s3/bench.json 2> gpurun_out/s3/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
j=json.loads(open('gpurun_out/s3/bench.json').read().strip().splitlines()[-1])
print('value',j['value'],'ms',j['ms_per_step'],'kernel',j['roofline']['kernel_ms'],'frac',j['roofline']['frac'])
for k,v in j['secondary'].items():
    print(k, {a:b for a,b in v.items() if a not in ('what',)})
PY
tail -3 gpurun_out/s3/bench.err
