#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s11
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/s11/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/s11/pytest.log
grep -n "FAILED\|passed\|failed\|rc=\|^E  " gpurun_out/s11/pytest.log | tail -12
for nu in 1.1 0.3 2.2; do
timeout 600 python bench.py --nu $nu --steps 20 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('nu', j['config']['covparms'][2], j['value'], j['ms_per_step'], j['roofline']['kernel_ms'])"
done
GPV_NO_MATERN_TABLE=1 timeout 600 python bench.py --nu 1.1 --steps 5 --warmup 1 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no-table nu', j['config']['covparms'][2], j['value'], j['ms_per_step'], j['roofline']['kernel_ms'])"
timeout 600 python tools/estimate_bench.py 2>&1 | tail -3
