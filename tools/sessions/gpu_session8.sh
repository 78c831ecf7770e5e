#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s8
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/s8/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/s8/pytest.log
grep -n "FAILED\|passed\|failed\|rc=\|^E  " gpurun_out/s8/pytest.log | tail -20
for rep in 1 2; do
for lpc in 64 0; do
echo "== GPV_POST_LPC=$lpc" >> gpurun_out/s8/log.txt
if [ $lpc = 0 ]; then timeout 600 python bench.py --mode S --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'], j['roofline']['kernel_ms'], j['config']['loglik'])" >> gpurun_out/s8/log.txt
else GPV_POST_LPC=$lpc timeout 600 python bench.py --mode S --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'], j['roofline']['kernel_ms'], j['config']['loglik'])" >> gpurun_out/s8/log.txt
fi
done
done
cat gpurun_out/s8/log.txt
