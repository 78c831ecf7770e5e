#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s9
one() { echo "== $*" >> gpurun_out/s9/log.txt; env "$@" timeout 600 python bench.py --mode S --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'], j['roofline']['kernel_ms'])" >> gpurun_out/s9/log.txt; }
one A=1
one GPV_POST_T16=3.0 GPV_POST_T32=10
one GPV_POST_T16=6.5 GPV_POST_T32=10
one GPV_POST_T16=4.5 GPV_POST_T32=7
one GPV_POST_T16=4.5 GPV_POST_T32=14
one GPV_POST_T16=4.5 GPV_POST_T32=20
one GPV_POST_T16=8 GPV_POST_T32=20
one GPV_POST_T16=0 GPV_POST_T32=12
one A=1
cat gpurun_out/s9/log.txt
