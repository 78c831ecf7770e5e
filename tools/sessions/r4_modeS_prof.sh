#!/bin/bash
# round 4: posterior-pass tests, then the mode S trace / level summary / bench line that profiles/r04_modeS_* are copied from
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_prediction.py -q -x -m gpu 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/r04S; mkdir -p gpurun_out/r04S
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04S/trace -o run -- python3 bench.py --mode S --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r04S/bench_traced.json 2> gpurun_out/r04S/err.log
python3 tools/sgv_levels.py gpurun_out/r04S/trace > gpurun_out/r04S/levels.txt 2>&1
tail -3 gpurun_out/r04S/levels.txt
python bench.py --mode S --steps 30 --warmup 3 --no-cpu-baseline > gpurun_out/r04S/bench.json 2>/dev/null
for i in 1 2; do python bench.py --mode S --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('mode S', 'evals/s %.1f' % j['value'], 'ms %.4f' % j['ms_per_step'], 'set kernel %.4f' % j['roofline']['kernel_ms'])"; done
python3 -c "
import json
j=json.loads([l for l in open('gpurun_out/r04S/bench.json') if l.startswith('{')][-1]); print('mode S', j['value'], j['ms_per_step'])"
