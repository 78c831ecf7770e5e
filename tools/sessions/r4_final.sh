#!/bin/bash
# round 4, after the two-block dense top: the full GPU suite, the default bench line, and the mode S trace / level summary
cd $GRAFT_REPO_ROOT
bash tools/sessions/r4_full.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04S; rm -rf gpurun_out/r04S/trace
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04S/trace -- python3 bench.py --mode S --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r04S/bench.json 2> gpurun_out/r04S/err.log
python3 tools/sgv_levels.py gpurun_out/r04S/trace > gpurun_out/r04S/levels.txt 2>&1
tail -4 gpurun_out/r04S/levels.txt
for i in 1 2 3; do python bench.py --mode S --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('mode S', 'evals/s %.1f' % j['value'], 'ms %.4f' % j['ms_per_step'], 'set kernel %.4f' % j['roofline']['kernel_ms'])"; done
