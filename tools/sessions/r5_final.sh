#!/bin/bash
# round 5, final tree: full GPU suite, the default bench line (what the driver runs), mode S trace + level summary
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05f
GPV_PARITY_LOG=$GRAFT_REPO_ROOT/gpurun_out/r05f/parity_counts.jsonl timeout 3000 python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/r05f/gpu_tests.txt 2>&1
tail -12 gpurun_out/r05f/gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
( time python bench.py ) > gpurun_out/r05f/bench.json 2> gpurun_out/r05f/bench.err; tail -4 gpurun_out/r05f/bench.err
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/r05f/trace
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r05f/trace -- python3 bench.py --mode S --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r05f/benchS.json 2> gpurun_out/r05f/errS.log
python3 tools/sgv_levels.py gpurun_out/r05f/trace > gpurun_out/r05f/levels.txt 2>&1; tail -3 gpurun_out/r05f/levels.txt
bash tools/sessions/pmc_post.sh gpurun_out/r05f/pmc 2>&1 | grep "all levels"
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import json
j = json.loads([l for l in open('gpurun_out/r05f/bench.json') if l.startswith('{')][-1])
print('value', j['value'], 'frac', j['roofline']['frac'], 'traffic', j['roofline']['traffic'], 'from_idle', j['config']['from_idle']['value'])
s = j['secondary']
for k in ('mode_U', 'mode_S', 'mode_S_mean', 'mode_L_maxmin', 'C2', 'C4', 'dropin_U_D2H', 'per_rank_step'):
    v = s.get(k, {})
    print(k, {kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in v.items() if kk in ('value', 'ms_per_step', 'kernel_ms', 'fp64_frac', 'sets_kernel_ms', 'ms_per_call', 'kernel_ms_back_to_back', 'error')})
print('C5', json.dumps(s.get('C5_vl'))[:260])
print('mode_S parity', json.dumps(s['mode_S'].get('parity_in_run'))[:420])
print('parity_in_run', json.dumps(j.get('parity_in_run'))[:300])
PY
