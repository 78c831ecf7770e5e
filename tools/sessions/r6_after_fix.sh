#!/bin/bash
# round 6, after the pivot-row fix: the whole GPU suite, the fuzz sweeps again (VL loop, posterior pass, set kernel), the bench line
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06fix; mkdir -p $O
GPV_PARITY_LOG=$GRAFT_REPO_ROOT/$O/parity_counts.jsonl timeout 3000 python -m pytest tests -m gpu -x -q --durations=8 > $O/gpu_tests.txt 2>&1
tail -14 $O/gpu_tests.txt
timeout 1500 python tools/fuzz_vl.py 0 150 > $O/vl.txt 2>&1; tail -8 $O/vl.txt
timeout 1200 python tools/fuzz_posterior.py 9000 9400 --oracle > $O/posterior.txt 2>&1; tail -4 $O/posterior.txt
timeout 900 python tools/fuzz_more.py 24 424 > $O/sets.txt 2>&1; tail -7 $O/sets.txt
( time python bench.py ) > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err
python3 - <<'PY'
import json
j = json.loads([l for l in open('gpurun_out/r06fix/bench.json') if l.startswith('{')][-1])
print('value', j['value'], 'frac', j['roofline']['frac'], 'kernel_ms', j['roofline']['kernel_ms'], 'traffic', j['roofline']['traffic'])
s = j['secondary']
for k in ('mode_U', 'mode_S', 'mode_S_mean', 'mode_L_maxmin', 'C2', 'C4', 'dropin_U_D2H', 'per_rank_step'):
    v = s.get(k, {})
    print(k, {kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in v.items() if kk in ('value', 'ms_per_step', 'kernel_ms', 'fp64_frac', 'sets_kernel_ms', 'ms_per_call', 'kernel_ms_back_to_back', 'error')})
print('passroof', json.dumps(s['mode_S'].get('pass_roofline'))[:300])
print('modeS parity', json.dumps({k: v for k, v in s['mode_S']['parity_in_run'].items() if k != 'what'}))
print('C5', json.dumps({k: v for k, v in s['C5_vl'].items() if k not in ('what',)})[:900])
print('parity_in_run', json.dumps({k: v for k, v in j['parity_in_run'].items() if k != 'what'}))
PY
