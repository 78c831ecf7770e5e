#!/bin/bash
# full GPU suite, then the default bench line and the general-nu rate
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/full
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/full/tests.log 2>&1
tail -5 gpurun_out/full/tests.log
timeout 900 python bench.py > gpurun_out/full/bench.json 2> gpurun_out/full/bench.err
python3 - <<'PY'
import json
j=json.loads([l for l in open("gpurun_out/full/bench.json") if l.startswith("{")][-1])
print("value", j["value"], "ms", j["ms_per_step"], "kernel", j["roofline"]["kernel_ms"], "frac", j["roofline"]["frac"], "traffic", j["roofline"]["traffic"])
for k,v in j.get("secondary",{}).items():
    print(k, {a:(round(b,4) if isinstance(b,float) else b) for a,b in v.items() if a in ("value","ms_per_step","kernel_ms","sets_kernel_ms","frac","overhead_us","ms_per_nr_iter","vecchia_laplace_likelihood_s","ms_per_call","ms_per_call_fresh_outputs","error")})
PY
for nu in 1.1 0.3 2.2; do python bench.py --nu $nu --steps 40 --no-secondary --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('nu', j['config']['covparms'][2], 'evals/s %.1f' % j['value'], 'ms %.4f' % j['ms_per_step'], 'kernel %.4f' % j['roofline']['kernel_ms'], 'loglik', j['config']['loglik'])"; done
python tools/estimate_bench.py 2>&1 | tail -3
