#!/bin/bash
# round 4: the full GPU suite (caps enforced), then the default bench line
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4e; mkdir -p $O; rm -f $O/parity.jsonl
GPV_PARITY_LOG=$PWD/$O/parity.jsonl timeout 2400 python -m pytest tests -m gpu -q > $O/tests.log 2>&1
tail -15 $O/tests.log
timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err
python3 - <<'PY'
import json
j=json.loads([l for l in open("gpurun_out/r4e/bench.json") if l.startswith("{")][-1])
print("value", j["value"], "ms", j["ms_per_step"], "kernel", j["roofline"]["kernel_ms"], "frac", j["roofline"]["frac"], "from_idle", j["config"].get("from_idle",{}).get("value"))
for k,v in j.get("secondary",{}).items():
    print(k, {a:(round(b,4) if isinstance(b,float) else b) for a,b in v.items() if a in ("value","ms_per_step","kernel_ms","sets_kernel_ms","fp64_frac","overhead_us","ms_per_nr_iter","vecchia_laplace_likelihood_s","ms_per_call","error")})
print("parity", {k:v for k,v in j.get("parity_in_run",{}).items() if k!="what"})
PY
for nu in 1.1 0.3; do python bench.py --nu $nu --steps 40 --no-secondary --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('nu', j['config']['covparms'][2], 'evals/s %.1f' % j['value'], 'ms %.4f' % j['ms_per_step'], 'kernel %.4f' % j['roofline']['kernel_ms'], 'loglik', j['config']['loglik'])"; done
