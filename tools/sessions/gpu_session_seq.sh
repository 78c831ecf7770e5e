#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/seq
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/seq/tests.log 2>&1
tail -5 gpurun_out/seq/tests.log
python tools/comm_diag.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/seq/comm_diag.txt
timeout 900 python bench.py > gpurun_out/seq/bench.json 2> gpurun_out/seq/bench.err
python3 - <<'PY'
import json
j=json.loads([l for l in open("gpurun_out/seq/bench.json") if l.startswith("{")][-1])
print("value", j["value"], "ms", j["ms_per_step"], "kernel", j["roofline"]["kernel_ms"], "frac", j["roofline"]["frac"])
for k,v in j.get("secondary",{}).items():
    print(k, {a:(round(b,4) if isinstance(b,float) else b) for a,b in v.items() if a in ("value","ms_per_step","kernel_ms","sets_kernel_ms","frac","overhead_us","ms_per_nr_iter","vecchia_laplace_likelihood_s","ms_per_call","ms_per_call_fresh_outputs","error")})
PY
