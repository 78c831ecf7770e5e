#!/bin/bash
# round 4: the mean sweep with the top block's columns solved by one dense substitution: tests, then the Vecchia-Laplace step
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_prediction.py -q -x -m gpu 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_configs.py -q -x -m gpu -k "C5 or dense_top or laplace or VL" 2>&1 | tail -3
for t in 0 128; do
GPV_POST_TOP=$t python tools/vl_bench.py 2>/dev/null | python3 -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('TOP=$t', {k:j[k] for k in ('nr_loop_s','nr_iters','ms_per_nr_iter','laplace_loglik_s','loglik','rmse_latent')})"
done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/r4vl; mkdir -p gpurun_out/r4vl
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4vl -o run -- python3 tools/vl_trace.py > gpurun_out/r4vl/out.txt 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r4vl/**/*kernel_stats.csv", recursive=True)
for r in csv.DictReader(open(f[0])):
    if "mean" in r["Name"] or "top" in r["Name"]: print(r["Name"][:80], r["Calls"], "avg us %.1f" % (float(r["AverageNs"])/1e3), "tot ms %.2f" % (float(r["TotalDurationNs"])/1e6))
PY
