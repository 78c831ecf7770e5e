#!/bin/bash
cd $GRAFT_REPO_ROOT
python tools/dropin_timing.py 2>&1 | tail -4
echo ---- no staging
GPV_NO_D2H_STAGING=1 python tools/dropin_timing.py 2>&1 | tail -4
python - <<'PY'
import sys, time, numpy as np
sys.path.insert(0, '.')
import bench
import gpvecchia_amd as G
n, m = 1_000_000, 30
locs, z, revNN, revCond, a, b = bench.build_workload(n, m, 2, 0, 1)
for rep in range(2):
    print(bench.dropin_config(n, locs, revNN, revCond, [1.0, 0.02, 1.5], 0.1))
PY
