#!/bin/bash
cd $GRAFT_REPO_ROOT
GPV_TIMING=1 GPV_NO_D2H_STAGING=1 python tools/dropin_timing.py 2>&1 | grep -v "plan:" | tail -12
echo ---- staged
GPV_TIMING=1 python tools/dropin_timing.py 2>&1 | grep -v "plan:" | tail -12
python -m pytest tests/test_gpu_parity.py -q -x -k "dropin or kat or plan_cache or lentries" 2>&1 | tail -3
