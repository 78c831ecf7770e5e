#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s2
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/s2/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/s2/pytest.log
grep -n "FAILED\|passed\|failed" gpurun_out/s2/pytest.log | tail -15
timeout 600 python -m torch.distributed.run --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 tools/step_profile2.py > gpurun_out/s2/step.log 2>&1
cat gpurun_out/s2/step.log | grep -v "^W\|warn" | tail -20
