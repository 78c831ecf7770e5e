#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_distributed_nccl.py -q 2>&1 | tail -4
bash tools/sessions/r4_launcher.sh 2>&1 | tail -6
