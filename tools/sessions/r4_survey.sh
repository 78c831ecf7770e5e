#!/bin/bash
# round 4, first contact: the GPU suite in survey mode (rows that need the extended-precision adjudication are logged, caps
# not enforced), then the default bench line with parity_in_run
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4a; mkdir -p $O; rm -f $O/parity.jsonl
GPV_PARITY_SURVEY=1 GPV_PARITY_LOG=$PWD/$O/parity.jsonl timeout 2400 python -m pytest tests -m gpu -q > $O/tests.log 2>&1
tail -15 $O/tests.log
timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err
tail -c 1500 $O/bench.json; tail -5 $O/bench.err
