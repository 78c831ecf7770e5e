#!/bin/bash
# round-3 GPU session 1: parity suite, A/B of the set kernel (round-2 arithmetic vs the round-3 instruction diet), bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s1
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/s1/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/s1/pytest.log
tail -5 gpurun_out/s1/pytest.log
for i in 1 2; do
timeout 600 python tools/kbench.py --configs 30x2,20x2,60x3 gpvecchia_amd/libgpvecchia_hip_base.so gpvecchia_amd/libgpvecchia_hip.so >> gpurun_out/s1/kbench.log 2>&1
done
cat gpurun_out/s1/kbench.log
timeout 900 python bench.py > gpurun_out/s1/bench.json 2> gpurun_out/s1/bench.err; echo "bench rc=$?"
tail -c 6000 gpurun_out/s1/bench.json; tail -5 gpurun_out/s1/bench.err
