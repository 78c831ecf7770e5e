#!/bin/bash
# profiles: C3 (kernel stats + PMC + traffic), C4 PMC, mode S kernel trace
cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh r03 > gpurun_out/r03_profile.log 2>&1
bash tools/profile_round.sh r03C4 --config C4 --steps 10 > gpurun_out/r03C4_profile.log 2>&1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03S
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03S/trace -- python3 bench.py --mode S --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r03S/bench.json 2> gpurun_out/r03S/err.log
python3 tools/sgv_levels.py gpurun_out/r03S/trace > gpurun_out/r03S/levels.txt 2>&1
cat gpurun_out/r03S/levels.txt | tail -25
tail -c 600 gpurun_out/r03/bench.json
