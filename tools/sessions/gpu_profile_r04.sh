#!/bin/bash
# round-4 judged artefacts: C3 (kernel stats + PMC + traffic), C2, general nu, per-rank shard, mode S kernel trace
cd $GRAFT_REPO_ROOT
(rocm-smi --showclocks --showpower --showtemp 2>&1 | head -40) > gpurun_out/r04_box.txt; bash tools/profile_round.sh r04 > gpurun_out/r04_profile.log 2>&1
bash tools/profile_round.sh r04C2 --config C2 --steps 20 > gpurun_out/r04C2_profile.log 2>&1
bash tools/profile_round.sh r04nu11 --nu 1.1 > gpurun_out/r04nu11_profile.log 2>&1
bash tools/profile_round.sh r04shard --emulate-world 8 --steps 50 > gpurun_out/r04shard_profile.log 2>&1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04S
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04S/trace -- python3 bench.py --mode S --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r04S/bench.json 2> gpurun_out/r04S/err.log
python3 tools/sgv_levels.py gpurun_out/r04S/trace > gpurun_out/r04S/levels.txt 2>&1
tail -4 gpurun_out/r04S/levels.txt
tail -c 600 gpurun_out/r04/bench.json
