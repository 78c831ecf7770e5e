#!/bin/bash
# round-6 judged artefacts: the full GPU suite, then C3 (kernel stats + PMC + traffic), C2, C4, general nu, per-rank shard,
# mode S kernel trace + level counters
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
GPV_PARITY_LOG=$GRAFT_REPO_ROOT/gpurun_out/r06/parity_counts.jsonl timeout 3000 python -m pytest tests -m gpu -x -q --durations=12 > gpurun_out/r06/gpu_tests.txt 2>&1
tail -18 gpurun_out/r06/gpu_tests.txt
(rocm-smi --showclocks --showpower --showtemp 2>&1 | head -40) > gpurun_out/r06_box.txt; bash tools/profile_round.sh r06 > gpurun_out/r06_profile.log 2>&1
bash tools/profile_round.sh r06C2 --config C2 --steps 20 > gpurun_out/r06C2_profile.log 2>&1
bash tools/profile_round.sh r06C4 --config C4 --steps 10 > gpurun_out/r06C4_profile.log 2>&1
bash tools/profile_round.sh r06nu11 --nu 1.1 > gpurun_out/r06nu11_profile.log 2>&1
bash tools/profile_round.sh r06shard --emulate-world 8 --steps 50 > gpurun_out/r06shard_profile.log 2>&1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06S; rm -rf gpurun_out/r06S/trace
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r06S/trace -- python3 bench.py --mode S --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r06S/bench.json 2> gpurun_out/r06S/err.log
python3 tools/sgv_levels.py gpurun_out/r06S/trace > gpurun_out/r06S/levels.txt 2>&1
tail -4 gpurun_out/r06S/levels.txt
bash tools/sessions/pmc_post.sh gpurun_out/r06S/pmc 2>&1 | tail -3
cd $GRAFT_REPO_ROOT
tail -c 600 gpurun_out/r06/bench.json
