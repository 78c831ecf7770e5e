#!/bin/bash
# Collect the judged artefacts for one round on the GPU box: bench line, rocprofv3 kernel stats of the SAME
# command, and PMC passes (separate runs, as gpurun requires).  Usage: bash tools/profile_round.sh r02 [bench args]
R=${1:-r02}; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$R; mkdir -p $OUT
python3 bench.py --steps 20 --warmup 3 "$@" > $OUT/bench.json 2> $OUT/bench.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary "$@" > $OUT/bench_under_rocprof.json 2> $OUT/rocprof.err
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
         "GRBM_GUI_ACTIVE FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc$i -- python3 bench.py --steps 3 --warmup 1 --clock-warmup-s 0 --no-cpu-baseline --no-secondary "$@" > $OUT/pmc$i.log 2>&1
done
tail -c 3000 $OUT/bench.json
