cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc_r01
python3 tools/kbench.py --child --configs 30x2 --iters 1 > /dev/null 2>&1   # warm NN cache
for i in 1 2 3 4; do
  case $i in
   1) C="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM";;
   2) C="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE";;
   3) C="GRBM_GUI_ACTIVE FETCH_SIZE";;
   4) C="WRITE_SIZE SQ_INST_CYCLES_VMEM";;
  esac
  timeout 300 rocprofv3 --pmc $C --output-format csv -d gpurun_out/pmc_r01/p$i -- python3 tools/kbench.py --child --configs 30x2 --iters 1 > gpurun_out/pmc_r01/log$i.txt 2>&1
done
ls -R gpurun_out/pmc_r01 | head -30
