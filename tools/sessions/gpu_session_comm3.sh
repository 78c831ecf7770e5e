#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/comm3
echo "=== in-process tcp init + barrier"
COMM_DIAG_TORCH=1 COMM_DIAG_BARRIER=1 python tools/comm_diag.py 2>&1 | grep -v "amdgpu.ids\|socket.cpp" | grep -A4 "communicator attached" 
echo "=== under torchrun"
COMM_DIAG_TORCH=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 29611 tools/comm_diag.py 2>&1 | grep -v "amdgpu.ids\|socket.cpp" | grep -A4 "communicator attached\|env:" 
echo "=== under torchrun + barrier"
COMM_DIAG_TORCH=1 COMM_DIAG_BARRIER=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 29612 tools/comm_diag.py 2>&1 | grep -v "amdgpu.ids\|socket.cpp" | grep -A4 "communicator attached" 
echo "=== bench child route 0"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 29613 bench.py --gpus 1 --steps 400 --warmup 40 --emulate-world 8 --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('ms_per_step', j['ms_per_step'], 'kernel', j['roofline']['kernel_ms'])"
