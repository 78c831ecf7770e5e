#!/bin/bash
# library-owned RCCL communicator: tests, then the per-rank step by both routes
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/comm
timeout 900 python -m pytest tests/test_distributed_nccl.py tests/test_bench_launch.py -m gpu -x -q > gpurun_out/comm/tests.log 2>&1
tail -5 gpurun_out/comm/tests.log
for route in 0 1; do
  for rep in 1 2; do
    GPV_TORCH_ALLREDUCE=$route timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port $((29500+route*10+rep)) bench.py --gpus 1 --steps 400 --warmup 40 --emulate-world 8 --no-cpu-baseline --no-secondary > gpurun_out/comm/step_route${route}_$rep.json 2> gpurun_out/comm/step_route${route}_$rep.err
    python3 - <<PY
import json
try:
    j=json.loads([l for l in open("gpurun_out/comm/step_route${route}_$rep.json") if l.startswith("{")][-1])
    print("route", $route, "ms_per_step", round(j["ms_per_step"],4), "kernel_ms", round(j["roofline"]["kernel_ms"],4), "overhead_us", round(1e3*(j["ms_per_step"]-j["roofline"]["kernel_ms"]),1))
except Exception as e:
    print("route", $route, "failed", e)
PY
  done
done
tail -3 gpurun_out/comm/step_route0_1.err
