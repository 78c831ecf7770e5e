#!/bin/bash
# round 4: N = 1 under the launcher (the N > 1 code path at world 1) must reproduce the plain N = 1 line
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4v; mkdir -p $O
python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline > $O/plain.json 2> $O/plain.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 5 --no-secondary --no-cpu-baseline > $O/launcher.json 2> $O/launcher.err
GPV_TORCH_ALLREDUCE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 1 --steps 20 --warmup 5 --no-secondary --no-cpu-baseline > $O/launcher_torch.json 2> $O/launcher_torch.err
python3 - <<'PY'
import json
for f in ("plain","launcher","launcher_torch"):
    try:
        j=json.loads([l for l in open(f"gpurun_out/r4v/{f}.json") if l.startswith("{")][-1])
        print(f, "value %.1f"%j["value"], "ms %.4f"%j["ms_per_step"], "kernel %.4f"%j["roofline"]["kernel_ms"], "loglik %.6f"%j["config"]["loglik"], "| collective:", j["config"]["collective"][:90], "| ranks", j["config"]["ranks"], "rows_reduced", j["config"]["rows_reduced"])
    except Exception as e:
        print(f, "FAILED", e); print(open(f"gpurun_out/r4v/{f}.err").read()[-1500:])
PY
python bench.py --gpus 2 --steps 2 --warmup 1; echo "exit code of --gpus 2 on a 1-GPU box: $?"
