"""bench.py's N > 1 flow, run for real on a one-GPU box: `bench.py --gpus 2` starts its two ranks itself;
GPV_BENCH_BACKEND=gloo lets both share GPU 0 (each rank evaluates its own row shard with the HIP kernel, the 8 sums are
all-reduced over gloo on host copies — the sharding, the reduction, the self-check and the exit codes are the ones an
8-GPU run takes; only the transport differs).  Fresh child processes only: nothing here initialises the GPU in the
pytest process."""
import json
import os
import subprocess
import sys
import time

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N = 200_000


def _bench(extra, env=None, timeout=600):
    e = dict(os.environ, GPV_BENCH_BACKEND="gloo", PYTHONPATH=ROOT)
    e.pop("RANK", None); e.pop("WORLD_SIZE", None); e.pop("LOCAL_RANK", None)
    e.update(env or {})
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--n", str(N), "--steps", "5",
                        "--warmup", "1", "--no-cpu-baseline", "--no-secondary"] + extra,
                       capture_output=True, text=True, timeout=timeout, env=e)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    tag = "_".join(f"{k[10:]}-{v}" for k, v in sorted((env or {}).items())).replace(":", "-") or "plain"
    log = os.path.join(ROOT, "gpurun_out", f"nranks_{tag}.txt")          # the whole stderr of the launch, for a failure
    try:
        os.makedirs(os.path.dirname(log), exist_ok=True)
        with open(log, "w") as f:
            f.write(f"rc {r.returncode} extra {extra} env {env}\n" + r.stderr)
    except OSError:
        pass
    return r, (json.loads(lines[-1]) if lines else None), time.time() - t0


def test_two_ranks_on_one_gpu_reduce_every_row_and_match_the_unsharded_plan():
    r, out, _ = _bench([])
    assert r.returncode == 0, r.stderr[-3000:]
    assert out is not None and out["n_gpus"] == 2 and out["steps"] == 5
    assert out["config"]["ranks"] == 2 and out["config"]["rows_reduced"] == N
    assert out["config"]["sharding"] == "rows/2" and "gloo" in out["config"]["collective"]
    sc = out["self_check"]
    assert sc["ok"] and sc["rows_ok"] and sc["loglik_ok"] and sc["rel_diff"] <= 1e-12
    assert sc["loglik_nrank"] == out["config"]["loglik"]
    assert out["value"] > 0 and out["scaling"] == "strong"


@pytest.mark.parametrize("kind", ["data", "rows"])
def test_a_corrupted_shard_fails_the_self_check_with_exit_3(kind):
    # rank 1 evaluates one wrong datum ("data") or loses one of its rows ("rows"): the line is still printed, the job exits 3
    r, out, _ = _bench([], env={"GPV_BENCH_CORRUPT": f"{kind}:1"})
    assert r.returncode == 3, (r.returncode, r.stderr[-3000:])
    assert out is not None and not out["self_check"]["ok"]
    if kind == "rows":
        assert out["config"]["rows_reduced"] == N - 1 and not out["self_check"]["rows_ok"]
    else:
        assert out["self_check"]["rows_ok"] and not out["self_check"]["loglik_ok"]
    assert "self-check FAILED" in r.stderr


def test_a_stalled_rank_trips_the_wall_clock_guard_with_exit_4_and_no_hang():
    # rank 1 sleeps 120 s in front of its first evaluation; rank 0 waits in the all-reduce; its guard (2 s) ends the job
    r, out, elapsed = _bench(["--comm-guard-s", "2"], env={"GPV_BENCH_STALL_RANK": "1", "GPV_BENCH_STALL_S": "120"},
                             timeout=300)
    assert r.returncode == 4, (r.returncode, r.stderr[-3000:])
    assert out is None                                               # no line for a job that did not complete
    assert "exiting 4" in r.stderr
    assert elapsed < 100, elapsed                                    # ended by the guard, not by the sleeper


def test_eight_ranks_the_target_world_size_at_full_size_on_one_gpu():
    """SURVEY.md §8e at the machine's world size: `bench.py --gpus 8 --n 1000000` (BASELINE configs[2]) with eight fresh
    child ranks sharing GPU 0 over gloo.  Every rank owns 125 000 contiguous rows (src/U_NZentries.cpp:37-39: rows are
    independent), the all-reduced row count is n, the 8-rank log-likelihood equals the unsharded plan's to 1e-12."""
    e = dict(os.environ, GPV_BENCH_BACKEND="gloo", PYTHONPATH=ROOT)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        e.pop(k, None)
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--n", "1000000", "--steps", "5",
                        "--warmup", "1", "--no-cpu-baseline", "--no-secondary"], capture_output=True, text=True, timeout=900, env=e)
    elapsed = time.time() - t0
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "nranks_8.txt"), "w") as f:
            f.write(f"rc {r.returncode} elapsed {elapsed:.1f} s\n" + (lines[-1] if lines else "") + "\n" + r.stderr)
    except OSError:
        pass
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads(lines[-1])
    assert out["n_gpus"] == 8 and out["config"]["ranks"] == 8 and out["config"]["rows_reduced"] == 1_000_000
    assert out["config"]["sharding"] == "rows/8"
    sh = sorted(out["config"]["shards"], key=lambda x: x["rank"])
    assert [x["rank"] for x in sh] == list(range(8))
    assert all(x["rows"] == [125_000 * i, 125_000 * (i + 1)] for i, x in enumerate(sh))
    assert all(0 < x["device_mem_used_gb"] < 288 for x in sh)
    sc = out["self_check"]
    assert sc["ok"] and sc["rows_ok"] and sc["loglik_ok"] and sc["rel_diff"] <= 1e-12 and sc["ranks"] == 8
    assert out["roofline"]["sets_per_launch"] == 125_000
    print(f"8 ranks on one GPU: {elapsed:.1f} s, device memory in use {max(x['device_mem_used_gb'] for x in sh)} GB")
    assert elapsed <= 180, elapsed                                    # (eight torch imports and eight HIP contexts on one box)
