"""bench.py launch contract, the parts that need no GPU: `--gpus N` starts N ranks by itself or refuses loudly."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None):
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True,
                          timeout=300, env=dict(os.environ, **(env or {})))


def test_gpus_n_refuses_when_fewer_devices_are_visible():
    # this container has no GPU: the parent must not fall back to a 1-GPU measurement labelled as N
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("box has >= 2 GPUs")
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0
    assert "refusing" in r.stderr and '"metric"' not in r.stdout


def test_world_size_mismatch_is_an_error():
    r = _run(["--gpus", "4", "--steps", "1", "--warmup", "0"], env=dict(RANK="0", WORLD_SIZE="2", LOCAL_RANK="0"))
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr and '"metric"' not in r.stdout


def test_roofline_models():
    sys.path.insert(0, ROOT)
    import bench
    # 203 B (C3 mode L), 435 B (C3 mode U), 305 B (C2 U), 833 B (C4 U); 2.5e4 flop per set at C3
    assert bench.alg_bytes_per_set(31, 2, "L") == 203 and bench.alg_bytes_per_set(31, 2, "U") == 435
    assert bench.alg_bytes_per_set(21, 2, "U") == 305 and bench.alg_bytes_per_set(61, 3, "U") == 833
    assert abs(bench.flops_per_set(31, 2) - 25306.33) < 1
    assert bench.CONFIGS["C3"][:4] == (2, 1_000_000, 30, 2) and bench.CONFIGS["C4"][:5] == (3, 1_000_000, 60, 3, 0.5)
