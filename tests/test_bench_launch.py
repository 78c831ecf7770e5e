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


def test_cpu_baseline_times_the_c_function_alone(monkeypatch):
    """"wall-clock of the function only".  The callable between bench.py's two clock reads must be the ctypes
    function oracle_U_NZentries on pre-marshalled arguments — never the Python wrapper oracle.r_side.U_NZentries with its
    NumPy conversions (round 5's defect: 62 % of the timed interval was marshalling)."""
    import ctypes
    import numpy as np
    sys.path.insert(0, ROOT)
    import bench
    from oracle import r_side as R
    n, m = 3000, 10
    rng = np.random.default_rng(0)
    locs = rng.random((n, 2))
    NN = R.findOrderedNN(locs, m)
    revNN = np.nan_to_num(NN[:, ::-1], nan=0.0).astype(np.int64)
    revCond = np.where(revNN != 0, 0, -1).astype(np.int8)
    revCond[:, -1] = 1
    want = R.U_NZentries(2, n, locs, revNN, np.where(revCond < 0, 0, revCond), np.full(n, .1), np.full(n, .1), "matern",
                         [1., .1, 1.5])["Lentries"].copy()

    def boom(*a, **k):
        raise AssertionError("cpu_baseline went through the Python wrapper")
    monkeypatch.setattr(R, "U_NZentries", boom)
    seen = []
    kept = {}
    res = bench.cpu_baseline(locs, revNN, revCond, [1., .1, 1.5], .1, (0, n), repeats=2, keep=kept, probe=seen.append)
    assert seen and all(f is R._lib().oracle_U_NZentries for f in seen)
    assert all(isinstance(f, ctypes._CFuncPtr) for f in seen)
    assert np.array_equal(kept["Lentries"], want) and kept["rows"] == (0, n)
    assert res["timed_callable"] == "ctypes oracle_U_NZentries" and res["marshal_s"] > 0 and res["seconds"] > 0
    th = [e["threads"] for e in res["thread_sweep"]]
    assert th == sorted(th) and th[0] == 1 and th[-1] == res["logical_cpus"] and all(e["sets_per_s"] > 0 for e in res["thread_sweep"])
    assert res["cores"] in th and res["sets_per_s"] == max(e["sets_per_s"] for e in res["thread_sweep"] if e["rows"] == n)
    # a row sample (the extrapolated form) computes the same rows
    kept2 = {}
    res2 = bench.cpu_baseline(locs, revNN, revCond, [1., .1, 1.5], .1, (n - 500, n), repeats=1, keep=kept2, sweep=False)
    assert res2["extrapolated"] and np.array_equal(kept2["Lentries"], want[n - 500:])
