"""Developer tool: the random-shape parity test of tests/test_gpu_fuzz.py over many more seeds (one-off sweeps on a GPU box).

    python tools/fuzz_more.py [first_seed last_seed]
"""
import sys, os, traceback
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch  # noqa
import test_gpu_fuzz as F
lo, hi = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (24, 424)
bad = 0
for seed in range(lo, hi):
    try:
        F.test_random_shapes_against_oracle.__wrapped__(seed) if hasattr(F.test_random_shapes_against_oracle, "__wrapped__") else F.test_random_shapes_against_oracle(seed)
    except Exception as e:
        bad += 1
        print("SEED", seed, "FAILED:", repr(e)[:300])
print("extended fuzz done, seeds", lo, "to", hi - 1, "failures:", bad)
