#!/bin/bash
# round 6: general-nu table formats A/B (shipped 7 doubles + 4 floats / 8 segments per octave / two 4-wave workgroups per CU
# against GPV_MT_F64: 9 doubles, degree 8, 16 segments per octave, one 8-wave workgroup per CU), same box, alternating
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06nu; mkdir -p $O
F64=$GRAFT_REPO_ROOT/gpvecchia_amd/libgpvecchia_hip_f64.so
python3 - <<'PY' 2>&1 | tee $O/parity.txt
import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
os.environ["GPV_LIB"] = os.path.join(os.getcwd(), "gpvecchia_amd", "libgpvecchia_hip_f64.so")
import gpvecchia_amd as G
from oracle import r_side as R
sys.path.insert(0, "tests")
rng = np.random.default_rng(3)
n, m = 4000, 30
locs = rng.random((n, 2))
va = G.vecchia_specify(locs, m, ordering="maxmin", cond_yz="SGV")
prep = va["U_prep"]
tau = 0.05 + 0.1 * rng.random(n)
for nu, rg in ((0.3, 0.05), (1.1, 0.02), (1.1, 0.3), (2.2, 0.1), (7.5, 0.05), (0.9, 2.0)):
    cp = [1.3, rg, nu]
    out = G.U_NZentries(1, n, va["locsord"], prep["revNNarray"], prep["revCond"], tau, tau, "matern", cp)
    ref = R.U_NZentries(1, n, va["locsord"], prep["revNNarray"], np.where(prep["revCond"] < 0, 0, prep["revCond"]), tau, tau, "matern", cp)
    err = np.abs(out["Lentries"] - ref["Lentries"]).max(axis=1) / np.abs(ref["Lentries"]).max(axis=1)
    print(f"F64 table nu={nu} range={rg}: max row err {err.max():.2e}, rows beyond 1e-8: {(err > 1e-8).sum()}, beyond 1e-10: {(err > 1e-10).sum()}, median {np.median(err):.1e}")
PY
for rep in 1 2 3; do
  for lib in base f64; do
    if [ $lib = f64 ]; then export GPV_LIB=$F64; else unset GPV_LIB; fi
    python bench.py --nu 1.1 --no-secondary --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python3 -c "
import sys, json
j = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$lib rep $rep nu=1.1: evals/s %.1f kernel_ms %.4f frac %.4f' % (j['value'], j['roofline']['kernel_ms'], j['roofline']['frac']))" | tee -a $O/ab.txt
  done
done
export GPV_LIB=$F64
for mult in 2 4; do
  GPV_GRID_MULT=$mult python bench.py --nu 1.1 --no-secondary --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python3 -c "
import sys, json
j = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('f64 GRID_MULT=$mult nu=1.1: evals/s %.1f kernel_ms %.4f' % (j['value'], j['roofline']['kernel_ms']))" | tee -a $O/ab.txt
done
for nu in 0.3 2.2; do
  for lib in base f64; do
    if [ $lib = f64 ]; then export GPV_LIB=$F64; else unset GPV_LIB; fi
    python bench.py --nu $nu --no-secondary --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python3 -c "
import sys, json
j = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$lib nu=$nu: evals/s %.1f kernel_ms %.4f' % (j['value'], j['roofline']['kernel_ms']))" | tee -a $O/ab.txt
  done
done
unset GPV_LIB
# the changed default library: new GPU tests of this round + cpu_baseline evidence
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_prediction.py -m gpu -x -q -k "speculative or long_rows or generic or nu" 2>&1 | tail -5
python bench.py --no-secondary --steps 20 --warmup 5 2>/dev/null | python3 -c "
import sys, json
j = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('headline', j['value'], j['roofline']['frac']); print(json.dumps(j['cpu_baseline'])[:2500])" | tee $O/cpu_baseline.txt
