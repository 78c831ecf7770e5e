#!/bin/bash
# round 4: the two-block dense top (K = 128) against the one-block form (K = 64): parity with the level schedule, mode S
# same-box A/B, and where the new kernel's time goes (traced build)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4top2; mkdir -p $O
# the traced build (in-kernel stamps): python gpvecchia_amd/build.py --tag toptrace --flags=-DGPV_TOP_TRACE --plist 31
[ -f gpvecchia_amd/libgpvecchia_hiptoptrace.so ] || python gpvecchia_amd/build.py --tag toptrace --flags=-DGPV_TOP_TRACE --plist 31 > /dev/null 2>&1
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_prediction.py -q -x -m gpu 2>&1 | tail -5
for t in 64 128 64 128; do
  GPV_POST_TOP=$t python bench.py --mode S --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('TOP=$t', 'evals/s %.1f' % j['value'], 'ms %.4f' % j['ms_per_step'], 'set kernel %.4f' % j['roofline']['kernel_ms'])"
done
GPV_LIB=$GRAFT_REPO_ROOT/gpvecchia_amd/libgpvecchia_hiptoptrace.so python bench.py --mode S --steps 3 --warmup 1 --no-cpu-baseline --clock-warmup-s 0 2>&1 | grep "gpv top2" | tail -3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o run -- python3 bench.py --mode S --steps 10 --warmup 2 --no-cpu-baseline --clock-warmup-s 0 > $O/prof.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r4top2/prof/**/*kernel_stats.csv", recursive=True)
for r in csv.DictReader(open(f[0])):
    if "gpv" in r["Name"]: print(r["Name"][:90], r["Calls"], r["AverageNs"])
PY
python tools/sgv_levels.py $O/prof | tail -8
