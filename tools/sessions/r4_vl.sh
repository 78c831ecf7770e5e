#!/bin/bash
# round 4: kernel time of one Vecchia-Laplace Newton step (C5: n = 5e5)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/r4vl; mkdir -p gpurun_out/r4vl
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4vl -o run -- python3 tools/vl_trace.py > gpurun_out/r4vl/out.txt 2>&1
tail -2 gpurun_out/r4vl/out.txt
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r4vl/**/*kernel_stats.csv", recursive=True)
for r in csv.DictReader(open(f[0])):
    print(r["Name"][:100], r["Calls"], "avg us %.1f" % (float(r["AverageNs"])/1e3), "tot ms %.2f" % (float(r["TotalDurationNs"])/1e6))
PY
