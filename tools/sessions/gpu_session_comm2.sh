#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/comm2
python tools/comm_diag.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/comm2/windows.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/comm2/trace -- python3 tools/comm_diag.py > gpurun_out/comm2/trace.log 2>&1
python3 - <<'PY'
import csv,glob,collections
f=sorted(glob.glob('gpurun_out/comm2/trace/**/*kernel_trace.csv',recursive=True))[-1]
c=collections.Counter(); d=collections.Counter()
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"][:70]; c[k]+=1; d[k]+=int(r["End_Timestamp"])-int(r["Start_Timestamp"])
for k,v in c.most_common(12): print(v, round(d[k]/v/1e3,1), "us", k)
PY
