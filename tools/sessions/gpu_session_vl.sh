#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/vl
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/vl/trace -- python3 tools/vl_trace.py > gpurun_out/vl/out.txt 2>&1
tail -2 gpurun_out/vl/out.txt | cut -c1-400
python3 - <<'PY'
import csv,glob,collections,os
f=sorted(glob.glob('gpurun_out/vl/trace/**/*kernel_trace.csv',recursive=True), key=os.path.getmtime)[-1]
rows=list(csv.DictReader(open(f))); rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# the last Newton step = from the last but one gpv_vl_update kernel to the last
upd=[i for i,r in enumerate(rows) if "vl_update" in r["Kernel_Name"]]
a,b=upd[-2],upd[-1]
ev=rows[a:b+1]
agg=collections.OrderedDict(); prev=None
for r in ev:
    n=r["Kernel_Name"].split('(')[0][-48:]
    d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
    g=(int(r["Start_Timestamp"])-prev)/1e3 if prev else 0
    prev=int(r["End_Timestamp"])
    x=agg.setdefault(n,[0,0.0,0.0]); x[0]+=1; x[1]+=d; x[2]+=g
for k,v in agg.items(): print(f"{k:50s} n={v[0]:3d} sum={v[1]:8.1f} us gaps_before={v[2]:7.1f}")
print('one Newton step, span us', (int(ev[-1]["End_Timestamp"])-int(ev[0]["End_Timestamp"]))/1e3)
PY
