#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s12
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/s12/trace -- python3 tools/vl_bench.py > gpurun_out/s12/vl.json 2> gpurun_out/s12/err.log
cat gpurun_out/s12/vl.json
python3 - <<'PY'
import csv,glob,collections
f=glob.glob('gpurun_out/s12/trace/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# last NR step: between the last two vl_prepare kernels
prep=[i for i,r in enumerate(rows) if "vl_prepare" in r["Kernel_Name"]]
a,b=prep[-2],prep[-1]
ev=rows[a:b]
agg=collections.OrderedDict(); prev=None
for r in ev:
    n=r["Kernel_Name"].split('(')[0][-46:]
    d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
    g=(int(r["Start_Timestamp"])-prev)/1e3 if prev else 0
    prev=int(r["End_Timestamp"])
    x=agg.setdefault(n,[0,0.0,0.0]); x[0]+=1; x[1]+=d; x[2]+=g
for k,v in agg.items(): print(f"{k:48s} n={v[0]:3d} sum={v[1]:8.1f} us gaps={v[2]:7.1f}")
print('span us', (int(rows[b]["Start_Timestamp"])-int(rows[a]["Start_Timestamp"]))/1e3)
PY
