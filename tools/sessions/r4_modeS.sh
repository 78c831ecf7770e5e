#!/bin/bash
# round 4: why the set kernel is 0.15 ms slower inside mode S: the same launch without the posterior pass (GPV_POST_SKIP),
# and its counters inside mode S against stand-alone
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4q; mkdir -p $O
for e in 0 1; do
  if [ $e = 1 ]; then export GPV_POST_SKIP=1; else unset GPV_POST_SKIP; fi
  python bench.py --mode S --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('POST_SKIP=$e', 'evals/s %.1f' % j['value'], 'ms %.4f' % j['ms_per_step'], 'set kernel %.4f' % j['roofline']['kernel_ms'])"
done
unset GPV_POST_SKIP
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR" \
         "GRBM_GUI_ACTIVE FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C --output-format csv -d $O/pmcS$i -- python3 bench.py --mode S --steps 3 --warmup 1 --clock-warmup-s 0 --no-cpu-baseline > $O/pmcS$i.log 2>&1
  timeout 300 rocprofv3 --pmc $C --output-format csv -d $O/pmcL$i -- python3 bench.py --steps 3 --warmup 1 --clock-warmup-s 0 --no-cpu-baseline --no-secondary > $O/pmcL$i.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
for tag in ("S","L"):
    tot=collections.OrderedDict()
    for i in range(1,7):
        fs=glob.glob(f"gpurun_out/r4q/pmc{tag}{i}/**/*counter_collection.csv", recursive=True)
        if not fs: continue
        rows=[r for r in csv.DictReader(open(fs[0])) if "gpv_sets_kernel" in r["Kernel_Name"]]
        if not rows: continue
        last=max(int(r["Dispatch_Id"]) for r in rows)
        for r in rows:
            if int(r["Dispatch_Id"])==last: tot[r["Counter_Name"]]=tot.get(r["Counter_Name"],0)+float(r["Counter_Value"])
    print(tag, dict(tot))
PY
