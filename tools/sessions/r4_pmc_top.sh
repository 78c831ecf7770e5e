#!/bin/bash
# round 4: instruction counters of the dense top block kernel (and the leaf / MODE 1 launches next to it), mode S of bench.py
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_top; rm -rf $OUT; mkdir -p $OUT
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
         "SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM_RD"; do
  i=$((i+1))
  GPV_NO_GRAPH=1 timeout 300 rocprofv3 --pmc $C --output-format csv -d $OUT/p$i -- python3 bench.py --mode S --steps 2 --warmup 1 --no-cpu-baseline --clock-warmup-s 0 > $OUT/log$i.txt 2>&1
done
python3 - <<'PY'
import csv, glob, collections, json
res = collections.defaultdict(dict)
for p in (1, 2):
    for f in glob.glob(f"gpurun_out/pmc_top/p{p}/**/*counter_collection.csv", recursive=True):
        rows = list(csv.DictReader(open(f)))
        last = max(int(r["Dispatch_Id"]) for r in rows if "gpv_sets_kernel" in r["Kernel_Name"])
        for r in rows:
            if int(r["Dispatch_Id"]) > last and any(t in r["Kernel_Name"] for t in ("top2", "leaf", "<16, 1", "sum_pair")):
                key = r["Kernel_Name"].split("(")[0][-40:]
                res[key][r["Counter_Name"]] = res[key].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
json.dump(res, open("gpurun_out/pmc_top/summary.json", "w"), indent=1)
for k, v in res.items(): print(k, {a: round(b) for a, b in v.items()})
PY
