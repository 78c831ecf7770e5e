# Developer tool: PMC counters of the posterior-pass level kernels (separate passes, no tracing), mode S of bench.py.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=${1:-gpurun_out/pmc_post}; mkdir -p $OUT   # usage: pmc_post.sh [outdir]  (GPV_LIB selects the library)
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
         "GRBM_GUI_ACTIVE FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "WRITE_SIZE"; do
  i=$((i+1))
  GPV_NO_GRAPH=1 timeout 300 rocprofv3 --pmc $C --output-format csv -d $OUT/p$i -- python3 bench.py --mode S --steps 2 --warmup 1 --no-cpu-baseline --clock-warmup-s 0 > $OUT/log$i.txt 2>&1
done
OUT=$OUT python3 - <<'PY'
import csv, glob, collections, json, os
OUT = os.environ['OUT']
res = {}
for p in range(1, 6):
    for f in glob.glob(f"{OUT}/p{p}/**/*counter_collection.csv", recursive=True):
        rows = list(csv.DictReader(open(f)))
        last_sets = max(int(r["Dispatch_Id"]) for r in rows if "gpv_sets_kernel" in r["Kernel_Name"])
        per = collections.defaultdict(lambda: collections.defaultdict(float))
        order = []
        for r in rows:
            d = int(r["Dispatch_Id"])
            if d > last_sets and "posterior_le" in r["Kernel_Name"]:
                if d not in order: order.append(d)
                per[d][r["Counter_Name"]] += float(r["Counter_Value"])
                per[d]["_kernel"] = r["Kernel_Name"][:60]
                per[d]["_grid"] = r.get("Grid_Size", "")
        for li, d in enumerate(sorted(order)):
            res.setdefault(li, {}).update({k: v for k, v in per[d].items()})
import sys
sys.path.insert(0, os.getcwd())
import bench
res["_meta"] = {"posterior_code_sha256": bench.posterior_code_sha256(),
                "what": "per-level PMC sums of bench.py --mode S (n = 1e6, m = 30, maxmin + SGV), GPV_NO_GRAPH=1, one evaluation"}
json.dump(res, open(f"{OUT}/levels_pmc.json", "w"), indent=0)
res.pop("_meta")
tot = collections.defaultdict(float)
for li, d in res.items():
    for k, v in d.items():
        if isinstance(v, float): tot[k] += v
print("all levels:", {k: round(v) for k, v in tot.items()})
for li in (0, 1, 2, 4, 8, 12, 16, 24, 32, 48, 64, 96, 128):
    if li in res: print(li, {k: (round(v) if isinstance(v, float) else v) for k, v in res[li].items()})
PY
