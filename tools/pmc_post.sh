# Developer tool: HBM traffic counters of the posterior-pass level kernels (one PMC pass, no tracing).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc_post
# FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950 (and a refused counter set leaves the process hanging: keep the timeouts)
timeout 240 rocprofv3 --pmc GRBM_GUI_ACTIVE FETCH_SIZE --output-format csv -d gpurun_out/pmc_post/p1 -- python3 tools/kbench.py --child --configs 30x2 --sgv --iters 1 > gpurun_out/pmc_post/log1.txt 2>&1
timeout 240 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/pmc_post/p2 -- python3 tools/kbench.py --child --configs 30x2 --sgv --iters 1 > gpurun_out/pmc_post/log2.txt 2>&1
python3 - <<'PY'
import csv, glob, collections
for p in ("p1", "p2"):
    for f in glob.glob(f"gpurun_out/pmc_post/{p}/**/*counter_collection.csv", recursive=True):
        rows = list(csv.DictReader(open(f)))
        # last evaluation: dispatches after the last conditioning-set kernel
        disp = sorted({int(r["Dispatch_Id"]) for r in rows})
        last_sets = max(int(r["Dispatch_Id"]) for r in rows if "gpv_sets_kernel" in r["Kernel_Name"])
        agg = collections.defaultdict(float); per = collections.defaultdict(dict)
        for r in rows:
            d = int(r["Dispatch_Id"])
            if d > last_sets and "posterior_level" in r["Kernel_Name"]:
                agg[r["Counter_Name"]] += float(r["Counter_Value"])
                per[d][r["Counter_Name"]] = float(r["Counter_Value"])
        print(p, dict(agg))
        ds = sorted(per)
        for i in (0, 1, 2, 5, 10, 20, 40):
            if i < len(ds): print("  level", i, per[ds[i]])
PY
