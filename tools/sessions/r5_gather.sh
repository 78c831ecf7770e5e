#!/bin/bash
# round 5: cost of scattered 16-byte gathers (tools/ubench/gather_lines.hip) from an Infinity-Cache-resident buffer and from
# a 2 GB one, with FETCH_SIZE and the L2 hit / miss counts per kernel
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5g; mkdir -p $O
for mb in 2048 128; do ./tools/ubench/gather_lines $mb 3 | tee $O/times_$mb.txt; done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for mb in 2048 128; do
  rm -rf $O/pmcF_$mb $O/pmcT_$mb
  timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmcF_$mb -- ./tools/ubench/gather_lines $mb 1 > $O/pmcF_$mb.log 2>&1
  timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $O/pmcT_$mb -- ./tools/ubench/gather_lines $mb 1 > $O/pmcT_$mb.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
for mb in (2048, 128):
    for tag in ("F", "T"):
        fs = glob.glob(f"gpurun_out/r5g/pmc{tag}_{mb}/**/*counter_collection.csv", recursive=True)
        if not fs:
            print(mb, tag, "no counters"); continue
        rows = list(csv.DictReader(open(fs[0])))
        agg = collections.OrderedDict()
        for r in rows:
            if "gather" not in r["Kernel_Name"]:
                continue
            key = (r["Kernel_Name"].split("(")[0], r["Counter_Name"])
            agg.setdefault(key, []).append(float(r["Counter_Value"]))
        for (k, c), v in agg.items():
            print(mb, "MB", k, c, "last dispatch", v[-1], "dispatches", len(v))
PY
