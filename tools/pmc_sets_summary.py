"""Developer tool: print the counters of the LAST gpv_sets_kernel dispatch of each PMC pass written by tools/sessions/pmc_r01.sh."""
import csv
import glob
import sys

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc_r01"
tot = {}
for f in sorted(glob.glob(root + "/**/*counter_collection.csv", recursive=True)):
    rows = [r for r in csv.DictReader(open(f)) if "gpv_sets_kernel" in r["Kernel_Name"]]
    if not rows:
        continue
    last = max(int(r["Dispatch_Id"]) for r in rows)
    for r in rows:
        if int(r["Dispatch_Id"]) == last:
            tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
for k, v in sorted(tot.items()):
    print(f"{k:28s} {v:.6g}")
if "SQ_WAVES" in tot and "SQ_INSTS_VALU" in tot:
    print("VALU inst per wave", tot["SQ_INSTS_VALU"] / tot["SQ_WAVES"])
if "GRBM_GUI_ACTIVE" in tot and "SQ_ACTIVE_INST_VALU" in tot:
    cyc = tot["GRBM_GUI_ACTIVE"] / 8
    print("VALU busy", tot["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / cyc, "LDS busy", tot.get("SQ_LDS_IDX_ACTIVE", 0) / 256 / cyc)
