"""Developer tool: summarise the posterior-pass level kernels of the LAST evaluation in a rocprofv3 kernel trace.

    rocprofv3 --kernel-trace -d gpurun_out/sgvN -o runc --output-format csv -- python3 tools/kbench.py --child --configs 30x2 --sgv --iters 2
    python tools/sgv_levels.py gpurun_out/sgvN/runc_kernel_trace.csv
"""
import csv
import glob
import sys

path = sys.argv[1]
import os
files = [path] if os.path.isfile(path) else glob.glob(path + "/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(files[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last evaluation = kernels after the last conditioning-set kernel
last = max(i for i, r in enumerate(rows) if "gpv_sets_kernel" in r["Kernel_Name"])
ev = rows[last:]
t0 = int(ev[0]["Start_Timestamp"])
lev = [r for r in ev if "posterior_le" in r["Kernel_Name"]]          # level kernels and the leaf kernel of level 0
print("eval span ms", (int(ev[-1]["End_Timestamp"]) - t0) / 1e6, "kernels", len(ev))
tot = 0
for i, r in enumerate(lev):
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    if True:
        print(i, "WPC8" if "<8>" in r["Kernel_Name"] else "WPC1", "grid", r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size"),
              "us", round(d, 1), "gap_us", round((int(r["Start_Timestamp"]) - int(lev[i - 1]["End_Timestamp"])) / 1e3, 1) if i else 0)
print("levels", len(lev), "sum kernel us", round(tot, 1), "span us", (int(lev[-1]["End_Timestamp"]) - int(lev[0]["Start_Timestamp"])) / 1e3)
w8 = [r for r in lev if "<8>" in r["Kernel_Name"]]
print("WPC8 levels", len(w8), "sum us", round(sum((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in w8), 1))
