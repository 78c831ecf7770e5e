"""Developer tool: turn the scratch output of tools/profile_round.sh (gpurun_out/<tag>/) into the tracked artefacts under
profiles/: <name>_bench.json, <name>_kernel_stats.csv, <name>_pmc_pass<i>.csv (set-kernel rows only), <name>_pmc_summary.json
and <round>_pmc_traffic.json (the per-launch HBM traffic bench.py quotes while the kernel source hash matches).

    python tools/collect_profiles.py gpurun_out/r02 r02
    python tools/collect_profiles.py gpurun_out/r02C4 r02_C4 notraffic     (another config: everything but r02_pmc_traffic.json)
"""
import csv
import glob
import hashlib
import json
import os
import shutil
import sys

src, name = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
line = [l for l in open(os.path.join(src, "bench.json")).read().splitlines() if l.startswith("{")][-1]
open(os.path.join(P, name + "_bench.json"), "w").write(line + "\n")
newest = lambda fs: sorted(fs, key=os.path.getmtime)[-1:]          # gpurun merges into the same scratch directory run after run
st = newest(glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True))
if st:
    shutil.copy(st[0], os.path.join(P, name + "_kernel_stats.csv"))
# the launches of the set kernel one by one (the bench command precedes its K timed steps with a clock warm-up, whose first
# ~25 launches run at a rising shader clock): overall mean, mean of the last K = the timed region
tr = newest(glob.glob(os.path.join(src, "stats", "**", "*kernel_trace.csv"), recursive=True))
if tr:
    rows = [r for r in csv.DictReader(open(tr[0])) if "gpv_sets_kernel" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    us = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
    K = json.loads(line)["steps"]
    json.dump({"kernel": rows[0]["Kernel_Name"].split("(")[0] if rows else None, "launches": len(us),
               "mean_us_all_launches": sum(us) / max(len(us), 1),
               "timed_region": {"launches": K, "mean_us": sum(us[-K:]) / max(len(us[-K:]), 1), "min_us": min(us[-K:]), "max_us": max(us[-K:])},
               "first_launches_us": [round(u, 1) for u in us[:30]],
               "bench_kernel_ms_hipEvents_same_run": json.loads(line)["roofline"]["kernel_ms"]},
              open(os.path.join(P, name + "_kernel_launches.json"), "w"), indent=1)
tot = {}
for i in (1, 2, 3, 4):
    fs = newest(glob.glob(os.path.join(src, f"pmc{i}", "**", "*counter_collection.csv"), recursive=True))
    if not fs:
        continue
    rows = list(csv.DictReader(open(fs[0])))
    keep = [r for r in rows if "gpv_sets_kernel" in r["Kernel_Name"]]
    with open(os.path.join(P, f"{name}_pmc_pass{i}.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(keep)
    last = max(int(r["Dispatch_Id"]) for r in keep)
    for r in keep:
        if int(r["Dispatch_Id"]) == last:
            tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
b = json.loads(line)
# the sets ONE launch of THIS process handles: a rank's row shard when the bench line says so (sharding "rows/N", or the
# emulated per-rank load of --emulate-world E), not the whole data set
nsets = b["config"]["n"]
shard = str(b["config"].get("sharding", "rows/1")).split("/")[-1]
if shard.isdigit() and int(shard) > 1:
    nsets = nsets // int(shard)
spl = b.get("roofline", {}).get("sets_per_launch")
if spl:
    nsets = int(spl)
if "GRBM_GUI_ACTIVE" in tot:
    cyc = tot["GRBM_GUI_ACTIVE"] / 8
    tot["_derived"] = {
        "kernel_cycles": cyc,
        "valu_busy_frac": tot.get("SQ_ACTIVE_INST_VALU", 0) * 4 / 1024 / cyc,
        "lds_busy_frac": tot.get("SQ_LDS_IDX_ACTIVE", 0) / 256 / cyc,
        "lds_conflict_frac": tot.get("SQ_LDS_BANK_CONFLICT", 0) / max(tot.get("SQ_LDS_IDX_ACTIVE", 1), 1),
        "valu_insts_per_set": tot.get("SQ_INSTS_VALU", 0) / nsets,
        "fetch_bytes_raw_x2": 2.0 * tot.get("FETCH_SIZE", 0) * 1024,
        "write_bytes": tot.get("WRITE_SIZE", 0) * 1024,
    }
json.dump(tot, open(os.path.join(P, name + "_pmc_summary.json"), "w"), indent=1)
if "FETCH_SIZE" in tot and not (len(sys.argv) > 3 and sys.argv[3] == "notraffic"):
    sys.path.insert(0, ROOT)
    from bench import kernel_code_sha256
    json.dump({
        "hbm_bytes_per_launch": (2.0 * tot["FETCH_SIZE"] + tot.get("WRITE_SIZE", 0)) * 1024,
        "fetch_size_kb_raw": tot["FETCH_SIZE"], "write_size_kb": tot.get("WRITE_SIZE", 0),
        "kernel_code_sha256": kernel_code_sha256(),
        "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/profile_round.sh), last launch of the set "
                "kernel in bench.py at n=1e6 m=30 mode L.  Units KB -> x1024; FETCH_SIZE doubled as MI355X_MICROARCH.md "
                "prescribes for gfx950 (the counter tallies 128-byte fabric requests at 64 bytes); WRITE_SIZE as read.  "
                "bench.py quotes the figure only while gpv_sets_kernel.hpp, comments and whitespace removed, hashes to kernel_code_sha256.",
    }, open(os.path.join(P, name.split("_")[0] + "_pmc_traffic.json"), "w"), indent=1)
print(json.dumps(tot.get("_derived", {}), indent=1))
print(line[:400])
