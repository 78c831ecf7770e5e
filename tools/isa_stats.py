"""Developer tool: static instruction mix of one conditioning-set kernel instantiation, from the built objects.
    python tools/isa_stats.py gpvecchia_amd/csrc/build/sets_p31.o Li31ELi2ELi1E"""
import collections
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
obj, pat = sys.argv[1], sys.argv[2]
with tempfile.TemporaryDirectory() as td:
    fat, co = os.path.join(td, "x.fat"), os.path.join(td, "x.co")
    subprocess.check_call([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat])
    subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}",
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"])
    notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", co], capture_output=True, text=True).stdout
blocks = re.split(r"\n(?=[0-9a-f]+ <)", dis)
for b in blocks:
    head = b.split("\n", 1)[0]
    if pat not in head or "gpv_sets_kernel" not in head:
        continue
    ins = [l.split()[0] for l in b.split("\n")[1:] if l.strip() and not l.strip().endswith(":") and "\t" in l]
    ins = [l.strip().split()[0] for l in b.split("\n")[1:] if l.startswith("\t") or l.startswith(" ")]
    ins = [i for i in ins if re.match(r"^[a-z]", i)]
    cls = collections.Counter()
    for i in ins:
        k = "valu" if i.startswith("v_") else "salu" if i.startswith("s_") else "lds" if i.startswith("ds_") else \
            "vmem" if i.startswith(("global_", "buffer_", "flat_", "scratch_")) else "other"
        cls[k] += 1
    top = collections.Counter(ins).most_common(28)
    print(head)
    print(dict(cls), "total", len(ins))
    print(top)
    for key in ("vgpr_count", "sgpr_count", "sgpr_spill_count", "vgpr_spill_count", "group_segment_fixed_size"):
        m = re.search(r"\.name:\s+" + re.escape(head.split("<")[1].split(">")[0]) + r"[\s\S]*?\." + key + r":\s+(\d+)", notes)
    print(re.findall(r"\.group_segment_fixed_size: (\d+)[\s\S]{0,300}?" + re.escape(head.split("<")[1].split(">")[0])
                     + r"[\s\S]{0,400}?\.sgpr_spill_count: (\d+)[\s\S]{0,200}?\.vgpr_count:\s+(\d+)", notes))
