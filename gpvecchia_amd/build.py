"""Build libgpvecchia_hip.so (gfx950) in-tree with hipcc.

    python -m gpvecchia_amd.build [--force] [--jobs N]

One translation unit per compiled row length (csrc/gpv_plist.h) so the kernels
compile in parallel; objects are cached under csrc/build/ by source mtime.
hipcc cross-compiles without a GPU, so this also runs in the CPU-only container.
"""
from __future__ import annotations

import argparse
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
BUILD = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "libgpvecchia_hip.so")
ARCH = "gfx950"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
SPLIT_FROM_P = 41          # row lengths from here on compile one TU per spatial dimension (minutes per instantiation)
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-variable"]


def plist():
    txt = open(os.path.join(CSRC, "gpv_plist.h")).read()
    line = [l for l in txt.splitlines() if l.startswith("#define GPV_P_LIST")][0]
    return [int(x) for x in re.findall(r"X\((\d+)\)", line)]


PLIST_ONLY = None          # tuning builds: compile these row lengths only


def _newest(paths):
    return max(os.path.getmtime(p) for p in paths)


def _compile(args):
    src, obj, extra, deps, force = args
    if not force and os.path.exists(obj) and os.path.getmtime(obj) >= _newest(deps + [src]):
        return obj, 0.0
    import time
    t0 = time.time()
    cmd = [HIPCC] + FLAGS + extra + ["-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
    return obj, time.time() - t0


def build(force: bool = False, jobs: int | None = None, verbose: bool = False, tag: str = "",
          extra_flags: list[str] | None = None, only: list[int] | None = None) -> str:
    """tag/extra_flags build a tuning variant (libgpvecchia_hip<tag>.so, objects under build<tag>/);
    the default (no tag) is the product library."""
    global BUILD, LIB
    if tag:
        BUILD = os.path.join(CSRC, "build" + tag)
        LIB = os.path.join(HERE, f"libgpvecchia_hip{tag}.so")
    extra_flags = list(extra_flags or [])
    if only:
        if not tag:
            raise ValueError("a shortened row-length list is for tagged tuning builds only")
        extra_flags.append("-DGPV_P_LIST(X)=" + " ".join(f"X({p})" for p in sorted(only)))
    os.makedirs(BUILD, exist_ok=True)
    H = lambda *names: [os.path.join(CSRC, f) for f in names]
    pub = os.path.join(os.path.dirname(HERE), "include", "gpvecchia.h")
    internal = H("gpv_internal.h", "gpv_bessel.hpp")
    kern = internal + H("gpv_sets_kernel.hpp", "gpv_reduce_tail.hpp", "gpv_plist.h")   # what the conditioning-set kernel TUs include
    work = []
    inst = os.path.join(CSRC, "gpv_sets_inst.hip")
    for P in sorted(only or plist(), reverse=True):          # longest compiles first
        if P >= SPLIT_FROM_P:                        # one TU per spatial dimension + the function that picks among them
            for d in (3, 2, 1, 0):
                work.append((inst, os.path.join(BUILD, f"sets_p{P}_d{d}.o"),
                             [f"-DGPV_INST_P={P}", f"-DGPV_INST_DIM={d}"] + extra_flags, kern, force))
            work.append((inst, os.path.join(BUILD, f"sets_p{P}.o"), [f"-DGPV_INST_P={P}", "-DGPV_INST_DISPATCH"] + extra_flags,
                         kern, force))
        else:
            work.append((inst, os.path.join(BUILD, f"sets_p{P}.o"), [f"-DGPV_INST_P={P}"] + extra_flags, kern, force))
    work.append((os.path.join(CSRC, "gpv_aux_kernels.hip"), os.path.join(BUILD, "aux.o"), list(extra_flags),
                 kern + H("gpv_plist.h"), force))
    work.append((os.path.join(CSRC, "gpv_api.hip"), os.path.join(BUILD, "api.o"), list(extra_flags), internal + [pub] + H("gpv_laplace.h", "gpv_generic.h", "gpv_posterior_ext.h"), force))
    work.append((os.path.join(CSRC, "gpv_posterior.hip"), os.path.join(BUILD, "posterior.o"), list(extra_flags),
                 internal + H("gpv_posterior_ext.h"), force))
    work.append((os.path.join(CSRC, "gpv_laplace.hip"), os.path.join(BUILD, "laplace.o"), [], H("gpv_laplace.h"), force))
    work.append((os.path.join(CSRC, "gpv_sets_generic.hip"), os.path.join(BUILD, "generic.o"), list(extra_flags),
                 kern + H("gpv_generic.h"), force))
    work.append((os.path.join(CSRC, "gpv_order.cpp"), os.path.join(BUILD, "order.o"), ["-x", "c++"], [pub], force))
    work.append((os.path.join(CSRC, "gpv_nn.hip"), os.path.join(BUILD, "nn.o"), ["-ffp-contract=off"], internal + [pub], force))
    jobs = jobs or min(8, os.cpu_count() or 1)
    with ThreadPoolExecutor(jobs) as ex:
        res = list(ex.map(_compile, work))
    objs = [o for o, _ in res]
    if verbose:
        for o, t in res:
            if t:
                print(f"  compiled {os.path.basename(o)} in {t:.1f}s", file=sys.stderr)
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < _newest(objs):
        cmd = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed: {r.stdout}\n{r.stderr}")
    return LIB


def dpp_hazards(obj: str):
    """Verification of a built set-kernel object: the sweeps read other lanes' registers through DPP operands of INLINE-ASM
    instructions (v_fmac_f64_dpp ... row_newbcast), which hipcc's hazard recogniser does not look into -- a VALU write of a VGPR
    needs two wait states before a DPP read of it, and under register pressure the compiler fetches parked values back from
    AGPRs (v_accvgpr_read) right in front of their use.  Disassembles the gfx950 code object inside `obj` and returns
    (DPP instructions, hazards): reads whose source register was written by one of the two preceding instructions (s_nop N
    counts as N + 1 wait states).  (None, None) for an object without device code (the per-P dispatch units)."""
    import tempfile
    llvm = "/opt/rocm/lib/llvm/bin"
    with tempfile.TemporaryDirectory() as td:
        fat, co = os.path.join(td, "x.fat"), os.path.join(td, "x.co")
        subprocess.check_call([f"{llvm}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat])
        r = subprocess.run([f"{llvm}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}",
                            f"--targets=hipv4-amdgcn-amd-amdhsa--{ARCH}", f"--output={co}"], capture_output=True, text=True)
        if r.returncode != 0:
            return None, None
        dis = subprocess.run([f"{llvm}/llvm-objdump", "-d", "--no-show-raw-insn", co], capture_output=True, text=True).stdout

    def regs(tok):
        m = re.match(r"v\[(\d+):(\d+)\]", tok)
        if m:
            return set(range(int(m.group(1)), int(m.group(2)) + 1))
        m = re.match(r"v(\d+)$", tok)
        return {int(m.group(1))} if m else set()
    ndpp = bad = 0
    for blk in re.split(r"\n(?=[0-9a-f]+ <)", dis):
        if "gpv_sets_kernel" not in blk.split("\n", 1)[0]:
            continue
        hist = []                                     # (VGPRs written, wait states the instruction provides)
        for line in blk.split("\n")[1:]:
            t = line.strip().split("//")[0].strip()
            if not t or t.endswith(":"):
                continue
            op, _, rest = t.partition(" ")
            ops = [o.strip() for o in rest.split(",")]
            if op == "s_nop":
                hist.append((set(), int(ops[0], 0) + 1))
                continue
            if "row_newbcast" in t or "row_bcast" in t or "_dpp" in op:
                ndpp += 1
                src = regs(ops[1].lstrip("-").split()[0]) if len(ops) > 1 else set()
                ws = 0
                for w, states in reversed(hist[-4:]):
                    if ws >= 2:
                        break
                    if w & src:
                        bad += 1
                        break
                    ws += states
            hist.append((regs(ops[0].split()[0]) if op.startswith("v_") and ops and ops[0] else set(), 1))
    return ndpp, bad


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--jobs", type=int, default=None)
    ap.add_argument("--tag", default="")
    ap.add_argument("--flags", default="", help="extra hipcc flags for the kernel TUs, space separated")
    ap.add_argument("--plist", default="", help="tagged builds: compile these row lengths only, e.g. 21,31,61")
    ap.add_argument("--check-dpp", action="store_true", help="scan the built set-kernel objects for DPP read hazards (dpp_hazards)")
    a = ap.parse_args()
    print(build(a.force, a.jobs, verbose=True, tag=a.tag, extra_flags=a.flags.split(),
                only=[int(x) for x in a.plist.split(",") if x] or None))
    if a.check_dpp:
        import glob
        tot = 0
        for o in sorted(glob.glob(os.path.join(BUILD, "sets_p*.o"))):
            nd, bad = dpp_hazards(o)
            if nd is not None:
                tot += bad
                print(f"  {os.path.basename(o)}: {nd} DPP instructions, {bad} hazard(s)")
        print("DPP hazards:", tot)
        sys.exit(1 if tot else 0)
