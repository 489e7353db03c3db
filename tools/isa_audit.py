#!/usr/bin/env python3
"""Static look at the gfx950 code of the HIP kernels for the three things round 4 found BY READING THE DISASSEMBLY of the BFS kernel
and that no counter shows directly (profiles/r4_bfs_notes.txt):

  * a wait for ALL outstanding memory operations right behind a load (`global_load ...` then `s_waitcnt vmcnt(0)` within a few
    instructions): loads the source issues "together" that the compiler serialised -- a load inside an `if`, merged with the
    not-loaded case through a copy, or a value used where it is requested;
  * LDS permutes (`ds_bpermute_b32`, what `__shfl*` compiles to) where a DPP move / rotate / lane read does the same without an
    LDS round trip;
  * scratch traffic (spilled registers).

    python tools/isa_audit.py [file.hip ...]      (default: every .hip under graphgan_amd/csrc)

Cross-compiles with hipcc (no GPU needed) and prints one line per kernel.  It does not judge: a load-then-wait pair in a prologue is
free, a permute outside a loop is noise.  It tells where to look."""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "--offload-arch=gfx950", "-munsafe-fp-atomics",
         "-S", "--cuda-device-only", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "graphgan_amd", "csrc")]
NEAR = 6  # instructions between a load and a full wait that still count as "right behind it"


def demangle(names):
    try:
        exe = next(e for e in ("/opt/rocm/lib/llvm/bin/llvm-cxxfilt", "/usr/bin/c++filt") if os.path.exists(e))
        out = subprocess.run([exe] + names, capture_output=True, text=True, check=True).stdout.split("\n")
        return [o.strip() for o in out[:len(names)]]
    except Exception:
        return names


def audit(path):
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "k.s")
        extra = ["-mllvm", "-amdgpu-mfma-vgpr-form", "-mllvm", "-pragma-unroll-threshold=200000"] if path.endswith("all_score.hip") else []  # (K7FLAGS of the Makefile)
        r = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + extra + ["-o", asm, path], capture_output=True, text=True)
        if r.returncode != 0:
            print("%s: does not compile stand-alone (%s)" % (os.path.relpath(path, ROOT), r.stderr.strip().split("\n")[-1][:120]))
            return
        text = open(asm).read()
    rows = []
    for m in re.finditer(r"^(_Z\w+):\s*; @", text, re.M):
        name = m.group(1)
        end = text.find(".Lfunc_end", m.end())
        body = [l.strip() for l in text[m.end():end].split("\n")]
        ins = [l for l in body if l and not l.startswith((".", ";", "_Z")) and not l.endswith(":")]
        if not any(l.startswith("s_endpgm") for l in ins):
            continue  # a device function, not a kernel
        loads = [i for i, l in enumerate(ins) if l.startswith(("global_load", "buffer_load", "flat_load"))]
        serial = 0
        for i in loads:
            for j in range(i + 1, min(i + 1 + NEAR, len(ins))):
                if ins[j].startswith(("global_load", "buffer_load", "flat_load")):
                    break
                if ins[j].startswith("s_waitcnt") and "vmcnt(0)" in ins[j]:
                    serial += 1
                    break
        rows.append((name, len(ins), len(loads), serial, sum(l.startswith("ds_bpermute") for l in ins),
                     sum("_dpp" in l for l in ins), sum(l.startswith("scratch_") for l in ins),
                     sum(l.startswith("s_waitcnt") and "vmcnt(0)" in l for l in ins)))
    names = demangle([r[0] for r in rows])
    print("%s" % os.path.relpath(path, ROOT))
    print("  %-78s %7s %6s %10s %9s %5s %8s %9s" % ("kernel", "instr", "loads", "load+wait0", "bpermute", "dpp", "scratch", "vmcnt(0)"))
    for r, n in zip(rows, names):
        n = re.sub(r"\(.*", "", n).replace("void ", "").replace("gg::", "")
        print("  %-78s %7d %6d %10d %9d %5d %8d %9d" % ((n[:78],) + r[1:]))


def main():
    files = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "graphgan_amd", "csrc", "*.hip")))
    for f in files:
        audit(f)


if __name__ == "__main__":
    main()
