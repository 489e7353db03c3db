"""The wide all-pairs kernel (graphgan_amd/csrc/all_score.hip, all_score_reduce_bf16_x16_kernel) holds data of hand-placed
loads in registers the compiler believes are already written; the build audits the generated assembly for anything that
touches such a register while its load is in flight (csrc/audit_inflight_regs.py).  This checks the auditor itself: the
assembly of the current build passes, and a copy with one injected register read fails."""
import os
import re
import subprocess
import sys

import pytest

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "graphgan_amd", "csrc")
ASM = os.path.join(CSRC, "all_score.s")
AUDIT = os.path.join(CSRC, "audit_inflight_regs.py")


@pytest.mark.skipif(not os.path.exists(ASM), reason="all_score.s is written by the build (make -C graphgan_amd/csrc)")
def test_auditor_accepts_the_build_and_rejects_a_touched_register(tmp_path):
    ok = subprocess.run([sys.executable, AUDIT, ASM], capture_output=True, text=True)
    assert ok.returncode == 0, ok.stderr
    assert "8 instantiations" in ok.stdout
    lines = open(ASM).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_ZN2gg32all_score_reduce_bf16_x16_kernelILi16ELi4ELb1E.*:", l))
    header = next(i for i in range(start, len(lines)) if "Loop Header" in lines[i] or "Inner Loop" in lines[i])
    load = [i for i in range(header, header + 3000) if "global_load_dwordx4" in lines[i] and "#ASMSTART" in lines[i - 1]][3]
    reg = re.search(r"v\[(\d+):", lines[load]).group(1)
    bad = lines[:load + 3] + ["\tv_mov_b32_e32 v250, v%s" % reg] + lines[load + 3:]
    p = tmp_path / "bad.s"
    p.write_text("\n".join(bad))
    res = subprocess.run([sys.executable, AUDIT, str(p)], capture_output=True, text=True)
    assert res.returncode != 0 and "in flight" in res.stderr


def test_product_path_never_touches_the_oracle():
    """oracle/ is test infrastructure: nothing under graphgan_amd/ may import, load or execute it (a product path that routed
    through the CPU restatement would void every parity claim), and bench.py may name it only inside its cpu_baseline legs."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pat = re.compile(r"(from\s+oracle\b|import\s+oracle\b|oracle[/\\.]|walk_oracle|libwalk_oracle)")
    for dirpath, _, files in os.walk(os.path.join(root, "graphgan_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h", ".sh")) or f == "Makefile":
                src = open(os.path.join(dirpath, f), errors="replace").read()
                hits = [ln for ln in src.splitlines() if pat.search(ln) and not ln.lstrip().startswith(("#", "//", "*"))]
                # comments may cite the oracle (what a kernel is checked against); code may not use it
                hits = [ln for ln in hits if "import" in ln or "CDLL" in ln or "dlopen" in ln or "open(" in ln]
                assert not hits, (os.path.join(dirpath, f), hits[:3])
    bench = open(os.path.join(root, "bench.py")).read()
    for m in re.finditer(r"^\s*from oracle import|^\s*import oracle", bench, flags=re.M):
        # every import sits inside one of the CPU-baseline functions (all three run behind the timed region)
        head = bench[:m.start()]
        fn = re.findall(r"^def (\w+)\(", head, flags=re.M)[-1]
        assert fn in ("cpu_baseline", "cpu_baseline_faithful", "cpu_baseline_threads"), fn
