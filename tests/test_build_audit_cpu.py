"""Build audits.  The wide all-pairs kernel (graphgan_amd/csrc/all_score.hip, all_score_reduce_bf16_x32_kernel) must keep its
operands in registers: the build fails if an instantiation uses scratch (csrc/check_no_scratch.sh) -- checked here on the
auditor itself: the remarks of the current build pass, a copy with one spilled instantiation fails.  And the product path
never touches the oracle."""
import os
import re
import subprocess

import pytest

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "graphgan_amd", "csrc")
REMARKS = os.path.join(CSRC, "all_score.remarks")
CHECK = os.path.join(CSRC, "check_no_scratch.sh")


@pytest.mark.skipif(not os.path.exists(REMARKS), reason="all_score.remarks is written by the build (make -C graphgan_amd/csrc)")
def test_scratch_check_accepts_the_build_and_rejects_a_spill(tmp_path):
    ok = subprocess.run(["bash", CHECK, REMARKS], capture_output=True, text=True)
    assert ok.returncode == 0, ok.stderr
    assert "no scratch, no spill" in ok.stdout
    text = open(REMARKS).read()
    m = re.search(r"(Function Name: _ZN2gg32all_score_reduce_bf16_x32_kernel.*?ScratchSize \[bytes/lane\]: )0", text, flags=re.S)
    assert m
    bad = text[:m.end() - 1] + "832" + text[m.end():]
    p = tmp_path / "bad.remarks"
    p.write_text(bad)
    res = subprocess.run(["bash", CHECK, str(p)], capture_output=True, text=True)
    assert res.returncode != 0 and "spills" in res.stderr


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
