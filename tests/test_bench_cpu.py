"""CPU-side pieces of bench.py (no GPU): the faithful CPU baseline leg, the PMC-summary lookup, the defaults the driver
relies on (`python bench.py` with no flags = 1 GPU, a few timed steps)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_defaults_and_faithful_baseline(monkeypatch):
    import bench
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = bench.parse()
    assert a.gpus == 1 and 1 <= a.steps <= 50 and a.warmup >= 1 and a.nodes == 1_000_000 and a.emb == 128
    r = bench.cpu_baseline_faithful(2)  # two all_score recomputations on CA-GrQc + the C oracle's walks
    assert r["value"] > 0 and r["unit"] == "edges/s" and r["kind"] == "port" and "faithful" in r["flavour"]


def test_pmc_summary_lookup_takes_the_latest_round():
    import bench
    traffic, src = bench.pmc_traffic("level_score_kernel")
    files = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc_hbm_traffic.json"))
    assert src == os.path.join("profiles", files[-1]) and traffic > 1e8
    data = json.load(open(os.path.join(ROOT, src)))
    assert any("level_score_kernel" in k for k in data)
    assert bench.pmc_traffic("no_such_kernel") == (None, None)
