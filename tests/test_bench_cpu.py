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
    from graphgan_amd import workloads
    assert a.roots == workloads.BENCH_ROOTS and a.continuity_roots == workloads.BENCH_ROOTS_ROUND2
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


def test_gpus_n_launches_its_own_ranks(tmp_path, monkeypatch):
    """`python bench.py --gpus N` with no launcher around it starts N ranks itself (it used to exit): every rank
    gets the environment torch.distributed.run would export, rank 0's stdout is the bench's stdout, a failing rank fails
    the whole run.  The ranks here are a stand-in script (no GPU in this container)."""
    import subprocess
    import bench
    child = tmp_path / "child.py"
    child.write_text(
        "import json, os, sys\n"
        "r = int(os.environ['RANK'])\n"
        "open(os.path.join(os.path.dirname(__file__), 'rank%d' % r), 'w').write(' '.join(sys.argv[1:]))\n"
        "print(json.dumps({k: os.environ[k] for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}))\n"
        "sys.exit(int(os.environ.get('CHILD_FAIL_RANK', '-1')) == r)\n")
    driver = ("import sys; sys.argv = ['bench.py', '--gpus', '3', '--steps', '2']; sys.path.insert(0, %r); import bench; "
              "bench.self_launch(3, %r)" % (ROOT, str(child)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, "-c", driver], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0, r.stderr
    line = json.loads(r.stdout)   # exactly one line: rank 0's
    assert line["RANK"] == "0" and line["WORLD_SIZE"] == "3" and line["MASTER_ADDR"] == "127.0.0.1" and int(line["MASTER_PORT"]) > 0
    assert sorted(f for f in os.listdir(tmp_path) if f.startswith("rank")) == ["rank0", "rank1", "rank2"]
    assert (tmp_path / "rank2").read_text() == "--gpus 3 --steps 2"
    r = subprocess.run([sys.executable, "-c", driver], capture_output=True, text=True, env=dict(env, CHILD_FAIL_RANK="1"), timeout=120)
    assert r.returncode != 0 and "rank exit codes [0, 1, 0]" in r.stderr
    # main() takes that path exactly when no launcher exported WORLD_SIZE
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert 'if "WORLD_SIZE" not in os.environ and args.gpus > 1:' in src and "must be launched with" not in src


def test_comm_model_prices_the_exchange_on_xgmi():
    """bench.comm_model: bytes per strategy, the time they take over P - 1 point-to-point links, what stays exposed and the
    weak-scaling efficiency that follows (a MODEL: no multi-GPU run exists).  Sanity of its arithmetic on the bench's counts."""
    import bench
    m = bench.comm_model(1_000_000, 128, 445_000, 488_000, 646_000, 3_630_000, step_ms=4.05, d_pass_ms=1.1, g_walk_ms=1.34)
    row = 4.0 * 130
    for P in (2, 4, 8):
        e = m["P=%d" % P]
        f = (P - 1.0) / P
        want = sum(f * t * row + f * min(1_000_000, P * t) * row for t in (445_000, 488_000))
        assert abs(e["owner_partitioned_sparse"] - want) < 1.0
        assert e["owner_partitioned_sparse_bf16"] < 0.52 * e["owner_partitioned_sparse"]           # half the row bytes (+ the ids)
        t32, t16 = e["modelled_time_owner_fp32"], e["modelled_time_owner_bf16"]
        rate = (P - 1) * bench.XGMI_LINK_GBS * 1e9 * bench.XGMI_EFFICIENCY
        assert abs(t32["d_exchange_ms"] + t32["g_exchange_ms"] - (1e3 * want / rate + 4 * bench.HOST_ROUND_TRIP_MS)) < 1e-9
        # the D exchange hides behind the G-mode walks as far as they reach; the G exchange is exposed in full
        assert abs(t32["exposed_ms_per_step"] - (max(0.0, 1.1 + t32["d_exchange_ms"] - 1.34) + t32["g_exchange_ms"])) < 1e-12
        assert 0.0 < t32["weak_scaling_efficiency_estimate"] < t16["weak_scaling_efficiency_estimate"] < 1.0
    assert m["P=8"]["modelled_time_owner_fp32"]["weak_scaling_efficiency_estimate"] < 0.8     # not near-linear at this batch: said so in DESIGN section 7
    assert m["assumptions"]["xgmi_link_GBs"] == 153.0
