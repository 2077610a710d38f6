"""The driver parses ONE stdout line of bench.py: it must stay short (round 5's 22 KB line was not parsed) and carry the contract's
keys plus `roofline` and `cpu_baseline`.  compact_line() is exercised on committed full result objects of earlier runs."""
import glob
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

FULLS = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_bench_full.json")))


@pytest.mark.parametrize("path", FULLS, ids=[os.path.basename(p) for p in FULLS])
def test_compact_line_is_short_and_complete(path):
    import bench
    full = json.load(open(path))
    if "metric" not in full:            # (2-rank dry-run logs hold other things)
        pytest.skip("not a bench result object")
    line = bench.compact_line(full, "bench_full.json")
    assert "\n" not in line and len(line) < bench.COMPACT_LIMIT
    o = json.loads(line)
    for k in bench.COMPACT_KEYS:
        assert k in o, k
    assert o["dtype"] == "f32" and o["data"] == "synthetic" and o["higher_is_better"] is True
    assert "workload" in o["config"] and "model" not in o["config"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms_avg"):
        assert k in o["roofline"], k
    assert abs(o["roofline"]["frac"] - o["roofline"]["achieved"] / o["roofline"]["peak"]) < 1e-3
    if full.get("cpu_baseline"):
        for k in ("value", "unit", "cores", "kind", "sample"):
            assert o["cpu_baseline"][k] is not None, k
    assert abs(o["value"] - full["value"]) <= 1e-6 * full["value"]
    # no prose: every string value is short
    def strings(x):
        if isinstance(x, dict):
            for v in x.values():
                yield from strings(v)
        elif isinstance(x, str):
            yield x
    assert max(len(s_) for s_ in strings(o)) <= 160


def test_compact_line_survives_oversized_extras():
    import bench
    full = json.load(open(FULLS[-1]))
    full["comm_error"] = "x" * 100000
    full["per_rank_users_per_s"] = [1.0] * 4096
    line = bench.compact_line(full, "bench_full.json")
    assert len(line) < bench.COMPACT_LIMIT
    assert json.loads(line)["value"]


def test_gpus_mismatch_with_launcher_world_is_refused():
    """`--gpus 4` under a launcher that started 2 ranks must not print a line labelled with either number."""
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr and r.stdout.strip() == ""
