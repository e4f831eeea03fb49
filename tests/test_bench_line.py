"""bench.py's ONE stdout line stays inside what the driver captures (round 5's 20.9 KB line was cut at
~8 KB and the round's measurement was lost).  CPU: the compaction on the committed full records of earlier rounds and on an
inflated record; GPU: the real script end to end."""
import glob
import importlib.util
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
            "data", "config", "roofline", "cpu_baseline")
ROOFLINE = ("bound", "achieved", "peak", "unit", "frac", "traffic")
CPU = ("value", "unit", "cores", "kind", "sample")


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _check(line, full, limit):
    assert "\n" not in line
    assert len(line.encode()) <= limit, len(line)
    d = json.loads(line)
    for k in CONTRACT:
        assert k in d, k
    assert d["config"]["workload"]
    assert abs(d["value"] - full["value"]) <= 1e-4 * abs(full["value"])
    assert abs(d["ms_per_step"] - full["ms_per_step"]) <= 1e-4 * abs(full["ms_per_step"])
    for k in ("metric", "unit", "n_gpus", "steps", "warmup", "dtype", "scaling", "higher_is_better", "data"):
        assert d[k] == full[k], k
    if full.get("roofline"):
        for k in ROOFLINE:
            assert k in d["roofline"], k
        assert abs(d["roofline"]["frac"] - full["roofline"]["frac"]) <= 1e-4
        assert abs(d["roofline"]["achieved"] / d["roofline"]["peak"] - d["roofline"]["frac"]) <= 1e-3
    if full.get("cpu_baseline"):
        for k in CPU:
            assert k in d["cpu_baseline"], k
    return d


def test_committed_full_records_compact_under_the_limit():
    b = _bench()
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_bench_cfg*.json")))
    assert len(files) >= 10
    seen_other = False
    for f in files:
        txt = open(f).read().strip()
        if not txt.startswith("{"):
            continue
        full = json.loads(txt.splitlines()[-1])
        if "value" not in full or full.get("value") is None:
            continue
        line = b.compact_line(full, "bench_details.json")
        d = _check(line, full, b.LINE_LIMIT)
        if "other_configs" in full:
            seen_other = True
            assert set(d["other_configs"]) == set(full["other_configs"])     # every BASELINE config still on the line
            assert d["other_configs"]["cfg4"]["value"] > 0
            assert len(d["other_configs"]["cfg5"]["renders"]) == len(full["other_configs"]["cfg5"]["renders"])
    assert seen_other


def test_an_inflated_record_drops_optional_parts_never_contract_keys():
    b = _bench()
    full = json.loads(open(os.path.join(ROOT, "profiles", "r05_bench_cfg3_driver_args_20frames.json")).read().strip().splitlines()[-1])
    assert len(json.dumps(full)) > 20000          # the line that was lost
    big = json.loads(json.dumps(full))
    big["other_configs"]["cfg5"]["renders"] = big["other_configs"]["cfg5"]["renders"] * 40
    big["roofline"]["limiter"] = "x" * 5000
    big["cpu_baseline"]["sample"] = "y" * 5000
    big["config"]["workload"] = "cfg3: " + "z" * 3000
    line = b.compact_line(big, "bench_details.json")
    d = _check(line, big, b.LINE_LIMIT)
    assert "other_configs" not in d and "roofline_stages" in d
    # a tighter limit drops more, still valid
    line = b.compact_line(big, None, limit=2500)
    _check(line, big, 2500)


def test_details_file_round_trips(tmp_path):
    b = _bench()
    full = json.loads(open(os.path.join(ROOT, "profiles", "r05_bench_cfg3_driver_args_20frames.json")).read().strip().splitlines()[-1])
    p = b.write_details(full, str(tmp_path / "sub" / "details.json"))
    assert p and json.load(open(p)) == full


@pytest.mark.gpu
def test_real_bench_output_is_one_short_parseable_line(tmp_path):
    det = str(tmp_path / "details.json")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--cpu-budget", "1",
                        "--repeats", "2", "--no-other-configs", "--details", det], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines[:3]
    assert len(lines[0].encode()) <= 6000
    d = json.loads(lines[0])
    full = json.load(open(det))
    _check(lines[0], full, 6000)
    assert d["roofline"]["frac"] > 0 and d["cpu_baseline"]["value"] > 0
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
