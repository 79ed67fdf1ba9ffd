"""bench.py's ONE stdout line must stay driver-parseable: the driver keeps an 8 KB tail of stdout, and round 5's 21 KB line
came back as `parsed: null`.  Built here from canned full records -- no GPU."""
import io
import json
import os
import sys
from contextlib import redirect_stderr, redirect_stdout

import pytest

from conftest import ROOT

sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _full_records():
    d = os.path.join(ROOT, "profiles")
    return sorted(f for f in os.listdir(d) if f.endswith("_bench.json") and f[:3] in ("r05", "r06"))


@pytest.mark.parametrize("name", _full_records())
def test_compact_line_is_small_and_complete(name, tmp_path):
    full = json.load(open(os.path.join(ROOT, "profiles", name)))
    if "full_record" in full and "roofline" in full and "peak_def" not in (full["roofline"] or {}) and "legs" in full:
        pytest.skip("already a compact line")
    out, err = io.StringIO(), io.StringIO()
    with redirect_stdout(out), redirect_stderr(err):
        bench.emit(full, str(tmp_path / "bench_full.json"))
    lines = out.getvalue().splitlines()
    assert len(lines) == 1, "exactly one stdout line"
    line = lines[-1]
    assert len(line) <= bench.COMPACT_LIMIT < 6000, len(line)
    c = json.loads(line)
    assert json.loads(json.dumps(c)) == c
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in c, k
    assert c["value"] == full["value"] and c["ms_per_step"] == full["ms_per_step"] and c["config"]["workload"]
    r = c["roofline"]
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_us", "notional_hbm_frac",
              "step_frac", "step_notional_hbm_frac"):
        assert k in r, k
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 2e-3
    b = c["cpu_baseline"]
    assert b["value"] > 0 and b["cores"] >= 1 and b["kind"] in ("port", "reference") and b["sample"]
    assert c["host_span"]["value"] > 0 and c["host_span"]["vs_device_resident"] > 0
    # the full record went to the side file and to stderr, complete
    assert json.load(open(tmp_path / "bench_full.json"))["roofline"]["per_kernel"]
    assert "bench full record: {" in err.getvalue()


def test_compact_line_sheds_optional_groups_rather_than_growing():
    full = json.load(open(os.path.join(ROOT, "profiles", _full_records()[0])))
    full["config"]["workload"] = "w" * 1500                       # an absurd mandatory field
    full["arithmetics"] = {("arith%d" % i): {"value": 1.0, "ms_per_step": 1.0, "roofline": {"kernel": "k" * 60}} for i in range(12)}
    line = json.dumps(bench.compact_line(full))
    assert len(line) <= bench.COMPACT_LIMIT
    c = json.loads(line)
    assert "roofline" in c and "cpu_baseline" in c and c["value"] == full["value"]


def test_compact_line_without_optional_legs():
    """A multi-rank / --no-extras run has no cpu_baseline, no arithmetics, no legs: still one valid line."""
    full = json.load(open(os.path.join(ROOT, "profiles", _full_records()[0])))
    for k in ("cpu_baseline", "arithmetics", "host_span"):
        full[k] = None
    for k in ("single_fragment", "batch_4", "batch_8", "e2e_extract_features", "host_span_auto_batch", "sharded_pipeline"):
        full["config"].pop(k, None)
    full["rccl"] = {"backend": "nccl", "rccl_ranks": 8, "per_rank_ms_per_step": [1.2] * 8, "gather_crc_ok": None}
    c = bench.compact_line(full)
    assert c["cpu_baseline"] is None and c["rccl"]["rccl_ranks"] == 8 and len(json.dumps(c)) < 3000
