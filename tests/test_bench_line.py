"""The driver parses ONE JSON line from bench.py's stdout; round 4's 20.6 KB line came back as `parsed: null`.  The compact record
must stay below 4 KB whatever the extras hold, and must carry the contract's keys with `roofline` and `cpu_baseline`."""
import json
import os

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
            "data", "config", "roofline", "cpu_baseline")


def _full():
    out = json.load(open(os.path.join(ROOT, "profiles", "r04_bench_default.json")))       # the line that defeated the driver
    out["cpu_baseline_column"] = {"value": 1234.5678, "unit": "column solves/s", "cores": 16, "kind": "port", "sample": "s" * 400}
    out["cpu_baseline_sw"] = {"value": 8.9, "unit": "SW time-steps/s (upper bound: operator work only)", "cores": 16, "kind": "port",
                              "sample": "t" * 400, "includes": "i" * 300, "excludes": "e" * 300}
    return out


def test_compact_line_is_small_and_round_trips():
    out = _full()
    assert len(json.dumps(out)) > 15000
    rec, line = bench.compact_record(out, "gpurun_out/bench_extras.json")
    assert len(line) < 4096 and "\n" not in line
    back = json.loads(line)
    assert back == json.loads(json.dumps(rec))
    for k in CONTRACT:
        assert k in back, k
    assert back["value"] == float("%.6g" % out["value"]) and back["n_gpus"] == 1 and back["dtype"] == "f64"
    r = back["roofline"]
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_kernel_us", "bytes_per_launch", "whole_operator"):
        assert k in r, k
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0 < r["frac"] <= 1
    assert back["roofline_cold"]["frac"] > r["frac"] * 0.5
    for k in ("cpu_baseline", "cpu_baseline_column", "cpu_baseline_sw"):
        assert {"value", "unit", "cores", "kind"} <= set(back[k]), k
    assert back["cpu_baseline"]["matrix_free_value"] > 0
    assert back["summary"]["column_solves_per_s"] > 0 and back["summary"]["sw_steps_per_s_config3"] > 0
    assert "workload" in back["config"] and "model" not in back["config"]


def test_compact_line_survives_failed_and_missing_extras():
    out = _full()
    out["roofline_cold"] = {"error": "RuntimeError: " + "x" * 3000, "where": ["a"] * 3}
    out["cpu_baseline"] = {"error": "OSError: " + "y" * 3000}
    for k in ("column", "sw", "families_cold", "box_p4", "column_box_p4"):
        out.pop(k, None)
    out["extras_watchdog"] = {"note": "extras did not finish", "in_flight": "weak_scaled"}
    rec, line = bench.compact_record(out)
    assert len(line) < 4096
    back = json.loads(line)
    assert "error" in back["roofline_cold"] and "error" in back["cpu_baseline"] and back["extras_watchdog"]["in_flight"] == "weak_scaled"
    assert set(back["extras_with_errors"]) >= {"roofline_cold", "cpu_baseline"}
    # a bare headline (N > 1, every extra skipped) still makes a record
    bare = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                "vs_baseline", "dtype", "data", "config")}
    assert len(bench.compact_record(bare)[1]) < 1500
