"""The driver's line (bench.py: driver_line / emit).  Round 5's line carried nested objects inside `config` and had grown to
20 KB; the driver recorded `"parsed": null` for it (BENCH_r05.json).  These tests hold the line to the format the driver
did parse (BENCH_r04.json): one strict-JSON line under 8 KB whose `config` holds scalars only, with `roofline` and
`cpu_baseline` present, built here from round 5's full 20 KB record and from degenerate records."""
import io
import json
import math
import os
import sys
from contextlib import redirect_stdout

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

CANNED = os.path.join(ROOT, "profiles", "r05", "bench_driver_with_legs.json")


def _no_constants(token):
    raise ValueError("non-finite JSON constant in the driver's line: " + token)


def _canned():
    lines = [ln for ln in open(CANNED).read().splitlines() if ln.startswith("{")]
    return json.loads(lines[-1])


def _check(line_text):
    assert "\n" not in line_text
    assert len(line_text.encode()) < bench.LINE_LIMIT
    line = json.loads(line_text, parse_constant=_no_constants)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in line, k
    for k, v in line["config"].items():
        assert v is None or isinstance(v, (str, int, float, bool)), (k, v)
    assert "workload" in line["config"]
    for k, v in line.items():
        if k not in ("config", "roofline", "cpu_baseline"):
            assert v is None or isinstance(v, (str, int, float, bool)), (k, type(v))
    for obj in ("roofline", "cpu_baseline"):
        for k, v in (line.get(obj) or {}).items():
            assert v is None or isinstance(v, (str, int, float, bool)), (obj, k)
    rl = line["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rl, k
    assert rl["frac"] == pytest.approx(rl["achieved"] / rl["peak"])
    return line


def test_line_from_round5_full_record():
    res = _canned()
    assert len(json.dumps(res)) > 16000            # the record that was not parsed
    line = _check(json.dumps(bench.driver_line(res, "gpurun_out/bench_full_humanoid_n1.json"), allow_nan=False))
    assert line["value"] == res["value"]
    assert line["cpu_baseline"]["value"] == res["cpu_baseline"]["value"]
    assert line["cpu_baseline"]["kind"] == "reference" and line["cpu_baseline"]["cores"] == res["cpu_baseline"]["cores"]
    assert line["cpu_baseline"]["testspeed_value"] == res["cpu_baseline"]["testspeed_regime"]["value"]
    # the legs' heads travel as top-level scalars
    for name in ("cube", "flex", "slider_crank"):
        assert line[f"leg_{name}_value"] == res["configs"][name]["value"]
        assert line[f"leg_{name}_parity_ok"] is True
    assert line["parity_ok"] is True and line["configs_ok"] is True
    assert line["testspeed_regime_value"] == res["testspeed_regime"]["value"]
    assert line["full_record"].endswith(".json")


def test_line_survives_non_finite_and_failed_legs():
    res = _canned()
    res["roofline"]["traffic"] = float("nan")
    res["parity_sample"]["reference_glibc"]["identical_input_steps"]["max_rel_err"] = float("inf")
    res["configs"]["cube"] = {"error": "x" * 5000, "returncode": 1}
    res["api_regime"] = {"error": "boom" * 200}
    res["cpu_baseline"]["sample"] = "s" * 4000
    res["config"]["mapping"] = "m" * 4000
    line = _check(json.dumps(bench.driver_line(res, None), allow_nan=False))
    assert line["roofline"]["traffic"] is None
    assert line["parity_glibc_max_rel_err"] is None
    assert len(line["leg_cube_error"]) <= 200


def test_line_size_gate_drops_detail_not_contract_keys():
    res = _canned()
    for i in range(40):                              # many more legs than there are configurations
        res["configs"][f"extra{i}"] = dict(res["configs"]["cube"])
    line = _check(json.dumps(bench.driver_line(res, None), allow_nan=False))
    assert "cpu_baseline" in line and "value" in line


def test_minimal_record_without_extras():
    res = {k: _canned()[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                     "scaling", "vs_baseline", "dtype", "data", "config", "roofline")}
    res["n_gpus"] = 8
    _check(json.dumps(bench.driver_line(res, None), allow_nan=False))


def test_emit_prints_one_parsable_last_line_and_writes_the_full_record(tmp_path, monkeypatch):
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    buf = io.StringIO()
    with redirect_stdout(buf):
        print("noise before the line")
        bench.emit(_canned(), "humanoid")
    last = buf.getvalue().rstrip("\n").splitlines()[-1]
    line = _check(last)
    full = json.load(open(os.path.join(str(tmp_path), line["full_record"])), parse_constant=_no_constants)
    assert full["configs"]["flex"]["value"] == line["leg_flex_value"]
    assert math.isfinite(line["roofline"]["frac"])
