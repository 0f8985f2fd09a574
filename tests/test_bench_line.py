"""The committed bench line of the round (profiles/r04_bench.json, written by bench.py on the GPU box) keeps the
driver's contract: the keys of the JSON line, the roofline object of the dominant kernel and the CPU baseline."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line(name):
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        pytest.skip(f"{name} not present")
    with open(path) as f:
        return json.loads(f.readlines()[-1])


def test_default_line_keeps_the_contract():
    d = _line("r04_bench.json")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "variants/s" and d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and d["dtype"] == "f64"
    assert "workload" in d["config"] and "model" not in d["config"]
    # value = variants of all timed steps / time
    assert abs(d["value"] * d["ms_per_step"] * 1e-3 / d["config"]["variants_per_step_per_gpu"] - 1) < 0.01
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert (r["bound"] == "hbm") == (r["unit"] == "GB/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0 < r["frac"] < 1
    assert r["traffic"] is None or r["traffic"] > 0.9 * r["algorithmic_bytes_per_launch"]
    assert 0 < r["hbm"]["frac"] < 1
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["sample"] and c["parity_ok"] is True


@pytest.mark.parametrize("name", ["r04_bench_c2.json", "r04_bench_c4.json", "r04_bench_k5.json", "r04_bench_k13.json"])
def test_other_configurations_name_their_form(name):
    d = _line(name)
    assert d["roofline"]["bounds"]["form"] in ("two planes", "three planes")
    assert d["value"] > 0 and d["roofline"]["frac"] > 0
