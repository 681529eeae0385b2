"""The committed driver-facing line (profiles/r02_bench_line.json = `python bench.py` on one MI355X) against the bench contract: required keys,
BASELINE.json's metric / config, and the arithmetic that ties value, ms_per_step and the roofline object together (SURVEY.md §8(d):
32^3 f32 br=1 beta=0 is 65 536 flop and 12 288 algorithmic bytes per problem)."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LINE = os.path.join(ROOT, "profiles", "r02_bench_line.json")
pytestmark = pytest.mark.skipif(not os.path.exists(LINE), reason="no committed bench line")


@pytest.fixture(scope="module")
def line():
    return json.load(open(LINE))


def test_contract_keys(line):
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["unit"] == "GFLOP/s" and line["higher_is_better"] is True and line["scaling"] == "weak"
    assert line["dtype"] == "f32" and line["data"] == "synthetic" and line["vs_baseline"] is None     # BASELINE.md has no MI355X number
    assert "workload" in line["config"] and "model" not in line["config"]
    assert line["verified"] is True and line["verify"]["normf_rel_max"] < 1e-5
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in line["roofline"], key
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in line["cpu_baseline"], key
    assert line["cpu_baseline"]["kind"] in ("reference", "port")


def test_metric_is_baselines(line):
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    text = json.dumps(base).lower()
    assert "gflop" in text and "brgemm" in text
    assert "32" in line["metric"] and "f32" in line["metric"].lower()
    assert line["config"]["per_gpu_batch"] == 4096                     # configs[1]: fp32 32^3, batch 4096


def test_value_and_roofline_arithmetic(line):
    batch, flops, nbytes = line["config"]["per_gpu_batch"], 2 * 32 ** 3, 3 * 32 * 32 * 4
    us = line["ms_per_step"] * 1e3
    assert line["value"] == pytest.approx(batch * flops / us * 1e-3, rel=2e-3)                 # GFLOP/s from the step time
    r = line["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s"
    assert r["algorithmic_bytes_per_launch"] == batch * nbytes
    assert r["achieved"] == pytest.approx(batch * nbytes / r["kernel_us"] * 1e-3, rel=2e-3)    # GB/s from the kernel time
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=2e-3)
    assert 0.95 < r["traffic"] / r["algorithmic_bytes_per_launch"] < 1.10                      # PMC traffic: no wasted re-reads
    assert r["kernel_us"] <= us * 1.02                                                         # a launch cannot take longer than a step
    assert line["timed_region_s"] >= 0.5


def test_every_sweep_entry_was_verified(line):
    for group in ("sweep", "reuse", "ragged"):
        for label, e in line[group].items():
            assert e["verified"] is True, (group, label)
            assert e["frac_hbm"] == pytest.approx(e["GB/s"] / 8000.0, abs=2e-3), (group, label)
