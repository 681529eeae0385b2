"""The driver-facing output of `python bench.py` against the bench contract.

Two committed artefacts of one run on one MI355X: the full record (`profiles/rNN_bench_detail.json`, what bench.py writes to bench_detail.json
and stderr) and the compact line (`profiles/rNN_bench_line.json`, the LAST stdout line: the only thing the driver parses, from an 8 KB tail of
stdout -- round 2's 20 KB line arrived without its head and was recorded as `parsed: null`).  Checked: the line fits, required keys, BASELINE.json's
metric / config, the arithmetic that ties value, ms_per_step and the roofline object together (32^3 f32 br=1 beta=0 is 65 536 flop
and 12 288 algorithmic bytes per problem), and that the compaction of the full record reproduces the committed line."""
import glob
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def latest(pattern):
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    return found[-1] if found else None


DETAIL = latest("r*_bench_detail.json") or latest("r02_bench_line.json")      # round 2 committed the full record under the `line` name
LINE = latest("r*_bench_line.json")
pytestmark = pytest.mark.skipif(DETAIL is None, reason="no committed bench record")


@pytest.fixture(scope="module")
def detail():
    return json.load(open(DETAIL))


@pytest.fixture(scope="module")
def line(detail):
    """the committed compact line; for a record that predates the compact format, what bench.compact_line makes of it"""
    import bench
    got = json.load(open(LINE))
    if "sweep_fields" not in got and "configs" not in got and "detail" not in got:
        return bench.compact_line(detail, os.path.join(ROOT, "bench_detail.json"))
    return got


def test_line_fits_the_driver_tail(detail, line):
    import bench
    assert len(json.dumps(line, separators=(",", ":"))) < 4608                       # the driver keeps an 8 KB tail
    # the worst case the code can produce: every optional object present, long kernel names and sample texts
    fat = dict(detail)
    fat.setdefault("configs", {k: {"frac_hbm": 0.7123, "verified": True, "cpu_baseline": {"value": 123456.78}} for k in
                               ("c3_csr15", "c3_csr10", "c3_fsspmdm", "c3_fsspmdm_n1e6", "c4_bcsc_bf16", "c4_bcsc_f32", "c4_bcsc_u8i8", "c5_fused", "variantB_f32_m32_br4096",
                                "variantB_f32_m32_br65536")})
    fat["cpu_baseline"] = dict(fat.get("cpu_baseline") or {"value": 1.0, "unit": "GFLOP/s", "cores": 1, "kind": "reference"}, sample="x" * 500)
    fat["config"] = dict(fat["config"], kernel="k" * 120, workload="w" * 200)
    fat["pipelined"] = dict({"lanes": 4}, **{f"{dt}_m{m}_b4096": {"frac_hbm": 0.87654, "verified": True} for dt in ("f32", "bf16") for m in (16, 23, 32, 64)})
    fat["mfma_power_roof_TF"] = {"bf16": 1692.8, "f32": 143.9}
    fat["tpp"] = {k: {"frac_hbm": 0.7123, "verified": True, "cpu_baseline": {"GB/s": 123.45}} for k in
                  ("copy_f32", "transpose_f32", "vnni2_bf16", "c5_bias_add_tiles", "c5_relu_tiles", "reduce_rows_f32", "reduce_cols_f32", "gather_cols_f32", "meqn_simple_f32", "packed_gemm_9x9x9")}
    text = json.dumps(bench.compact_line(fat, os.path.join(ROOT, "bench_detail.json")), separators=(",", ":"))
    assert len(text) < 5632, len(text)


def test_contract_keys(line):
    for key in ("metric", "value", "unit", "n_gpus", "steps", "steps_timed", "warmup", "ms_per_step", "timed_region_s", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "verified", "pct_mfma_peak"):
        assert key in line, key
    assert line["unit"] == "GFLOP/s" and line["higher_is_better"] is True and line["scaling"] == "weak"
    assert line["dtype"] == "f32" and line["data"] == "synthetic" and line["vs_baseline"] is None     # BASELINE.md has no MI355X number
    assert "workload" in line["config"] and "kernel" in line["config"] and "model" not in line["config"]
    assert line["verified"] is True
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel_us", "algorithmic_bytes_per_launch"):
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
    if r["traffic"] is not None:
        assert 0.95 < r["traffic"] / r["algorithmic_bytes_per_launch"] < 1.10                  # PMC traffic: no wasted re-reads
    assert r["kernel_us"] <= us * 1.02                                                         # a launch cannot take longer than a step
    assert line["timed_region_s"] >= 0.5


def test_every_sweep_entry_was_verified(detail, line):
    for group in ("sweep", "reuse", "ragged"):
        for label, e in detail[group].items():
            assert e["verified"] is True, (group, label)
            assert e["frac_hbm"] == pytest.approx(e["GB/s"] / 8000.0, abs=2e-3), (group, label)
            assert line[group][label][0] == pytest.approx(e["frac_hbm"], abs=1e-3), (group, label)     # the compact pair is [frac_hbm, pct_mfma_peak]
            assert line[group][label][1] == pytest.approx(e["pct_mfma_peak"], abs=0.06), (group, label)
    assert line["sweep_verified"] is True


def test_baseline_configs_are_in_the_line(detail, line):
    """BASELINE configs #3 / #4 / #5 run inside bench.py (round-2 review: they were builder-run only) -- records from round 3 on"""
    if "configs" not in detail:
        pytest.skip("record predates the configs leg")
    for key in ("c3_csr15", "c3_csr10", "c3_fsspmdm", "c4_bcsc_bf16", "c5_fused"):
        e = detail["configs"][key]
        assert "error" not in e, (key, e)
        assert e["verified"] is True, key
        assert line["configs"][key][0] == pytest.approx(e["frac_hbm"], abs=1e-4)
        assert e["frac_hbm"] == pytest.approx(e["algorithmic_bytes_per_launch"] / e["us_per_launch"] * 1e-3 / 8000.0, rel=2e-3)
    assert line["configs_verified"] is True


def _run_plain(argv, extra_env=None):
    """the PLAIN command (no launcher), dry: ranks, barriers and the one JSON line with no device"""
    import subprocess
    env = dict(os.environ, BENCH_DRY="1", BENCH_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env or {})
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    last = [ln for ln in res.stdout.splitlines() if ln.strip()][-1]
    return json.loads(last), res


@pytest.mark.parametrize("n", [2, 3])
def test_gpus_n_starts_n_ranks_by_itself(n):
    """`python bench.py --gpus N --steps K --warmup W` -- the command shape of BENCH_rNN.json.cmd -- must yield N ranks without an external launcher
    (round 4: the flag was parsed and dropped, so a scaling curve could not be produced).  Dry: gloo, no kernel."""
    line, res = _run_plain(["--gpus", str(n), "--steps", "5", "--warmup", "1"])
    assert line["n_gpus"] == n and line["dry"] is True
    assert line["problems_owned_by_all_ranks"] == n * 4096            # weak scaling: every rank owns its own batch
    assert "torch.distributed.run" in res.stderr and f"--nproc-per-node={n}" in res.stderr


def test_gpus_n_config5_shards_the_total():
    line, _ = _run_plain(["--gpus", "2", "--config", "5", "--total", "1001", "--gather"])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["problems_owned_by_all_ranks"] == 1001


def test_gpus_1_is_one_process():
    """N = 1 stays the plain single-process run (no launcher, no process group)"""
    line, res = _run_plain(["--gpus", "1", "--steps", "5", "--warmup", "1"])
    assert line["n_gpus"] == 1 and "torch.distributed.run" not in res.stderr


def test_gpus_n_without_devices_refuses(monkeypatch):
    """not dry, no GPUs: `--gpus 2` must refuse loudly instead of measuring one device twice"""
    import subprocess
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("box has the devices")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "BENCH_DRY", "BENCH_DEVICE")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 2 and "needs 2 visible GPUs" in res.stderr


def test_launcher_environment_wins():
    """under the driver's own launcher (WORLD_SIZE set) bench.py must NOT spawn again"""
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert '"WORLD_SIZE" not in os.environ and args.gpus > 1' in src
