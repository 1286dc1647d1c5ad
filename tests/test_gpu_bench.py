"""bench.py keeps its contract (one JSON line with the driver's keys, `roofline` and `cpu_baseline` objects, the schedule named in
`config.workload`) in both modes - checked at a small preset so that the test takes seconds; the numbers mean nothing at that size."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra):
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU is visible")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--model", "tiny", "--batch", "4", "--tokens", "12", "--steps", "4", "--warmup", "1"] + extra,
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_default_mode_prints_the_contract_line():
    d = _run(["--cpu-windows", "1", "--cpu-tokens", "4"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 1 and d["unit"] == "audio-sec/sec" and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["value"] > 0 and abs(d["value"] - 30.0 * 4 * 4 / (d["ms_per_step"] * 4 / 1e3)) / d["value"] < 0.01
    assert "LANES" in d["config"]["workload"] and "value_one_batch_in_flight" in d and d["value_one_batch_in_flight"] > 0
    roof = d["roofline"]
    assert roof["bound"] in ("hbm", "mfma") and roof["peak"] > 0 and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
    cpu = d["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["value"] > 0 and cpu["cores"] >= 1 and "sample" in cpu
    lat = d["latency"]
    assert lat["b1_large_v3_ms_per_token"]["value"] > 0 and lat["cfg5_beam5_chunk_ms"]["value"] > 0 and lat["cfg2_small_b1_ms_per_window"]["value"] > 0


def test_pool_mode_runs_the_product_api_on_one_device():
    d = _run(["--pool", "--gpus", "1"])
    assert d["n_gpus"] == 1 and d["value"] > 0 and "--pool" in d["config"]["workload"] and d["config"]["broadcast"] == "none"
