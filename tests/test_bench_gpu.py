"""bench.py keeps the driver's contract: one JSON line with the agreed keys."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
            "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"]


@pytest.mark.gpu
def test_bench_line_has_the_contract_keys(native):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "300", "--warmup", "30",
                        "--worlds", "512", "--cpu-views", "256", "--no-extra"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    for k in REQUIRED:
        assert k in out, k
    assert out["n_gpus"] == 1 and out["steps"] == 300 and out["warmup"] == 30
    assert out["unit"] == "views/s" and out["higher_is_better"] is True and out["scaling"] == "weak"
    assert out["vs_baseline"] is None and out["data"] == "synthetic" and out["dtype"] == "f32"
    assert "workload" in out["config"] and "model" not in out["config"]
    r = out["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    # 512 views x (64*64*8 + 2*44 + 28) algorithmic bytes per launch
    assert r["bytes_per_launch"] == 512 * (64 * 64 * 8 + 2 * 44 + 28)
    # value = views per second over the timed region
    assert abs(out["value"] - 512 / (out["ms_per_step"] * 1e-3)) / out["value"] < 1e-6
    c = out["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "views/s"
