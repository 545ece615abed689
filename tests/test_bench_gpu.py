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
    assert out["settle_s"] == 0.25 and r["frac"] == r["frac_kernel"] and 0 < r["frac_wall"]
    assert r["traffic"] is None and r["traffic_source"] is None     # no counters for this configuration
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    # 512 views x (64*64*8 + 2*44 + 28) algorithmic bytes per launch
    assert r["bytes_per_launch"] == 512 * (64 * 64 * 8 + 2 * 44 + 28)
    # value = views per second over the timed region
    assert abs(out["value"] - 512 / (out["ms_per_step"] * 1e-3)) / out["value"] < 1e-6
    c = out["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "views/s"


@pytest.mark.gpu
def test_bench_under_torchrun_runs_rccl_for_real(native):
    # the nccl (= RCCL) path had never executed.  One rank on the one
    # GPU of this box: process-group init with device binding, barriers, the
    # MAX all-reduce of the timing and all_gather_into_tensor on the renderer's
    # DLPack tensors all go through RCCL.
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                        "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "100", "--warmup", "10",
                        "--worlds", "256", "--gather", "--no-cpu-baseline", "--no-extra"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    g = out["with_gather"]
    assert g["backend"] == "nccl" and g["own_slab_intact"] is True and g["value"] > 0
    assert out["n_gpus"] == 1 and out["roofline"]["traffic"] is None
    assert out["roofline"]["traffic_source"] is None


@pytest.mark.gpu
def test_bench_line_on_the_bvh_path(native):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "50", "--warmup", "5",
                        "--worlds", "128", "--cubes", "40", "--no-cpu-baseline", "--no-extra"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert "bvh" in out["roofline"]["kernel"] and out["roofline"]["bytes_per_launch"] == 128 * (64 * 64 * 8 + 41 * 44 + 28)
    assert out["roofline"]["frac_wall"] <= out["roofline"]["frac_kernel"] * 1.05
