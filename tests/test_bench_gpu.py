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
                        "--worlds", "512", "--cpu-views", "256", "--no-extra", "--strong-worlds", "2048"],
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
    assert "unbinned scalar oracle" in c["sample"]
    assert out["ranks_seen"] == 1 and out["launcher"] == "none"
    # outputs below 256 MiB are laid out deterministically: no candidates were timed
    assert out["placement"] == {"tries": 1, "candidates_us": [], "kept_us": None}
    st = out["also_strong"]
    assert st["scaling"] == "strong" and st["unit"] == "views/s" and "with_gather" not in st
    assert abs(st["value"] - 2048 / (st["ms_per_step"] * 1e-3)) / st["value"] < 1e-6
    assert "also" not in out and "also_bvh" not in out           # (--no-extra)


@pytest.mark.gpu
def test_bench_extras_configs1_and_the_bvh_path(native):
    # without --no-extra the line carries BASELINE configs[1] (`also`) and the BVH path on a scene that takes it by
    # itself, against its own (VALU-issue) roofline (`also_bvh`)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "100", "--warmup", "10",
                        "--worlds", "256", "--no-strong", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads([l for l in p.stdout.splitlines() if l.strip()][-1])
    assert out["also"]["unit"] == "views/s" and out["also"]["kernel_us"] > 0
    bv = out["also_bvh"]
    assert bv["render_path"] == "bvh" and bv["unit"] == "views/s" and bv["kernel_us"] > 0
    assert bv["roofline"]["bound"] == "valu-issue" and 0 < bv["roofline"]["frac"] < 1
    assert abs(bv["roofline"]["peak"] - 1228.8) < 1e-6 and isinstance(bv["roofline"]["stale"], bool)
    lp = out["also_loop"]
    assert lp["ms_per_iteration"] > lp["ms_update_alone"] > 0 and lp["iterations"] == 100
    # the same loop captured once into a hipGraph and replayed: never slower than launching it piece by piece
    assert 0 < lp["ms_per_iteration_graph_replay"] <= lp["ms_per_iteration"] * 1.05
    # BASELINE configs[2] and configs[4] (default dispatch and the BVH path the config names) at full size
    cf = out["also_configs"]
    assert [c["render_path"] for c in cf] == ["raster", "bvh", "raster"]
    assert cf[0]["bytes_per_launch"] == 4096 * (128 * 128 * 8 + 3 * 44 + 28)
    assert cf[1]["bytes_per_launch"] == cf[2]["bytes_per_launch"] == 4096 * (256 * 256 * 12 + 2 * 44 + 28)
    for c in cf:
        assert 0.3 < c["frac_kernel"] < 1.0 and c["kernel_us"] > 0 and c["placement"]["tries"] >= 1, c
    # (round 3: 845 us through the BVH path; the flat kernel: 475 - 600 by placement mode)
    assert cf[1]["kernel_us"] < 760.0
    # the one-Manager multi-shard form: host cost of a step() by how the launches are enqueued
    md = out["also_multidev"]
    assert md["eight_shards_calling_thread"] > md["eight_shards_threads"] > md["eight_shards_threads_async"] > 0
    assert md["eight_shards_threads_async"] < 1.5 * md["one_shard"] and md["device_us_per_step"] > 0


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
                        "--worlds", "256", "--gather", "--no-cpu-baseline", "--no-extra", "--no-strong"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    g = out["with_gather"]
    assert g["backend"] == "nccl" and g["own_slab_intact"] is True and g["value"] > 0
    assert out["n_gpus"] == 1 and out["roofline"]["traffic"] is None
    assert out["roofline"]["traffic_source"] is None
    assert out["ranks_seen"] == 1 and out["launcher"] == "external"


@pytest.mark.gpu
def test_bench_through_its_own_launcher_one_rank_rccl(native):
    # `bench.py --gpus N` started plainly starts the ranks itself.  One
    # rank through that same path (--spawn): the child gets RANK / LOCAL_RANK / MASTER_*
    # from the launcher and brings the RCCL group up with device binding.
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("RANK", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--spawn", "--steps", "100",
                        "--warmup", "10", "--worlds", "256", "--gather", "--no-cpu-baseline", "--no-extra",
                        "--strong-worlds", "1024"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out["launcher"] == "self" and out["ranks_seen"] == 1 and out["n_gpus"] == 1
    assert out["with_gather"]["backend"] == "nccl" and out["with_gather"]["own_slab_intact"] is True
    assert out["also_strong"]["scaling"] == "strong" and out["also_strong"]["value"] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("strong_worlds", [1024, 1023], ids=["even-shards", "ragged-shards"])
def test_bench_gpus_2_rehearsed_on_one_gpu(native, strong_worlds):
    # Two ranks started by bench.py itself, both on cuda:0 (MRX_BENCH_REHEARSAL=1: gloo for
    # the collectives, the numbers mean nothing): the N > 1 control flow end to end --
    # world shards by rank, MAX over ranks, the configs[3] strong leg with 16384 / N worlds per
    # rank and its gather through sharding.gather_slabs.
    env = dict(os.environ, MRX_BENCH_REHEARSAL="1")
    env.pop("RANK", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "100",
                        "--warmup", "10", "--worlds", "256", "--strong-worlds", str(strong_worlds)],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and out["launcher"] == "self"
    assert out["config"]["views_total"] == 512 and out["scaling"] == "weak"
    st = out["also_strong"]
    assert "512 per GPU" in st["workload"] and st["scaling"] == "strong"      # (rank 0: 512 of 1023 too)
    g = st["with_gather"]
    # ragged shards: 512 + 511 views, padded to one fused collective, trimmed
    assert g["gathered_views"] == strong_worlds and g["own_slab_intact"] is True and g["backend"] == "gloo"
    assert ("ragged" in g["collective"]) == (strong_worlds % 2 == 1)
    assert "cpu_baseline" not in out and "also" not in out


@pytest.mark.gpu
def test_bench_line_on_the_bvh_path(native):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "50", "--warmup", "5",
                        "--worlds", "128", "--cubes", "40", "--no-cpu-baseline", "--no-extra", "--no-strong"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert "bvh" in out["roofline"]["kernel"] and out["roofline"]["bytes_per_launch"] == 128 * (64 * 64 * 8 + 41 * 44 + 28)
    assert out["roofline"]["frac_wall"] <= out["roofline"]["frac_kernel"] * 1.05


@pytest.mark.gpu
def test_bvh_workload_with_committed_sq_counters_reports_its_own_roofline(native):
    # the BVH kernel is issue-bound, not HBM-bound -- for the workload whose SQ counters
    # are committed (profiles/pmc_latest.json, "sq" of 1024x64x64+cubes40) the line's roofline is the VALU issue
    # rate, with the HBM fraction kept beside it
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "100", "--warmup", "10",
                        "--worlds", "1024", "--cubes", "40", "--no-cpu-baseline", "--no-extra", "--no-strong"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    r = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])["roofline"]
    import bench
    fresh = bench.pmc_table()["1024x64x64+cubes40"]["sq"].get("kernel_hash") == bench.bvh_kernel_hash()
    if not fresh:
        # counters of another build of the kernel: the HBM roofline stays the line's own,
        # the VALU-issue estimate rides along marked stale
        assert r["bound"] == "hbm" and r["valu_issue_stale"]["stale"] is True
        assert "bvh" in r["kernel"] and 0.05 < r["frac"] < 0.5
        r = r["valu_issue_stale"]
    else:
        assert r["stale"] is False and 0.05 < r["frac_hbm"] < 0.5 and "bvh" in r["kernel"]
    # peak: 1024 SIMDs x 2.4 GHz / 2 cycles per wave64 instruction (MI355X_MICROARCH.md:54;
    # profiles/r04_valu_issue.txt)
    assert r["bound"] == "valu-issue" and r["unit"] == "G wave-instr/s" and abs(r["peak"] - 1228.8) < 1e-6
    assert 0.1 < r["frac"] < 1.0 and r["valu_per_wave"] > 300
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
