"""bench.py --gpus N started plainly is its own launcher: it
starts N ranks before touching torch or HIP, relays rank 0's line, and fails when
a rank fails.  MRX_BENCH_DRY=1 runs the ranks' distributed control flow without a
renderer (gloo), so the launcher can be exercised here, where there is no GPU."""
import json
import os
import subprocess
import sys

from tests.conftest import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def _run(args, **env):
    e = dict(os.environ, MRX_BENCH_DRY="1", **env)
    e.pop("RANK", None)
    e.pop("WORLD_SIZE", None)
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, timeout=300,
                          cwd=ROOT, env=e)


def test_gpus_2_starts_two_ranks_and_relays_one_line():
    p = _run(["--gpus", "2"])
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out["ranks_seen"] == 2 and out["n_gpus"] == 2 and out["launcher"] == "self"
    assert out["max_over_ranks"] == 2.0             # the MAX reduction saw rank 1's value
    assert out["strong_worlds_total"] == 16384      # the configs[3] shards add up


def test_a_failing_rank_fails_the_job():
    p = _run(["--gpus", "2"], MRX_BENCH_DRY_FAIL_RANK="1")
    assert p.returncode == 3
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert "rank 1 exited with code 3" in p.stderr


def test_spawn_goes_through_the_launcher_for_one_rank():
    p = _run(["--gpus", "1", "--spawn"])
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert out["ranks_seen"] == 1 and out["launcher"] == "self" and out["local_rank"] == 0


def test_the_launcher_process_never_imports_torch():
    # the parent must not initialise a GPU: it imports neither torch nor the renderer
    code = ("import sys, runpy; sys.argv = ['bench.py', '--gpus', '2']\n"
            "try:\n    runpy.run_path(%r, run_name='__main__')\n"
            "except SystemExit as e:\n    rc = e.code\n"
            "assert 'torch' not in sys.modules and 'madrona_renderer' not in sys.modules, sorted(sys.modules)[:5]\n"
            "sys.exit(rc)\n" % BENCH)
    e = dict(os.environ, MRX_BENCH_DRY="1")
    e.pop("RANK", None)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd=ROOT, env=e)
    assert p.returncode == 0, p.stderr[-2000:]
    assert json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])["ranks_seen"] == 2
