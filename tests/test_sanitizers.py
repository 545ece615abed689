"""AddressSanitizer + UndefinedBehaviorSanitizer runs of the CPU-side code (SURVEY.md section 5,
VERDICT r2 item 9): the oracle (`make -C oracle asan`) and the host code that reads untrusted
input -- OBJ / MTL / PNG / KTX2 / BC7 readers and the BLAS builder (`python -m
madrona_renderer_amd.build --asan`).  CPU only: GPU AddressSanitizer is not available on the
pool.  A sanitizer report aborts the driver, so a zero exit status is the assertion."""
import glob
import os
import subprocess

import pytest

from madrona_renderer_amd import build
from tests.conftest import ROOT

ENV = dict(os.environ, ASAN_OPTIONS="abort_on_error=0:detect_leaks=1:exitcode=99",
           UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")


@pytest.fixture(scope="module")
def host_driver():
    return build.build_asan()


def _run(cmd, timeout=600):
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=ENV, cwd=ROOT)
    assert p.returncode == 0, (cmd, p.stdout[-1500:], p.stderr[-3000:])
    assert "Sanitizer" not in p.stderr and "runtime error" not in p.stderr, p.stderr[-3000:]
    return p.stdout


def test_oracle_under_asan_and_ubsan():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], stdout=subprocess.DEVNULL)
    out = _run([os.path.join(ROOT, "oracle", "_asan", "oracle_asan_driver")])
    assert "rasterizer:" in out and "raytracer:" in out and "checksum" in out


def test_asset_readers_on_the_shipped_files(host_driver):
    files = sorted(glob.glob(os.path.join(ROOT, "data", "*"))) + \
        sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "*.ktx2")))
    out = _run([host_driver, "parse"] + files)
    for f in files:
        assert f + ": ok" in out, out


@pytest.mark.parametrize("rel", ["data/cube.obj", "data/cube.mtl", "data/cube.png", "tests/golden/bc7_modes.ktx2",
                                 "tests/golden/bc7_modes_zlib.ktx2", "tests/golden/bc7_modes_zstd.ktx2",
                                 "tests/golden/rgba8_5x3.ktx2"])
def test_asset_readers_on_damaged_files(host_driver, rel):
    # random bytes overwritten, truncations, extreme 32-bit fields, runs of garbage: parse errors
    # are fine, memory errors are not (this found the OBJ reader walking off an embedded NUL)
    out = _run([host_driver, "fuzz", os.path.join(ROOT, rel), "300", "20261004"])
    assert "300 iterations" in out


@pytest.mark.parametrize("tris,kind", [(33, 0), (5000, 0), (4000, 1), (900, 2), (40000, 0)])
def test_blas_builder_under_sanitizers(host_driver, tris, kind):
    # random clouds, a sliver chain with exponentially growing gaps (lopsided SAH splits: the
    # recursion cap and the balanced rebuild), coincident centroids (median split by index)
    out = _run([host_driver, "blas", str(tris), "5", str(kind)])
    assert "stack bound holds" in out
