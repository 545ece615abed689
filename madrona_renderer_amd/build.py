"""In-tree build of the native pieces (no cmake needed):

  libmrx_hip.so         C-ABI + HIP kernels for gfx950      (hipcc)
  libmadrona_mi355.so   madRender::Manager C++ API          (g++)
  madrona_renderer.*.so Python module, reference API        (g++ / pybind11)

Everything lands next to this file so the built objects travel with the
source tree (they are git-ignored, not gpurun-ignored).
"""
import os
import shutil
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
ARCH = "gfx950"

HIP_SOURCES = ["raster.hip", "bvh.hip", "bvh.cpp", "mrx_api.cpp", "assets.cpp", "ktx2.cpp"]
HIP_DEPS = HIP_SOURCES + ["raster.hpp", "raster_dev.hpp", "bvh.hpp", "assets.hpp", "bc7_tables.inc",
                          "../../include/mrx.h"]
MGR_SOURCES = ["manager.cpp"]
MGR_DEPS = MGR_SOURCES + ["../../include/madrona_mi355/manager.hpp",
                          "../../include/madrona_mi355/types.hpp", "../../include/mrx.h"]
PY_SOURCES = ["py_module.cpp", "manager.cpp"]


def ext_suffix():
    return sysconfig.get_config_var("EXT_SUFFIX") or ".so"


def lib_path():
    return os.path.join(HERE, "libmrx_hip.so")


def mgr_path():
    return os.path.join(HERE, "libmadrona_mi355.so")


def module_path():
    return os.path.join(HERE, "madrona_renderer" + ext_suffix())


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in deps)


def _run(cmd, verbose):
    if verbose:
        print("+", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the HIP library cannot be built")
    return exe


def build_hip(force=False, verbose=False, extra_flags=()):
    out = lib_path()
    if force or _stale(out, HIP_DEPS):
        cmd = [hipcc(), "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC",
               "-shared", "-ffp-contract=off", "-fno-fast-math",
               # per-pixel state lives in small unrolled arrays; LLVM's AMDGPU
               # promote-alloca pass turns them into 16-wide vector registers
               # whose every conditional update copies the whole tuple
               "-mllvm", "-disable-promote-alloca-to-vector",
               # the SLP vectorizer packs scalar f32 math into v_pk_* pairs and
               # pays for it in v_mov shuffles and ~40 extra VGPRs (no rate gain
               # on gfx950: packed f32 issues at the scalar-f32 rate)
               "-fno-slp-vectorize",
               # the command processor writes the first 12 dwords of a kernel's arguments into SGPRs at wave
               # launch: the group kernel's argument header (raster.hpp GroupHeader)
               "-mllvm", "-amdgpu-kernarg-preload-count=12",
               "-Wall", "-Wno-unused-function"]
        cmd += list(extra_flags)
        # e.g. MRX_EXTRA_HIPCC_FLAGS=-DMRX_BVH_DIAG=1 python -m madrona_renderer_amd.build --force
        # (the BVH kernel's in-kernel stamps and phase switches: scripts/bvh_stamps.py, bvh_ablate.py)
        cmd += os.environ.get("MRX_EXTRA_HIPCC_FLAGS", "").split()
        cmd += [os.path.join(CSRC, s) for s in HIP_SOURCES]
        cmd += ["-lz", "-ldl", "-Wl,-rpath,/opt/rocm/lib", "-o", out]
        _run(cmd, verbose)
    return out


def build_manager(force=False, verbose=False):
    out = mgr_path()
    if force or _stale(out, MGR_DEPS) or \
            os.path.getmtime(out) < os.path.getmtime(lib_path()):
        cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall"]
        cmd += [os.path.join(CSRC, s) for s in MGR_SOURCES]
        cmd += ["-L" + HERE, "-lmrx_hip", "-Wl,-rpath,$ORIGIN", "-o", out]
        _run(cmd, verbose)
    return out


def build_module(force=False, verbose=False):
    import pybind11
    out = module_path()
    if force or _stale(out, PY_SOURCES + MGR_DEPS) or \
            os.path.getmtime(out) < os.path.getmtime(lib_path()):
        cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall",
               "-fvisibility=hidden",
               "-I" + pybind11.get_include(),
               "-I" + sysconfig.get_paths()["include"]]
        cmd += [os.path.join(CSRC, s) for s in PY_SOURCES]
        cmd += ["-L" + HERE, "-lmrx_hip", "-Wl,-rpath,$ORIGIN", "-o", out]
        _run(cmd, verbose)
    return out


def headless_path():
    return os.path.join(HERE, "renderer_headless")


def build_headless(force=False, verbose=False):
    """The reference's headless CLI (src/headless.cpp) over the MI355X Manager."""
    out = headless_path()
    deps = ["headless.cpp", "assets.cpp", "assets.hpp"] + MGR_DEPS
    if force or _stale(out, deps) or os.path.getmtime(out) < os.path.getmtime(lib_path()):
        cmd = ["g++", "-O2", "-std=c++17", "-Wall",
               "-DMRX_DATA_DIR=\"%s\"" % os.path.join(ROOT, "data"),
               os.path.join(CSRC, "headless.cpp"), os.path.join(CSRC, "manager.cpp"),
               os.path.join(CSRC, "assets.cpp"), os.path.join(CSRC, "ktx2.cpp"),
               "-L" + HERE, "-lmrx_hip", "-lz", "-ldl", "-lpthread", "-Wl,-rpath,$ORIGIN", "-o", out]
        _run(cmd, verbose)
    return out


def asan_driver_path():
    return os.path.join(ROOT, "build", "host_asan_driver")


def build_asan(force=False, verbose=False):
    """AddressSanitizer + UBSan build of the host code that reads untrusted input (OBJ / MTL /
    PNG / KTX2 / BC7 readers, BLAS builder) with its driver, csrc/host_asan_driver.cpp.  Host
    compiler only (the HIP runtime header bvh.hpp pulls in is declarations); run by
    tests/test_sanitizers.py on the CPU -- never on the GPU box."""
    out = asan_driver_path()
    srcs = ["host_asan_driver.cpp", "assets.cpp", "ktx2.cpp", "bvh.cpp"]
    deps = srcs + ["assets.hpp", "bvh.hpp", "raster.hpp", "bc7_tables.inc"]
    if force or _stale(out, deps):
        os.makedirs(os.path.dirname(out), exist_ok=True)
        cmd = ["g++", "-O1", "-g", "-std=c++17", "-Wall", "-fsanitize=address,undefined",
               "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer",
               "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include"]
        cmd += [os.path.join(CSRC, s) for s in srcs]
        cmd += ["-lz", "-ldl", "-o", out]
        _run(cmd, verbose)
    return out


def build_all(force=False, verbose=False):
    build_hip(force, verbose)
    build_manager(force, verbose)
    build_module(force, verbose)
    build_headless(force, verbose)
    return lib_path(), mgr_path(), module_path()


if __name__ == "__main__":
    if "--asan" in sys.argv:
        print(build_asan(force="--force" in sys.argv, verbose=True))
    else:
        print("\n".join(build_all(force="--force" in sys.argv, verbose=True)))
