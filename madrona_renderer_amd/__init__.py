"""madrona_renderer_amd -- MI355X-native batch renderer behind the
``madrona_renderer`` API of llGuy/madrona_renderer.

The product is native: ``libmrx_hip.so`` (C-ABI + hand-written HIP kernels for
gfx950), ``libmadrona_mi355.so`` (the C++ ``madRender::Manager``) and the
compiled Python module ``madrona_renderer``.  This package only locates / loads
them and offers scene helpers; it contains no rendering code and no CPU
fallback -- loading fails loudly when the native pieces are missing.
"""
import ctypes
import importlib.machinery
import importlib.util
import os
import sys

from . import build as _build

__all__ = ["load_module", "load_capi", "lib_path", "module_path", "scenes"]

_MODULE = None
_CAPI = None


def _torch_first():
    """PyTorch-ROCm ships its own libamdhip64; libmrx_hip.so links the system
    one.  Whichever is loaded first serves both (same SONAME), and torch fails
    to initialise ("No HIP GPUs are available") when it finds the other copy
    already resident -- so torch, when present, is imported before the native
    library is mapped."""
    try:
        import torch  # noqa: F401
    except ImportError:
        pass


def lib_path():
    return _build.lib_path()


def module_path():
    return _build.module_path()


def load_module():
    """Import the compiled ``madrona_renderer`` extension (reference API) and
    register it under that name."""
    global _MODULE
    if _MODULE is not None:
        return _MODULE
    _torch_first()
    path = module_path()
    if not os.path.exists(path):
        raise ImportError(
            "native module %s is missing: run `python -c 'import __graft_entry__ "
            "as g; g.build()'` (needs hipcc); there is no Python fallback" % path)
    loader = importlib.machinery.ExtensionFileLoader("madrona_renderer", path)
    spec = importlib.util.spec_from_file_location("madrona_renderer", path, loader=loader)
    mod = importlib.util.module_from_spec(spec)
    loader.exec_module(mod)
    sys.modules.setdefault("madrona_renderer", mod)
    _MODULE = mod
    return mod


def load_capi():
    """ctypes handle on libmrx_hip.so (the C-ABI of include/mrx.h)."""
    global _CAPI
    if _CAPI is None:
        _torch_first()
        path = lib_path()
        if not os.path.exists(path):
            raise ImportError("native library %s is missing (build it with hipcc)" % path)
        _CAPI = ctypes.CDLL(path)
    return _CAPI
