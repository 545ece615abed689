"""Is the ~20 % 'fast mode' of the large configurations (C3 / C5) the XCD phase of the queue?
A one-world renderer on the same stream launches ONE workgroup per step and so shifts the XCD
the next launch's workgroup 0 lands on by one."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MRX_PLACEMENT_TRIES"] = "1"
os.environ["MRX_DEBUG_STAMPS"] = "1"
import numpy as np
import torch
import madrona_renderer_amd as pkg
from madrona_renderer_amd import scenes
lib = pkg.load_capi()
lib.mrx_debug_stamps.restype = ctypes.c_int64
lib.mrx_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
K = {"HL": dict(num_worlds=4096), "C3": dict(num_worlds=4096, width=128, height=128, with_wall=True),
     "C5": dict(num_worlds=4096, width=256, height=256, textured=True, render_mode="Raytracer"),
     "C2": dict(num_worlds=1024), "C4": dict(num_worlds=2048)}
name = sys.argv[1] if len(sys.argv) > 1 else "C3"
R = scenes.make_renderer(scenes.synthetic_scene(**K[name]))
os.environ.pop("MRX_DEBUG_STAMPS")
pad = scenes.make_renderer(scenes.synthetic_scene(1))
n = 30 if name == "C5" else 200
buf = np.zeros(1 << 22, np.uint64)


def xcc0():
    R.step(); R.sync()
    k = lib.mrx_debug_stamps(ctypes.c_void_p(R.native_handle()), buf.ctypes.data, buf.size)
    st = buf[:k].reshape(-1, 4, 8)
    return int((st[0, 0, 7] >> np.uint64(32)) & np.uint64(0xF))


hip = ctypes.CDLL("libamdhip64.so")
R.time_renders(3 * n)
for i in range(9):
    us = min(R.time_renders(n) for _ in range(3)) / n * 1000
    print(f"{name}: stream #{i} (0 = null): workgroup 0 on XCC {xcc0()}   {us:8.2f} us/render", flush=True)
    s_ = ctypes.c_void_p()
    assert hip.hipStreamCreateWithFlags(ctypes.byref(s_), 1) == 0
    R.set_stream(s_.value)
    R.time_renders(n)
