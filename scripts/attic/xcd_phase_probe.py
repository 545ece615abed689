"""Does the XCD a launch's workgroup 0 lands on depend on the stream (hardware queue)?
Headline batch; MRX_DEBUG_STAMPS records HW_REG_XCC_ID per workgroup."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MRX_DEBUG_STAMPS"] = "1"
os.environ["MRX_PLACEMENT_TRIES"] = "1"
import numpy as np
import torch
import madrona_renderer_amd as pkg
from madrona_renderer_amd import scenes
hip = ctypes.CDLL("libamdhip64.so")
lib = pkg.load_capi()
lib.mrx_debug_stamps.restype = ctypes.c_int64
lib.mrx_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
r = scenes.make_renderer(scenes.synthetic_scene(4096))


def look(label):
    t0 = time.time()
    while time.time() - t0 < 0.1:
        r.time_renders(50)
    us = min(r.time_renders(300) for _ in range(3)) / 300 * 1000
    xs = []
    for _ in range(4):
        r.step(); r.sync()
        buf = np.zeros(1024 * 4 * 8, np.uint64)
        n = lib.mrx_debug_stamps(ctypes.c_void_p(r.native_handle()), buf.ctypes.data, buf.size)
        st = buf[:n].reshape(-1, 4, 8)
        xcc = ((st[:, 0, 7] >> np.uint64(32)) & np.uint64(0xF)).astype(int)
        xs.append((int(xcc[0]), int(((xcc - xcc[0]) % 8 != (np.arange(len(xcc)) % 8)).sum())))
    print(f"{label:28s} {us:7.2f} us   XCC of workgroup 0 over four launches (and workgroups off the round-robin): {xs}", flush=True)


look("null stream")
for i in range(6):
    s = ctypes.c_void_p()
    assert hip.hipStreamCreateWithFlags(ctypes.byref(s), 1) == 0
    r.set_stream(s.value)
    look(f"created stream #{i + 1}")
