"""Renderer on its own BLOCKING stream (hipStreamCreate, default flags): speed next to other active
streams, and ordering against torch's default (null) stream."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from madrona_renderer_amd import scenes
os.environ["MRX_PLACEMENT_TRIES"] = "1"
hip = ctypes.CDLL("libamdhip64.so")
hl = scenes.synthetic_scene(4096)
r = scenes.make_renderer(hl)


def t(label):
    t0 = time.time()
    while time.time() - t0 < 0.2:
        r.time_renders(50)
    us = min(r.time_renders(400) for _ in range(5)) / 400 * 1000
    print(f"{label:70s} {us:7.2f} us", flush=True)


t("fresh, null stream")
s1 = torch.cuda.Stream()
with torch.cuda.stream(s1):
    x = torch.ones(1024, device="cuda") * 2
torch.cuda.synchronize()
t("null stream, after a torch side stream worked")
blocking = ctypes.c_void_p()
assert hip.hipStreamCreate(ctypes.byref(blocking)) == 0
r.set_stream(blocking.value)
t("own blocking stream (hipStreamCreate)")
for i in range(3):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        x = torch.ones(1024, device="cuda") * 2
    torch.cuda.synchronize()
    t(f"own blocking stream, {i + 2} torch side streams have worked")
# ordering against the null stream: a long null-stream job, then a pose write, then step() -- no host sync
pos = r.instance_position_tensor().to_torch()
before = r.depth_tensor().to_torch().clone()
y = torch.randn(4096, 4096, device="cuda")
for _ in range(30):
    y = (y @ y) * 1e-4                      # ~10 ms on the null stream
pos[:, 2] += 1.0                            # null stream, behind the matmuls
r.step()                                    # own blocking stream
after = r.depth_tensor().to_torch().clone() # null stream: must wait for the render
torch.cuda.synchronize()
r.step(); r.sync()
settled = r.depth_tensor().to_torch().clone()
print("render saw the pose write:", bool((after != before).any().item()),
      " reader on the null stream saw the finished render:", bool(torch.equal(after, settled)))
