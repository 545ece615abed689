import os, sys
sys.path.insert(0, os.getcwd())
import torch
from madrona_renderer_amd import scenes
os.environ["MRX_SHARD_TRACE"] = "1"
os.environ["MRX_SHARD_THREADS"] = "2"
for worlds in (16384, 512):
    for n in (2, 4, 8):
        d = scenes.synthetic_scene(worlds)
        r = scenes.make_renderer(d, device_ids=[0] * n)
        ss = [torch.cuda.Stream() for _ in range(n)]
        for i, s in enumerate(ss):
            r.set_stream(s.cuda_stream, shard=i)
        r.time_steps_host(40)
        print(worlds, n, [round(r.time_steps_host(40), 2) for _ in range(5)], flush=True)
        del r
