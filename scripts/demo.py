#!/usr/bin/env python3
"""The calling sequence of the reference's scripts/test.py (4 worlds, cube.obj
plus one hand-specified triangle, Raytracer mode, 64x64, instance positions
nudged every step) against the MI355X module -- without the matplotlib window:
the last frame is written as a tiled PNG instead.

    python scripts/demo.py [steps] [out.png]
"""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

import madrona_renderer_amd
m = madrona_renderer_amd.load_module()      # == `import madrona_renderer as m`

DATA = os.path.join(ROOT, "data")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 16
out = sys.argv[2] if len(sys.argv) > 2 else "demo.png"

asset_paths = [m.ImportedAsset(path=os.path.join(DATA, "cube.obj"), mat_id=0)]
additional_mats = [m.AdditionalMaterial(color=[1, 1, 1, 1], texture_id=0, roughness=0.8,
                                        metalness=0.2)]
texture_paths = [os.path.join(DATA, "cube.png")]
instances = [
    m.ImportedInstance(position=[0.0, 0.0, 15.0], rotation=[0.707107, 0.707107, 0.0, 0.0],
                       scale=[3.0, 3.0, 3.0], object_id=0),
    m.ImportedInstance(position=[0.0, 0.0, 15.0], rotation=[0.707107, 0.707107, 0.0, 0.0],
                       scale=[10.0, 10.0, 10.0], object_id=1),
]
cameras = [m.ImportedCamera(position=[-22.343935, -21.845375, 27.061676],
                            rotation=[0.913407, -0.112268, 0.047731, -0.388336])]
num_worlds = 4
world_inits = [m.WorldInit(num_instances=2, instance_offset=0, num_cameras=1, camera_offset=0)
               for _ in range(num_worlds)]

renderer = m.MadronaRenderer(
    gpu_id=0, num_worlds=num_worlds, render_mode=m.RenderMode.Raytracer,
    batch_render_view_width=64, batch_render_view_height=64,
    asset_paths=asset_paths,
    mesh_vertices=np.array([[0, 0, 0], [5, 0, 10], [10, 0, 0]], dtype=np.float32),
    mesh_uvs=np.zeros((3, 2), dtype=np.float32),
    mesh_indices=np.array([0, 1, 2], dtype=np.uint32),
    mesh_vertex_offsets=np.array([0], dtype=np.uint32),
    mesh_indices_offsets=np.array([0], dtype=np.uint32),
    mesh_materials=np.array([-1], dtype=np.int32),
    instances=instances, materials=additional_mats, texture_paths=texture_paths,
    cameras=cameras, worlds=world_inits)

positions = renderer.instance_position_tensor().to_torch()
for _ in range(steps):
    positions[0][2] += 1.0
    positions[2][2] += 2.0
    positions[4][2] += 1.5
    positions[6][2] += 0.5
    renderer.step()
    rgb = renderer.rgb_tensor().to_torch()
    cpu = rgb.cpu()

grid_h = math.ceil(math.sqrt(num_worlds))
grid_w = math.ceil(num_worlds / grid_h)
canvas = np.zeros((grid_h * 64, grid_w * 64, 4), np.uint8)
for i in range(num_worlds):
    # Raytracer storage is [x][y]: transpose as scripts/test.py:160 does
    canvas[(i // grid_w) * 64:(i // grid_w + 1) * 64, (i % grid_w) * 64:(i % grid_w + 1) * 64] = \
        cpu[i].transpose(0, 1).numpy()
from PIL import Image
Image.fromarray(canvas).save(out)
print("rendered", steps, "steps;", tuple(rgb.shape), rgb.dtype, rgb.device, "->", out)
