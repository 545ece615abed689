"""Procedural meshes and many-instance worlds for the BVH-path tests: the
reference ships only 12-triangle assets (data/*.obj), so the large worlds the
ray-trace path exists for (BASELINE configs[4]) are
generated here, deterministically."""
import math
import os

import numpy as np

from madrona_renderer_amd import scenes

CUBE = os.path.join(scenes.DATA_DIR, "cube.obj")
PLANE = os.path.join(scenes.DATA_DIR, "plane.obj")
IDENT = (1.0, 0.0, 0.0, 0.0)


def grid_mesh(nu, nv, fn):
    """(verts [N,3], uvs [N,2], indices [K]) of a (nu x nv)-quad parametric
    surface fn(u, v) -> xyz with u, v in [0, 1]; two triangles per quad."""
    us, vs = np.meshgrid(np.linspace(0, 1, nu + 1), np.linspace(0, 1, nv + 1), indexing="ij")
    verts = np.asarray([fn(u, v) for u, v in zip(us.ravel(), vs.ravel())], np.float32)
    uvs = np.stack([us.ravel() * 4.0, vs.ravel() * 4.0], axis=1).astype(np.float32)
    idx = []
    for i in range(nu):
        for j in range(nv):
            a = i * (nv + 1) + j
            b, c, d = a + 1, a + nv + 1, a + nv + 2
            idx += [a, c, b, b, c, d]
    return verts, uvs, np.asarray(idx, np.uint32)


def sphere(nu=32, nv=16, r=1.0):
    def f(u, v):
        th, ph = 2 * math.pi * u, math.pi * v
        return (r * math.sin(ph) * math.cos(th), r * math.sin(ph) * math.sin(th), r * math.cos(ph))
    return grid_mesh(nu, nv, f)


def torus(nu=48, nv=24, R=2.0, r=0.7):
    def f(u, v):
        th, ph = 2 * math.pi * u, 2 * math.pi * v
        return ((R + r * math.cos(ph)) * math.cos(th), (R + r * math.cos(ph)) * math.sin(th),
                r * math.sin(ph))
    return grid_mesh(nu, nv, f)


def terrain(n=48, size=30.0, seed=5):
    rng = np.random.default_rng(seed)
    k = rng.uniform(0.5, 3.0, size=(4, 2))
    ph = rng.uniform(0, 6.28, size=4)

    def f(u, v):
        x, y = (u - 0.5) * size, (v - 0.5) * size
        z = sum(0.6 * math.sin(k[i, 0] * x * 0.4 + k[i, 1] * y * 0.4 + ph[i]) for i in range(4))
        return (x, y, z)
    return grid_mesh(n, n, f)


def pack_meshes(meshes):
    """[(verts, uvs, indices, material)] -> the raw-geometry kwargs of SceneDesc."""
    verts, uvs, idx, voff, ioff, mats = [], [], [], [], [], []
    nv = ni = 0
    for v, t, i, m in meshes:
        voff.append(nv)
        ioff.append(ni)
        verts.append(v)
        uvs.append(t)
        idx.append(i)
        mats.append(m)
        nv += len(v)
        ni += len(i)
    return dict(mesh_vertices=np.concatenate(verts).astype(np.float32),
                mesh_uvs=np.concatenate(uvs).astype(np.float32),
                mesh_indices=np.concatenate(idx).astype(np.uint32),
                mesh_vertex_offsets=np.asarray(voff, np.uint32),
                mesh_indices_offsets=np.asarray(ioff, np.uint32),
                mesh_materials=np.asarray(mats, np.int32))


def random_quat(rng):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    return tuple(float(np.float32(x)) for x in q)


# (lives in the package: bench.py uses it for its --cubes workloads)
cube_field = scenes.cube_field


def mesh_worlds(num_worlds, width=64, height=64, mode="Rasterizer", seed=3):
    """Worlds of large meshes (terrain 9800 + torus 2304 + two spheres of 1024 triangles = 14,152 per
    world, three-level BLASes), one camera per world on a ring: the shape of world the BLAS
    traversal exists for (perf scripts; tests/test_bvh_gpu.py has the parity scene)."""
    sph, tor, ter = sphere(32, 16), torus(48, 24), terrain(70)
    geo = pack_meshes([(sph[0], sph[1], sph[2], 0), (tor[0], tor[1], tor[2], 1), (ter[0], ter[1], ter[2], 2)])
    inst, cams, worlds = [], [], []
    for w in range(num_worlds):
        rng = np.random.default_rng(seed * 7919 + w)
        i0 = len(inst)
        inst.append(((0.0, 0.0, -1.0), IDENT, (1.0, 1.0, 1.0), 2))
        inst.append(((0.0, 0.0, 2.5), random_quat(rng), (1.5, 1.0, 0.7), 1))
        for _ in range(2):
            p = rng.uniform(-6, 6, 2)
            inst.append(((float(np.float32(p[0])), float(np.float32(p[1])), 2.0), random_quat(rng), (1.2, 1.2, 1.2), 0))
        az = float(rng.uniform(0, 2 * math.pi))
        r, h = float(rng.uniform(9, 20)), float(rng.uniform(3, 12))
        eye = tuple(float(np.float32(x)) for x in (r * math.cos(az), r * math.sin(az), h))
        cams.append((eye, scenes.look_at(eye, (0.0, 0.0, 1.0))))
        worlds.append((4, i0, 1, len(cams) - 1))
    return scenes.SceneDesc(
        num_worlds=num_worlds, render_mode=mode, width=width, height=height, asset_paths=[],
        materials=[((0.9, 0.8, 0.7, 1.0), -1, 0.5, 0.5), ((0.3, 0.5, 0.9, 1.0), -1, 0.5, 0.5),
                   ((0.4, 0.7, 0.3, 1.0), -1, 0.5, 0.5)],
        instances=inst, cameras=cams, worlds=worlds, **geo)


def mesh_scene_random_cameras(seed, width, height, mode):
    """Two worlds of terrain / torus / spheres (three-level BLASes) seen by eight random cameras: on the
    terrain, inside or next to an object's box, on a ring, anywhere looking anywhere -- the cases the BVH
    path's behind-the-eye culling and tile-corner classification decide."""
    rng = np.random.default_rng(seed)
    sph, tor, ter = sphere(24, 12), torus(32, 16), terrain(40)
    geo = pack_meshes([(sph[0], sph[1], sph[2], 0), (tor[0], tor[1], tor[2], 1), (ter[0], ter[1], ter[2], 2)])
    inst = [((0.0, 0.0, -1.0), IDENT, (1.0, 1.0, 1.0), 2),
            ((0.0, 0.0, 2.5), random_quat(rng), (1.5, 1.0, 0.7), 1),
            ((4.0, -3.0, 2.0), random_quat(rng), (1.2, -1.2, 1.2), 0),
            ((-5.0, 2.0, 1.5), random_quat(rng), tuple(float(x) for x in rng.uniform(0.3, 2.0, 3)), 0)]
    cams = []
    for _ in range(8):
        kind = int(rng.integers(0, 4))
        if kind == 0:       # on / just above the terrain, looking along it
            eye = (float(rng.uniform(-12, 12)), float(rng.uniform(-12, 12)), float(rng.uniform(-0.8, 1.5)))
            tgt = (float(rng.uniform(-12, 12)), float(rng.uniform(-12, 12)), float(rng.uniform(-1, 3)))
            cams.append((eye, scenes.look_at(eye, tgt)))
        elif kind == 1:     # inside / next to an object's box
            c = inst[int(rng.integers(1, 4))][0]
            eye = tuple(float(np.float32(c[i] + rng.normal() * 0.8)) for i in range(3))
            cams.append((eye, random_quat(rng)))
        elif kind == 2:     # ring, looking at the centre
            az, r, h = rng.uniform(0, 6.28), rng.uniform(3, 25), rng.uniform(0.5, 14)
            eye = (float(r * math.cos(az)), float(r * math.sin(az)), float(h))
            cams.append((eye, scenes.look_at(eye, (0.0, 0.0, 1.0))))
        else:               # anywhere, any direction
            eye = tuple(float(np.float32(x)) for x in rng.normal(size=3) * 8.0)
            cams.append((eye, random_quat(rng)))
    return scenes.SceneDesc(
        num_worlds=2, render_mode=mode, width=width, height=height, asset_paths=[],
        materials=[((0.9, 0.8, 0.7, 1.0), -1, 0.5, 0.5), ((0.3, 0.5, 0.9, 1.0), -1, 0.5, 0.5),
                   ((0.4, 0.7, 0.3, 1.0), -1, 0.5, 0.5)],
        instances=inst, cameras=cams, worlds=[(4, 0, 5, 0), (3, 0, 3, 5)], **geo)
