"""Scene descriptions for the batch renderer: the reference's demo scene and
the deterministic synthetic N-world scenes the benchmark and parity tests use.

A ``SceneDesc`` carries exactly the keyword arguments of the reference's
``MadronaRenderer(...)`` constructor (/root/reference/src/bindings.cpp:206-222)
as plain Python / numpy values, so the same description can be handed to the
product (``make_renderer``) and to the test oracle.

Synthetic scenes follow SURVEY.md section 8(d): counter-based RNG
``u(k) = (splitmix64(seed ^ k) >> 40) * 2**-24`` with seed 0x4D52584D.
"""
import math
import os
from dataclasses import dataclass, field

import numpy as np

SEED = 0x4D52584D
DATA_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                        "data")


@dataclass
class SceneDesc:
    num_worlds: int
    render_mode: str = "Rasterizer"          # or "Raytracer"
    width: int = 64
    height: int = 64
    asset_paths: list = field(default_factory=list)     # [(path, mat_id)]
    mesh_vertices: np.ndarray = None                    # [N,3] f32
    mesh_uvs: np.ndarray = None                         # [N,2] f32
    mesh_indices: np.ndarray = None                     # [K] u32
    mesh_vertex_offsets: np.ndarray = None              # [M] u32
    mesh_indices_offsets: np.ndarray = None             # [M] u32
    mesh_materials: np.ndarray = None                   # [M] i32
    materials: list = field(default_factory=list)       # [(rgba, tex, rough, metal)]
    texture_paths: list = field(default_factory=list)
    instances: list = field(default_factory=list)       # [(pos, rot, scale, obj)]
    cameras: list = field(default_factory=list)         # [(pos, rot)]
    worlds: list = field(default_factory=list)          # [(ni, io, nc, co)]
    # rows per world at least (spare rows start hidden and unbound: refresh_objects)
    max_instances_per_world: int = 0

    def __post_init__(self):
        if self.mesh_vertices is None:
            self.mesh_vertices = np.zeros((0, 3), np.float32)
            self.mesh_uvs = np.zeros((0, 2), np.float32)
            self.mesh_indices = np.zeros(0, np.uint32)
            self.mesh_vertex_offsets = np.zeros(0, np.uint32)
            self.mesh_indices_offsets = np.zeros(0, np.uint32)
            self.mesh_materials = np.zeros(0, np.int32)

    @property
    def num_views(self):
        return sum(w[2] for w in self.worlds)

    def shard(self, rank, world_size):
        """Contiguous world range of ``rank``: each
        rank's views form one slab of the global [views,H,W,C] tensors."""
        lo, hi = shard_range(self.num_worlds, rank, world_size)
        d = SceneDesc(**{k: getattr(self, k) for k in self.__dataclass_fields__})
        d.worlds = self.worlds[lo:hi]
        d.num_worlds = hi - lo
        return d


def shard_range(num_worlds, rank, world_size):
    """[lo, hi) of the worlds rank owns: contiguous, sizes differ by <= 1."""
    base, rem = divmod(int(num_worlds), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


# --------------------------------------------------------------------------
def _splitmix64(x):
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def uniform(k, seed=SEED):
    """u(k) in [0,1), exact in float32."""
    k = np.asarray(k, dtype=np.uint64)
    bits = _splitmix64(np.uint64(seed) ^ k) >> np.uint64(40)
    return bits.astype(np.float64) * 2.0 ** -24


def _quat_from_basis(right, fwd, up):
    """Unit quaternion (w,x,y,z) of the rotation whose columns are
    (right, fwd, up) = camera local +X, +Y, +Z in world space."""
    m = np.stack([right, fwd, up], axis=1)
    tr = m[0, 0] + m[1, 1] + m[2, 2]
    if tr > 0:
        s = math.sqrt(tr + 1.0) * 2
        q = (0.25 * s, (m[2, 1] - m[1, 2]) / s, (m[0, 2] - m[2, 0]) / s,
             (m[1, 0] - m[0, 1]) / s)
    elif m[0, 0] > m[1, 1] and m[0, 0] > m[2, 2]:
        s = math.sqrt(1.0 + m[0, 0] - m[1, 1] - m[2, 2]) * 2
        q = ((m[2, 1] - m[1, 2]) / s, 0.25 * s, (m[0, 1] + m[1, 0]) / s,
             (m[0, 2] + m[2, 0]) / s)
    elif m[1, 1] > m[2, 2]:
        s = math.sqrt(1.0 + m[1, 1] - m[0, 0] - m[2, 2]) * 2
        q = ((m[0, 2] - m[2, 0]) / s, (m[0, 1] + m[1, 0]) / s, 0.25 * s,
             (m[1, 2] + m[2, 1]) / s)
    else:
        s = math.sqrt(1.0 + m[2, 2] - m[0, 0] - m[1, 1]) * 2
        q = ((m[1, 0] - m[0, 1]) / s, (m[0, 2] + m[2, 0]) / s,
             (m[1, 2] + m[2, 1]) / s, 0.25 * s)
    return tuple(float(np.float32(c)) for c in q)


def look_at(eye, target):
    """Camera looks along local +Y, +X right, +Z up, no roll."""
    eye = np.asarray(eye, np.float64)
    fwd = np.asarray(target, np.float64) - eye
    fwd /= np.linalg.norm(fwd)
    right = np.cross(fwd, np.array([0.0, 0.0, 1.0]))
    right /= np.linalg.norm(right)
    up = np.cross(right, fwd)
    return _quat_from_basis(right, fwd, up)


def _f32t(*xs):
    return tuple(float(np.float32(x)) for x in xs)


def synthetic_scene(num_worlds, width=64, height=64, with_wall=False,
                    textured=False, render_mode="Rasterizer", data_dir=None,
                    first_world=0):
    """Cube + ground plane (+ wall) per world, one camera per world on a ring
    looking at the origin."""
    dd = DATA_DIR if data_dir is None else data_dir
    cube_mat = 1 if textured else 0
    assets = [(os.path.join(dd, "cube.obj"), cube_mat),
              (os.path.join(dd, "plane.obj"), 0)]
    if with_wall:
        assets.append((os.path.join(dd, "wall_render.obj"), 0))
    materials = [((0.588, 0.588, 0.588, 1.0), -1, 0.8, 0.2),
                 ((1.0, 1.0, 1.0, 1.0), 0, 0.8, 0.2)]
    n_inst = 3 if with_wall else 2
    # world ids are global: rank r of a sharded job passes first_world = its
    # offset and gets exactly the rows it would own of the whole job's scene
    w = np.arange(first_world, first_world + num_worlds, dtype=np.uint64)
    u = [uniform(w * np.uint64(16) + np.uint64(j)) for j in range(12)]
    instances, cameras, worlds = [], [], []
    for i in range(num_worlds):
        s = 1.0 + 2.0 * u[2][i]
        th = 2.0 * math.pi * u[3][i]
        instances.append((_f32t(0, 0, 0), (1.0, 0.0, 0.0, 0.0), (1.0, 1.0, 1.0), 1))
        instances.append((_f32t(-4 + 8 * u[0][i], -4 + 8 * u[1][i], 0.5 * s),
                          _f32t(math.cos(th / 2), 0, 0, math.sin(th / 2)),
                          _f32t(s, s, s), 0))
        if with_wall:
            ya = u[6][i] * math.pi
            instances.append((_f32t(-6 + 12 * u[4][i], 6 + 2 * u[5][i], 0),
                              _f32t(math.cos(ya / 2), 0, 0, math.sin(ya / 2)),
                              (1.0, 1.0, 1.0), 2))
        r = 10.0 + 6.0 * u[7][i]
        hgt = 3.0 + 5.0 * u[8][i]
        az = 2.0 * math.pi * u[9][i]
        eye = _f32t(r * math.cos(az), r * math.sin(az), hgt)
        cameras.append((eye, look_at(eye, (0.0, 0.0, 1.0))))
        worlds.append((n_inst, i * n_inst, 1, i))
    return SceneDesc(num_worlds=num_worlds, render_mode=render_mode,
                     width=width, height=height, asset_paths=assets,
                     materials=materials,
                     texture_paths=[os.path.join(dd, "cube.png")],
                     instances=instances, cameras=cameras, worlds=worlds)


def cube_field(num_worlds, cubes, width=64, height=64, mode="Rasterizer", textured=False,
               seed=1, spread=9.0, first_world=0):
    """`cubes` cubes scattered over the ground plane of every world (12 * cubes
    + 2 triangles per world), one camera per world on a ring -- the many-instance
    shape of worlds the reference's TLAS is for.  Worlds differ (own rows)."""
    mats = [((0.9, 0.7, 0.5, 1.0), 0 if textured else -1, 0.5, 0.5), ((0.3, 0.6, 0.3, 1.0), -1, 0.5, 0.5)]
    inst, cams, worlds = [], [], []
    for w in range(first_world, first_world + num_worlds):
        rng = np.random.default_rng(seed * 100003 + w)
        i0 = len(inst)
        inst.append(((0.0, 0.0, 0.0), (1.0, 0.0, 0.0, 0.0), (1.0, 1.0, 1.0), 1))
        for _ in range(cubes):
            p = rng.uniform(-spread, spread, 2)
            s = float(rng.uniform(0.4, 1.3))
            th = float(rng.uniform(0, 2 * math.pi))
            inst.append(((float(np.float32(p[0])), float(np.float32(p[1])), float(np.float32(0.5 * s))),
                         tuple(float(np.float32(x)) for x in (math.cos(th / 2), 0, 0, math.sin(th / 2))),
                         (float(np.float32(s)),) * 3, 0))
        az = float(rng.uniform(0, 2 * math.pi))
        r, h = float(rng.uniform(11, 17)), float(rng.uniform(3, 9))
        eye = tuple(float(np.float32(x)) for x in (r * math.cos(az), r * math.sin(az), h))
        cams.append((eye, look_at(eye, (0.0, 0.0, 0.5))))
        worlds.append((cubes + 1, i0, 1, len(cams) - 1))
    return SceneDesc(
        num_worlds=num_worlds, render_mode=mode, width=width, height=height,
        asset_paths=[(os.path.join(DATA_DIR, "cube.obj"), 0), (os.path.join(DATA_DIR, "plane.obj"), 1)], materials=mats,
        texture_paths=[os.path.join(DATA_DIR, "cube.png")],
        instances=inst, cameras=cams, worlds=worlds)


def demo_scene(num_worlds=4, render_mode="Raytracer", width=64, height=64,
               data_dir=None):
    """The literal scene of /root/reference/scripts/test.py:11-130: cube.obj +
    one hand-specified triangle, every world aliasing the same table rows."""
    dd = DATA_DIR if data_dir is None else data_dir
    return SceneDesc(
        num_worlds=num_worlds, render_mode=render_mode, width=width,
        height=height,
        asset_paths=[(os.path.join(dd, "cube.obj"), 0)],
        mesh_vertices=np.array([[0, 0, 0], [5, 0, 10], [10, 0, 0]], np.float32),
        mesh_uvs=np.zeros((3, 2), np.float32),
        mesh_indices=np.array([0, 1, 2], np.uint32),
        mesh_vertex_offsets=np.array([0], np.uint32),
        mesh_indices_offsets=np.array([0], np.uint32),
        mesh_materials=np.array([-1], np.int32),
        materials=[((1.0, 1.0, 1.0, 1.0), 0, 0.8, 0.2)],
        texture_paths=[os.path.join(dd, "cube.png")],
        instances=[((0.0, 0.0, 15.0), (0.707107, 0.707107, 0.0, 0.0),
                    (3.0, 3.0, 3.0), 0),
                   ((0.0, 0.0, 15.0), (0.707107, 0.707107, 0.0, 0.0),
                    (10.0, 10.0, 10.0), 1)],
        cameras=[((-22.343935, -21.845375, 27.061676),
                  (0.913407, -0.112268, 0.047731, -0.388336))],
        worlds=[(2, 0, 1, 0)] * num_worlds)


def make_renderer(desc, gpu_id=0, device_ids=None):
    """Instantiate the product renderer (compiled ``madrona_renderer`` module,
    HIP only) from a SceneDesc, with the reference's constructor kwargs.
    ``device_ids`` = [d0, d1, ...] makes the one renderer span several devices
    (contiguous world ranges, one shard per listed device)."""
    from . import load_module
    m = load_module()
    extra = {}
    if device_ids is not None:
        extra["device_ids"] = [int(d) for d in device_ids]
    if desc.max_instances_per_world:
        extra["max_instances_per_world"] = int(desc.max_instances_per_world)
    return m.MadronaRenderer(
        gpu_id=gpu_id,
        num_worlds=desc.num_worlds,
        render_mode=getattr(m.RenderMode, desc.render_mode),
        batch_render_view_width=desc.width,
        batch_render_view_height=desc.height,
        asset_paths=[m.ImportedAsset(path=p, mat_id=i) for p, i in desc.asset_paths],
        mesh_vertices=np.ascontiguousarray(desc.mesh_vertices, np.float32).reshape(-1, 3),
        mesh_uvs=np.ascontiguousarray(desc.mesh_uvs, np.float32).reshape(-1, 2),
        mesh_indices=np.ascontiguousarray(desc.mesh_indices, np.uint32),
        mesh_vertex_offsets=np.ascontiguousarray(desc.mesh_vertex_offsets, np.uint32),
        mesh_indices_offsets=np.ascontiguousarray(desc.mesh_indices_offsets, np.uint32),
        mesh_materials=np.ascontiguousarray(desc.mesh_materials, np.int32),
        materials=[m.AdditionalMaterial(color=list(c), texture_id=t, roughness=r,
                                        metalness=me)
                   for c, t, r, me in desc.materials],
        texture_paths=list(desc.texture_paths),
        instances=[m.ImportedInstance(position=list(p), rotation=list(q),
                                      scale=list(s), object_id=o)
                   for p, q, s, o in desc.instances],
        cameras=[m.ImportedCamera(position=list(p), rotation=list(q))
                 for p, q in desc.cameras],
        worlds=[m.WorldInit(num_instances=a, instance_offset=b, num_cameras=c,
                            camera_offset=d) for a, b, c, d in desc.worlds],
        **extra)
