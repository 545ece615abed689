"""ctypes front-end of the CPU oracle (oracle/raster_oracle.c).

TEST INFRASTRUCTURE ONLY -- see the header of raster_oracle.c.  Imported by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the
product package.  PARITY UNPINNED (the reference ships no arithmetic, tests or
fixtures for this path: SURVEY.md section 8c); pinned instead by the anchors of
tests/test_oracle_anchors.py.

The scene description it consumes is any object with the attributes of the
reference's ``MadronaRenderer(...)`` keyword arguments
(/root/reference/src/bindings.cpp:206-222): ``asset_paths`` [(path, mat_id)],
``mesh_vertices`` ... ``mesh_materials``, ``materials`` [(rgba, texture_id,
roughness, metalness)], ``texture_paths``, ``instances`` [(pos, rot, scale,
object_id)], ``cameras`` [(pos, rot)], ``worlds`` [(num_instances,
instance_offset, num_cameras, camera_offset)], ``render_mode`` ("Rasterizer" |
"Raytracer"), ``width``, ``height``.

Asset ingestion here is deliberately a second, independent implementation
(pure Python OBJ reader, Pillow PNG decode) of what the product does in C++
(madrona_renderer_amd/csrc/assets.cpp), so the parity tests also cross-check
the loaders.
"""
import ctypes
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

# Build-defined constants of the rendering spec.
VFOV_DEG = 90.0          # /root/reference/src/sim.cpp:170
RASTER_ZNEAR = 0.001     # /root/reference/src/sim.cpp:170
RT_ZNEAR = 0.1           # /root/reference/src/mgr.cpp:477
RT_ZFAR = 1000.0         # /root/reference/src/mgr.cpp:478
LIGHT_DIR = (1.0, -1.0, -0.05)   # /root/reference/src/mgr.cpp:357
AMBIENT = 0.25
DIFFUSE = 0.75
DEFAULT_COLOR = (1.0, 1.0, 1.0, 1.0)


class _Scene(ctypes.Structure):
    _fields_ = [
        ("tri_pos", ctypes.c_void_p), ("tri_uv", ctypes.c_void_p),
        ("tri_mat", ctypes.c_void_p), ("obj_first_tri", ctypes.c_void_p),
        ("obj_num_tris", ctypes.c_void_p), ("num_objects", ctypes.c_int32),
        ("tri_orient", ctypes.c_void_p), ("tri_bbmin", ctypes.c_void_p),
        ("tri_bbmax", ctypes.c_void_p),
        ("mat_color", ctypes.c_void_p), ("mat_tex", ctypes.c_void_p),
        ("num_materials", ctypes.c_int32),
        ("tex_data", ctypes.c_void_p), ("tex_offset", ctypes.c_void_p),
        ("tex_w", ctypes.c_void_p), ("tex_h", ctypes.c_void_p),
        ("num_textures", ctypes.c_int32),
        ("inst_pos", ctypes.c_void_p), ("inst_rot", ctypes.c_void_p),
        ("inst_scale", ctypes.c_void_p), ("inst_obj", ctypes.c_void_p),
        ("inst_obj0", ctypes.c_void_p),
        ("world_inst_start", ctypes.c_void_p),
        ("cam_pos", ctypes.c_void_p), ("cam_rot", ctypes.c_void_p),
        ("view_world", ctypes.c_void_p), ("num_views", ctypes.c_int32),
        ("width", ctypes.c_int32), ("height", ctypes.c_int32),
        ("sx", ctypes.c_float), ("ox", ctypes.c_float),
        ("sz", ctypes.c_float), ("oz", ctypes.c_float),
        ("inv_near", ctypes.c_float), ("inv_far", ctypes.c_float),
        ("s6b_pad", ctypes.c_float),
        ("to_light", ctypes.c_float * 3),
        ("ambient", ctypes.c_float), ("diffuse", ctypes.c_float),
        ("default_color", ctypes.c_float * 4),
        ("transposed", ctypes.c_int32),
    ]


def build(force=False):
    """Compile liboracle.so with gcc (oracle/Makefile)."""
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "raster_oracle.c")
    if force or not os.path.exists(so) or \
            os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"],
                              stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        _LIB = ctypes.CDLL(so)
        _LIB.orc_render.restype = ctypes.c_int
        _LIB.orc_render.argtypes = [ctypes.POINTER(_Scene), ctypes.c_int,
                                    ctypes.c_int] + [ctypes.c_void_p] * 4 + \
                                   [ctypes.c_int]
        _LIB.orc_num_procs.restype = ctypes.c_int
    return _LIB


# --------------------------------------------------------------------------
# independent asset readers
# --------------------------------------------------------------------------
def parse_obj(path, with_materials=False, with_objects=False):
    """Wavefront OBJ -> (tri_pos [T,3,3] f32, tri_uv [T,3,2] f32).

    Polygons fan-triangulated; ``vt`` optional.  With ``with_materials`` also
    returns (tri_mtl [T] index into names or -1, names, mtllib paths resolved
    against the OBJ's directory); with ``with_objects`` the first triangle of
    each ``o`` / ``g`` block that holds faces last (FlatScene makes one object of
    a file, as /root/reference/src/mgr.cpp:301-303,340-345 does, unless
    MRX_OBJ_SPLIT_BLOCKS=1)."""
    vs, vts, tris_p, tris_t = [], [], [], []
    tri_mtl, names, libs, cur = [], [], [], -1
    obj_start = [0]
    base = os.path.dirname(path)
    with open(path, "r") as f:
        for line in f:
            parts = line.split()
            if not parts:
                continue
            if parts[0] in ("o", "g"):
                if len(tris_p) > obj_start[-1]:
                    obj_start.append(len(tris_p))
            if parts[0] == "usemtl" and len(parts) > 1:
                name = line.split(None, 1)[1].strip()
                if name not in names:
                    names.append(name)
                cur = names.index(name)
            elif parts[0] == "mtllib" and len(parts) > 1:
                name = line.split(None, 1)[1].strip()
                libs.append(name if os.path.isabs(name) else os.path.join(base, name))
            if parts[0] == "v":
                vs.append([float(x) for x in parts[1:4]])
            elif parts[0] == "vt":
                vts.append([float(x) for x in parts[1:3]])
            elif parts[0] == "f":
                corners = []
                for c in parts[1:]:
                    fields = c.split("/")
                    vi = int(fields[0])
                    vi = vi - 1 if vi > 0 else len(vs) + vi
                    ti = None
                    if len(fields) > 1 and fields[1] != "":
                        ti = int(fields[1])
                        ti = ti - 1 if ti > 0 else len(vts) + ti
                    corners.append((vi, ti))
                for j in range(1, len(corners) - 1):
                    tri = (corners[0], corners[j], corners[j + 1])
                    tris_p.append([vs[c[0]] for c in tri])
                    tris_t.append([vts[c[1]] if c[1] is not None else [0.0, 0.0]
                                   for c in tri])
                    tri_mtl.append(cur)
    pos = np.asarray(tris_p, dtype=np.float64).astype(np.float32).reshape(-1, 3, 3)
    uv = np.asarray(tris_t, dtype=np.float64).astype(np.float32).reshape(-1, 3, 2)
    out = (pos, uv)
    if with_materials:
        out += (np.asarray(tri_mtl, dtype=np.int32), names, libs)
    if with_objects:
        out += (obj_start,)
    return out


def parse_mtl(path):
    """Wavefront MTL -> [(name, Kd rgb as float32, map_Kd path or None)]."""
    out = []
    base = os.path.dirname(path)
    try:
        f = open(path, "r")
    except OSError:
        return out
    with f:
        for line in f:
            parts = line.split()
            if not parts:
                continue
            if parts[0] == "newmtl" and len(parts) > 1:
                out.append([line.split(None, 1)[1].strip(), [1.0, 1.0, 1.0], None])
            elif out and parts[0] == "Kd" and len(parts) >= 4:
                out[-1][1] = [float(np.float32(float(x))) for x in parts[1:4]]
            elif out and parts[0] == "map_Kd" and len(parts) > 1:
                name = parts[-1]
                out[-1][2] = name if os.path.isabs(name) else os.path.join(base, name)
    return [tuple(m) for m in out]


def shell_orientation(tri_pos):
    """S6b, per triangle of one object: (orient [T] f32, bbmin [T,3], bbmax [T,3]).

    Triangles joined through a shared (undirected) edge form a shell, vertices
    welded by exact position.  A shell is closed and consistently wound when
    every directed edge occurs exactly once and so does its reverse; its
    triangles then carry orient = the sign of the enclosed volume, else 0.
    The box is the shell's, padded by 1e-4 of its extent + 1e-6 (float32)."""
    n = len(tri_pos)
    orient = np.zeros(n, np.float32)
    bmin = np.zeros((n, 3), np.float32)
    bmax = np.zeros((n, 3), np.float32)
    if n == 0:
        return orient, bmin, bmax
    ids, tv = {}, []
    for t in range(n):
        tv.append([ids.setdefault(tuple(float(x) + 0.0 for x in tri_pos[t, c]), len(ids))
                   for c in range(3)])                              # -0 -> +0
    parent = list(range(n))

    def find(x):
        while parent[x] != x:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x
    first = {}
    for t in range(n):
        for c in range(3):
            e = tuple(sorted((tv[t][c], tv[t][(c + 1) % 3])))
            if e in first:
                parent[find(t)] = find(first[e])
            else:
                first[e] = t
    shells = {}
    for t in range(n):
        shells.setdefault(find(t), []).append(t)
    for members in shells.values():
        edges, bad, vol = set(), len(members) < 4, 0.0
        for t in members:
            v = tv[t]
            if len(set(v)) < 3:
                bad = True
            for c in range(3):
                e = (v[c], v[(c + 1) % 3])
                if e in edges:
                    bad = True
                edges.add(e)
            a, b, c3 = (tri_pos[t, i].astype(np.float64) for i in range(3))
            vol += float(a @ np.cross(b, c3))
        if not bad and any((j, i) not in edges for (i, j) in edges):
            bad = True
        pts = tri_pos[members].reshape(-1, 3).astype(np.float32)
        lo, hi = pts.min(axis=0), pts.max(axis=0)
        pad = np.float32(1e-4) * (hi - lo) + np.float32(1e-6)
        orient[members] = 0.0 if bad else (1.0 if vol > 0 else -1.0 if vol < 0 else 0.0)
        bmin[members] = lo - pad
        bmax[members] = hi + pad
    return orient, bmin, bmax


def _dds_bc7(blocks, w, h):
    """BC7 blocks wrapped as an in-memory DDS file (DX10 header, DXGI_FORMAT_BC7_UNORM)."""
    import struct
    hdr = b"DDS " + struct.pack("<7I", 124, 0x1007 | 0x80000, h, w, len(blocks), 0, 1) + bytes(44)
    hdr += struct.pack("<2I4s5I", 32, 0x4, b"DX10", 0, 0, 0, 0, 0)
    hdr += struct.pack("<5I", 0x1000, 0, 0, 0, 0)
    return hdr + struct.pack("<5I", 98, 3, 0, 1, 0) + blocks


def decode_bc7(blocks, width, height):
    """BC7 blocks -> RGBA8 [height, width, 4] through Pillow's DDS reader."""
    import io
    from PIL import Image
    bw, bh = (width + 3) // 4, (height + 3) // 4
    with Image.open(io.BytesIO(_dds_bc7(bytes(blocks), bw * 4, bh * 4))) as im:
        return np.asarray(im.convert("RGBA"), dtype=np.uint8)[:height, :width].copy()


def decode_ktx2(path):
    """KTX 2.0 (BC7 or RGBA8 base level, supercompression none / Zstandard / ZLIB) ->
    RGBA8.  An independent reading of the container (struct), of BC7 (Pillow) and of
    Zstandard (pyarrow's codec)."""
    import struct
    import zlib
    with open(path, "rb") as f:
        data = f.read()
    if data[:12] != bytes([0xAB, 0x4B, 0x54, 0x58, 0x20, 0x32, 0x30, 0xBB, 0x0D, 0x0A, 0x1A, 0x0A]):
        raise OSError("not a KTX2 file")
    vk, _ts, w, h, d, layers, faces, levels, scheme = struct.unpack_from("<9I", data, 12)
    off, length, _ulen = struct.unpack_from("<3Q", data, 80)
    if d > 1 or layers > 1 or faces != 1 or vk not in (37, 43, 145, 146) or scheme not in (0, 2, 3):
        raise OSError("unsupported KTX2")
    payload = data[off:off + length]
    if scheme == 3:
        payload = zlib.decompress(payload)
    elif scheme == 2:
        import pyarrow
        payload = pyarrow.decompress(payload, decompressed_size=int(_ulen), codec="zstd").to_pybytes()
    if vk in (37, 43):
        return np.frombuffer(payload[:w * h * 4], np.uint8).reshape(h, w, 4).copy()
    return decode_bc7(payload[:((w + 3) // 4) * ((h + 3) // 4) * 16], w, h)


def decode_image(path):
    """Image file -> RGBA8 array [h, w, 4]: .ktx2 as above, the rest by Pillow."""
    if path.lower().endswith(".ktx2"):
        return decode_ktx2(path)
    from PIL import Image
    with Image.open(path) as im:
        return np.asarray(im.convert("RGBA"), dtype=np.uint8).copy()


# --------------------------------------------------------------------------
# scene flattening (mirrors /root/reference/src/mgr.cpp:214-363 ordering and
# /root/reference/src/sim.cpp:135-176 world assembly)
# --------------------------------------------------------------------------
def _f32(x):
    return np.float32(x)


def projection_constants(width, height, transposed):
    th = float(np.float32(math.tan(VFOV_DEG * math.pi / 360.0)))
    asp = float(width) / float(height)
    sx = np.float32(2.0 * th * asp / width)
    ox = np.float32((1.0 / width - 1.0) * th * asp)
    sz = np.float32(-2.0 * th / height)
    oz = np.float32((1.0 - 1.0 / height) * th)
    return sx, ox, sz, oz


def to_light_vector():
    d = np.asarray(LIGHT_DIR, dtype=np.float64)
    return (-d / math.sqrt(float(d @ d))).astype(np.float32)


class FlatScene:
    """Flattened arrays in the layout raster_oracle.c reads."""

    def __init__(self, desc, base_dir=None):
        mode = getattr(desc, "render_mode", "Rasterizer")
        mode = getattr(mode, "name", mode)
        self.raytracer = (str(mode) == "Raytracer")
        self.width = int(desc.width)
        self.height = int(desc.height)
        if self.raytracer:
            # /root/reference/src/mgr.cpp:443 -- square, res = view width
            self.height = self.width

        def _path(p):
            return p if (base_dir is None or os.path.isabs(p)) \
                else os.path.join(base_dir, p)

        # -- objects: disk assets first, then one object per raw mesh
        pos_l, uv_l, mat_l, first, count = [], [], [], [], []
        ntri = 0
        mats = list(desc.materials)
        file_mats = []              # (Kd, map_Kd) appended after the API materials
        for path, mat_id in desc.asset_paths:
            p, t, tri_mtl, names, libs, starts = parse_obj(_path(path), with_materials=True,
                                                           with_objects=True)
            pos_l.append(p)
            uv_l.append(t)
            # intended semantics of the disabled block mgr.cpp:339-349: mat_id
            # indexes the additional materials; with -1 a face keeps the
            # material its OBJ names through mtllib / usemtl
            tm = np.full(len(p), int(mat_id), dtype=np.int32)
            if int(mat_id) < 0 and names:
                lib = [m for ml in libs for m in parse_mtl(ml)]
                name_to_mat = []
                for n in names:
                    hit = next((m for m in lib if m[0] == n), None)
                    if hit is None:
                        name_to_mat.append(-1)
                    else:
                        name_to_mat.append(len(mats) + len(file_mats))
                        file_mats.append((hit[1], hit[2]))
                for i, k in enumerate(tri_mtl):
                    if k >= 0:
                        tm[i] = name_to_mat[k]
            mat_l.append(tm)
            # one object per asset file (importFromDisk(..., true) and objects[i] <-> asset i,
            # mgr.cpp:301-303,340-345): `o` / `g` blocks are meshes of the one object;
            # MRX_OBJ_SPLIT_BLOCKS=1 opts into one object per block
            if os.environ.get("MRX_OBJ_SPLIT_BLOCKS", "")[:1] != "1":
                starts = [0]
            for o, t0 in enumerate(starts):
                t1 = starts[o + 1] if o + 1 < len(starts) else len(p)
                first.append(ntri + t0)
                count.append(t1 - t0)
            ntri += len(p)
        verts = np.asarray(desc.mesh_vertices, dtype=np.float32).reshape(-1, 3)
        uvs = np.asarray(desc.mesh_uvs, dtype=np.float32).reshape(-1, 2)
        idx = np.asarray(desc.mesh_indices, dtype=np.uint32).reshape(-1)
        voff = np.asarray(desc.mesh_vertex_offsets, dtype=np.uint32).reshape(-1)
        ioff = np.asarray(desc.mesh_indices_offsets, dtype=np.uint32).reshape(-1)
        mmat = np.asarray(desc.mesh_materials, dtype=np.int32).reshape(-1)
        for m in range(len(voff)):
            v0 = int(voff[m])
            v1 = int(voff[m + 1]) if m + 1 < len(voff) else len(verts)
            i0 = int(ioff[m])
            i1 = int(ioff[m + 1]) if m + 1 < len(ioff) else len(idx)
            nt = (i1 - i0) // 3
            ii = idx[i0:i0 + 3 * nt].astype(np.int64).reshape(-1, 3)
            mv, mu = verts[v0:v1], uvs[v0:v1]
            pos_l.append(mv[ii].reshape(-1, 3, 3))
            uv_l.append(mu[ii].reshape(-1, 3, 2))
            mat_l.append(np.full(nt, int(mmat[m]), dtype=np.int32))
            first.append(ntri)
            count.append(nt)
            ntri += nt
        self.tri_pos = np.ascontiguousarray(
            np.concatenate(pos_l) if pos_l else np.zeros((0, 3, 3)), np.float32)
        self.tri_uv = np.ascontiguousarray(
            np.concatenate(uv_l) if uv_l else np.zeros((0, 3, 2)), np.float32)
        self.tri_mat = np.ascontiguousarray(
            np.concatenate(mat_l) if mat_l else np.zeros(0), np.int32)
        self.obj_first_tri = np.asarray(first, dtype=np.int32)
        self.obj_num_tris = np.asarray(count, dtype=np.int32)
        T = len(self.tri_pos)
        self.tri_orient = np.zeros(T, np.float32)
        self.tri_bbmin = np.zeros((T, 3), np.float32)
        self.tri_bbmax = np.zeros((T, 3), np.float32)
        for f, c in zip(first, count):
            o, lo, hi = shell_orientation(self.tri_pos[f:f + c])
            self.tri_orient[f:f + c] = o
            self.tri_bbmin[f:f + c] = lo
            self.tri_bbmax[f:f + c] = hi

        # -- materials / textures (API ones first, then the files' own)
        texels, offs, tw, th = [], [], [], []
        o = 0

        def add_texture(img):
            nonlocal o
            offs.append(o)
            th.append(img.shape[0])
            tw.append(img.shape[1])
            texels.append(img.reshape(-1, 4))
            o += img.shape[0] * img.shape[1]

        for p in desc.texture_paths:
            add_texture(decode_image(_path(p)))
        mat_color = [list(m[0]) for m in mats]
        mat_tex = [int(m[1]) for m in mats]
        file_tex = {}
        for kd, map_kd in file_mats:
            tex = -1
            if map_kd:
                if map_kd not in file_tex:
                    try:
                        if not map_kd.lower().endswith((".png", ".ktx2")):
                            raise OSError("only PNG and KTX2 textures are read")
                        img = decode_image(map_kd)
                        file_tex[map_kd] = len(texels)
                        add_texture(img)
                    except OSError:
                        file_tex[map_kd] = -1
                tex = file_tex[map_kd]
            mat_color.append([kd[0], kd[1], kd[2], 1.0])
            mat_tex.append(tex)
        self.mat_color = np.asarray(mat_color, dtype=np.float32).reshape(-1, 4)
        self.mat_tex = np.asarray(mat_tex, dtype=np.int32)
        self.tex_data = np.ascontiguousarray(
            np.concatenate(texels) if texels else np.zeros((1, 4)), np.uint8)
        self.tex_offset = np.asarray(offs if offs else [0], dtype=np.int64)
        self.tex_w = np.asarray(tw if tw else [0], dtype=np.int32)
        self.tex_h = np.asarray(th if th else [0], dtype=np.int32)
        self.num_textures = len(texels)

        # -- world assembly: per-world copies, world-major
        inst = list(desc.instances)
        cams = list(desc.cameras)
        ipos, irot, iscl, iobj, wstart = [], [], [], [], [0]
        cpos, crot, vworld = [], [], []
        # max_instances_per_world: a world owns that many rows at least; the spare ones
        # start hidden and unbound (ObjectID -1, identity pose) -- the reference sizes its
        # renderer by maxInstancesPerWorld (/root/reference/src/mgr.cpp:378-388) and creates
        # renderables at run time (src/sim.inl:5-8)
        cap = int(getattr(desc, "max_instances_per_world", 0) or 0)
        for w, (ni, io, nc, co) in enumerate(desc.worlds):
            for r in inst[io:io + ni]:
                ipos.append(r[0]); irot.append(r[1]); iscl.append(r[2])
                iobj.append(r[3])
            for _ in range(max(0, cap - ni)):
                ipos.append((0.0, 0.0, 0.0)); irot.append((1.0, 0.0, 0.0, 0.0)); iscl.append((1.0, 1.0, 1.0))
                iobj.append(-1)
            wstart.append(len(ipos))
            for r in cams[co:co + nc]:
                cpos.append(r[0]); crot.append(r[1]); vworld.append(w)
        self.inst_pos = np.asarray(ipos, dtype=np.float32).reshape(-1, 3)
        self.inst_rot = np.asarray(irot, dtype=np.float32).reshape(-1, 4)
        self.inst_scale = np.asarray(iscl, dtype=np.float32).reshape(-1, 3)
        self.inst_obj = np.asarray(iobj, dtype=np.int32)
        # inst_obj0 = the object each row is BOUND to (geometry, triangle slots, segmask
        # label); inst_obj = the live ObjectID column, of which only the sign is read
        # (negative hides).  refresh_objects() re-binds, as mrx_refresh_objects does.
        self.inst_obj0 = self.inst_obj.copy()
        self.world_inst_start = np.asarray(wstart, dtype=np.int32)
        self.cam_pos = np.asarray(cpos, dtype=np.float32).reshape(-1, 3)
        self.cam_rot = np.asarray(crot, dtype=np.float32).reshape(-1, 4)
        self.view_world = np.asarray(vworld, dtype=np.int32)
        self.num_views = len(vworld)

    def refresh_objects(self):
        """Bind every row whose live id is non-negative to that object (rows holding a
        negative id stay bound to what they drew): include/mrx.h, mrx_refresh_objects."""
        live = self.inst_obj >= 0
        self.inst_obj0[live] = self.inst_obj[live]

    def _struct(self):
        s = _Scene()
        keep = []

        def ptr(a):
            keep.append(a)
            return a.ctypes.data if a.size else None
        s.tri_pos = ptr(self.tri_pos); s.tri_uv = ptr(self.tri_uv)
        s.tri_mat = ptr(self.tri_mat)
        s.obj_first_tri = ptr(self.obj_first_tri)
        s.obj_num_tris = ptr(self.obj_num_tris)
        s.num_objects = len(self.obj_first_tri)
        s.tri_orient = ptr(self.tri_orient)
        s.tri_bbmin = ptr(self.tri_bbmin); s.tri_bbmax = ptr(self.tri_bbmax)
        s.mat_color = ptr(self.mat_color); s.mat_tex = ptr(self.mat_tex)
        s.num_materials = len(self.mat_tex)
        s.tex_data = ptr(self.tex_data); s.tex_offset = ptr(self.tex_offset)
        s.tex_w = ptr(self.tex_w); s.tex_h = ptr(self.tex_h)
        s.num_textures = self.num_textures
        s.inst_pos = ptr(self.inst_pos); s.inst_rot = ptr(self.inst_rot)
        s.inst_scale = ptr(self.inst_scale); s.inst_obj = ptr(self.inst_obj)
        s.inst_obj0 = ptr(self.inst_obj0)
        s.world_inst_start = ptr(self.world_inst_start)
        s.cam_pos = ptr(self.cam_pos); s.cam_rot = ptr(self.cam_rot)
        s.view_world = ptr(self.view_world)
        s.num_views = self.num_views
        s.width = self.width; s.height = self.height
        sx, ox, sz, oz = projection_constants(self.width, self.height,
                                              self.raytracer)
        s.sx, s.ox, s.sz, s.oz = sx, ox, sz, oz
        znear = _f32(RT_ZNEAR if self.raytracer else RASTER_ZNEAR)
        s.inv_near = _f32(1.0) / znear
        s.inv_far = (_f32(1.0) / _f32(RT_ZFAR)) if self.raytracer else _f32(0.0)
        th = float(np.float32(math.tan(VFOV_DEG * math.pi / 360.0)))
        asp = float(self.width) / float(self.height)
        s.s6b_pad = _f32(float(znear) * math.sqrt(1.0 + th * th * (1.0 + asp * asp)) * 1.001)
        tl = to_light_vector()
        s.to_light = (ctypes.c_float * 3)(*[float(x) for x in tl])
        s.ambient = AMBIENT
        s.diffuse = DIFFUSE
        s.default_color = (ctypes.c_float * 4)(*DEFAULT_COLOR)
        s.transposed = 1 if self.raytracer else 0
        return s, keep

    def render(self, view_begin=0, view_end=None, num_threads=0,
               want_ids=True):
        """-> dict(rgb [V,H,W,4] u8, depth [V,H,W] f32, tri_id, segmask)."""
        if view_end is None:
            view_end = self.num_views
        V = self.num_views
        # storage is [view][slow][fast]; raster (H, W); Raytracer (res, res)
        nslow = self.width if self.raytracer else self.height
        nfast = self.height if self.raytracer else self.width
        rgb = np.zeros((V, nslow, nfast, 4), dtype=np.uint8)
        depth = np.zeros((V, nslow, nfast), dtype=np.float32)
        tri = np.full((V, nslow, nfast), -1, dtype=np.int32) if want_ids else None
        seg = np.full((V, nslow, nfast), -1, dtype=np.int32) if want_ids else None
        s, keep = self._struct()
        used = lib().orc_render(ctypes.byref(s), int(view_begin), int(view_end),
                                rgb.ctypes.data, depth.ctypes.data,
                                tri.ctypes.data if want_ids else None,
                                seg.ctypes.data if want_ids else None,
                                int(num_threads))
        del keep
        return {"rgb": rgb, "depth": depth, "tri_id": tri, "segmask": seg,
                "threads": used}


def num_procs():
    return lib().orc_num_procs()
