"""ctypes binding of the CPU oracle (oracle/libptoracle.so) for the tests.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg load this."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "libptoracle.so")

NSTAT = 10
STAT_NAMES = ["rays", "box_tests", "obj_tests", "sphere_tests", "tri_tests", "sockets",
              "hit_sphere", "hit_plane", "hit_triangle", "material_fetch"]


class PtoTexture(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("rgba", C.c_void_p)]


class PtoScene(C.Structure):
    _fields_ = [
        ("materials", C.c_void_p), ("n_materials", C.c_int),
        ("spheres", C.c_void_p), ("n_spheres", C.c_int),
        ("planes", C.c_void_p), ("n_planes", C.c_int),
        ("triangles", C.c_void_p), ("n_triangles", C.c_int),
        ("bvh", C.c_void_p), ("n_nodes", C.c_int),
        ("object_ids", C.c_void_p), ("n_object_ids", C.c_int),
        ("transforms", C.c_void_p), ("n_transforms", C.c_int),
        ("textures", C.c_void_p), ("n_textures", C.c_int),
        ("order", C.c_int), ("refraction_mode", C.c_int), ("use_emission", C.c_int), ("early_t_break", C.c_int),
        ("n_proc", C.c_int), ("proc_builtin", C.c_void_p),
        ("triangle_uvs", C.c_void_p), ("n_triangle_uvs", C.c_int),
    ]


class PtoTrace(C.Structure):
    _fields_ = [("primary_id", C.c_void_p), ("pixel_stats", C.c_void_p), ("totals", C.c_void_p)]


_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB):
        build()
    lib = C.CDLL(LIB)
    lib.pto_render.restype = C.c_int
    lib.pto_render.argtypes = [C.POINTER(PtoScene), C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                               C.POINTER(PtoTrace), C.c_int]
    lib.pto_hash.restype = C.c_float
    lib.pto_hash.argtypes = [C.c_float, C.c_float, C.c_float]
    lib.pto_pcg_next.restype = C.c_float
    lib.pto_pcg_next.argtypes = [C.POINTER(C.c_float)]
    lib.pto_noise_next.restype = C.c_float
    lib.pto_noise_next.argtypes = [C.POINTER(C.c_float)]
    lib.pto_rand.restype = C.c_float
    lib.pto_rand.argtypes = [C.c_float]
    lib.pto_math_probe.restype = None
    lib.pto_math_probe.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.pto_offset_ray.restype = None
    lib.pto_offset_ray.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    # builder
    lib.pbo_build.restype = C.c_void_p
    lib.pbo_build.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.pbo_assemble.restype = C.c_void_p
    lib.pbo_assemble.argtypes = [C.c_int, C.c_int] + [C.c_void_p] * 4 + [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    lib.pbo_error.restype = C.c_int
    lib.pbo_error.argtypes = [C.c_void_p]
    lib.pbo_node_count.restype = C.c_int
    lib.pbo_node_count.argtypes = [C.c_void_p]
    lib.pbo_object_id_count.restype = C.c_int
    lib.pbo_object_id_count.argtypes = [C.c_void_p]
    lib.pbo_counts.restype = None
    lib.pbo_counts.argtypes = [C.c_void_p, C.c_void_p]
    lib.pbo_flatten.restype = C.c_int
    lib.pbo_flatten.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.pbo_refit.restype = C.c_int
    lib.pbo_refit.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.pbo_free.restype = None
    lib.pbo_free.argtypes = [C.c_void_p]
    lib.pbo_is_aabb_valid.restype = C.c_int
    lib.pbo_is_aabb_valid.argtypes = [C.c_void_p]
    lib.pbo_merge_aabb.restype = None
    lib.pbo_merge_aabb.argtypes = [C.c_void_p] * 3
    lib.pbo_xform_aabb.restype = None
    lib.pbo_xform_aabb.argtypes = [C.c_void_p] * 3
    lib.pbo_godot_sort_desc.restype = None
    lib.pbo_godot_sort_desc.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
    _lib = lib
    return lib


PROC_BUILTIN = {"checker_board.comp": 1, "checker_board_spatial.comp": 2}


def render(packed, push: bytes, image: np.ndarray, groups_x: int, groups_y: int, refraction_mode=2, use_emission=False,
           early_t_break=False, trace=True, nthreads: int = 0, order=None, triangle_uvs=None):
    """One dispatch on the oracle.  image: [H,W,4] float32, updated in place.  Returns dict with
    primary ids, per-pixel stats and totals when trace is set."""
    lib = load()
    h, w = image.shape[:2]
    assert image.dtype == np.float32 and image.flags["C_CONTIGUOUS"]
    keep = []

    def buf(b):
        a = np.frombuffer(b, dtype=np.uint8).copy()
        keep.append(a)
        return a.ctypes.data

    s = PtoScene()
    s.materials, s.n_materials = buf(packed.materials), len(packed.materials) // 64
    s.spheres, s.n_spheres = buf(packed.spheres), len(packed.spheres) // 32
    s.planes, s.n_planes = buf(packed.planes), len(packed.planes) // 32
    s.triangles, s.n_triangles = buf(packed.triangles), len(packed.triangles) // 80
    s.bvh, s.n_nodes = buf(packed.bvh), len(packed.bvh) // 32
    s.object_ids, s.n_object_ids = buf(packed.object_ids), len(packed.object_ids) // 4
    s.transforms, s.n_transforms = buf(packed.transforms), len(packed.transforms) // 64
    texs = (PtoTexture * max(1, len(packed.sampled_textures)))()
    for i, img in enumerate(packed.sampled_textures):
        if img is None:
            texs[i].width, texs[i].height, texs[i].rgba = 8, 8, None
        else:
            a = np.ascontiguousarray(img, dtype=np.float32)
            keep.append(a)
            texs[i].width, texs[i].height, texs[i].rgba = a.shape[1], a.shape[0], a.ctypes.data
    s.textures, s.n_textures = C.cast(texs, C.c_void_p), len(packed.sampled_textures)
    s.order = packed.bvh_order if order is None else order
    s.refraction_mode, s.use_emission, s.early_t_break = refraction_mode, int(use_emission), int(early_t_break)
    pb = np.array([PROC_BUILTIN[os.path.basename(p)] for p in packed.proc_tex_paths] or [0], dtype=np.int32)
    keep.append(pb)
    s.n_proc, s.proc_builtin = len(packed.proc_tex_paths), pb.ctypes.data
    if triangle_uvs is not None:  # the real-UV extension: [N,6] float32, one row per triangle record
        tu = np.ascontiguousarray(triangle_uvs, dtype=np.float32).reshape(-1, 6)
        keep.append(tu)
        s.triangle_uvs, s.n_triangle_uvs = tu.ctypes.data, tu.shape[0]
    out = {}
    tr = PtoTrace()
    if trace:
        out["primary_id"] = np.full((h, w), -1, dtype=np.int32)
        out["pixel_stats"] = np.zeros((h, w, NSTAT), dtype=np.int32)
        out["totals"] = np.zeros(NSTAT, dtype=np.int64)
        tr.primary_id = out["primary_id"].ctypes.data
        tr.pixel_stats = out["pixel_stats"].ctypes.data
        tr.totals = out["totals"].ctypes.data
    if nthreads <= 0:
        nthreads = min(os.cpu_count() or 1, 16)
    rc = lib.pto_render(C.byref(s), push, image.ctypes.data, w, h, groups_x, groups_y, C.byref(tr) if trace else None, nthreads)
    if rc != 0:
        raise RuntimeError(f"pto_render failed: {rc}")
    return out


class OracleBvhBuilder:
    """Builder object for work_dispatcher.create_scene_bvh backed by the CPU restatement."""

    def __init__(self):
        self.lib = load()
        self.trees = []

    def build(self, order, bvh_type, aabbs6, object_ids, mesh_bvhs=(), mesh_transforms12=None, mesh_indices=None):
        a = np.ascontiguousarray(aabbs6, dtype=np.float32).reshape(-1, 6)
        ids = np.ascontiguousarray(object_ids, dtype=np.int32)
        nm = len(mesh_bvhs)
        mh = (C.c_void_p * max(1, nm))(*list(mesh_bvhs))
        mt = np.ascontiguousarray(np.array(mesh_transforms12 if nm else [0.0], dtype=np.float32))
        mi = np.ascontiguousarray(np.array(mesh_indices if nm else [0], dtype=np.int32))
        t = self.lib.pbo_build(order, bvh_type, a.shape[0], a.ctypes.data, ids.ctypes.data, nm, mh, mt.ctypes.data, mi.ctypes.data)
        if not t:
            raise RuntimeError("pbo_build failed")
        if self.lib.pbo_error(t) != 0:
            raise RuntimeError(f"pbo_build error {self.lib.pbo_error(t)}")
        return t

    def read(self, t):
        n, ni = self.lib.pbo_node_count(t), self.lib.pbo_object_id_count(t)
        nodes = np.zeros(n * 32, dtype=np.uint8)
        ids = np.zeros(max(ni, 1), dtype=np.int32)
        rc = self.lib.pbo_flatten(t, nodes.ctypes.data, ids.ctypes.data)
        if rc != 0:
            raise RuntimeError(f"pbo_flatten failed: {rc}")
        return nodes.tobytes(), ids[:ni].copy()

    def refit(self, t, object_ids, values12, node_bytes: bytearray):
        """The reference's update_aabb chains, one per event in order; rewrites the queued nodes' records in
        node_bytes (the image read() returned) and returns their flat indices in queue order."""
        ids = np.ascontiguousarray(object_ids, dtype=np.int32)
        v = np.ascontiguousarray(values12, dtype=np.float32).reshape(-1, 12)
        buf = np.frombuffer(node_bytes, dtype=np.uint8)
        queued = np.zeros(self.lib.pbo_node_count(t), dtype=np.int32)
        n = self.lib.pbo_refit(t, len(ids), ids.ctypes.data, v.ctypes.data, buf.ctypes.data, queued.ctypes.data)
        if n < 0:
            raise RuntimeError("pbo_refit: unknown object or mesh")
        return queued[:n].copy()

    def counts(self, t):
        c = np.zeros(4, dtype=np.int32)
        self.lib.pbo_counts(t, c.ctypes.data)
        return dict(inner=int(c[0]), leaf=int(c[1]), objects=int(c[2]), mesh_objects=int(c[3]))

    def free(self, t):
        self.lib.pbo_free(t)

    def release(self, scene_handle, mesh_handles=()):
        """pbo_free releases a scene tree together with the mesh trees merged into it."""
        self.lib.pbo_free(scene_handle)


def algorithmic_bytes(totals, n_planes_padded: int, pixels: int) -> int:
    """Algorithmic bytes of one dispatch in the reference's AoS layout, no cache credit:
    B = 32*box_tests + 4*obj_tests + 32*sphere_tests + 48*tri_tests + 32*P*rays
        + 32*hit_sphere + 32*hit_plane + 80*hit_triangle + 64*material_fetch + 64*sockets
        + pixels*(32 + 128)   (image load+store, push constants)."""
    t = dict(zip(STAT_NAMES, [int(x) for x in totals]))
    return (32 * t["box_tests"] + 4 * t["obj_tests"] + 32 * t["sphere_tests"] + 48 * t["tri_tests"]
            + 32 * n_planes_padded * t["rays"] + 32 * t["hit_sphere"] + 32 * t["hit_plane"] + 80 * t["hit_triangle"]
            + 64 * t["material_fetch"] + 64 * t["sockets"] + pixels * (32 + 128))
