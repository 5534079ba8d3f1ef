"""Host-side mirror of PTWorkDispatcher / PTRenderWindow for the render dispatch.

Same names, argument meaning and byte layouts as
proprietary_dataclasses/pt_work_dispatcher.gd and pt_render_window.gd; the
`rd` member is a PtkRenderingDevice (libptk) instead of Godot's
RenderingDevice.  BVH construction is delegated to a builder object: the
product's DeviceBvhBuilder runs the HIP builder of libptk.
"""
from __future__ import annotations

import struct
import time
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np

from . import scene as sc
from .rendering_device import PtkRenderingDevice

# pt_work_dispatcher.gd:8-19
MAX_TEXTURE_COUNT = 512
SPHERE_COUNT_STEP = 64
PLANE_COUNT_STEP = 16
TRIANGLE_COUNT_STEP = 16
MATERIAL_COUNT_STEP = 16
BVH_COUNT_STEP = 64
NODE_BYTE_SIZE = 32

# pt_work_dispatcher.gd:76-109
IMAGE_SET_INDEX, IMAGE_BUFFER_BIND = 0, 0
OBJECT_SET_INDEX, MATERIALS_BIND, SPHERES_BIND, PLANES_BIND, TRIANGLES_BIND = 2, 0, 1, 2, 3
BVH_SET_INDEX, BVH_BIND, OBJECT_ID_BIND = 3, 0, 1
TRIANGLE_SET_INDEX, TRIANGLE_VERTEX_BIND, TRIANGLE_UV_BIND, TRIANGLE_INDEX_BIND = 4, 0, 1, 2
TEXTURE_SET_INDEX, TEXTURE_BIND = 5, 0
TRANSFORM_SET_INDEX, TRANSFORM_BIND = 6, 0


def ceil_snap(x: int, step: int) -> int:
    """Smallest multiple of step strictly greater than x (pt_work_dispatcher.gd:163-164)."""
    return (x // step + 1) * step


@dataclass
class PTRenderWindow:
    """Render settings that feed the push constants (pt_render_window.gd:16-140)."""

    render_width: int = 1920
    render_height: int = 1080
    enable_multisampling: bool = True
    max_samples: int = 16
    max_ray_depth: int = 1
    max_refraction_bounces: int = 4
    use_bvh: bool = True
    display_node_count: bool = True
    display_object_count: bool = False
    object_display_threshold: int = 40
    node_display_threshold: int = 50
    render_mode: int = 0  # DEFAULT, BVH_DEPTH, NORMAL_VIEW
    x_offset: int = 0
    y_offset: int = 0
    work_group_width: Optional[int] = None
    work_group_height: Optional[int] = None
    work_group_depth: int = 1
    frame: int = 0

    USE_BVH, MULTISAMPLE, SHOW_NODE_COUNT, SHOW_OBJECT_COUNT = 1, 2, 4, 8

    def __post_init__(self):
        if self.work_group_width is None:
            self.work_group_width = self.render_width // 8
        if self.work_group_height is None:
            self.work_group_height = self.render_height // 8

    @property
    def flags(self) -> int:
        f = 0
        if self.use_bvh:
            f |= self.USE_BVH
        if self.enable_multisampling:
            f |= self.MULTISAMPLE
        if self.display_node_count:
            f |= self.SHOW_NODE_COUNT
        if self.display_object_count:
            f |= self.SHOW_OBJECT_COUNT
        return f

    def get_render_mode(self) -> int:
        return self.render_mode


def push_constant_bytes(camera: sc.PTCamera, window: PTRenderWindow, time: float) -> bytes:
    """_push_constant_byte_array (pt_work_dispatcher.gd:771-808) with the wall-clock `time` injected."""
    b = camera.to_byte_array()
    b += struct.pack(
        "<13i",
        window.flags, window.get_render_mode(), 0, 0, 0,
        window.x_offset, window.y_offset,
        window.node_display_threshold, window.object_display_threshold,
        window.render_width, window.render_height,
        window.max_ray_depth, window.max_refraction_bounces,
    )
    b += struct.pack("<f", time)
    b += struct.pack("<2f", float(window.frame), float(window.max_samples))
    assert len(b) == 128
    return b


def sample_time(k: int) -> float:
    """The reference uses fmod(ticks_msec / 1e5, 0.1) (pt_work_dispatcher.gd:782-784); runs that must be
    reproducible use 16 ms per dispatch instead of the wall clock."""
    import math

    return math.fmod(16.0 * k / 100_000.0, (10.0 * 1000) / 100_000.0)


# ---- byte arrays (pt_work_dispatcher.gd:656-768) ---------------------------------------
def create_materials_byte_array(scene: sc.PTScene, buffer_size: int = 0) -> bytes:
    parts = []
    for m in scene.materials:
        parts.append((m if m is not None else sc.PTMaterial()).to_byte_array())
    size = len(scene.materials)
    if buffer_size == 0:
        buffer_size = ceil_snap(size, MATERIAL_COUNT_STEP)
    parts += [sc.PTMaterial().to_byte_array()] * (buffer_size - size)
    return b"".join(parts)


def create_spheres_byte_array(scene: sc.PTScene, buffer_size: int = 0) -> bytes:
    parts = [s.to_byte_array() for s in scene.spheres]
    size = len(scene.spheres)
    if buffer_size == 0:
        buffer_size = ceil_snap(size, SPHERE_COUNT_STEP)
    parts.append(bytes(32 * (buffer_size - size)))
    return b"".join(parts)


def create_planes_byte_array(scene: sc.PTScene, buffer_size: int = 0) -> bytes:
    parts = [p.to_byte_array() for p in scene.planes]
    size = len(scene.planes)
    if buffer_size == 0:
        buffer_size = ceil_snap(size, PLANE_COUNT_STEP)
    parts.append(bytes(32 * (buffer_size - size)))
    return b"".join(parts)


def create_triangles_byte_array(scene: sc.PTScene, buffer_size: int = 0) -> bytes:
    parts = []
    for t in scene.triangles:
        if isinstance(t, sc.PTTriangleSet):
            mesh = next(m for (_, m, s) in scene._tri_sets if s is t)
            rec = t.records(scene.tri_set_material_indices(mesh, t), scene.get_texture_id(t.texture))
            parts.append(rec.tobytes())
        else:
            parts.append(t.to_byte_array())
    size = scene.n_triangles
    if buffer_size == 0:
        buffer_size = ceil_snap(size, TRIANGLE_COUNT_STEP)
    # the reference pads with 32 bytes per missing record, not 80 (:717-718)
    parts.append(bytes(32 * (buffer_size - size)))
    return b"".join(parts)


def create_transform_byte_array(scene: sc.PTScene) -> bytes:
    if scene.meshes:
        return b"".join(m.to_byte_array() for m in scene.meshes)
    return bytes(64)


# ---- BVH creation (pt_scene.gd:659-666, pt_mesh.gd:101, pt_bvh_tree.gd:164-189) -----------
class DeviceBvhBuilder:
    """Builds trees with libptk's HIP builder."""

    def __init__(self, rd: PtkRenderingDevice):
        self.rd = rd
        self.library_seconds = 0.0  # wall time spent inside ptk_bvh_build (mesh trees and scene tree together)

    def build(self, order, bvh_type, aabbs6, object_ids, mesh_bvhs=(), mesh_transforms12=None, mesh_indices=None):
        t0 = time.perf_counter()
        tree = self.rd.bvh_build(order, bvh_type, aabbs6, object_ids, mesh_bvhs, mesh_transforms12, mesh_indices)
        self.library_seconds += time.perf_counter() - t0
        return tree

    def read(self, handle):
        return self.rd.bvh_read(handle)

    def release(self, scene_handle, mesh_handles=()):
        """Frees the trees' handles (and with them the builder's device buffers) once their bytes are read."""
        self.rd.free_rid(scene_handle)
        for h in mesh_handles:
            self.rd.free_rid(h)


def _object_lists(spheres: Sequence[sc.PTSphere], triangles, scene: sc.PTScene):
    """AABBs and object ids in create_disconnected_node_list order (spheres, then triangles)."""
    aabbs: List[np.ndarray] = []
    ids: List[np.ndarray] = []
    if spheres:
        aabbs.append(np.stack([s.get_bvh_aabb().to6() for s in spheres]))
        ids.append(np.array([sc.make_object_id(scene.get_object_index(s), sc.SPHERE) for s in spheres], dtype=np.int32))
    for t in triangles:
        if isinstance(t, sc.PTTriangleSet):
            first = scene.tri_set_first_index(t)
            aabbs.append(t.bvh_aabbs())
            ids.append((np.arange(len(t), dtype=np.int32) + first) + np.int32(sc.TRIANGLE << 24))
        else:
            aabbs.append(t.get_bvh_aabb().to6()[None, :])
            ids.append(np.array([sc.make_object_id(scene.get_object_index(t), sc.TRIANGLE)], dtype=np.int32))
    if not aabbs:
        return np.zeros((0, 6), dtype=np.float32), np.zeros((0,), dtype=np.int32)
    return np.concatenate(aabbs).astype(np.float32), np.concatenate(ids).astype(np.int32)


def build_scene_tree(scene: sc.PTScene, builder, order: int = 8, bvh_type: Optional[int] = None):
    """PTMesh._ready builds each mesh tree, then PTScene.create_BVH builds the scene tree over the
    scene's own objects plus one socket per mesh.  Returns (scene tree handle, mesh tree handles), or
    (None, []) for an empty scene; the caller releases them (builder.release)."""
    bvh_type = scene.default_bvh if bvh_type is None else bvh_type
    mesh_handles, mesh_t, mesh_i = [], [], []
    for mi, mesh in enumerate(scene.meshes):
        sph = [o for o in mesh.objects if o.get_type() == sc.SPHERE]
        tri = [o for o in mesh.objects if o.get_type() == sc.TRIANGLE]
        if mesh.triangle_set is not None and len(mesh.triangle_set):
            tri.append(mesh.triangle_set)
        aabbs, ids = _object_lists(sph, tri, scene)
        if aabbs.shape[0] == 0:
            continue
        mesh_handles.append(builder.build(mesh.bvh_order, mesh.bvh_type, aabbs, ids))
        mesh_t.append(mesh.transform.to_list())
        mesh_i.append(mi)
    aabbs, ids = _object_lists(scene.scene_spheres, scene.scene_triangles, scene)
    if aabbs.shape[0] == 0 and not mesh_handles:
        return None, []
    return builder.build(order, bvh_type, aabbs, ids, mesh_handles, mesh_t, mesh_i), mesh_handles


def create_scene_bvh(scene: sc.PTScene, builder, order: int = 8, bvh_type: Optional[int] = None):
    """build_scene_tree, then the tree's bytes: (node_bytes, object_ids) in GPU order, or (None, None) for an
    empty scene."""
    h, mesh_handles = build_scene_tree(scene, builder, order, bvh_type)
    if h is None:
        return None, None
    try:
        return builder.read(h)
    finally:
        builder.release(h, mesh_handles)


def create_bvh_byte_array(node_bytes: Optional[bytes], buffer_size: int = 0) -> bytes:
    """_create_bvh_byte_array padding rule (:723-744)."""
    if node_bytes is None:
        node_bytes = bytes(NODE_BYTE_SIZE)
    size = len(node_bytes) // NODE_BYTE_SIZE
    if buffer_size == 0:
        buffer_size = ceil_snap(size, BVH_COUNT_STEP)
    return node_bytes + bytes(NODE_BYTE_SIZE * (buffer_size - size))


def create_object_id_byte_array(object_ids) -> bytes:
    if object_ids is None or len(object_ids) == 0:
        return bytes(8)
    return np.ascontiguousarray(object_ids, dtype=np.int32).tobytes()


@dataclass
class PackedScene:
    """Everything PTWorkDispatcher.create_buffers uploads, as bytes."""

    materials: bytes
    spheres: bytes
    planes: bytes
    triangles: bytes
    bvh: bytes
    object_ids: bytes
    transforms: bytes
    proc_tex_paths: List[str]
    sampled_textures: list
    bvh_order: int


def pack_scene(scene: sc.PTScene, builder, order: int = 8, bvh_type: Optional[int] = None, tree_bytes=None) -> PackedScene:
    """Everything create_buffers uploads.  tree_bytes = (node_bytes, object_ids) when the tree's bytes are
    already at hand (a tree bound on the device), else the tree is built with `builder` and read back."""
    nodes, ids = tree_bytes if tree_bytes is not None else create_scene_bvh(scene, builder, order, bvh_type)
    return PackedScene(
        materials=create_materials_byte_array(scene),
        spheres=create_spheres_byte_array(scene),
        planes=create_planes_byte_array(scene),
        triangles=create_triangles_byte_array(scene),
        bvh=create_bvh_byte_array(nodes),
        object_ids=create_object_id_byte_array(ids),
        transforms=create_transform_byte_array(scene),
        proc_tex_paths=[t.texture_path for t in scene.procedural_textures[1:]],
        sampled_textures=[t.image for t in scene.sampled_textures],
        bvh_order=scene.bvh_order,
    )


class PTWorkDispatcher:
    """create_buffers / create_compute_list on top of libptk."""

    def __init__(self, rd: Optional[PtkRenderingDevice] = None, device_ordinal: int = 0):
        self.rd = rd if rd is not None else PtkRenderingDevice(device_ordinal)
        self._scene: Optional[sc.PTScene] = None
        self.packed: Optional[PackedScene] = None
        self.rids_to_free: List[int] = []
        self.image_buffer = 0
        self.material_buffer = self.sphere_buffer = self.plane_buffer = self.triangle_buffer = 0
        self.bvh_buffer = self.object_id_buffer = self.transform_buffer = 0
        self.bvh_library_seconds = (0.0, 0.0)
        self.triangle_vertex_buffer = self.triangle_uv_buffer = self.triangle_index_buffer = 0

    def set_scene(self, scene: sc.PTScene):
        self._scene = scene

    def load_shader(self, refraction_mode=2, use_emission=False, early_t_break=False, packed: Optional[PackedScene] = None):
        """Pipeline selection: PTUtils.load_shader's text edits become ptk_configure arguments."""
        p = packed if packed is not None else self.packed
        self.rd.configure(p.bvh_order if p else 8, refraction_mode, use_emission, early_t_break,
                          p.proc_tex_paths if p else ())

    def _create_uniform(self, data: bytes, set_index: int, binding: int) -> int:
        h = self.rd.storage_buffer_create(set_index, binding, data)
        self.rids_to_free.append(h)
        return h

    def create_buffers(self, width: int, height: int, packed: Optional[PackedScene] = None, order: int = 8,
                       bvh_type: Optional[int] = None):
        assert self._scene is not None or packed is not None, "Set a scene before trying to create gpu buffers."
        bound = None
        if packed is None:
            # the tree is built on the GPU and bound where it lies (ptk_bvh_bind); its bytes are read back only
            # to complete the host-side mirror of the buffers
            builder = DeviceBvhBuilder(self.rd)
            tree, mesh_trees = build_scene_tree(self._scene, builder, order, bvh_type)
            if tree is not None:
                t0 = time.perf_counter()
                bound = self.rd.bvh_bind(tree)
                # what the library itself took for the scene's trees: ptk_bvh_build calls, ptk_bvh_bind
                self.bvh_library_seconds = (builder.library_seconds, time.perf_counter() - t0)
                builder.release(tree, mesh_trees)  # the bound tree keeps what it needs for refits
                n_nodes, n_ids = self.rd.bvh_sizes_of_bound(bound)
                nodes = self.rd.buffer_get_data(bound[0], 0, n_nodes * NODE_BYTE_SIZE)
                ids = np.frombuffer(self.rd.buffer_get_data(bound[1], 0, max(n_ids, 1) * 4), dtype=np.int32)[:n_ids]
                packed = pack_scene(self._scene, None, order, bvh_type, tree_bytes=(nodes, ids))
            else:
                packed = pack_scene(self._scene, None, order, bvh_type, tree_bytes=(None, None))
        self.packed = packed
        self.image_buffer = self.rd.image_create(width, height)
        self.rids_to_free.append(self.image_buffer)
        self.material_buffer = self._create_uniform(packed.materials, OBJECT_SET_INDEX, MATERIALS_BIND)
        self.sphere_buffer = self._create_uniform(packed.spheres, OBJECT_SET_INDEX, SPHERES_BIND)
        self.plane_buffer = self._create_uniform(packed.planes, OBJECT_SET_INDEX, PLANES_BIND)
        self.triangle_buffer = self._create_uniform(packed.triangles, OBJECT_SET_INDEX, TRIANGLES_BIND)
        if bound is not None:
            self.bvh_buffer, self.object_id_buffer = bound
            self.rids_to_free += [self.bvh_buffer, self.object_id_buffer]
        else:
            self.object_id_buffer = self._create_uniform(packed.object_ids, BVH_SET_INDEX, OBJECT_ID_BIND)
            self.bvh_buffer = self._create_uniform(packed.bvh, BVH_SET_INDEX, BVH_BIND)
        # set 4 is bound but never read by the kernel; placeholders like create_triangle_buffers (:265-269)
        self.triangle_vertex_buffer = self._create_uniform(bytes(12), TRIANGLE_SET_INDEX, TRIANGLE_VERTEX_BIND)
        self.triangle_uv_buffer = self._create_uniform(bytes(12), TRIANGLE_SET_INDEX, TRIANGLE_UV_BIND)
        self.triangle_index_buffer = self._create_uniform(bytes(12), TRIANGLE_SET_INDEX, TRIANGLE_INDEX_BIND)
        for slot, img in enumerate(packed.sampled_textures):
            self.rids_to_free.append(self.rd.texture_create(slot, img))
        self.transform_buffer = self._create_uniform(packed.transforms, TRANSFORM_SET_INDEX, TRANSFORM_BIND)
        # record counts the buffers hold (pt_work_dispatcher.gd:56-63)
        self.material_buffer_size = len(packed.materials) // 64
        self.sphere_buffer_size = len(packed.spheres) // 32
        self.plane_buffer_size = len(packed.planes) // 32
        # (the triangle buffer's padding is 32 bytes per missing record, :717-718, so its byte length does not give the count)
        self.triangle_buffer_size = ceil_snap(self._scene.n_triangles, TRIANGLE_COUNT_STEP) if self._scene is not None else len(packed.triangles) // 80
        self.bvh_buffer_size = len(packed.bvh) // NODE_BYTE_SIZE

    # ---- buffer growth (pt_work_dispatcher.gd:511-637) --------------------------------------------------
    def _replace_uniform(self, old: int, data: bytes, set_index: int, binding: int) -> int:
        """free_rid + create_*_buffer + uniform_set_create: the new buffer takes the old one's binding."""
        if old in self.rids_to_free:
            self.rids_to_free.remove(old)
        self.rd.free_rid(old)
        return self._create_uniform(data, set_index, binding)

    def expand_object_buffer(self, object_type: int, steps: int = 0) -> bool:
        """Grows one object buffer to fit the scene (steps < 1) or by `steps` count steps, refilled from the
        scene's current objects; False when it already fits (:511-567)."""
        scn = self._scene
        table = {
            sc.SPHERE: ("sphere_buffer", "sphere_buffer_size", len(scn.spheres), SPHERE_COUNT_STEP, create_spheres_byte_array, SPHERES_BIND),
            sc.PLANE: ("plane_buffer", "plane_buffer_size", len(scn.planes), PLANE_COUNT_STEP, create_planes_byte_array, PLANES_BIND),
            sc.TRIANGLE: ("triangle_buffer", "triangle_buffer_size", scn.n_triangles, TRIANGLE_COUNT_STEP,
                          create_triangles_byte_array, TRIANGLES_BIND),
        }
        if object_type not in table:
            return False
        buf, size_name, count, step, packer, binding = table[object_type]
        size = getattr(self, size_name)
        if steps < 1:
            if size >= count:
                return False
            new_size = ceil_snap(count, step)
            if new_size <= size:
                return False
        else:
            new_size = size + step * steps
        setattr(self, size_name, new_size)
        setattr(self, buf, self._replace_uniform(getattr(self, buf), packer(scn, new_size), OBJECT_SET_INDEX, binding))
        return True

    def check_object_buffer_size(self) -> bool:
        """Expands every object buffer the scene has outgrown (:570-581)."""
        grown = False
        for t in (sc.SPHERE, sc.PLANE, sc.TRIANGLE):
            grown = self.expand_object_buffer(t) or grown
        return grown

    def expand_material_buffer(self, steps: int = 0) -> bool:
        scn = self._scene
        if steps < 1:
            if self.material_buffer_size >= len(scn.materials):
                return False
            new_size = ceil_snap(len(scn.materials), MATERIAL_COUNT_STEP)
            if new_size <= self.material_buffer_size:
                return False
        else:
            new_size = self.material_buffer_size + MATERIAL_COUNT_STEP * steps
        self.material_buffer_size = new_size
        self.material_buffer = self._replace_uniform(self.material_buffer, create_materials_byte_array(scn, new_size), OBJECT_SET_INDEX, MATERIALS_BIND)
        return True

    def expand_bvh_buffer(self, node_bytes: bytes, object_ids, steps: int = 0) -> bool:
        """Grows the node buffer for a tree that no longer fits (:584-603) and refills it and the id table from the
        given tree; with a tree that still fits, both are rewritten in place."""
        n = len(node_bytes) // NODE_BYTE_SIZE
        grow = steps >= 1 or self.bvh_buffer_size < n
        if grow:
            self.bvh_buffer_size = self.bvh_buffer_size + BVH_COUNT_STEP * steps if steps >= 1 else ceil_snap(n, BVH_COUNT_STEP)
            self.bvh_buffer = self._replace_uniform(self.bvh_buffer, create_bvh_byte_array(node_bytes, self.bvh_buffer_size), BVH_SET_INDEX, BVH_BIND)
        else:
            data = create_bvh_byte_array(node_bytes, self.bvh_buffer_size)
            self.rd.buffer_update(self.bvh_buffer, 0, len(data), data)
        self.object_id_buffer = self._replace_uniform(self.object_id_buffer, create_object_id_byte_array(object_ids), BVH_SET_INDEX, OBJECT_ID_BIND)
        return grow

    def create_compute_list(self, window: PTRenderWindow, camera: Optional[sc.PTCamera] = None, time: float = 0.0,
                            sync: bool = False):
        cam = camera if camera is not None else self._scene.camera
        push = push_constant_bytes(cam, window, time)
        self.rd.compute_list_dispatch(push, window.work_group_width, window.work_group_height, window.work_group_depth)
        if sync:
            self.rd.sync()

    def free_rids(self):
        for h in self.rids_to_free:
            self.rd.free_rid(h)
        self.rids_to_free = []
