"""Host-side mirror of the reference's scene classes, as far as the render hot
path needs them: the byte records each object contributes and the index rules
that tie them together.

Mirrors (same names, same meaning):
  PTMaterial.to_byte_array          proprietary_dataclasses/public/resources/pt_material.gd:79-107
  PTSphere / PTPlane / PTTriangle   proprietary_dataclasses/public/primitives/*.gd (to_byte_array, _get_aabb, get_bvh_aabb)
  PTObject._get_property_byte_array proprietary_dataclasses/public/pt_object.gd:275-289, object ids :195-205
  PTMesh.to_byte_array              proprietary_dataclasses/public/pt_mesh.gd:161-163
  PTCamera.to_byte_array            proprietary_dataclasses/public/pt_camera.gd:104-136
  PTScene material / texture tables proprietary_dataclasses/public/pt_scene.gd:56-59,124-137,310-332,428-457
  PTObjectContainer index rule      proprietary_dataclasses/pt_object_container.gd:123-131

Everything here is plain data handling on the host; the hot path itself lives
in the HIP library.  Large triangle sets are held as numpy arrays (one
PTTriangleSet = N PTTriangle nodes with identity local transform) so that
packing a 262 144-triangle mesh does not loop in Python.
"""
from __future__ import annotations

import math
import struct
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

from .godot_math import AABB, Transform3D, f32, normalized, cross, vec3, xform_aabb

# PTObject.ObjectType, pt_object.gd:37-43
NOT_OBJECT, SPHERE, PLANE, TRIANGLE, OBJECT_TYPE_MAX = 0, 1, 2, 3, 4
# PTBVHTree.BVHType, pt_bvh_tree.gd:40-46
BVH_EMPTY, BVH_X_SORTED, BVH_Y_SORTED, BVH_Z_SORTED, BVH_XYZ_SORTED = 0, 1, 2, 3, 4
BVH_SAH = 5  # binned surface-area-heuristic builder of libptk (no counterpart in the plugin)


def make_object_id(index: int, type_: int) -> int:
    """pt_object.gd:195-196"""
    return index + (type_ << 24)


@dataclass(eq=False)
class PTMaterial:
    albedo: tuple = (1.0, 1.0, 1.0)
    roughness: float = 0.0
    metallic: float = 0.0
    opacity: float = 1.0
    reflectivity: float = 1.0
    IOR: float = 1.0
    dielectric_priority: int = 0
    is_emissive: bool = False

    def to_byte_array(self) -> bytes:
        return struct.pack(
            "<8f8i",
            self.albedo[0], self.albedo[1], self.albedo[2],
            self.roughness, self.metallic, self.opacity, self.reflectivity, self.IOR,
            self.dielectric_priority, int(self.is_emissive), 0, 0, 0, 0, 0, 0,
        )


@dataclass(eq=False)
class PTTextureAbstract:
    is_spatial: bool = False
    IS_SAMPLE = 1
    IS_SPATIAL = 2

    def get_type(self) -> int:
        t = int(self.is_spatial) * self.IS_SPATIAL
        if isinstance(self, PTSampledTexture):
            t += self.IS_SAMPLE
        return t

    def get_texture_id(self, index: int) -> int:
        return ((self.get_type() << 30) ^ index) & 0xFFFFFFFF


@dataclass(eq=False)
class PTProceduralTexture(PTTextureAbstract):
    texture_path: str = ""


@dataclass(eq=False)
class PTSampledTexture(PTTextureAbstract):
    image: Optional[np.ndarray] = None  # H x W x 4 float32 (RGBAF), None = missing texture


@dataclass(eq=False)
class PTObject:
    transform: Transform3D = field(default_factory=Transform3D)
    material: Optional[PTMaterial] = None
    texture: Optional[PTTextureAbstract] = None
    _scene: Optional["PTScene"] = None
    _mesh: Optional["PTMesh"] = None

    @property
    def is_meshlet(self) -> bool:
        return self._mesh is not None

    @property
    def global_transform(self) -> Transform3D:
        if self._mesh is not None:
            return self._mesh.global_transform.mul(self.transform)
        return self.transform

    def get_type(self) -> int:
        return OBJECT_TYPE_MAX

    def _actual_material(self):
        m = self.material
        if self.is_meshlet and self._mesh.override_material is not None:
            m = self._mesh.override_material
        elif self.is_meshlet and self._mesh.default_material is not None and self.material is None:
            m = self._mesh.default_material
        return m

    def _get_property_byte_array(self) -> bytes:
        return struct.pack(
            "<iIii",
            self._scene.get_material_index(self._actual_material()),
            self._scene.get_texture_id(self.texture),
            0, 0,
        )

    def get_bvh_aabb(self) -> AABB:
        if self.is_meshlet:
            return xform_aabb(self.transform, self._get_aabb())
        return xform_aabb(self.global_transform, self._get_aabb())


@dataclass(eq=False)
class PTSphere(PTObject):
    radius: float = 1.0

    @property
    def position(self):
        return self.transform.origin

    def get_type(self) -> int:
        return SPHERE

    def _get_aabb(self) -> AABB:
        r = f32(self.radius)
        rv = vec3(r, r, r)
        return AABB(self.position - rv, rv * f32(2)).abs()

    def get_bvh_aabb(self) -> AABB:  # pt_sphere.gd:62-65: no transform applied
        return self._get_aabb()

    def to_byte_array(self) -> bytes:
        p = self.position
        return struct.pack("<4f", p[0], p[1], p[2], self.radius) + self._get_property_byte_array()


@dataclass(eq=False)
class PTPlane(PTObject):
    normal: tuple = (0.0, 1.0, 0.0)
    distance: float = 0.0

    def get_type(self) -> int:
        return PLANE

    def to_byte_array(self) -> bytes:
        n = self.normal
        return struct.pack("<4f", n[0], n[1], n[2], self.distance) + self._get_property_byte_array()


@dataclass(eq=False)
class PTTriangle(PTObject):
    vertex1: tuple = (-1.0, 0.0, 0.0)  # Vector3.LEFT
    vertex2: tuple = (0.0, 0.0, -1.0)  # Vector3.FORWARD
    vertex3: tuple = (0.0, 0.0, 0.0)

    def get_type(self) -> int:
        return TRIANGLE

    def _get_aabb(self) -> AABB:  # pt_triangle.gd:107-112
        v1, v2, v3 = (np.asarray(v, dtype=np.float32) for v in (self.vertex1, self.vertex2, self.vertex3))
        temp = AABB(v1, v2 - v1).abs()
        temp = temp.expand(v1)
        temp = temp.expand(v2)
        return temp.expand(v3).abs()

    def to_byte_array(self) -> bytes:  # pt_triangle.gd:125-150
        t = self.transform if self.is_meshlet else self.global_transform
        p1 = t.xform(np.asarray(self.vertex1, dtype=np.float32))
        p2 = t.xform(np.asarray(self.vertex2, dtype=np.float32))
        p3 = t.xform(np.asarray(self.vertex3, dtype=np.float32))
        n = normalized(cross(p3 - p1, p2 - p1))
        return (
            struct.pack("<16f", p1[0], p1[1], p1[2], 0, p2[0], p2[1], p2[2], 0, p3[0], p3[1], p3[2], 0, n[0], n[1], n[2], 0)
            + self._get_property_byte_array()
        )


@dataclass(eq=False)
class PTTriangleSet:
    """N triangles with identity local transform, stored as arrays.

    Equivalent to N PTTriangle children created by
    PTObjectContainer.mesh_to_pttriangles (pt_object_container.gd:163-186).
    vertices: [N,3,3] float32.  materials: optional per-triangle PTMaterial
    choice as (palette, index array)."""

    vertices: np.ndarray = None
    palette: Optional[List[PTMaterial]] = None
    palette_index: Optional[np.ndarray] = None
    texture: Optional[PTTextureAbstract] = None

    def __len__(self):
        return int(self.vertices.shape[0])

    def bvh_aabbs(self) -> np.ndarray:
        """[N,6] position,size per pt_triangle.gd:107-112 followed by Transform3D(identity) * aabb."""
        v = self.vertices.astype(np.float32)
        v1, v2, v3 = v[:, 0], v[:, 1], v[:, 2]
        zero = np.float32(0)
        size0 = v2 - v1
        pos = v1 + np.where(size0 < zero, size0, zero)  # AABB(v1, v2 - v1).abs()
        size = np.abs(size0)
        for p in (v1, v2, v3):  # expand
            begin = pos
            end = pos + size
            begin = np.where(p < begin, p, begin)
            end = np.where(p > end, p, end)
            pos, size = begin, end - begin
        pos = pos + np.where(size < zero, size, zero)  # .abs()
        size = np.abs(size)
        # identity Transform3D * AABB: origin(0) + 1*min + 0*min + 0*min  (zero products keep the value)
        mn, mx = pos, pos + size
        tmin = (np.float32(0) + mn)
        tmax = (np.float32(0) + mx)
        return np.concatenate([tmin, tmax - tmin], axis=1).astype(np.float32)

    def records(self, mtl_index: np.ndarray, texture_id: int) -> np.ndarray:
        """[N,20] float32 view of the 80-byte triangle records (identity transform)."""
        v = self.vertices.astype(np.float32)
        n = v.shape[0]
        rec = np.zeros((n, 20), dtype=np.float32)
        p1, p2, p3 = v[:, 0], v[:, 1], v[:, 2]
        rec[:, 0:3], rec[:, 4:7], rec[:, 8:11] = p1, p2, p3
        a, b = p3 - p1, p2 - p1
        c = np.stack(
            [a[:, 1] * b[:, 2] - a[:, 2] * b[:, 1], a[:, 2] * b[:, 0] - a[:, 0] * b[:, 2], a[:, 0] * b[:, 1] - a[:, 1] * b[:, 0]],
            axis=1,
        ).astype(np.float32)
        lsq = (c[:, 0] * c[:, 0] + c[:, 1] * c[:, 1]) + c[:, 2] * c[:, 2]
        length = np.sqrt(lsq).astype(np.float32)
        with np.errstate(divide="ignore", invalid="ignore"):
            nrm = np.where(lsq[:, None] == 0, np.float32(0), c / length[:, None]).astype(np.float32)
        rec[:, 12:15] = nrm
        ints = rec.view(np.int32)
        ints[:, 16] = mtl_index
        ints.view(np.uint32)[:, 17] = np.uint32(texture_id)
        return rec


@dataclass(eq=False)
class PTMesh:
    transform: Transform3D = field(default_factory=Transform3D)
    bvh_type: int = BVH_X_SORTED
    bvh_order: int = 8
    default_material: Optional[PTMaterial] = None
    override_material: Optional[PTMaterial] = None
    objects: List[PTObject] = field(default_factory=list)
    triangle_set: Optional[PTTriangleSet] = None
    scene: Optional["PTScene"] = None

    @property
    def global_transform(self) -> Transform3D:
        return self.transform  # meshes are direct children of the scene root (a plain Node)

    def add_object(self, obj: PTObject):
        """pt_mesh.gd:117-131: the object joins the mesh and, through it, the scene."""
        obj._mesh = self
        self.objects.append(obj)
        if self.scene is not None:
            self.scene.add_object(obj)

    def to_byte_array(self) -> bytes:
        """global_transform.affine_inverse() as a column-major mat4 with last row (0,0,0,1)."""
        inv = self.global_transform.affine_inverse()
        out = []
        for c in range(3):
            col = inv.column(c)
            out += [col[0], col[1], col[2], 0.0]
        out += [inv.origin[0], inv.origin[1], inv.origin[2], 1.0]
        return struct.pack("<16f", *[float(x) for x in out])


@dataclass(eq=False)
class PTCamera:
    transform: Transform3D = field(default_factory=Transform3D)
    fov: float = 75.0
    focal_length: float = 1.0
    gamma: float = 1.0 / 2.2
    aspect_ratio_x: int = 16
    aspect_ratio_y: int = 9

    # pt_camera.gd:16-25
    camera_settings_values = {
        "top_down": ((0, 8, -15), (0, 0, -6), 106.0),
        "corner": ((-11, 3, -11), (0, 0, 0), 106.0),
        "book_ex": ((13, 2, 3), (0, 0, 0), 20 * 16 / 9.0),
        "center": ((0, 0, 1), (0, 0, 0), 106.0),
        "left": ((0, 0, 1), (-1, 0, 1), 106.0),
        "right": ((0, 0, 1), (1, 0, 1), 106.0),
        "middle": ((7, 2, -3), (0, 0, 0), 30 * 16 / 9.0),
        "cornell": ((-1, 0.5, 0.5), (1, 0.5, 0.5), 106.0),
    }
    camera_setting_names = ["none", "top_down", "corner", "book_ex", "center", "left", "right", "middle", "cornell"]

    @property
    def aspect_ratio(self) -> float:
        return float(self.aspect_ratio_x) / float(self.aspect_ratio_y)

    def set_camera_setting(self, name: str):
        pos, look, fov = self.camera_settings_values[name]
        self.transform = Transform3D.looking_at(pos, look)
        self.fov = fov / self.aspect_ratio

    def viewport(self):
        # fov is a Camera3D property stored as float32
        theta = math.radians(float(f32(self.fov)))
        h = 2 * math.tan(theta / 2.0) * self.focal_length
        return h * self.aspect_ratio, h

    def to_byte_array(self) -> bytes:
        w, h = self.viewport()
        t = self.transform
        pos, right, up, fwd = t.origin, t.column(0), t.column(1), t.column(2)
        vals = [pos[0], pos[1], pos[2], self.focal_length, right[0], right[1], right[2], w,
                up[0], up[1], up[2], h, fwd[0], fwd[1], fwd[2], self.gamma]
        return struct.pack("<16f", *[float(x) for x in vals])


class PTScene:
    """Object / material / texture bookkeeping of pt_scene.gd (registration order matters)."""

    def __init__(self, bvh_order: int = 8, default_bvh: int = BVH_X_SORTED):
        self.bvh_order = bvh_order
        self.default_bvh = default_bvh
        air = PTMaterial(opacity=0.0, reflectivity=0.0)
        self.materials: List[Optional[PTMaterial]] = [air, None]  # pt_scene.gd:124-137
        self._material_to_index = {None: 1}
        self.textures: list = [None]
        self.sampled_textures: List[PTSampledTexture] = []
        self.procedural_textures: list = [None]
        self._texture_to_texture_id = {None: 0}
        # unpacked_objects: every object incl. meshlets, per type, in registration order
        self.spheres: List[PTSphere] = []
        self.planes: List[PTPlane] = []
        self.triangles: list = []  # PTTriangle or (PTTriangleSet, local index) expanded lazily
        self._tri_sets: list = []  # (first_index, PTMesh, PTTriangleSet)
        # scene_objects: objects owned by the scene itself (not meshlets)
        self.scene_spheres: List[PTSphere] = []
        self.scene_triangles: List[PTTriangle] = []
        self.meshes: List[PTMesh] = []
        self.camera: Optional[PTCamera] = None
        self._object_index = {}
        self.n_triangles = 0

    # -- tables ------------------------------------------------------------
    def get_material_index(self, material) -> int:
        return self._material_to_index[material]

    def get_texture_id(self, texture) -> int:
        return self._texture_to_texture_id[texture]

    def get_object_index(self, obj) -> int:
        return self._object_index[id(obj)]

    def get_mesh_index(self, mesh) -> int:
        return self.meshes.index(mesh)

    def _add_material(self, material):
        if material is None:
            return 1
        if material not in self._material_to_index:
            self._material_to_index[material] = len(self.materials)
            self.materials.append(material)
        return self._material_to_index[material]

    def _add_texture(self, texture):
        if texture is None:
            return 0
        if texture not in self._texture_to_texture_id:
            if isinstance(texture, PTSampledTexture):
                index = len(self.sampled_textures)
                self.sampled_textures.append(texture)
            else:
                index = len(self.procedural_textures)
                self.procedural_textures.append(texture)
            self.textures.append(texture)
            self._texture_to_texture_id[texture] = texture.get_texture_id(index)
        return self._texture_to_texture_id[texture]

    # -- registration ------------------------------------------------------
    def add_object(self, obj: PTObject):
        obj._scene = self
        t = obj.get_type()
        arr = {SPHERE: self.spheres, PLANE: self.planes, TRIANGLE: self.triangles}[t]
        if t == TRIANGLE:
            self._object_index[id(obj)] = self.n_triangles
            self.n_triangles += 1
        else:
            self._object_index[id(obj)] = len(arr)
        arr.append(obj)
        if not obj.is_meshlet:
            if t == SPHERE:
                self.scene_spheres.append(obj)
            elif t == TRIANGLE:
                self.scene_triangles.append(obj)
        self._add_material(obj._actual_material())
        self._add_texture(obj.texture)

    def add_mesh(self, mesh: PTMesh):
        """PTMesh._ready + PTScene.add_mesh (pt_scene.gd:240-275): imported triangles join the
        scene-wide arrays, then the mesh and its override/default materials are registered."""
        mesh.scene = self
        self.meshes.append(mesh)
        ts = mesh.triangle_set
        if ts is not None and len(ts):
            self._tri_sets.append((self.n_triangles, mesh, ts))
            self.triangles.append(ts)
            self.n_triangles += len(ts)
            if ts.palette is not None:
                for m in ts.palette:
                    self._add_material(m)
            self._add_texture(ts.texture)
        if mesh.override_material is not None:
            self._add_material(mesh.override_material)
        if mesh.default_material is not None:
            self._add_material(mesh.default_material)

    def tri_set_first_index(self, ts: PTTriangleSet) -> int:
        for first, _, s in self._tri_sets:
            if s is ts:
                return first
        raise KeyError("triangle set not registered")

    def tri_set_material_indices(self, mesh: PTMesh, ts: PTTriangleSet) -> np.ndarray:
        n = len(ts)
        if mesh.override_material is not None:
            return np.full(n, self.get_material_index(mesh.override_material), dtype=np.int32)
        if ts.palette is not None:
            lut = np.array([self.get_material_index(m) for m in ts.palette], dtype=np.int32)
            return lut[ts.palette_index]
        if mesh.default_material is not None:
            return np.full(n, self.get_material_index(mesh.default_material), dtype=np.int32)
        return np.full(n, 1, dtype=np.int32)
