"""A RenderingDevice with Godot's own method names and signatures on top of libptk.

PTWorkDispatcher drives its compute shader through Godot's RenderingDevice
(proprietary_dataclasses/pt_work_dispatcher.gd:146-160, 297-497, 642-653) and PTRenderer touches the same object
directly (pt_renderer.gd:327, 658-788).  This class answers exactly those calls -- same names, same argument
order -- so that a dispatcher written against Godot's API runs unchanged when its `rd` is one of these.  It is
the ctypes twin of gdextension/src/ptk_godot_rd.cpp; both are thin: every call ends in one entry point of
include/ptk.h.

    shader_compile_spirv_from_source / shader_create_from_spirv / compute_pipeline_create
                                   -> ptk_configure_from_source (nothing is compiled: kernel variants are selected)
    storage_buffer_create          -> ptk_buffer_create(PTK_SET_NONE)
    texture_create                 -> ptk_image_create (storage usage) / ptk_texture_create(slot -1) (sampling usage)
    uniform_set_create             -> remembered; compute_list_bind_uniform_set -> ptk_buffer_bind / ptk_texture_bind
    compute_list_dispatch          -> ptk_dispatch with the bytes of compute_list_set_push_constant
    buffer_update, texture_get_data, free_rid, submit, sync -> their ptk_ namesakes

There is no CPU path: construction raises without the HIP library or a GPU.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np

from .rendering_device import PTK_SET_NONE, PtkError, PtkRenderingDevice


class RID:
    """Godot's opaque resource id.  kind tells the facade what the id stands for; handle is the libptk handle
    (buffers, images, textures) or a facade-local number (shaders, pipelines, samplers, uniform sets)."""

    __slots__ = ("kind", "handle")

    def __init__(self, kind: str = "", handle: int = 0):
        self.kind = kind
        self.handle = handle

    def is_valid(self) -> bool:
        return self.handle != 0

    def get_id(self) -> int:
        return self.handle

    def __eq__(self, other):
        return isinstance(other, RID) and other.kind == self.kind and other.handle == self.handle

    def __hash__(self):
        return hash((self.kind, self.handle))

    def __repr__(self):
        return f"RID({self.kind}:{self.handle})"


class RDShaderSource:
    def __init__(self):
        self.source_compute = ""


class RDShaderSPIRV:
    def __init__(self, source: str = ""):
        self._source = source
        self.compile_error_compute = ""


class RDUniform:
    def __init__(self):
        self.uniform_type = 0
        self.binding = 0
        self._ids: List[RID] = []

    def add_id(self, rid: RID):
        self._ids.append(rid)

    def get_ids(self) -> List[RID]:
        return list(self._ids)

    def clear_ids(self):
        self._ids = []


class RDTextureFormat:
    def __init__(self):
        self.format = 0
        self.texture_type = 0
        self.width = 1
        self.height = 1
        self.depth = 1
        self.array_layers = 1
        self.mipmaps = 1
        self.usage_bits = 0


class RDTextureView:
    pass


class RDSamplerState:
    """Defaults are Godot's: nearest filtering, clamp to edge -- the only sampler the kernels implement."""

    def __init__(self):
        self.mag_filter = 0  # SAMPLER_FILTER_NEAREST
        self.min_filter = 0
        self.repeat_u = 2    # SAMPLER_REPEAT_MODE_CLAMP_TO_EDGE
        self.repeat_v = 2


class RenderingDevice:
    """See the module docstring.  Constants carry Godot's values (RenderingDevice enum)."""

    UNIFORM_TYPE_SAMPLER = 0
    UNIFORM_TYPE_SAMPLER_WITH_TEXTURE = 1
    UNIFORM_TYPE_TEXTURE = 2
    UNIFORM_TYPE_IMAGE = 3
    UNIFORM_TYPE_STORAGE_BUFFER = 8
    DATA_FORMAT_R8G8B8A8_UNORM = 36
    DATA_FORMAT_R32G32B32A32_SFLOAT = 109
    TEXTURE_TYPE_2D = 1
    TEXTURE_USAGE_SAMPLING_BIT = 1
    TEXTURE_USAGE_COLOR_ATTACHMENT_BIT = 2
    TEXTURE_USAGE_STORAGE_BIT = 8
    TEXTURE_USAGE_CAN_UPDATE_BIT = 64
    TEXTURE_USAGE_CAN_COPY_FROM_BIT = 128

    def __init__(self, device_ordinal: int = 0, ptk: Optional[PtkRenderingDevice] = None):
        self.ptk = ptk if ptk is not None else PtkRenderingDevice(device_ordinal)
        self._next_local = 1
        self._uniform_sets = {}      # local id -> (shader_set, [(uniform_type, binding, [RID])])
        self._list_open = False
        self._push: Optional[bytes] = None
        self._pipeline_bound = False
        self._bound_set = {}         # set index -> local id of the uniform set whose resources are placed there now

    # ---- helpers ---------------------------------------------------------------------------------------
    def _local(self, kind: str) -> RID:
        r = RID(kind, self._next_local)
        self._next_local += 1
        return r

    def close(self):
        self.ptk.close()

    # ---- shader / pipeline (pt_work_dispatcher.gd:436-445) -------------------------------------------------
    def shader_compile_spirv_from_source(self, shader_source: RDShaderSource, allow_cache: bool = True) -> RDShaderSPIRV:
        spirv = RDShaderSPIRV(shader_source.source_compute)
        try:
            self.ptk.shader_source_info(shader_source.source_compute)
        except PtkError as e:  # Godot reports compile errors in the SPIR-V object
            spirv.compile_error_compute = str(e)
        return spirv

    def shader_create_from_spirv(self, spirv_data: RDShaderSPIRV, name: str = "") -> RID:
        if spirv_data.compile_error_compute:
            return RID()
        try:
            self.ptk.configure_from_source(spirv_data._source)
        except PtkError as e:
            spirv_data.compile_error_compute = str(e)
            return RID()  # Godot returns an invalid RID and prints the error
        return self._local("shader")

    def compute_pipeline_create(self, shader: RID, specialization_constants: Sequence = ()) -> RID:
        if not shader.is_valid() or shader.kind != "shader":
            return RID()
        return self._local("pipeline")

    # ---- resources ---------------------------------------------------------------------------------------
    def storage_buffer_create(self, size_bytes: int, data: bytes = b"", usage: int = 0) -> RID:
        b = bytes(data) if len(data) else bytes(size_bytes)
        if len(b) != size_bytes:
            raise ValueError("storage_buffer_create: data must be size_bytes long")
        return RID("buffer", self.ptk.storage_buffer_create(PTK_SET_NONE, 0, b))

    def texture_create(self, format: RDTextureFormat, view: RDTextureView, data: Sequence = ()) -> RID:
        if format.usage_bits & self.TEXTURE_USAGE_STORAGE_BIT:
            if format.format != self.DATA_FORMAT_R32G32B32A32_SFLOAT:
                raise PtkError("storage images are rgba32f (pt_work_dispatcher.gd:388-393)")
            h = self.ptk.image_create(format.width, format.height)
            if len(data):
                self.ptk.texture_update(h, np.frombuffer(bytes(data[0]), np.float32).reshape(format.height, format.width, 4))
            return RID("image", h)
        if not len(data):
            raise PtkError("a sampled texture needs its level-0 data")
        raw = bytes(data[0])
        n = format.width * format.height * 4
        if format.format == self.DATA_FORMAT_R32G32B32A32_SFLOAT:
            rgba = np.frombuffer(raw, np.float32, count=n)  # level 0 comes first; the mip chain behind it is not sampled
        elif format.format == self.DATA_FORMAT_R8G8B8A8_UNORM:
            rgba = np.frombuffer(raw, np.uint8, count=n).astype(np.float32) / np.float32(255.0)  # what UNORM sampling returns
        else:
            raise PtkError(f"texture format {format.format} is not one the plugin creates")
        return RID("texture", self.ptk.texture_create(-1, rgba.reshape(format.height, format.width, 4)))

    def sampler_create(self, state: RDSamplerState) -> RID:
        if (state.mag_filter, state.min_filter) != (0, 0) or (state.repeat_u, state.repeat_v) != (2, 2):
            raise PtkError("only Godot's default sampler (nearest, clamp to edge) is implemented")
        return self._local("sampler")

    def uniform_set_create(self, uniforms: Sequence[RDUniform], shader: RID, shader_set: int) -> RID:
        recorded = [(u.uniform_type, u.binding, u.get_ids()) for u in uniforms if u is not None]
        rid = self._local("uniform_set")
        self._uniform_sets[rid.handle] = (shader_set, recorded)
        return rid

    def buffer_update(self, buffer: RID, offset: int, size_bytes: int, data: bytes):
        self.ptk.buffer_update(buffer.handle, offset, size_bytes, data)

    def buffer_get_data(self, buffer: RID, offset_bytes: int = 0, size_bytes: int = 0) -> bytes:
        return self.ptk.buffer_get_data(buffer.handle, offset_bytes, size_bytes)

    def texture_get_data(self, texture: RID, layer: int) -> bytes:
        return self.ptk.texture_get_data(texture.handle).tobytes()

    def texture_update(self, texture: RID, layer: int, data: bytes):
        w, h = self.ptk._images[texture.handle]
        self.ptk.texture_update(texture.handle, np.frombuffer(bytes(data), np.float32).reshape(h, w, 4))

    def free_rid(self, rid: RID):
        if rid.kind in ("buffer", "image", "texture"):
            self.ptk.free_rid(rid.handle)
        elif rid.kind == "uniform_set":
            self._uniform_sets.pop(rid.handle, None)
            for k in [k for k, v in self._bound_set.items() if v == rid.handle]:
                del self._bound_set[k]
        # shaders, pipelines and samplers hold nothing on the device

    # ---- compute list (pt_work_dispatcher.gd:470-497) -------------------------------------------------------
    def compute_list_begin(self) -> int:
        if self._list_open:
            raise PtkError("compute_list_begin: a compute list is already open")
        self._list_open = True
        self._push = None
        self._pipeline_bound = False
        return 1

    def compute_list_bind_compute_pipeline(self, compute_list: int, compute_pipeline: RID):
        if not compute_pipeline.is_valid() or compute_pipeline.kind != "pipeline":
            raise PtkError("compute_list_bind_compute_pipeline: invalid pipeline")
        self._pipeline_bound = True

    def compute_list_bind_uniform_set(self, compute_list: int, uniform_set: RID, set_index: int):
        shader_set, recorded = self._uniform_sets[uniform_set.handle]
        if shader_set != set_index:
            raise PtkError(f"uniform set was created for set {shader_set}, bound at {set_index}")
        if self._bound_set.get(set_index) == uniform_set.handle:
            return  # the plugin binds every set before every dispatch; nothing moved
        self._bound_set[set_index] = uniform_set.handle
        for utype, binding, ids in recorded:
            if utype == self.UNIFORM_TYPE_STORAGE_BUFFER or utype == self.UNIFORM_TYPE_IMAGE:
                self.ptk.buffer_bind(ids[0].handle, set_index, binding)
            elif utype == self.UNIFORM_TYPE_SAMPLER_WITH_TEXTURE:
                # ids alternate sampler, texture: pair k is element k of the shader's sampler array
                for k in range(len(ids) // 2):
                    self.ptk.texture_bind(ids[2 * k + 1].handle, k)
            else:
                raise PtkError(f"uniform type {utype} is not one the plugin binds")

    def compute_list_set_push_constant(self, compute_list: int, buffer: bytes, size_bytes: int):
        self._push = bytes(buffer[:size_bytes])

    def compute_list_dispatch(self, compute_list: int, x_groups: int, y_groups: int, z_groups: int):
        if not self._list_open or not self._pipeline_bound or self._push is None:
            raise PtkError("compute_list_dispatch needs an open list with a pipeline and push constants")
        self.ptk.compute_list_dispatch(self._push, x_groups, y_groups, z_groups)

    def compute_list_end(self):
        self._list_open = False

    def submit(self):
        pass  # dispatches are queued on the context's stream as they are recorded

    def sync(self):
        self.ptk.sync()
