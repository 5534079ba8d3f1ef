"""A work dispatcher that knows nothing but Godot's RenderingDevice API.

It follows PTWorkDispatcher call for call (proprietary_dataclasses/pt_work_dispatcher.gd:168-190 create_buffers,
:297-405 textures and image, :408-445 uniform sets and shader, :470-497 compute list, :642-653 _create_uniform)
and PTUtils.load_shader's text edits (pt_utils.gd:140-195), so that running it against
path-tracer_amd/godot_rd.RenderingDevice shows what the plugin's own dispatcher would get from that object: the
only thing that differs from the plugin is which class `rd` is.  The byte arrays come from the packers of
path-tracer_amd/work_dispatcher.py (a PackedScene)."""
import importlib

pta = importlib.import_module("path-tracer_amd")
grd = importlib.import_module("path-tracer_amd.godot_rd")
wd = pta.work_dispatcher

RD = grd.RenderingDevice

# A stand-in for shaders/ray_tracer.comp with just the lines PTUtils.load_shader edits and the switches the
# library reads: the shader itself does not travel to the GPU box.
SHADER_TEMPLATE = """#version 450
#define use_refraction_tracking {refraction}
#define USE_EMISSION {emission}
#define use_early_t_break {early_t}
const int order = 2; // set by the script
vec2 sphere_get_uv(vec3 pos) {{
    return vec2(atan(-pos.z, pos.x), acos(pos.y));
}}
//procedural_texture_function_definition_hook
vec4 procedural_texture(vec3 pos, uint function) {{
    if (function == 0) {{ return vec4(1); }}
    //procedural_texture_function_call_hook
    return vec4(0);
}}
void main() {{}}
"""

# Stand-ins for the scene's procedural-texture files (own wording; registered with the library as the built-in
# functions they compute, since it recognises files by their text).
CHECKER_XZ = """
// checker in x and z
vec4 procedural_texture(vec3 pos) {
    vec3 cell = floor(pos * 2.0);
    return bool(mod(cell.x + cell.z, 2.0)) ? vec4(0.9, 0.9, 0.9, 1) : vec4(0.2, 0.3, 0.1, 1);
}
"""
CHECKER_XYZ = """
/* checker in x, y and z */
vec4 procedural_texture(vec3 pos) {
    vec3 cell = floor(pos * 2.0);
    return bool(mod(cell.x + cell.y + cell.z, 2.0)) ? vec4(0.9, 0.9, 0.9, 1) : vec4(0.2, 0.3, 0.1, 1);
}
"""
STAND_IN_FILES = {"checker_board.comp": CHECKER_XZ, "checker_board_spatial.comp": CHECKER_XYZ}


def load_shader_source(template: str, bvh_order: int, procedural_texts) -> "grd.RDShaderSource":
    """PTUtils.load_shader (pt_utils.gd:140-195): the order constant, then every procedural texture file with its
    function renamed and a call spliced into the selector."""
    res = template.replace("const int order = 2;", "const int order = %s;" % bvh_order)
    definitions, calls = "", ""
    for i, text in enumerate(procedural_texts, start=1):
        definitions += text.replace("procedural_texture", "_procedural_texture_function" + str(i))
        calls += "else if (function == %s) {\n\t\treturn _procedural_texture_function%s(pos);\n\t}\n\t" % (i, i)
    res = res.replace("//procedural_texture_function_definition_hook", definitions)
    res = res.replace("//procedural_texture_function_call_hook", calls)
    src = grd.RDShaderSource()
    src.source_compute = res
    return src


class UniformStorage:
    """pt_work_dispatcher.gd:811-851."""

    SIZES = {wd.IMAGE_SET_INDEX: 1, wd.OBJECT_SET_INDEX: 4, wd.BVH_SET_INDEX: 2, wd.TRIANGLE_SET_INDEX: 3,
             wd.TEXTURE_SET_INDEX: 1, wd.TRANSFORM_SET_INDEX: 1}

    def __init__(self):
        self.sets = {k: [None] * n for k, n in self.SIZES.items()}

    def get_set_uniforms(self, index):
        return self.sets[index]

    def set_uniform(self, set_index, bind_index, uniform):
        self.sets[set_index][bind_index] = uniform


class GodotStyleDispatcher:
    SET_ORDER = (wd.IMAGE_SET_INDEX, wd.OBJECT_SET_INDEX, wd.BVH_SET_INDEX, wd.TRIANGLE_SET_INDEX, wd.TEXTURE_SET_INDEX,
                 wd.TRANSFORM_SET_INDEX)

    def __init__(self, rd, is_local=True):
        self.rd = rd
        self.is_local_renderer = is_local
        self.uniforms = UniformStorage()
        self.rids_to_free = []
        self.set_rids = {}
        self.shader = self.pipeline = None

    # -- pt_work_dispatcher.gd:436-445
    def load_shader(self, shader_source):
        spirv = self.rd.shader_compile_spirv_from_source(shader_source)
        self.shader = self.rd.shader_create_from_spirv(spirv)
        assert self.shader.is_valid(), spirv.compile_error_compute
        self.rids_to_free.append(self.shader)
        self.pipeline = self.rd.compute_pipeline_create(self.shader)

    # -- :642-653
    def _create_uniform(self, data: bytes, set_index: int, binding: int):
        buffer = self.rd.storage_buffer_create(len(data), data)
        self.rids_to_free.append(buffer)
        uniform = grd.RDUniform()
        uniform.uniform_type = RD.UNIFORM_TYPE_STORAGE_BUFFER
        uniform.binding = binding
        uniform.add_id(buffer)
        self.uniforms.set_uniform(set_index, binding, uniform)
        return buffer

    # -- :374-405
    def _create_image_buffer(self, width, height):
        tf = grd.RDTextureFormat()
        tf.format = RD.DATA_FORMAT_R32G32B32A32_SFLOAT
        tf.texture_type = RD.TEXTURE_TYPE_2D
        tf.width, tf.height = width, height
        tf.usage_bits = (RD.TEXTURE_USAGE_SAMPLING_BIT + RD.TEXTURE_USAGE_COLOR_ATTACHMENT_BIT + RD.TEXTURE_USAGE_STORAGE_BIT +
                         RD.TEXTURE_USAGE_CAN_UPDATE_BIT + RD.TEXTURE_USAGE_CAN_COPY_FROM_BIT)
        image = self.rd.texture_create(tf, grd.RDTextureView(), [])
        self.rids_to_free.append(image)
        uniform = grd.RDUniform()
        uniform.uniform_type = RD.UNIFORM_TYPE_IMAGE
        uniform.binding = wd.IMAGE_BUFFER_BIND
        uniform.add_id(image)
        self.uniforms.set_uniform(wd.IMAGE_SET_INDEX, wd.IMAGE_BUFFER_BIND, uniform)
        return image

    # -- :350-370; the grid's bytes are PTUtils.create_missing_texture_grid(8) (pt_utils.gd:122-137)
    def _create_fill_texture(self):
        black, pink = [0, 0, 0, 56], [56, 16, 56, 56]
        grid = []
        for i in range(8):
            for j in range(8):
                grid += (pink, black)[((i % 2) + (j % 2)) % 2]
        tf = grd.RDTextureFormat()
        tf.format = RD.DATA_FORMAT_R8G8B8A8_UNORM
        tf.texture_type = RD.TEXTURE_TYPE_2D
        tf.width = tf.height = 8
        tf.usage_bits = RD.TEXTURE_USAGE_SAMPLING_BIT + RD.TEXTURE_USAGE_CAN_UPDATE_BIT
        tex = self.rd.texture_create(tf, grd.RDTextureView(), [bytes(grid)])
        self.rids_to_free.append(tex)
        return tex

    # -- :297-347
    def create_texture_buffers(self, sampled_textures):
        uniform = grd.RDUniform()
        uniform.binding = wd.TEXTURE_BIND
        uniform.uniform_type = RD.UNIFORM_TYPE_SAMPLER_WITH_TEXTURE
        fill_texture = self._create_fill_texture()
        sampler = self.rd.sampler_create(grd.RDSamplerState())
        for img in sampled_textures:
            if img is None:
                uniform.add_id(sampler)
                uniform.add_id(fill_texture)
                continue
            tf = grd.RDTextureFormat()
            tf.format = RD.DATA_FORMAT_R32G32B32A32_SFLOAT
            tf.texture_type = RD.TEXTURE_TYPE_2D
            tf.width, tf.height = img.shape[1], img.shape[0]
            tf.mipmaps = 1
            tf.usage_bits = RD.TEXTURE_USAGE_SAMPLING_BIT + RD.TEXTURE_USAGE_CAN_UPDATE_BIT
            tex = self.rd.texture_create(tf, grd.RDTextureView(), [img.astype("float32").tobytes()])
            self.rids_to_free.append(tex)
            uniform.add_id(sampler)
            uniform.add_id(tex)
        for _ in range(wd.MAX_TEXTURE_COUNT - len(sampled_textures)):
            uniform.add_id(sampler)
            uniform.add_id(fill_texture)
        self.uniforms.set_uniform(wd.TEXTURE_SET_INDEX, wd.TEXTURE_BIND, uniform)

    # -- :408-433
    def bind_set(self, index):
        self.set_rids[index] = self.rd.uniform_set_create(self.uniforms.get_set_uniforms(index), self.shader, index)

    def bind_sets(self):
        for index in self.SET_ORDER:
            self.bind_set(index)

    # -- :168-190 (the byte arrays are the packers' output)
    def create_buffers(self, width, height, packed):
        self.image_buffer = self._create_image_buffer(width, height)
        self.material_buffer = self._create_uniform(packed.materials, wd.OBJECT_SET_INDEX, wd.MATERIALS_BIND)
        self.sphere_buffer = self._create_uniform(packed.spheres, wd.OBJECT_SET_INDEX, wd.SPHERES_BIND)
        self.plane_buffer = self._create_uniform(packed.planes, wd.OBJECT_SET_INDEX, wd.PLANES_BIND)
        self.triangle_buffer = self._create_uniform(packed.triangles, wd.OBJECT_SET_INDEX, wd.TRIANGLES_BIND)
        self.object_id_buffer = self._create_uniform(packed.object_ids, wd.BVH_SET_INDEX, wd.OBJECT_ID_BIND)
        self.bvh_buffer = self._create_uniform(packed.bvh, wd.BVH_SET_INDEX, wd.BVH_BIND)
        self.triangle_vertex_buffer = self._create_uniform(bytes(12), wd.TRIANGLE_SET_INDEX, wd.TRIANGLE_VERTEX_BIND)
        self.triangle_uv_buffer = self._create_uniform(bytes(12), wd.TRIANGLE_SET_INDEX, wd.TRIANGLE_UV_BIND)
        self.triangle_index_buffer = self._create_uniform(bytes(12), wd.TRIANGLE_SET_INDEX, wd.TRIANGLE_INDEX_BIND)
        self.create_texture_buffers(packed.sampled_textures)
        self.transform_buffer = self._create_uniform(packed.transforms, wd.TRANSFORM_SET_INDEX, wd.TRANSFORM_BIND)
        self.bind_sets()

    # -- :470-497
    def create_compute_list(self, window, push_bytes: bytes):
        compute_list = self.rd.compute_list_begin()
        self.rd.compute_list_bind_compute_pipeline(compute_list, self.pipeline)
        for index in self.SET_ORDER:
            self.rd.compute_list_bind_uniform_set(compute_list, self.set_rids[index], index)
        self.rd.compute_list_set_push_constant(compute_list, push_bytes, len(push_bytes))
        self.rd.compute_list_dispatch(compute_list, window.work_group_width, window.work_group_height, window.work_group_depth)
        self.rd.compute_list_end()
        if self.is_local_renderer:
            self.rd.submit()
            self.rd.sync()

    # -- :511-567, one case of it: a grown buffer is a new buffer behind a new uniform set
    def replace_buffer(self, attr: str, data: bytes, set_index: int, binding: int):
        old = getattr(self, attr)
        self.rids_to_free.remove(old)
        self.rd.free_rid(old)
        setattr(self, attr, self._create_uniform(data, set_index, binding))
        self.bind_set(set_index)

    # -- :448-454
    def free_rids(self):
        for rid in self.rids_to_free:
            self.rd.free_rid(rid)
        self.rids_to_free = []
