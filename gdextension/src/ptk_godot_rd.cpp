// ptk_godot_rd.cpp -- see ptk_godot_rd.h.
#include "ptk_godot_rd.h"

#include <godot_cpp/core/error_macros.hpp>
#include <godot_cpp/variant/utility_functions.hpp>

#include <map>

namespace godot {

namespace {
// width and height of the images this class created, for texture_get_data (RenderingDevice knows its textures' formats)
std::map<uint64_t, std::pair<int, int>>& image_sizes() {
    static std::map<uint64_t, std::pair<int, int>> m;
    return m;
}
}  // namespace

PtkGodotRenderingDevice::PtkGodotRenderingDevice() { open(0); }

RID PtkGodotRenderingDevice::to_rid(uint64_t id) { return UtilityFunctions::rid_from_int64((int64_t)id); }

void PtkGodotRenderingDevice::report(const char* what) const {
    ERR_PRINT(String("PT: ") + what + ": " + core.last_error().c_str());
}

int PtkGodotRenderingDevice::open(int device) {
    if (!core.open(device)) {
        report("ptk_create");
        return PTK_E_HIP;
    }
    return PTK_OK;
}

String PtkGodotRenderingDevice::last_error() const { return String(core.last_error().c_str()); }

Ref<RDShaderSPIRV> PtkGodotRenderingDevice::shader_compile_spirv_from_source(const Ref<RDShaderSource>& shader_source, bool) {
    // Nothing is compiled.  The text travels on in the SPIR-V object's error-free compute slot: the source's own bytes
    // stand where the bytecode would be, and shader_create_from_spirv reads them back.
    Ref<RDShaderSPIRV> spirv;
    spirv.instantiate();
    ERR_FAIL_COND_V(shader_source.is_null(), spirv);
    const CharString text = shader_source->get_stage_source(RenderingDevice::SHADER_STAGE_COMPUTE).utf8();
    int32_t info[21];
    if (ptk_shader_source_info(core.context(), text.get_data(), (size_t)text.length(), info) != PTK_OK) {
        spirv->set_stage_compile_error(RenderingDevice::SHADER_STAGE_COMPUTE, String(ptk_last_error(core.context())));
        return spirv;
    }
    PackedByteArray bytes;
    bytes.resize(text.length());
    memcpy(bytes.ptrw(), text.get_data(), (size_t)text.length());
    spirv->set_stage_bytecode(RenderingDevice::SHADER_STAGE_COMPUTE, bytes);
    return spirv;
}

RID PtkGodotRenderingDevice::shader_create_from_spirv(const Ref<RDShaderSPIRV>& spirv_data, const String&) {
    ERR_FAIL_COND_V(spirv_data.is_null(), RID());
    const String err = spirv_data->get_stage_compile_error(RenderingDevice::SHADER_STAGE_COMPUTE);
    ERR_FAIL_COND_V_MSG(!err.is_empty(), RID(), String("PT: shader source refused: ") + err);
    const PackedByteArray text = spirv_data->get_stage_bytecode(RenderingDevice::SHADER_STAGE_COMPUTE);
    const uint64_t id = core.shader_create_from_source((const char*)text.ptr(), (size_t)text.size());
    if (!id) {
        report("shader_create_from_spirv");
        return RID();
    }
    return to_rid(id);
}

RID PtkGodotRenderingDevice::compute_pipeline_create(const RID& shader, const TypedArray<Ref<RefCounted>>&) {
    const uint64_t id = core.compute_pipeline_create(from_rid(shader));
    if (!id) report("compute_pipeline_create");
    return id ? to_rid(id) : RID();
}

RID PtkGodotRenderingDevice::storage_buffer_create(int64_t size_bytes, const PackedByteArray& data, int64_t) {
    ERR_FAIL_COND_V(size_bytes <= 0 || (data.size() != 0 && data.size() != size_bytes), RID());
    PackedByteArray zeros;
    if (data.size() == 0) zeros.resize(size_bytes), zeros.fill(0);
    const uint64_t id = core.storage_buffer_create((size_t)size_bytes, data.size() ? data.ptr() : zeros.ptr());
    if (!id) report("storage_buffer_create");
    return id ? to_rid(id) : RID();
}

RID PtkGodotRenderingDevice::texture_create(const Ref<RDTextureFormat>& format, const Ref<RDTextureView>&, const TypedArray<PackedByteArray>& data) {
    ERR_FAIL_COND_V(format.is_null(), RID());
    PackedByteArray level0;
    if (data.size() > 0) level0 = data[0];
    const int w = (int)format->get_width(), h = (int)format->get_height();
    const uint64_t id = core.texture_create((int)format->get_format(), w, h, (int)format->get_usage_bits(),
                                            level0.size() ? level0.ptr() : nullptr, (size_t)level0.size());
    if (!id) {
        report("texture_create");
        return RID();
    }
    if (format->get_usage_bits() & RenderingDevice::TEXTURE_USAGE_STORAGE_BIT) image_sizes()[id] = {w, h};
    return to_rid(id);
}

RID PtkGodotRenderingDevice::sampler_create(const Ref<RDSamplerState>& state) {
    // the kernels implement Godot's default sampler (nearest, clamp to edge), the one the plugin creates
    ERR_FAIL_COND_V(state.is_null(), RID());
    ERR_FAIL_COND_V_MSG(state->get_mag_filter() != RenderingDevice::SAMPLER_FILTER_NEAREST || state->get_min_filter() != RenderingDevice::SAMPLER_FILTER_NEAREST ||
                            state->get_repeat_u() != RenderingDevice::SAMPLER_REPEAT_MODE_CLAMP_TO_EDGE ||
                            state->get_repeat_v() != RenderingDevice::SAMPLER_REPEAT_MODE_CLAMP_TO_EDGE,
                        RID(), "PT: only the default sampler state is implemented");
    return to_rid(core.sampler_create());
}

RID PtkGodotRenderingDevice::uniform_set_create(const TypedArray<Ref<RDUniform>>& uniforms, const RID&, int64_t shader_set) {
    std::vector<ptk_rd::Uniform> list;
    for (int64_t i = 0; i < uniforms.size(); i++) {
        const Ref<RDUniform> u = uniforms[i];
        if (u.is_null()) continue;
        ptk_rd::Uniform c;
        c.type = (int)u->get_uniform_type();
        c.binding = (int)u->get_binding();
        const TypedArray<RID> ids = u->get_ids();
        for (int64_t k = 0; k < ids.size(); k++) c.ids.push_back(from_rid(ids[k]));
        list.push_back(std::move(c));
    }
    return to_rid(core.uniform_set_create(std::move(list), (int)shader_set));
}

int PtkGodotRenderingDevice::buffer_update(const RID& buffer, int64_t offset, int64_t size_bytes, const PackedByteArray& data) {
    ERR_FAIL_COND_V(size_bytes < 0 || size_bytes > data.size(), ERR_INVALID_PARAMETER);
    if (core.buffer_update(from_rid(buffer), (size_t)offset, (size_t)size_bytes, data.ptr()) != PTK_OK) {
        report("buffer_update");
        return ERR_INVALID_PARAMETER;
    }
    return OK;
}

PackedByteArray PtkGodotRenderingDevice::buffer_get_data(const RID& buffer, int64_t offset_bytes, int64_t size_bytes) {
    PackedByteArray out;
    out.resize(size_bytes);
    if (core.buffer_get_data(from_rid(buffer), (size_t)offset_bytes, (size_t)size_bytes, out.ptrw()) != PTK_OK) {
        report("buffer_get_data");
        out.clear();
    }
    return out;
}

PackedByteArray PtkGodotRenderingDevice::texture_get_data(const RID& texture, int64_t) {
    PackedByteArray out;
    auto it = image_sizes().find(from_rid(texture));
    ERR_FAIL_COND_V_MSG(it == image_sizes().end(), out, "PT: texture_get_data serves the storage image");
    out.resize((int64_t)it->second.first * it->second.second * 16);
    if (core.texture_get_data(from_rid(texture), out.ptrw(), (size_t)out.size()) != PTK_OK) {
        report("texture_get_data");
        out.clear();
    }
    return out;
}

PackedByteArray PtkGodotRenderingDevice::texture_get_display(const RID& texture, int transfer) {
    PackedByteArray out;
    auto it = image_sizes().find(from_rid(texture));
    ERR_FAIL_COND_V_MSG(it == image_sizes().end(), out, "PT: texture_get_display serves the storage image");
    out.resize((int64_t)it->second.first * it->second.second * 4);
    if (core.texture_get_display(from_rid(texture), out.ptrw(), (size_t)out.size(), transfer) != PTK_OK) {
        report("texture_get_display");
        out.clear();
    }
    return out;
}

void PtkGodotRenderingDevice::free_rid(const RID& rid) {
    image_sizes().erase(from_rid(rid));
    if (core.free_rid(from_rid(rid)) != PTK_OK) report("free_rid");
}

int64_t PtkGodotRenderingDevice::compute_list_begin() {
    if (core.compute_list_begin() != PTK_OK) report("compute_list_begin");
    return 1;
}
void PtkGodotRenderingDevice::compute_list_bind_compute_pipeline(int64_t, const RID& compute_pipeline) {
    if (core.compute_list_bind_compute_pipeline(from_rid(compute_pipeline)) != PTK_OK) report("compute_list_bind_compute_pipeline");
}
void PtkGodotRenderingDevice::compute_list_bind_uniform_set(int64_t, const RID& uniform_set, int64_t set_index) {
    if (core.compute_list_bind_uniform_set(from_rid(uniform_set), (int)set_index) != PTK_OK) report("compute_list_bind_uniform_set");
}
void PtkGodotRenderingDevice::compute_list_set_push_constant(int64_t, const PackedByteArray& buffer, int64_t size_bytes) {
    ERR_FAIL_COND(size_bytes > buffer.size());
    if (core.compute_list_set_push_constant(buffer.ptr(), (size_t)size_bytes) != PTK_OK) report("compute_list_set_push_constant");
}
void PtkGodotRenderingDevice::compute_list_dispatch(int64_t, int64_t x_groups, int64_t y_groups, int64_t z_groups) {
    if (core.compute_list_dispatch((int)x_groups, (int)y_groups, (int)z_groups) != PTK_OK) report("compute_list_dispatch");
}
void PtkGodotRenderingDevice::compute_list_end() { core.compute_list_end(); }
void PtkGodotRenderingDevice::submit() { core.submit(); }
void PtkGodotRenderingDevice::sync() {
    if (core.sync() != PTK_OK) report("sync");
}

void PtkGodotRenderingDevice::_bind_methods() {
    ClassDB::bind_method(D_METHOD("open", "device"), &PtkGodotRenderingDevice::open);
    ClassDB::bind_method(D_METHOD("last_error"), &PtkGodotRenderingDevice::last_error);
    ClassDB::bind_method(D_METHOD("shader_compile_spirv_from_source", "shader_source", "allow_cache"), &PtkGodotRenderingDevice::shader_compile_spirv_from_source, DEFVAL(true));
    ClassDB::bind_method(D_METHOD("shader_create_from_spirv", "spirv_data", "name"), &PtkGodotRenderingDevice::shader_create_from_spirv, DEFVAL(String()));
    ClassDB::bind_method(D_METHOD("compute_pipeline_create", "shader", "specialization_constants"), &PtkGodotRenderingDevice::compute_pipeline_create,
                         DEFVAL(TypedArray<Ref<RefCounted>>()));
    ClassDB::bind_method(D_METHOD("storage_buffer_create", "size_bytes", "data", "usage"), &PtkGodotRenderingDevice::storage_buffer_create, DEFVAL(PackedByteArray()),
                         DEFVAL(0));
    ClassDB::bind_method(D_METHOD("texture_create", "format", "view", "data"), &PtkGodotRenderingDevice::texture_create, DEFVAL(TypedArray<PackedByteArray>()));
    ClassDB::bind_method(D_METHOD("sampler_create", "state"), &PtkGodotRenderingDevice::sampler_create);
    ClassDB::bind_method(D_METHOD("uniform_set_create", "uniforms", "shader", "shader_set"), &PtkGodotRenderingDevice::uniform_set_create);
    ClassDB::bind_method(D_METHOD("buffer_update", "buffer", "offset", "size_bytes", "data"), &PtkGodotRenderingDevice::buffer_update);
    ClassDB::bind_method(D_METHOD("buffer_get_data", "buffer", "offset_bytes", "size_bytes"), &PtkGodotRenderingDevice::buffer_get_data, DEFVAL(0), DEFVAL(0));
    ClassDB::bind_method(D_METHOD("texture_get_data", "texture", "layer"), &PtkGodotRenderingDevice::texture_get_data);
    ClassDB::bind_method(D_METHOD("texture_get_display", "texture", "transfer"), &PtkGodotRenderingDevice::texture_get_display);
    ClassDB::bind_method(D_METHOD("free_rid", "rid"), &PtkGodotRenderingDevice::free_rid);
    ClassDB::bind_method(D_METHOD("compute_list_begin"), &PtkGodotRenderingDevice::compute_list_begin);
    ClassDB::bind_method(D_METHOD("compute_list_bind_compute_pipeline", "compute_list", "compute_pipeline"), &PtkGodotRenderingDevice::compute_list_bind_compute_pipeline);
    ClassDB::bind_method(D_METHOD("compute_list_bind_uniform_set", "compute_list", "uniform_set", "set_index"), &PtkGodotRenderingDevice::compute_list_bind_uniform_set);
    ClassDB::bind_method(D_METHOD("compute_list_set_push_constant", "compute_list", "buffer", "size_bytes"), &PtkGodotRenderingDevice::compute_list_set_push_constant);
    ClassDB::bind_method(D_METHOD("compute_list_dispatch", "compute_list", "x_groups", "y_groups", "z_groups"), &PtkGodotRenderingDevice::compute_list_dispatch);
    ClassDB::bind_method(D_METHOD("compute_list_end"), &PtkGodotRenderingDevice::compute_list_end);
    ClassDB::bind_method(D_METHOD("submit"), &PtkGodotRenderingDevice::submit);
    ClassDB::bind_method(D_METHOD("sync"), &PtkGodotRenderingDevice::sync);
}

}  // namespace godot
