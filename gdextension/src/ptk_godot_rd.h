// ptk_godot_rd.h -- a GDExtension class with RenderingDevice's own method names and signatures on top of libptk.
//
// PTWorkDispatcher keeps a RenderingDevice in `rd` (proprietary_dataclasses/pt_work_dispatcher.gd:51, 146-160) and
// calls storage_buffer_create, texture_create, sampler_create, uniform_set_create, shader_compile_spirv_from_source,
// shader_create_from_spirv, compute_pipeline_create, compute_list_*, submit, sync and free_rid on it (:297-497,
// 642-653); PTRenderer reaches through wd.rd for buffer_update, texture_get_data and free_rid (pt_renderer.gd:327,
// 658-788).  PtkGodotRenderingDevice answers those calls with the same argument lists, taking Godot's own RDUniform,
// RDShaderSource, RDShaderSPIRV, RDTextureFormat, RDTextureView and RDSamplerState objects and handing out RIDs, so
// the plugin's GDScript stays as it is except for the line that declares `rd` (typed RenderingDevice, which an
// extension class cannot be) and the two lines that create it.
//
// All decisions live in ptk_rd_core.h (plain C++, compiled and tested without Godot); this file converts types.
// Like the rest of gdextension/ it needs godot-cpp 4.3 to build, which the repository's build image lacks.
#pragma once

#include <godot_cpp/classes/rd_sampler_state.hpp>
#include <godot_cpp/classes/rd_shader_source.hpp>
#include <godot_cpp/classes/rd_shader_spirv.hpp>
#include <godot_cpp/classes/rd_texture_format.hpp>
#include <godot_cpp/classes/rd_texture_view.hpp>
#include <godot_cpp/classes/rd_uniform.hpp>
#include <godot_cpp/classes/ref_counted.hpp>
#include <godot_cpp/classes/rendering_device.hpp>
#include <godot_cpp/core/class_db.hpp>
#include <godot_cpp/variant/packed_byte_array.hpp>
#include <godot_cpp/variant/rid.hpp>
#include <godot_cpp/variant/typed_array.hpp>

#include "ptk_rd_core.h"

namespace godot {

class PtkGodotRenderingDevice : public RefCounted {
    GDCLASS(PtkGodotRenderingDevice, RefCounted)

    ptk_rd::Core core;
    static RID to_rid(uint64_t id);
    static uint64_t from_rid(const RID& rid) { return (uint64_t)rid.get_id(); }
    void report(const char* what) const;  // ERR_PRINT of the core's last error, as RenderingDevice prints its own

protected:
    static void _bind_methods();

public:
    PtkGodotRenderingDevice();

    int open(int device);  // another GPU than 0 (one object per GPU)
    String last_error() const;

    Ref<RDShaderSPIRV> shader_compile_spirv_from_source(const Ref<RDShaderSource>& shader_source, bool allow_cache = true);
    RID shader_create_from_spirv(const Ref<RDShaderSPIRV>& spirv_data, const String& name = String());
    RID compute_pipeline_create(const RID& shader, const TypedArray<Ref<RefCounted>>& specialization_constants = TypedArray<Ref<RefCounted>>());

    RID storage_buffer_create(int64_t size_bytes, const PackedByteArray& data = PackedByteArray(), int64_t usage = 0);
    RID texture_create(const Ref<RDTextureFormat>& format, const Ref<RDTextureView>& view, const TypedArray<PackedByteArray>& data = TypedArray<PackedByteArray>());
    RID sampler_create(const Ref<RDSamplerState>& state);
    RID uniform_set_create(const TypedArray<Ref<RDUniform>>& uniforms, const RID& shader, int64_t shader_set);

    int buffer_update(const RID& buffer, int64_t offset, int64_t size_bytes, const PackedByteArray& data);
    PackedByteArray buffer_get_data(const RID& buffer, int64_t offset_bytes = 0, int64_t size_bytes = 0);
    PackedByteArray texture_get_data(const RID& texture, int64_t layer);
    void free_rid(const RID& rid);

    int64_t compute_list_begin();
    void compute_list_bind_compute_pipeline(int64_t compute_list, const RID& compute_pipeline);
    void compute_list_bind_uniform_set(int64_t compute_list, const RID& uniform_set, int64_t set_index);
    void compute_list_set_push_constant(int64_t compute_list, const PackedByteArray& buffer, int64_t size_bytes);
    void compute_list_dispatch(int64_t compute_list, int64_t x_groups, int64_t y_groups, int64_t z_groups);
    void compute_list_end();
    void submit();
    void sync();

    // not RenderingDevice's: the display bytes for an ImageTexture (see PtkRenderingDevice::texture_get_display)
    PackedByteArray texture_get_display(const RID& texture, int transfer);
};

}  // namespace godot
