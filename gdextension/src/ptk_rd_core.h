// ptk_rd_core.h -- the bookkeeping between Godot's RenderingDevice calls and libptk's C ABI, free of Godot types.
//
// PTWorkDispatcher speaks RenderingDevice (proprietary_dataclasses/pt_work_dispatcher.gd:297-497, 642-653):
// resources are created without a place, uniforms collect them, uniform sets are bound into a compute list, the
// list is dispatched.  libptk wants to know which buffer sits at which (set, binding).  This class keeps what
// lies between -- uniform sets, the open compute list, ids of things that hold nothing on the device -- in plain
// C++, so that it compiles and is tested without godot-cpp (tests/c_abi/rd_core_harness.cpp); the GDExtension
// class in ptk_godot_rd.cpp only converts Godot's argument types and calls it.  The ctypes twin is
// path-tracer_amd/godot_rd.py.
#pragma once

#include <stddef.h>
#include <stdint.h>

#include <map>
#include <string>
#include <utility>
#include <vector>

#include "ptk.h"

namespace ptk_rd {

// RenderingDevice enum values the plugin uses
enum { UNIFORM_TYPE_SAMPLER_WITH_TEXTURE = 1, UNIFORM_TYPE_IMAGE = 3, UNIFORM_TYPE_STORAGE_BUFFER = 8 };
enum { DATA_FORMAT_R8G8B8A8_UNORM = 36, DATA_FORMAT_R32G32B32A32_SFLOAT = 109 };
enum { TEXTURE_USAGE_STORAGE_BIT = 8 };

struct Uniform {
    int type = 0, binding = 0;
    std::vector<uint64_t> ids;  // RDUniform.get_ids(), as the ids this class handed out
};

class Core {
    // ids of shaders, pipelines, samplers and uniform sets carry this bit: nothing of them lives in libptk
    static constexpr uint64_t kLocal = 1ull << 62;
    ptk_ctx* ctx = nullptr;
    uint64_t next_local = 1;
    std::map<uint64_t, char> local_kind;  // 's'hader, 'p'ipeline, sa'm'pler, 'u'niform set
    std::map<uint64_t, std::pair<int, std::vector<Uniform>>> uniform_sets;
    std::map<int, uint64_t> bound_set;    // set index -> uniform set whose resources are placed there now
    bool list_open = false, pipeline_bound = false, have_push = false;
    unsigned char push[128];
    std::string error;

    uint64_t make_local(char kind) {
        const uint64_t id = kLocal | next_local++;
        local_kind[id] = kind;
        return id;
    }
    int fail(const std::string& what) {
        error = what;
        return PTK_E_INVALID;
    }
    int pass(int rc) {
        if (rc != PTK_OK) error = ptk_last_error(ctx);
        return rc;
    }

public:
    Core() = default;
    Core(const Core&) = delete;
    Core& operator=(const Core&) = delete;
    ~Core() { close(); }

    // RenderingServer.create_local_rendering_device() / get_rendering_device(), pt_work_dispatcher.gd:146-160
    bool open(int device) {
        close();
        ctx = ptk_create(device);
        if (!ctx) error = ptk_last_error(nullptr);
        return ctx != nullptr;
    }
    void close() {
        if (ctx) ptk_destroy(ctx);
        ctx = nullptr;
        uniform_sets.clear();
        bound_set.clear();
        local_kind.clear();
    }
    ptk_ctx* context() const { return ctx; }
    const std::string& last_error() const { return error; }

    // shader_compile_spirv_from_source + shader_create_from_spirv, :440-441: 0 and last_error() on failure
    uint64_t shader_create_from_source(const char* glsl, size_t nbytes) {
        if (pass(ptk_configure_from_source(ctx, glsl, nbytes)) != PTK_OK) return 0;
        return make_local('s');
    }
    // compute_pipeline_create, :445
    uint64_t compute_pipeline_create(uint64_t shader) {
        auto it = local_kind.find(shader);
        if (it == local_kind.end() || it->second != 's') { fail("compute_pipeline_create: not a shader"); return 0; }
        return make_local('p');
    }
    // storage_buffer_create(size, bytes), :643
    uint64_t storage_buffer_create(size_t size_bytes, const void* data) {
        const uint64_t h = ptk_buffer_create(ctx, PTK_SET_NONE, 0, data, size_bytes);
        if (!h) error = ptk_last_error(ctx);
        return h;
    }
    // texture_create(format, view, [level 0]), :336, :367, :395.  Storage usage makes the rgba32f image; sampling
    // usage a texture of the sampler array (R8G8B8A8_UNORM bytes become value / 255, what sampling UNORM returns).
    uint64_t texture_create(int format, int width, int height, int usage_bits, const void* level0, size_t nbytes) {
        if (usage_bits & TEXTURE_USAGE_STORAGE_BIT) {
            if (format != DATA_FORMAT_R32G32B32A32_SFLOAT) { fail("storage images are rgba32f"); return 0; }
            const uint64_t h = ptk_image_create(ctx, width, height);
            if (!h) { error = ptk_last_error(ctx); return 0; }
            if (level0 && nbytes && pass(ptk_image_update(ctx, h, level0, nbytes)) != PTK_OK) { (void)ptk_free(ctx, h); return 0; }
            return h;
        }
        const size_t n = (size_t)width * (size_t)height * 4;
        uint64_t h = 0;
        if (format == DATA_FORMAT_R32G32B32A32_SFLOAT && level0 && nbytes >= n * 4) {
            h = ptk_texture_create(ctx, -1, width, height, (const float*)level0);  // (mip levels behind level 0 are not sampled)
        } else if (format == DATA_FORMAT_R8G8B8A8_UNORM && level0 && nbytes >= n) {
            std::vector<float> rgba(n);
            const unsigned char* p = (const unsigned char*)level0;
            for (size_t i = 0; i < n; i++) rgba[i] = (float)p[i] / 255.0f;
            h = ptk_texture_create(ctx, -1, width, height, rgba.data());
        } else {
            fail("texture_create: a sampled texture needs level-0 data in R32G32B32A32_SFLOAT or R8G8B8A8_UNORM");
            return 0;
        }
        if (!h) error = ptk_last_error(ctx);
        return h;
    }
    // sampler_create(RDSamplerState.new()), :313: Godot's default sampler is the one the kernels implement
    uint64_t sampler_create() { return make_local('m'); }
    // uniform_set_create(uniforms, shader, set), :410
    uint64_t uniform_set_create(std::vector<Uniform> uniforms, int shader_set) {
        const uint64_t id = make_local('u');
        uniform_sets[id] = {shader_set, std::move(uniforms)};
        return id;
    }

    int buffer_update(uint64_t buffer, size_t offset, size_t size_bytes, const void* data) {
        return pass(ptk_buffer_update(ctx, buffer, offset, data, size_bytes));
    }
    int buffer_get_data(uint64_t buffer, size_t offset, size_t size_bytes, void* dst) {
        return pass(ptk_buffer_read(ctx, buffer, offset, dst, size_bytes));
    }
    int texture_get_data(uint64_t image, void* dst, size_t nbytes) { return pass(ptk_image_read(ctx, image, dst, nbytes)); }
    int texture_get_display(uint64_t image, void* dst, size_t nbytes, int transfer) {
        return pass(ptk_image_read_display(ctx, image, dst, nbytes, transfer));
    }
    // free_rid, :448-465
    int free_rid(uint64_t rid) {
        if (rid & kLocal) {
            uniform_sets.erase(rid);
            for (auto it = bound_set.begin(); it != bound_set.end();) it = it->second == rid ? bound_set.erase(it) : std::next(it);
            local_kind.erase(rid);
            return PTK_OK;
        }
        return pass(ptk_free(ctx, rid));
    }

    // compute_list_begin .. compute_list_end, :478-492
    int compute_list_begin() {
        if (list_open) return fail("compute_list_begin: a compute list is already open");
        list_open = true;
        pipeline_bound = have_push = false;
        return PTK_OK;
    }
    int compute_list_bind_compute_pipeline(uint64_t pipeline) {
        auto it = local_kind.find(pipeline);
        if (!list_open || it == local_kind.end() || it->second != 'p') return fail("compute_list_bind_compute_pipeline: no open list or not a pipeline");
        pipeline_bound = true;
        return PTK_OK;
    }
    int compute_list_bind_uniform_set(uint64_t uniform_set, int set_index) {
        auto it = uniform_sets.find(uniform_set);
        if (!list_open || it == uniform_sets.end()) return fail("compute_list_bind_uniform_set: no open list or not a uniform set");
        if (it->second.first != set_index) return fail("compute_list_bind_uniform_set: the set was created for another index");
        auto cur = bound_set.find(set_index);
        if (cur != bound_set.end() && cur->second == uniform_set) return PTK_OK;  // the plugin binds every set before every dispatch
        for (const Uniform& u : it->second.second) {
            if (u.type == UNIFORM_TYPE_STORAGE_BUFFER || u.type == UNIFORM_TYPE_IMAGE) {
                if (u.ids.empty()) return fail("uniform without a resource");
                const int rc = pass(ptk_buffer_bind(ctx, u.ids[0], set_index, u.binding));
                if (rc != PTK_OK) return rc;
            } else if (u.type == UNIFORM_TYPE_SAMPLER_WITH_TEXTURE) {
                // ids alternate sampler, texture: pair k is element k of the shader's sampler array
                for (size_t k = 0; 2 * k + 1 < u.ids.size(); k++) {
                    const int rc = pass(ptk_texture_bind(ctx, u.ids[2 * k + 1], (int)k));
                    if (rc != PTK_OK) return rc;
                }
            } else {
                return fail("uniform type " + std::to_string(u.type) + " is not one the plugin binds");
            }
        }
        bound_set[set_index] = uniform_set;
        return PTK_OK;
    }
    int compute_list_set_push_constant(const void* bytes, size_t size_bytes) {
        if (!list_open || size_bytes != 128 || !bytes) return fail("compute_list_set_push_constant: the shader's block is 128 bytes");
        for (size_t i = 0; i < 128; i++) push[i] = ((const unsigned char*)bytes)[i];
        have_push = true;
        return PTK_OK;
    }
    int compute_list_dispatch(int x, int y, int z) {
        if (!list_open || !pipeline_bound || !have_push) return fail("compute_list_dispatch needs an open list with a pipeline and push constants");
        return pass(ptk_dispatch(ctx, push, x, y, z));
    }
    void compute_list_end() { list_open = false; }
    void submit() {}  // dispatches are queued on the context's stream as they are recorded
    int sync() { return pass(ptk_sync(ctx)); }
};

}  // namespace ptk_rd
