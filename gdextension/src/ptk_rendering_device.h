// ptk_rendering_device.h -- Godot 4.3 GDExtension class that stands where PTWorkDispatcher.rd stood.
//
// PTWorkDispatcher talks to a RenderingDevice (proprietary_dataclasses/pt_work_dispatcher.gd:146-160,
// 436-497, 642-653) and PTRenderer reaches through wd.rd for buffer_update / texture_get_data / free_rid
// (pt_renderer.gd:327, 658-697, 761-788).  This class offers those verbs on top of libptk's C ABI
// (include/ptk.h), one method per call, so the GDScript side changes only where it creates `rd` and where
// it creates uniforms (the (set, binding) pair now travels with storage_buffer_create).
//
// Built with godot-cpp (SConstruct next to this directory); links -lptk.  Not compiled in the build image
// of this repository, which has no godot-cpp: the C ABI underneath is what the test-suite exercises, from
// Python (path-tracer_amd/rendering_device.py, the same methods) and from plain C (tests/c_abi).
#pragma once

#include <godot_cpp/classes/image.hpp>
#include <godot_cpp/classes/image_texture.hpp>
#include <godot_cpp/classes/ref_counted.hpp>
#include <godot_cpp/core/class_db.hpp>
#include <godot_cpp/variant/dictionary.hpp>
#include <godot_cpp/variant/packed_byte_array.hpp>
#include <godot_cpp/variant/packed_float32_array.hpp>
#include <godot_cpp/variant/packed_int32_array.hpp>
#include <godot_cpp/variant/packed_int64_array.hpp>
#include <godot_cpp/variant/packed_string_array.hpp>
#include <godot_cpp/variant/string.hpp>
#include <godot_cpp/variant/transform3d.hpp>

#include <vector>

#include "ptk.h"

namespace godot {

class PtkRenderingDevice : public RefCounted {
    GDCLASS(PtkRenderingDevice, RefCounted)

    ptk_ctx* ctx = nullptr;
    int check(int rc, const char* what) const;  // prints ptk_last_error through ERR_PRINT when rc != PTK_OK

protected:
    static void _bind_methods();

public:
    PtkRenderingDevice();
    ~PtkRenderingDevice() override;

    // create_local_rendering_device() is the constructor; `device` picks the GPU (one context per GPU)
    int open(int device);
    bool is_open() const { return ctx != nullptr; }
    String last_error() const;

    // load_shader: PTUtils.load_shader's text edits become arguments (pt_utils.gd:140-195)
    int configure(int bvh_order, int refraction_mode, bool use_emission, bool early_t_break, const PackedStringArray& procedural_texture_paths);

    // _create_uniform (pt_work_dispatcher.gd:642-653) and friends
    int64_t storage_buffer_create(int set_index, int binding, const PackedByteArray& data);
    int buffer_update(int64_t buffer, int64_t offset, int64_t size_bytes, const PackedByteArray& data);
    PackedByteArray buffer_get_data(int64_t buffer, int64_t offset, int64_t size_bytes);
    int64_t image_create(int width, int height);
    int64_t texture_create(int slot, int width, int height, const PackedByteArray& rgbaf);  // empty array: filler grid
    int free_rid(int64_t rid);

    // create_compute_list (pt_work_dispatcher.gd:470-497)
    int compute_list_dispatch(const PackedByteArray& push_constants, int groups_x, int groups_y, int groups_z);
    int compute_list_dispatch_noalloc(const PackedByteArray& push_constants, int groups_x, int groups_y, int groups_z);
    int sync();

    // rd.texture_get_data(image, 0) (pt_renderer.gd:327): width*height*16 bytes of RGBAF
    PackedByteArray texture_get_data(int64_t image, int width, int height);
    // the display path (pt_renderer.gd:519-540 + shaders/canvas.gdshader): RGBA8 of the image, converted on the GPU
    PackedByteArray texture_get_display(int64_t image, int width, int height, int transfer);
    // convenience for the canvas: refreshes an ImageTexture the canvas material samples instead of the Texture2DRD
    int update_image_texture(int64_t image, int width, int height, const Ref<ImageTexture>& texture, int transfer);

    // PTBVHTree.create_bvh + index_tree + to_byte_array on the GPU (pt_bvh_tree.gd:164-189,488-524)
    int64_t bvh_build(int order, int type, const PackedFloat32Array& aabbs6, const PackedInt32Array& object_ids,
                      const PackedInt64Array& mesh_bvhs, const PackedFloat32Array& mesh_transforms12, const PackedInt32Array& mesh_indices);
    PackedByteArray bvh_get_nodes(int64_t bvh);
    PackedInt32Array bvh_get_object_ids(int64_t bvh);
    // create_object_id_buffer + create_bvh_buffer for a tree built here (pt_work_dispatcher.gd:184-185,723-758):
    // returns [nodes buffer, object-id buffer], bound at (3,0) and (3,1) without a trip through host memory
    PackedInt64Array bvh_bind(int64_t bvh);
    // PTBVHTree.update_aabb chains + the dirty-node uploads of pt_renderer.gd:771-788 for the bound tree.
    // object_ids[i] with values12[12*i .. 12*i+5] = the object's new box (position, size), or
    // mesh index | PTK_REFIT_MESH << 24 with the mesh's new Transform3D (basis rows, origin).  Returns the number of
    // nodes rewritten, or a negative error code.
    int bvh_refit(const PackedInt32Array& object_ids, const PackedFloat32Array& values12);

    // PTObjectContainer.mesh_to_pttriangles on the GPU (pt_object_container.gd:163-186): `surface` is the dictionary
    // of RenderingServer.mesh_get_surface(mesh.get_rid(), 0).  Returns {"records": PackedByteArray (80 B per triangle),
    // "aabbs": PackedFloat32Array (6 per triangle), "uvs": PackedFloat32Array (6 per triangle)}.
    Dictionary mesh_to_triangles(const Dictionary& surface, const Transform3D& transform, int material_index, int64_t texture_id);
    int set_option(int option, int value);

    // several GPUs of one node, one PtkRenderingDevice per process: RCCL through libptk
    static PackedByteArray comm_unique_id();  // made on rank 0, 128 bytes, handed to every rank by the host
    int comm_init(int world, int rank, const PackedByteArray& unique_id);
    int image_reduce_mean(int64_t image, int root);
    int image_gather_bands(int64_t image, const PackedInt32Array& first_row, const PackedInt32Array& n_rows, int root);
    int comm_destroy();

    int set_renderer(int renderer);
    float last_dispatch_ms();
};

}  // namespace godot
