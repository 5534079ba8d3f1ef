// ptk_rendering_device.cpp -- see ptk_rendering_device.h.
#include "ptk_rendering_device.h"

#include <godot_cpp/core/error_macros.hpp>
#include <godot_cpp/variant/utility_functions.hpp>

namespace godot {

PtkRenderingDevice::PtkRenderingDevice() { open(0); }

PtkRenderingDevice::~PtkRenderingDevice() {
    if (ctx) ptk_destroy(ctx);
}

int PtkRenderingDevice::open(int device) {
    if (ctx) ptk_destroy(ctx);
    ctx = ptk_create(device);
    if (!ctx) {
        ERR_PRINT(String("PT: ptk_create failed: ") + ptk_last_error(nullptr));
        return PTK_E_HIP;
    }
    return PTK_OK;
}

int PtkRenderingDevice::check(int rc, const char* what) const {
    if (rc != PTK_OK) ERR_PRINT(String("PT: ") + what + " failed: " + ptk_last_error(ctx));
    return rc;
}

String PtkRenderingDevice::last_error() const { return String(ptk_last_error(ctx)); }

int PtkRenderingDevice::configure(int bvh_order, int refraction_mode, bool use_emission, bool early_t_break,
                                  const PackedStringArray& procedural_texture_paths) {
    ERR_FAIL_NULL_V(ctx, PTK_E_STATE);
    std::vector<CharString> keep;
    std::vector<const char*> ptrs;
    for (int64_t i = 0; i < procedural_texture_paths.size(); i++) {
        keep.push_back(procedural_texture_paths[i].utf8());
        ptrs.push_back(keep.back().get_data());
    }
    return check(ptk_configure(ctx, bvh_order, refraction_mode, use_emission, early_t_break, ptrs.data(), (int)ptrs.size()), "configure");
}

int64_t PtkRenderingDevice::storage_buffer_create(int set_index, int binding, const PackedByteArray& data) {
    ERR_FAIL_NULL_V(ctx, 0);
    uint64_t h = ptk_buffer_create(ctx, set_index, binding, data.ptr(), (size_t)data.size());
    if (!h) check(PTK_E_INVALID, "storage_buffer_create");
    return (int64_t)h;
}

int PtkRenderingDevice::buffer_update(int64_t buffer, int64_t offset, int64_t size_bytes, const PackedByteArray& data) {
    ERR_FAIL_NULL_V(ctx, PTK_E_STATE);
    ERR_FAIL_COND_V(size_bytes < 0 || size_bytes > data.size(), PTK_E_INVALID);
    return check(ptk_buffer_update(ctx, (uint64_t)buffer, (size_t)offset, data.ptr(), (size_t)size_bytes), "buffer_update");
}

PackedByteArray PtkRenderingDevice::buffer_get_data(int64_t buffer, int64_t offset, int64_t size_bytes) {
    PackedByteArray out;
    ERR_FAIL_NULL_V(ctx, out);
    out.resize(size_bytes);
    if (check(ptk_buffer_read(ctx, (uint64_t)buffer, (size_t)offset, out.ptrw(), (size_t)size_bytes), "buffer_get_data") != PTK_OK) out.clear();
    return out;
}

int64_t PtkRenderingDevice::image_create(int width, int height) {
    ERR_FAIL_NULL_V(ctx, 0);
    uint64_t h = ptk_image_create(ctx, width, height);
    if (!h) check(PTK_E_INVALID, "image_create");
    return (int64_t)h;
}

int64_t PtkRenderingDevice::texture_create(int slot, int width, int height, const PackedByteArray& rgbaf) {
    ERR_FAIL_NULL_V(ctx, 0);
    ERR_FAIL_COND_V(rgbaf.size() != 0 && rgbaf.size() != (int64_t)width * height * 16, 0);
    uint64_t h = ptk_texture_create(ctx, slot, width, height, rgbaf.size() ? (const float*)rgbaf.ptr() : nullptr);
    if (!h) check(PTK_E_INVALID, "texture_create");
    return (int64_t)h;
}

int PtkRenderingDevice::free_rid(int64_t rid) {
    ERR_FAIL_NULL_V(ctx, PTK_E_STATE);
    return check(ptk_free(ctx, (uint64_t)rid), "free_rid");
}

int PtkRenderingDevice::compute_list_dispatch(const PackedByteArray& push_constants, int groups_x, int groups_y, int groups_z) {
    ERR_FAIL_NULL_V(ctx, PTK_E_STATE);
    ERR_FAIL_COND_V(push_constants.size() != 128, PTK_E_INVALID);
    return check(ptk_dispatch(ctx, push_constants.ptr(), groups_x, groups_y, groups_z), "compute_list_dispatch");
}

int PtkRenderingDevice::compute_list_dispatch_noalloc(const PackedByteArray& push_constants, int groups_x, int groups_y, int groups_z) {
    ERR_FAIL_NULL_V(ctx, PTK_E_STATE);
    ERR_FAIL_COND_V(push_constants.size() != 128, PTK_E_INVALID);
    return check(ptk_dispatch_noalloc(ctx, push_constants.ptr(), groups_x, groups_y, groups_z), "compute_list_dispatch_noalloc");
}

int PtkRenderingDevice::sync() {
    ERR_FAIL_NULL_V(ctx, PTK_E_STATE);
    return check(ptk_sync(ctx), "sync");
}

PackedByteArray PtkRenderingDevice::texture_get_data(int64_t image, int width, int height) {
    PackedByteArray out;
    ERR_FAIL_NULL_V(ctx, out);
    out.resize((int64_t)width * height * 16);
    if (check(ptk_image_read(ctx, (uint64_t)image, out.ptrw(), (size_t)out.size()), "texture_get_data") != PTK_OK) out.clear();
    return out;
}

PackedByteArray PtkRenderingDevice::texture_get_display(int64_t image, int width, int height, int transfer) {
    PackedByteArray out;
    ERR_FAIL_NULL_V(ctx, out);
    out.resize((int64_t)width * height * 4);
    if (check(ptk_image_read_display(ctx, (uint64_t)image, out.ptrw(), (size_t)out.size(), transfer), "texture_get_display") != PTK_OK) out.clear();
    return out;
}

int PtkRenderingDevice::update_image_texture(int64_t image, int width, int height, const Ref<ImageTexture>& texture, int transfer) {
    ERR_FAIL_COND_V(texture.is_null(), PTK_E_INVALID);
    PackedByteArray bytes = texture_get_display(image, width, height, transfer);
    if (bytes.is_empty()) return PTK_E_HIP;
    Ref<Image> img = Image::create_from_data(width, height, false, Image::FORMAT_RGBA8, bytes);
    if (texture->get_width() == width && texture->get_height() == height) texture->update(img);
    else texture->set_image(img);
    return PTK_OK;
}

int64_t PtkRenderingDevice::bvh_build(int order, int type, const PackedFloat32Array& aabbs6, const PackedInt32Array& object_ids,
                                      const PackedInt64Array& mesh_bvhs, const PackedFloat32Array& mesh_transforms12,
                                      const PackedInt32Array& mesh_indices) {
    ERR_FAIL_NULL_V(ctx, 0);
    ERR_FAIL_COND_V(aabbs6.size() != object_ids.size() * 6, 0);
    ERR_FAIL_COND_V(mesh_transforms12.size() != mesh_bvhs.size() * 12 || mesh_indices.size() != mesh_bvhs.size(), 0);
    std::vector<uint64_t> handles;
    for (int64_t i = 0; i < mesh_bvhs.size(); i++) handles.push_back((uint64_t)mesh_bvhs[i]);
    uint64_t h = ptk_bvh_build(ctx, order, type, (int)object_ids.size(), aabbs6.ptr(), object_ids.ptr(), (int)handles.size(),
                               handles.data(), mesh_transforms12.ptr(), mesh_indices.ptr());
    if (!h) check(PTK_E_INVALID, "bvh_build");
    return (int64_t)h;
}

PackedByteArray PtkRenderingDevice::bvh_get_nodes(int64_t bvh) {
    PackedByteArray out;
    ERR_FAIL_NULL_V(ctx, out);
    int32_t n_nodes = 0, n_ids = 0;
    if (check(ptk_bvh_sizes(ctx, (uint64_t)bvh, &n_nodes, &n_ids), "bvh_get_nodes") != PTK_OK) return out;
    out.resize((int64_t)n_nodes * 32);
    if (check(ptk_bvh_read(ctx, (uint64_t)bvh, out.ptrw(), nullptr), "bvh_get_nodes") != PTK_OK) out.clear();
    return out;
}

PackedInt32Array PtkRenderingDevice::bvh_get_object_ids(int64_t bvh) {
    PackedInt32Array out;
    ERR_FAIL_NULL_V(ctx, out);
    int32_t n_nodes = 0, n_ids = 0;
    if (check(ptk_bvh_sizes(ctx, (uint64_t)bvh, &n_nodes, &n_ids), "bvh_get_object_ids") != PTK_OK) return out;
    out.resize(n_ids);
    if (n_ids > 0 && check(ptk_bvh_read(ctx, (uint64_t)bvh, nullptr, out.ptrw()), "bvh_get_object_ids") != PTK_OK) out.clear();
    return out;
}

PackedInt64Array PtkRenderingDevice::bvh_bind(int64_t bvh) {
    PackedInt64Array out;
    ERR_FAIL_NULL_V(ctx, out);
    uint64_t nodes = 0, ids = 0;
    if (check(ptk_bvh_bind(ctx, (uint64_t)bvh, &nodes, &ids), "bvh_bind") != PTK_OK) return out;
    out.push_back((int64_t)nodes);
    out.push_back((int64_t)ids);
    return out;
}

int PtkRenderingDevice::bvh_refit(const PackedInt32Array& object_ids, const PackedFloat32Array& values12) {
    ERR_FAIL_NULL_V(ctx, -PTK_E_STATE);
    ERR_FAIL_COND_V(values12.size() != object_ids.size() * 12, -PTK_E_INVALID);
    std::vector<ptk_refit_event> events((size_t)object_ids.size());
    for (int64_t i = 0; i < object_ids.size(); i++) {
        events[(size_t)i].object_id = object_ids[i];
        for (int k = 0; k < 12; k++) events[(size_t)i].v[k] = values12[i * 12 + k];
    }
    int32_t updated = 0;
    const int rc = check(ptk_bvh_refit(ctx, events.data(), (int)events.size(), &updated), "bvh_refit");
    return rc == PTK_OK ? updated : -rc;
}

Dictionary PtkRenderingDevice::mesh_to_triangles(const Dictionary& surface, const Transform3D& transform, int material_index,
                                                 int64_t texture_id) {
    Dictionary out;
    ERR_FAIL_NULL_V(ctx, out);
    // keys of RenderingServer.mesh_get_surface (servers/rendering_server.cpp, Godot 4.3)
    const PackedByteArray vertex = surface.get("vertex_data", PackedByteArray());
    const PackedByteArray attribute = surface.get("attribute_data", PackedByteArray());
    const PackedByteArray index = surface.get("index_data", PackedByteArray());
    const AABB box = surface.get("aabb", AABB());
    const Vector4 uv_scale = surface.get("uv_scale", Vector4());
    ptk_mesh_surface s = {};
    s.format = (uint64_t)(int64_t)surface.get("format", 0);
    s.vertex_count = (int32_t)(int64_t)surface.get("vertex_count", 0);
    s.index_count = (int32_t)(int64_t)surface.get("index_count", 0);
    s.vertex_data = vertex.ptr(); s.vertex_bytes = (size_t)vertex.size();
    s.attribute_data = attribute.size() ? attribute.ptr() : nullptr; s.attribute_bytes = (size_t)attribute.size();
    s.index_data = index.ptr(); s.index_bytes = (size_t)index.size();
    s.aabb[0] = box.position.x; s.aabb[1] = box.position.y; s.aabb[2] = box.position.z;
    s.aabb[3] = box.size.x; s.aabb[4] = box.size.y; s.aabb[5] = box.size.z;
    s.uv_scale[0] = uv_scale.x; s.uv_scale[1] = uv_scale.y; s.uv_scale[2] = uv_scale.z; s.uv_scale[3] = uv_scale.w;
    const Basis& b = transform.basis;
    const float t12[12] = {b.rows[0].x, b.rows[0].y, b.rows[0].z, b.rows[1].x, b.rows[1].y, b.rows[1].z,
                           b.rows[2].x, b.rows[2].y, b.rows[2].z, transform.origin.x, transform.origin.y, transform.origin.z};
    const int64_t n = s.index_count / 3;
    PackedByteArray records;
    PackedFloat32Array aabbs, uvs;
    records.resize(n * 80);
    aabbs.resize(n * 6);
    uvs.resize(n * 6);
    int32_t n_triangles = 0;
    if (check(ptk_mesh_to_triangles(ctx, &s, t12, material_index, (uint32_t)texture_id, records.ptrw(), aabbs.ptrw(), uvs.ptrw(), &n_triangles),
              "mesh_to_triangles") != PTK_OK)
        return out;
    out["records"] = records;
    out["aabbs"] = aabbs;
    out["uvs"] = uvs;
    out["triangles"] = n_triangles;
    return out;
}

int PtkRenderingDevice::set_option(int option, int value) {
    ERR_FAIL_NULL_V(ctx, PTK_E_STATE);
    return check(ptk_set_option(ctx, option, value), "set_option");
}

PackedByteArray PtkRenderingDevice::comm_unique_id() {
    PackedByteArray id;
    id.resize(PTK_COMM_ID_BYTES);
    if (ptk_comm_unique_id(id.ptrw()) != PTK_OK) {
        ERR_PRINT(String("PT: comm_unique_id failed: ") + ptk_last_error(nullptr));
        id.clear();
    }
    return id;
}

int PtkRenderingDevice::comm_init(int world, int rank, const PackedByteArray& unique_id) {
    ERR_FAIL_NULL_V(ctx, PTK_E_STATE);
    ERR_FAIL_COND_V(unique_id.size() != PTK_COMM_ID_BYTES, PTK_E_INVALID);
    return check(ptk_comm_init(ctx, world, rank, unique_id.ptr()), "comm_init");
}

int PtkRenderingDevice::image_reduce_mean(int64_t image, int root) {
    ERR_FAIL_NULL_V(ctx, PTK_E_STATE);
    return check(ptk_image_reduce_mean(ctx, (uint64_t)image, root), "image_reduce_mean");
}

int PtkRenderingDevice::image_gather_bands(int64_t image, const PackedInt32Array& first_row, const PackedInt32Array& n_rows, int root) {
    ERR_FAIL_NULL_V(ctx, PTK_E_STATE);
    ERR_FAIL_COND_V(first_row.size() != n_rows.size(), PTK_E_INVALID);
    return check(ptk_image_gather_bands(ctx, (uint64_t)image, first_row.ptr(), n_rows.ptr(), root), "image_gather_bands");
}

int PtkRenderingDevice::comm_destroy() {
    ERR_FAIL_NULL_V(ctx, PTK_E_STATE);
    return check(ptk_comm_destroy(ctx), "comm_destroy");
}

int PtkRenderingDevice::set_renderer(int renderer) {
    ERR_FAIL_NULL_V(ctx, PTK_E_STATE);
    return check(ptk_set_renderer(ctx, renderer), "set_renderer");
}

float PtkRenderingDevice::last_dispatch_ms() {
    float ms = -1.0f;
    if (ctx) check(ptk_last_dispatch_ms(ctx, &ms), "last_dispatch_ms");
    return ms;
}

void PtkRenderingDevice::_bind_methods() {
    ClassDB::bind_method(D_METHOD("open", "device"), &PtkRenderingDevice::open);
    ClassDB::bind_method(D_METHOD("is_open"), &PtkRenderingDevice::is_open);
    ClassDB::bind_method(D_METHOD("last_error"), &PtkRenderingDevice::last_error);
    ClassDB::bind_method(D_METHOD("configure", "bvh_order", "refraction_mode", "use_emission", "early_t_break", "procedural_texture_paths"),
                         &PtkRenderingDevice::configure);
    ClassDB::bind_method(D_METHOD("storage_buffer_create", "set_index", "binding", "data"), &PtkRenderingDevice::storage_buffer_create);
    ClassDB::bind_method(D_METHOD("buffer_update", "buffer", "offset", "size_bytes", "data"), &PtkRenderingDevice::buffer_update);
    ClassDB::bind_method(D_METHOD("buffer_get_data", "buffer", "offset", "size_bytes"), &PtkRenderingDevice::buffer_get_data);
    ClassDB::bind_method(D_METHOD("image_create", "width", "height"), &PtkRenderingDevice::image_create);
    ClassDB::bind_method(D_METHOD("texture_create", "slot", "width", "height", "rgbaf"), &PtkRenderingDevice::texture_create);
    ClassDB::bind_method(D_METHOD("free_rid", "rid"), &PtkRenderingDevice::free_rid);
    ClassDB::bind_method(D_METHOD("compute_list_dispatch", "push_constants", "groups_x", "groups_y", "groups_z"),
                         &PtkRenderingDevice::compute_list_dispatch);
    ClassDB::bind_method(D_METHOD("compute_list_dispatch_noalloc", "push_constants", "groups_x", "groups_y", "groups_z"),
                         &PtkRenderingDevice::compute_list_dispatch_noalloc);
    ClassDB::bind_method(D_METHOD("sync"), &PtkRenderingDevice::sync);
    ClassDB::bind_method(D_METHOD("texture_get_data", "image", "width", "height"), &PtkRenderingDevice::texture_get_data);
    ClassDB::bind_method(D_METHOD("texture_get_display", "image", "width", "height", "transfer"), &PtkRenderingDevice::texture_get_display);
    ClassDB::bind_method(D_METHOD("update_image_texture", "image", "width", "height", "texture", "transfer"),
                         &PtkRenderingDevice::update_image_texture);
    ClassDB::bind_method(D_METHOD("bvh_build", "order", "type", "aabbs6", "object_ids", "mesh_bvhs", "mesh_transforms12", "mesh_indices"),
                         &PtkRenderingDevice::bvh_build);
    ClassDB::bind_method(D_METHOD("bvh_get_nodes", "bvh"), &PtkRenderingDevice::bvh_get_nodes);
    ClassDB::bind_method(D_METHOD("bvh_get_object_ids", "bvh"), &PtkRenderingDevice::bvh_get_object_ids);
    ClassDB::bind_method(D_METHOD("bvh_bind", "bvh"), &PtkRenderingDevice::bvh_bind);
    ClassDB::bind_method(D_METHOD("bvh_refit", "object_ids", "values12"), &PtkRenderingDevice::bvh_refit);
    ClassDB::bind_method(D_METHOD("mesh_to_triangles", "surface", "transform", "material_index", "texture_id"),
                         &PtkRenderingDevice::mesh_to_triangles);
    ClassDB::bind_method(D_METHOD("set_option", "option", "value"), &PtkRenderingDevice::set_option);
    ClassDB::bind_static_method("PtkRenderingDevice", D_METHOD("comm_unique_id"), &PtkRenderingDevice::comm_unique_id);
    ClassDB::bind_method(D_METHOD("comm_init", "world", "rank", "unique_id"), &PtkRenderingDevice::comm_init);
    ClassDB::bind_method(D_METHOD("image_reduce_mean", "image", "root"), &PtkRenderingDevice::image_reduce_mean);
    ClassDB::bind_method(D_METHOD("image_gather_bands", "image", "first_row", "n_rows", "root"), &PtkRenderingDevice::image_gather_bands);
    ClassDB::bind_method(D_METHOD("comm_destroy"), &PtkRenderingDevice::comm_destroy);
    ClassDB::bind_method(D_METHOD("set_renderer", "renderer"), &PtkRenderingDevice::set_renderer);
    ClassDB::bind_method(D_METHOD("last_dispatch_ms"), &PtkRenderingDevice::last_dispatch_ms);
    BIND_CONSTANT(PTK_DISPLAY_STORED);
    BIND_CONSTANT(PTK_DISPLAY_CANVAS);
    BIND_CONSTANT(PTK_DISPLAY_CANVAS_SRGB);
    BIND_CONSTANT(PTK_REFIT_MESH);
    BIND_CONSTANT(PTK_OPT_TRIANGLE_UVS);
    BIND_CONSTANT(PTK_OPT_LAUNCH_TIMING);
    BIND_CONSTANT(PTK_BVH_SAH);
    BIND_CONSTANT(PTK_RENDERER_AUTO);
    BIND_CONSTANT(PTK_RENDERER_SINGLE_KERNEL);
    BIND_CONSTANT(PTK_RENDERER_WAVEFRONT);
    BIND_CONSTANT(PTK_RENDERER_PERSISTENT);
}

}  // namespace godot
