// ptk_mesh.hip -- mesh ingestion on the GPU: a Godot ArrayMesh surface, as stored, to the plugin's 80-byte
// triangle records.
//
// The plugin imports a mesh with PTObjectContainer.mesh_to_pttriangles
// (proprietary_dataclasses/pt_object_container.gd:163-186): Mesh.surface_get_arrays(0), then one PTTriangle
// per index triplet, whose to_byte_array (public/primitives/pt_triangle.gd:125-150) and get_bvh_aabb
// (pt_triangle.gd:107-112, pt_object.gd:269-272) feed the triangle buffer and the BVH builder -- one GDScript
// object per triangle.  Here one lane does one triangle, straight from the surface's byte blocks:
//   * index triplet (16-bit indices up to 65536 vertices, 32-bit above),
//   * positions: float32 x,y,z (12-byte stride), or with ARRAY_FLAG_COMPRESS_ATTRIBUTES four unorm16 per
//     vertex (8-byte stride), position = aabb.position + (u16 / 65535) * aabb.size, the way
//     RenderingServer.mesh_surface_get_arrays decodes them,
//   * optional UVs from the attribute block (after an optional 4-byte colour; float32 pairs, or unorm16 pairs
//     under compression, re-centred and scaled by uv_scale when that is non-zero),
//   * the record {p1,0, p2,0, p3,0, normal,0, material, texture id, 0, 0} with normal =
//     (p3 - p1) x (p2 - p1) normalised (zero if degenerate), and the triangle's BVH box.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ptk_device.h"

namespace {

struct V3 {
    float x, y, z;
};
__device__ __forceinline__ V3 v3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ V3 sub(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }

__device__ __forceinline__ V3 load_position(const PtkMeshJob& J, uint32_t i) {
    if (J.compressed) {
        const uint16_t* u = (const uint16_t*)(J.vertex_data + (size_t)i * 8);
        // Vector3(float(u) / 65535.0): the division is in double, the result rounded to float32
        const float fx = (float)((double)(float)u[0] / 65535.0), fy = (float)((double)(float)u[1] / 65535.0),
                    fz = (float)((double)(float)u[2] / 65535.0);
        return v3(fx * J.aabb[3] + J.aabb[0], fy * J.aabb[4] + J.aabb[1], fz * J.aabb[5] + J.aabb[2]);
    }
    const float* p = (const float*)(J.vertex_data + (size_t)i * 12);
    return v3(p[0], p[1], p[2]);
}

__device__ __forceinline__ void load_uv(const PtkMeshJob& J, uint32_t i, float& u, float& v) {
    const uint8_t* a = J.attribute_data + (size_t)i * J.attribute_stride + J.uv_offset;
    if (J.compressed) {
        const uint16_t* q = (const uint16_t*)a;
        u = (float)((double)(float)q[0] / 65535.0);
        v = (float)((double)(float)q[1] / 65535.0);
        if (J.uv_rescale) { u = (u - 0.5f) * J.uv_scale[0]; v = (v - 0.5f) * J.uv_scale[1]; }
    } else {
        const float* q = (const float*)a;
        u = q[0]; v = q[1];
    }
}

// Transform3D * Vector3: basis rows dotted with v (left to right), plus origin
__device__ __forceinline__ V3 xform_point(const float* t, V3 v) {
    return v3(((t[0] * v.x + t[1] * v.y) + t[2] * v.z) + t[9], ((t[3] * v.x + t[4] * v.y) + t[5] * v.z) + t[10],
              ((t[6] * v.x + t[7] * v.y) + t[8] * v.z) + t[11]);
}

}  // namespace

__global__ __launch_bounds__(256) void ptk_mesh_kernel(const PtkMeshJob J) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= J.n_triangles) return;
    uint32_t idx[3];
    for (int k = 0; k < 3; k++)
        idx[k] = J.index32 ? ((const uint32_t*)J.index_data)[(size_t)t * 3 + k] : (uint32_t)((const uint16_t*)J.index_data)[(size_t)t * 3 + k];
    if (idx[0] >= J.vertex_count || idx[1] >= J.vertex_count || idx[2] >= J.vertex_count) {
        atomicAdd(J.error_count, 1u);
        return;
    }
    const V3 v1 = load_position(J, idx[0]), v2 = load_position(J, idx[1]), v3_ = load_position(J, idx[2]);

    // PTTriangle.to_byte_array: the points in the meshlet's transform, the normal from the transformed points
    V3 p1 = v1, p2 = v2, p3 = v3_;
    if (J.has_transform) { p1 = xform_point(J.transform, v1); p2 = xform_point(J.transform, v2); p3 = xform_point(J.transform, v3_); }
    const V3 a = sub(p3, p1), b = sub(p2, p1);
    V3 n = v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
    const float lsq = (n.x * n.x + n.y * n.y) + n.z * n.z;
    if (lsq == 0.0f) n = v3(0.0f, 0.0f, 0.0f);
    else { const float len = __builtin_sqrtf(lsq); n = v3(n.x / len, n.y / len, n.z / len); }
    float4* rec = (float4*)(J.records + (size_t)t * 80);
    rec[0] = make_float4(p1.x, p1.y, p1.z, 0.0f);
    rec[1] = make_float4(p2.x, p2.y, p2.z, 0.0f);
    rec[2] = make_float4(p3.x, p3.y, p3.z, 0.0f);
    rec[3] = make_float4(n.x, n.y, n.z, 0.0f);
    rec[4] = make_float4(__int_as_float(J.material_index), __uint_as_float(J.texture_id), 0.0f, 0.0f);

    if (J.aabbs) {
        // PTTriangle.set_aabb: AABB(v1, v2 - v1).abs(), expanded to v1, v2, v3, .abs(); then transform * aabb
        float pos[3], size[3];
        const float s0[3] = {v2.x - v1.x, v2.y - v1.y, v2.z - v1.z};
        const float o[3] = {v1.x, v1.y, v1.z};
        for (int i = 0; i < 3; i++) { pos[i] = o[i] + (s0[i] < 0.0f ? s0[i] : 0.0f); size[i] = fabsf(s0[i]); }
        const V3 pts[3] = {v1, v2, v3_};
        for (int k = 0; k < 3; k++) {
            const float p[3] = {pts[k].x, pts[k].y, pts[k].z};
            for (int i = 0; i < 3; i++) {
                float begin = pos[i], end = pos[i] + size[i];
                if (p[i] < begin) begin = p[i];
                if (p[i] > end) end = p[i];
                pos[i] = begin;
                size[i] = end - begin;
            }
        }
        for (int i = 0; i < 3; i++) { pos[i] = pos[i] + (size[i] < 0.0f ? size[i] : 0.0f); size[i] = fabsf(size[i]); }
        float* out = J.aabbs + (size_t)t * 6;
        // Transform3D::xform(AABB) (the identity when the caller gave no transform)
        for (int i = 0; i < 3; i++) {
            float lo = J.has_transform ? J.transform[9 + i] : 0.0f, hi = lo;
            for (int j = 0; j < 3; j++) {
                const float m = J.has_transform ? J.transform[i * 3 + j] : (i == j ? 1.0f : 0.0f);
                const float e = m * pos[j], f = m * (pos[j] + size[j]);
                if (e < f) { lo += e; hi += f; }
                else { lo += f; hi += e; }
            }
            out[i] = lo;
            out[3 + i] = hi - lo;
        }
    }
    if (J.uvs) {
        float* out = J.uvs + (size_t)t * 6;
        for (int k = 0; k < 3; k++) {
            float u = 0.0f, v = 0.0f;
            if (J.has_uv) load_uv(J, idx[k], u, v);
            out[2 * k] = u;
            out[2 * k + 1] = v;
        }
    }
}

extern "C" hipError_t ptk_launch_mesh(const PtkMeshJob* job, hipStream_t stream) {
    if (job->n_triangles == 0) return hipSuccess;
    hipLaunchKernelGGL(ptk_mesh_kernel, dim3((job->n_triangles + 255u) / 256u), dim3(256), 0, stream, *job);
    return hipGetLastError();
}
