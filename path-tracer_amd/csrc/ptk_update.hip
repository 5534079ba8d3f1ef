// ptk_update.hip -- device side of batched buffer updates.
//
// The plugin edits a scene with many small rd.buffer_update calls (one per moved object, one per dirty BVH
// node: proprietary_dataclasses/pt_renderer.gd:664-697,771-788).  The library gathers them on the host and
// brings them to the GPU in one piece: a staging block that holds a table of {destination, offset, size}
// entries followed by the payloads, copied with one transfer and scattered by this kernel, stream-ordered
// before the next dispatch.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ptk_device.h"
#include "ptk_math.h"

// one workgroup per entry; destinations of one batch never overlap (the host merges ranges)
__global__ __launch_bounds__(256) void ptk_scatter_kernel(const uint8_t* __restrict__ stage, uint32_t n_entries) {
    const uint32_t e = blockIdx.x;
    if (e >= n_entries) return;
    const PtkScatterEntry ent = ((const PtkScatterEntry*)stage)[e];
    uint8_t* dst = (uint8_t*)ent.dst;
    const uint8_t* src = stage + ent.src_offset;
    const uint32_t n = ent.nbytes;
    if ((((uintptr_t)dst | (uintptr_t)src | n) & 15u) == 0) {
        for (uint32_t i = threadIdx.x; i < n / 16; i += 256) ((uint4*)dst)[i] = ((const uint4*)src)[i];
    } else if ((((uintptr_t)dst | (uintptr_t)src | n) & 3u) == 0) {
        for (uint32_t i = threadIdx.x; i < n / 4; i += 256) ((uint32_t*)dst)[i] = ((const uint32_t*)src)[i];
    } else {
        for (uint32_t i = threadIdx.x; i < n; i += 256) dst[i] = src[i];
    }
}

extern "C" hipError_t ptk_launch_scatter(const void* d_stage, uint32_t n_entries, hipStream_t stream) {
    if (n_entries == 0) return hipSuccess;
    hipLaunchKernelGGL(ptk_scatter_kernel, dim3(n_entries), dim3(256), 0, stream, (const uint8_t*)d_stage, n_entries);
    return hipGetLastError();
}

// Leaf records: what a leaf tests for object-id slot i, by value, so that the traversal reads a leaf's primitives
// from consecutive addresses without going through the id table first.  Triangle, from the 80-byte record
// (pt_triangle.gd:125-150): {A.xyz, id} {E1.xyz, N.x} {E2.xyz, N.y} with E1 = B - A, E2 = C - A, N = E1 x E2 -- the
// first steps of the shader's triangle test (ray_tracer.comp:926-969), which do not depend on the ray and are made here
// once, by the same operations (hit_triangle_pre in ptk_render_core.h takes them up; N.z fits no word and is made
// there).  Sphere: {centre.xyz, id} {radius, 0, 0, 0} {0} from the 32-byte record; anything else (and an index outside
// its buffer, which validation has refused by then) zeros.
__global__ __launch_bounds__(256) void ptk_leaf_records_kernel(const int32_t* __restrict__ ids, int n_ids, const float4* __restrict__ triangles,
                                                               int n_tri, const float4* __restrict__ spheres, int n_sph, float4* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_ids) return;
    const int oid = ids[i];
    const int type = oid >> 24, oi = (int)((uint32_t)oid << 8) >> 8;  // sign-extended fields, as the shader's bitfieldExtract
    float4 r0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), r1 = r0, r2 = r0;
    if (type == 3 && oi >= 0 && oi < n_tri) {
        const float4 a = triangles[(size_t)oi * 5], b = triangles[(size_t)oi * 5 + 1], c = triangles[(size_t)oi * 5 + 2];
        const float e1x = b.x - a.x, e1y = b.y - a.y, e1z = b.z - a.z;
        const float e2x = c.x - a.x, e2y = c.y - a.y, e2z = c.z - a.z;
        r0 = make_float4(a.x, a.y, a.z, __int_as_float(oid));
        r1 = make_float4(e1x, e1y, e1z, e1y * e2z - e2y * e1z);
        r2 = make_float4(e2x, e2y, e2z, e1z * e2x - e2z * e1x);
    } else if (type == 1 && oi >= 0 && oi < n_sph) {
        const float4 s = spheres[(size_t)oi * 2];
        r0 = make_float4(s.x, s.y, s.z, __int_as_float(oid));
        r1 = make_float4(s.w, 0.0f, 0.0f, 0.0f);
    }
    out[(size_t)i * 3] = r0; out[(size_t)i * 3 + 1] = r1; out[(size_t)i * 3 + 2] = r2;
}

extern "C" hipError_t ptk_launch_leaf_records(const int32_t* ids, int n_ids, const void* triangles80, int n_tri, const void* spheres32, int n_sph,
                                              void* out48, hipStream_t stream) {
    if (n_ids <= 0) return hipSuccess;
    hipLaunchKernelGGL(ptk_leaf_records_kernel, dim3((n_ids + 255) / 256), dim3(256), 0, stream, ids, n_ids, (const float4*)triangles80, n_tri,
                       (const float4*)spheres32, n_sph, (float4*)out48);
    return hipGetLastError();
}

// The inverse of every mesh matrix, by the function the render kernels would otherwise run at each instanced hit
// (inverse44 in ptk_math.h: the same operations in the same order, so the same bits).
__global__ __launch_bounds__(64) void ptk_inverse_transforms_kernel(const float4* __restrict__ xf, int n, float4* __restrict__ out) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    m44 m;
    m.c0 = xf[(size_t)i * 4]; m.c1 = xf[(size_t)i * 4 + 1]; m.c2 = xf[(size_t)i * 4 + 2]; m.c3 = xf[(size_t)i * 4 + 3];
    const m44 r = inverse44(m);
    out[(size_t)i * 4] = r.c0; out[(size_t)i * 4 + 1] = r.c1; out[(size_t)i * 4 + 2] = r.c2; out[(size_t)i * 4 + 3] = r.c3;
}

extern "C" hipError_t ptk_launch_inverse_transforms(const void* xf64, int n, void* out64, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(ptk_inverse_transforms_kernel, dim3((n + 63) / 64), dim3(64), 0, stream, (const float4*)xf64, n, (float4*)out64);
    return hipGetLastError();
}
