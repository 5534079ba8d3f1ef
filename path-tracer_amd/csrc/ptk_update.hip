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
