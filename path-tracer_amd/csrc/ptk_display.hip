// ptk_display.hip -- the display side of the render image for gfx950.
//
// The plugin shows its rgba32f image through a full-screen quad whose fragment shader samples the image
// and applies `color.rgb = pow(color.rgb, vec3(2.2))` (shaders/canvas.gdshader:22-36); screenshots take the
// raw floats (pt_renderer.gd:320-350).  A host that presents the HIP image has to bring it into a Godot
// texture first; this kernel produces the 8-bit RGBA a Godot ImageTexture (FORMAT_RGBA8) holds, so that the
// copy over PCIe is 4 bytes per pixel instead of 16.
//
// transfer 0: clamp to [0,1] and quantise (the stored, gamma-space values: what the canvas shader would sample);
// transfer 1: canvas.gdshader's pow(rgb, 2.2) first (the linear ALBEDO it outputs), then quantise;
// transfer 2: as 1, then the sRGB encoding Godot's viewport applies to linear colours on their way to the screen.
// Alpha is clamped and quantised as it is.  Quantisation is round-half-up of v * 255.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ptk_device.h"
#include "ptk_math.h"

namespace {

PTK_DEV float clamp01(float v) { return (v != v) ? 0.0f : pmin(pmax(v, 0.0f), 1.0f); }
PTK_DEV uint32_t quant8(float v) { return (uint32_t)pf2i(clamp01(v) * 255.0f + 0.5f); }
PTK_DEV float srgb_encode(float c) {  // linear -> sRGB, the piecewise curve of IEC 61966-2-1
    return c <= 0.0031308f ? c * 12.92f : 1.055f * ppow(c, 1.0f / 2.4f) - 0.055f;
}
PTK_DEV float transfer_channel(float v, int transfer) {
    v = clamp01(v);
    if (transfer >= 1) v = (v > 0.0f) ? ppow(v, 2.2f) : 0.0f;  // canvas.gdshader:33
    if (transfer >= 2) v = srgb_encode(v);
    return v;
}

}  // namespace

// one lane = 4 consecutive pixels: 64 bytes in, one 16-byte store out
__global__ __launch_bounds__(256) void ptk_display_kernel(const float4* __restrict__ image, uint4* __restrict__ out, size_t n_quads,
                                                          size_t n_pixels, uint32_t* __restrict__ out_words, int transfer) {
    const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= n_quads) return;
    uint32_t w[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const size_t p = q * 4 + k;
        w[k] = 0;
        if (p < n_pixels) {
            const float4 c = image[p];
            w[k] = quant8(transfer_channel(c.x, transfer)) | (quant8(transfer_channel(c.y, transfer)) << 8) |
                   (quant8(transfer_channel(c.z, transfer)) << 16) | (quant8(c.w) << 24);
        }
    }
    if (q * 4 + 3 < n_pixels) out[q] = make_uint4(w[0], w[1], w[2], w[3]);
    else
        for (int k = 0; k < 4; k++)
            if (q * 4 + k < n_pixels) out_words[q * 4 + k] = w[k];
}

extern "C" hipError_t ptk_launch_display(const float4* image, void* out_rgba8, size_t n_pixels, int transfer, hipStream_t stream) {
    const size_t n_quads = (n_pixels + 3) / 4;
    hipLaunchKernelGGL(ptk_display_kernel, dim3((unsigned)((n_quads + 255) / 256)), dim3(256), 0, stream, image, (uint4*)out_rgba8, n_quads,
                       n_pixels, (uint32_t*)out_rgba8, transfer);
    return hipGetLastError();
}

// image *= s (the 1/world of a reduced sum of per-GPU means)
__global__ void ptk_scale_kernel(float4* p, size_t n, float s) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 v = p[i];
    v.x *= s; v.y *= s; v.z *= s; v.w *= s;
    p[i] = v;
}
extern "C" hipError_t ptk_launch_scale(float4* image, size_t n_pixels, float s, hipStream_t stream) {
    hipLaunchKernelGGL(ptk_scale_kernel, dim3((unsigned)((n_pixels + 255) / 256)), dim3(256), 0, stream, image, n_pixels, s);
    return hipGetLastError();
}
