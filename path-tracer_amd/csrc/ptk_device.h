// ptk_device.h -- structures shared by the host side of libptk and its kernels.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ptk.h"

// Widest node (children per inner node) the kernels unroll for.  The reference
// states that 8 is the only supported BVH order (pt_scene.gd:38-41).
#define PTK_MAX_WIDTH 8

// Built-in procedural textures (shaders/procedural_textures/checker_board.comp,
// checker_board_spatial.comp), selected by texture_path in ptk_configure.
#define PTK_PROC_CHECKER_BOARD 1
#define PTK_PROC_CHECKER_BOARD_SPATIAL 2
#define PTK_MAX_PROC 16

// Push constant block, ray_tracer.comp:416-434 (128 bytes).
struct PtkPush {
    float cam_pos[3]; float focal_length;
    float cam_right[3]; float viewport_width;
    float cam_up[3]; float viewport_height;
    float cam_forward[3]; float gamma;
    int32_t flags, render_mode, filler1, filler2, filler3;
    int32_t image_offset_x, image_offset_y;
    int32_t node_count_threshold, object_count_threshold;
    int32_t width, height, max_default_depth, max_refraction_bounces;
    float time, frame, max_samples;
};
static_assert(sizeof(PtkPush) == 128, "push constant block is 128 bytes");

struct PtkTexDesc {
    const float* rgba;  // level 0, RGBA32F; nullptr = filler grid
    int width, height;
};

struct PtkParams {
    const float4* nodes;       // 2 float4 per 32-byte node
    const int* object_ids;
    const float4* spheres;     // 2 float4 per 32-byte record
    const float4* planes;      // 2 float4 per 32-byte record
    const float4* triangles;   // 5 float4 per 80-byte record
    const float4* materials;   // 4 float4 per 64-byte record
    const float4* transforms;  // 4 float4 per 64-byte matrix
    const int* live_planes;    // indices of plane records with a non-zero normal
    int n_live_planes;
    int n_spheres, n_planes, n_triangles;  // buffer lengths (records), brute-force mode only
    float4* image;
    int img_w, img_h;
    const PtkTexDesc* textures;
    int n_textures;
    int n_proc;
    int proc_builtin[PTK_MAX_PROC];
    int refraction_mode, use_emission;
    int lds_cap;          // stack entries per lane held in LDS
    int* spill;           // stack entries beyond lds_cap: [entry][global lane]
    int* spill_t;
    size_t spill_stride;  // lanes in the launch
    int* trace_primary;
    int* trace_counters;
    PtkPush push;
};

extern "C" hipError_t ptk_launch_render(const PtkParams* p, int gx, int gy, int trace, int early_t,
                                        size_t lds_bytes, hipStream_t stream);
