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
// Digests (ptk_procedural_texture_digest) of those two files as the plugin ships them; tools/proc_texture_digests.py
// prints them from a checkout of the plugin.
#define PTK_DIGEST_CHECKER_BOARD 0x796fdc7af7a91a27ull
#define PTK_DIGEST_CHECKER_BOARD_SPATIAL 0x317ca6e7441a9819ull

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

// Wavefront renderer state: ray queues, hit records and per-pixel path state in HBM.
// A queue entry is two float4: {origin.xyz, bits(pixel slot)}, {direction.xyz, 0}, plus the path state
// in a parallel array that only the shading kernel touches; the hit record of
// entry i is hits[i] = {hit, object_id, transform_index, bits(t)} and hit_uv[i].
#define PTK_WF_MAX_ITER 64
#define PTK_WF_MAX_LAG 8       // launches a path may fall behind its bounce number by being carried over (see PtkWavefront::carry)
#ifndef PTK_PP_MAX_WG_WAVES
#define PTK_PP_MAX_WG_WAVES 12  // waves of a path-kernel workgroup (three per SIMD at its register count)
#endif
#define PTK_WF_WG_FLAG_BYTES 32 // LDS words the waves of a workgroup share besides the top blocks: queue dry, waves done, the claim word (wg_claim),
                                // the part of the queue the workgroup draws from
#define PTK_WF_HEADER_BYTES 4096 // counts, n_resume and the queue heads (one per part and launch) in front of the queues, zeroed per dispatch
#define PTK_WF_MAX_PARTS 8
#define PTK_WF_CARRY_PLANES 25 // words of search state carried besides the stack rows
#define PTK_WF_PREP_PLANES 10  // words of a started search (PtkWavefront::prep; wf_prepare in ptk_wavefront.hip)
// The top of many trees is a chain every ray walks in the same way: a root with one child, a mesh socket, a
// mesh root with one child (scene root -> PTMesh socket -> mesh tree is the plugin's usual layout).  Those
// steps need no per-lane node fetch: the host lists them here (they travel as kernel arguments) and every
// search runs them before its first fetched node.  A step is one box test (the only child of a one-child
// inner node) or the entry into a mesh socket.
#define PTK_MAX_CHAIN 6
struct PtkChainStep {
    float lo[3], hi[3];   // kind 0: the child's box (min - pad, max + pad as stored in the node buffer)
    int32_t kind;         // 0 box test, 1 mesh socket
    int32_t xf;           // kind 1: transform index
    float m[16];          // kind 1: that transform's matrix by value (the 64 bytes of the transform buffer's record),
                          // so that a search's start waits for no load of it
};
// Plane records travel as kernel arguments too when the scene has at most this many that can be hit (a plane whose
// normal is zero never is): every search tests them first (ray_tracer.comp:1204), and as arguments they cost its
// start neither the load of the list nor the loads of the records.
#define PTK_ARG_PLANES 4
struct PtkWavefront {
    float4* rays[2];
    int4* hits;
    float2* hit_uv;
    uint32_t* counts;  // counts[i] = rays the shading of launch i-1 queued for launch i (PTK_WF_MAX_ITER + 1 words)
    uint32_t* heads;   // heads[i * PTK_WF_MAX_PARTS + p] = entries of part p of launch i's queue handed to workgroups
                       // ((PTK_WF_MAX_ITER + 1) * PTK_WF_MAX_PARTS words; the path kernel: one word, its pixel counter)
    // Optional (default: one part).  The queue of a launch is cut into n_parts equal parts (1 or 8), each with a head of its own.  A workgroup draws
    // from part blockIdx.x % 8 -- workgroups are dealt round-robin over the chip's eight XCDs, so the workgroups that
    // share an L2 share a part, i.e. a stretch of the queue and with it (queues keep the order of the pixel slots they
    // stem from) a region of the image and of the scene -- and goes on to the next part when its own is used up.  Eight
    // heads also spread the device-wide atomics, which complete one at a time per address.
    uint32_t n_parts;
    // Carry-over: when the queue of launch i runs dry, a wave does not sit out the searches it still has in
    // flight but parks them -- ray, path state, search state, stack -- at the front of launch i+1's queue
    // (entries [0, n_resume[i+1])), where full waves pick them up again.  A launch then ends when its queue
    // is empty instead of when its slowest ray is done; a parked path is one launch behind its bounce number
    // (its own number travels in the ray's d.w), at most PTK_WF_MAX_LAG launches, which the host appends.
    uint32_t* n_resume;  // (PTK_WF_MAX_ITER + 2 words)
    uint32_t* carry[2];  // parked search state for the queue of that parity: [plane][carry_cap], PTK_WF_CARRY_PLANES planes
                         // then one per stack row of the LDS window; null = no carry-over
    uint32_t carry_cap;
    int32_t max_lag;     // launches a path may be behind its bounce number (<= PTK_WF_MAX_LAG)
    int32_t park_below;  // once the queue is dry, a wave parks its searches when at most this many are left (64: at once)
    int32_t leaf_split;  // a leaf of more objects than this is split between its lane and a helper lane (255: never)
    int32_t leaf_redo_all;  // test switch: every helped leaf is handed back to its owner, as if its helper had met the sphere corner
    uint32_t wg_claim;   // queue entries a traversal workgroup takes from the device-wide counter at a time (a multiple of 64)
    // A search's start -- planes, root box, the chain of one-child nodes and mesh sockets at the top of the tree -- is run
    // by the kernel that queues the ray (generate, shade), where all 64 lanes of a wave have one, instead of in the
    // traversal kernel's refill rounds, where a dozen lanes do: prep[parity] holds the started search of every queue
    // entry as planes [word][prep_cap] (PTK_WF_PREP_PLANES words, see wf_prepare).
    uint32_t* prep[2];
    uint32_t prep_cap;
    int32_t chain_xf;  // transform index of the mesh socket among the chain's steps (at most one: a mesh tree holds none), -1 if none
    float4* state[2];  // path state travelling with the queue entry, two float4 per entry:
                       // {throughput rgba}, {seed, current_IOR, bits(inside_of_count),
                       //  bits(refraction_bounces | inside-arrays-stored << 30)}
    int* inside;       // [10][n_slots]: in_mat[0..4], in_pri[0..4]
    uint32_t n_slots;  // groups_x * groups_y * 64
    int tiles_x;
    // Inner nodes compiled for the traversal kernel: one 256-byte block (16 float4) per inner node,
    // holding its eight child slots as planes: minx[8], miny[8], minz[8], maxx[8], maxy[8], maxz[8],
    // entry[8], 8 spare words.  An entry is what the traversal visits next: block index | child
    // count << 24 for an inner child, the leaf's packed node id (negative) for a leaf, or
    // 0x7F000000 | socket index for a mesh socket met at world level.
    const float4* blocks;
    const int2* sockets;     // per world-level mesh socket: {entry of the mesh's root, transform index}
    float root_min_x, root_min_y, root_min_z, root_max_x, root_max_y, root_max_z;  // box of the root node
    uint32_t root_entry;       // what a search visits first once the root box and the chain below are passed
    int32_t n_chain;
    PtkChainStep chain[PTK_MAX_CHAIN];
    uint32_t n_top;            // blocks [0, n_top) are the top of the tree in breadth-first order: the traversal
                               // kernel keeps a copy of them in LDS (272-byte stride) and reads them there
    uint32_t n_blocks, n_sockets, n_records;  // array lengths (checked builds)
    const float4* leaf_records;  // 3 float4 per object-id slot: the primitive a leaf tests, by value
};

struct PtkParams {
    const float4* nodes;       // 2 float4 per 32-byte node
    const int* object_ids;
    const float4* spheres;     // 2 float4 per 32-byte record
    const float4* planes;      // 2 float4 per 32-byte record
    const float4* triangles;   // 5 float4 per 80-byte record
    const float4* materials;   // 4 float4 per 64-byte record
    const float4* transforms;  // 4 float4 per 64-byte matrix
    const float4* inv_transforms;  // inverse44 of each of them (the shader takes inverse(mat4) at every instanced hit,
                                   // ray_tracer.comp:1021, 1065: the same arithmetic, done once per matrix when it changes)
    const float* triangle_uvs; // PTK_OPT_TRIANGLE_UVS: 6 floats per triangle record (corner UVs) from (4,1), else null
    int n_triangle_uvs;
    const int* live_planes;    // indices of plane records with a non-zero normal
    int n_live_planes;
    int n_arg_planes;          // >= 0: the live planes are arg_plane[0 .. n_arg_planes) below; -1: read the list
    int arg_plane_index[PTK_ARG_PLANES];
    float4 arg_plane[PTK_ARG_PLANES];  // {normal.xyz, distance}: the first 16 bytes of the plane's record
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
    // One-kernel renderer: tiles are not all alike (a tile of sky ends at its first search, a tile on a mesh runs
    // every bounce), and a launch lasts until its last tile is through.  A progressive render repeats the same frame,
    // so each dispatch records how long every tile took, and the next one starts the long tiles first.
    const uint32_t* tile_order;  // workgroup -> tile (row-major tile number); null = workgroup number itself
    uint32_t* tile_cost;         // tile -> ticks of a 100 MHz clock this dispatch spent on it; null = not recorded
    PtkWavefront wf;
    PtkPush push;
};

extern "C" hipError_t ptk_launch_render(const PtkParams* p, int gx, int gy, int trace, int early_t,
                                        size_t lds_bytes, hipStream_t stream);
// order[0 .. n) = the tiles sorted by the power of two of their recorded cost, costliest first (one workgroup).
extern "C" hipError_t ptk_launch_tile_order(const uint32_t* cost, uint32_t n, uint32_t* order, hipStream_t stream);
// Wavefront renderer: generate, then n_iter x (traverse, shade).  ev[2k], ev[2k+1] (may be null) are
// recorded around the k-th traversal launch.  n_wgs = persistent traversal workgroups (PTK_WF_WG_WAVES waves each),
// lds_bytes = their dynamic LDS: n_top * PTK_TOP_BLOCK_STRIDE + PTK_WF_WG_WAVES * (lds_cap + 1) * 256 + PTK_WF_WG_FLAG_BYTES.
extern "C" hipError_t ptk_launch_wavefront(const PtkParams* p, int gx, int gy, int n_iter, size_t lds_bytes,
                                           int n_wgs, hipEvent_t* ev, hipStream_t stream);
// Traversal workgroups are PTK_WF_WG_WAVES waves that share one LDS copy of the top blocks; *wgs_per_cu =
// resident workgroups per compute unit at that LDS size.
#define PTK_WF_WG_WAVES 16
#define PTK_LDS_BYTES_PER_CU 163840
#define PTK_TOP_BLOCK_STRIDE 272  /* bytes between top blocks in LDS: 256 + 16, so that lanes on different blocks spread over the banks */
extern "C" hipError_t ptk_wavefront_occupancy(size_t lds_bytes, int* wgs_per_cu);
// The path kernel runs in workgroups of wg_waves waves that share an LDS copy of the top node blocks (as the
// traversal kernel's do).  lds_per_wave: the wave's stack window; the launch takes n_wgs workgroups.
extern "C" hipError_t ptk_launch_persistent(const PtkParams* p, size_t lds_per_wave, int n_wgs, int wg_waves, hipStream_t stream);
// Picks the largest workgroup (12, 8, 4, 2 or 1 waves) that fits a compute unit with n_top top blocks in LDS.
extern "C" hipError_t ptk_persistent_occupancy(size_t lds_per_wave, uint32_t n_top, int* wg_waves, int* wgs_per_cu);
// Static device scratch of ptk_dispatch_noalloc (a file-scope __device__ array in ptk_persistent.hip):
// [0, 256) pixel counter, the rest traversal-stack spill.  _init resolves the current device's copy once.
#define PTK_NOALLOC_SCRATCH_BYTES ((size_t)8 << 20)
extern "C" hipError_t ptk_persistent_scratch_init(int device);
extern "C" void* ptk_persistent_scratch(int device);
// ptk_launch_persistent for a dispatch whose scratch is that array: waits (on `stream`) for the last launch on this GPU
// that used the array, launches, records its own end for the next one.  Allocates nothing.
extern "C" hipError_t ptk_launch_persistent_static(const PtkParams* p, size_t lds_per_wave, int n_wgs, int wg_waves, int device, hipStream_t stream);
// Display conversion (ptk_display.hip): rgba32f image -> RGBA8, n_pixels * 4 bytes at out_rgba8 (device, 16-byte aligned).
extern "C" hipError_t ptk_launch_display(const float4* image, void* out_rgba8, size_t n_pixels, int transfer, hipStream_t stream);
extern "C" hipError_t ptk_launch_scale(float4* image, size_t n_pixels, float s, hipStream_t stream);

// Mesh ingestion (ptk_mesh.hip): one Godot ArrayMesh surface to triangle records; all pointers are device pointers.
struct PtkMeshJob {
    const uint8_t* vertex_data;      // positions block: 12 bytes per vertex, or 8 (four unorm16) when compressed
    const uint8_t* attribute_data;   // per-vertex attributes: [colour 4 B][uv 8 B | 4 B compressed]...
    const uint8_t* index_data;
    uint32_t vertex_count, n_triangles;
    int32_t compressed, index32, has_uv, uv_rescale, has_transform;
    uint32_t attribute_stride, uv_offset;
    float aabb[6], uv_scale[4], transform[12];
    int32_t material_index;
    uint32_t texture_id;
    uint8_t* records;                // n_triangles * 80 bytes
    float* aabbs;                    // n_triangles * 6 (position, size) or null
    float* uvs;                      // n_triangles * 6 (uv1, uv2, uv3) or null
    uint32_t* error_count;           // triangles with an index outside the vertex array
};
extern "C" hipError_t ptk_launch_mesh(const PtkMeshJob* job, hipStream_t stream);

// Batched updates (ptk_update.hip): a staging block = n_entries PtkScatterEntry, then payload bytes.
struct PtkScatterEntry {
    uint64_t dst;         // device address
    uint32_t src_offset;  // of the payload inside the staging block (16-byte aligned)
    uint32_t nbytes;
};
extern "C" hipError_t ptk_launch_scatter(const void* d_stage, uint32_t n_entries, hipStream_t stream);
// Validation and derivation of the traversal structures on the device (ptk_derive.hip).
struct PtkDeriveInfo {
    uint32_t error;          // a check failed (the host's walk over its copy of the same bytes names it)
    uint32_t total;          // entries of the level list handed out so far
    uint32_t begin, end;     // the level the next launch processes: entries [begin, end)
    uint32_t level;          // its depth
    uint32_t finished;       // the list is complete: end = nodes reached, level = levels of the tree
    uint32_t done_blocks;    // workgroups of the running level launch that are through
    uint32_t stack_max;      // max over inner nodes of (stack entries below it + its children)
    uint32_t nested_socket;  // a mesh socket inside a mesh: block traversal does not cover the scene
    uint32_t n_blocks, n_sockets;  // inner nodes that get a block; world-level mesh sockets
    uint32_t has_tri;        // the id table names a triangle
    uint32_t pad[4];
};
struct PtkDerive {
    const int32_t* nodes;    // bound node buffer, 8 words per node
    const int32_t* ids;      // bound id table
    const float4* spheres; const float4* planes; const float4* triangles;
    int n_nodes, n_ids, n_sph, n_pln, n_tri, n_xf, n_mats;
    PtkDeriveInfo* info;
    uint32_t* ent_node;      // [n_nodes] the tree level by level: node | inside a mesh << 31
    uint32_t* ent_aux;       // [n_nodes] stack entries below the node | its level << 16
    uint32_t* visited;       // [(n_nodes + 31) / 32] zeroed
    int32_t* node_blk;       // [n_nodes] -1: the node's block
    int32_t* node_sock;      // [n_nodes] -1: the node's row of the socket table
    int32_t* node_slot;      // [n_nodes] -1: block * 8 + child slot that holds the node's box
    int32_t* tri_slot;       // [n_tri] -1: the id slot that names the triangle (-2: several do)
    int32_t* sph_slot;       // [n_sph] -1
    uint8_t* mat_used;       // [n_mats] 0: a primitive that can be hit has this material
    uint2* partial;          // [(n_nodes + 1023) / 1024 + 1] scan scratch
};
extern "C" hipError_t ptk_launch_derive_begin(const PtkDerive* d, hipStream_t stream);
extern "C" hipError_t ptk_launch_derive_levels(const PtkDerive* d, int n_levels, hipStream_t stream);
extern "C" hipError_t ptk_launch_derive_number(const PtkDerive* d, uint32_t n_reached, hipStream_t stream);
extern "C" hipError_t ptk_launch_derive_fill(const PtkDerive* d, uint32_t n_reached, float* blocks, int32_t* sockets, hipStream_t stream);

// inverse44 of n 64-byte matrices (ptk_update.hip), into PtkParams::inv_transforms.
extern "C" hipError_t ptk_launch_inverse_transforms(const void* xf64, int n, void* out64, hipStream_t stream);
// Leaf records (48 bytes per object-id slot) from the bound id table, triangle and sphere buffers, all on the device.
extern "C" hipError_t ptk_launch_leaf_records(const int32_t* ids, int n_ids, const void* triangles80, int n_tri, const void* spheres32, int n_sph,
                                              void* out48, hipStream_t stream);
