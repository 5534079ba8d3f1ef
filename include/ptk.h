/*
 * ptk.h -- C ABI of libptk, the MI355X render core that stands in for the
 * Godot RenderingDevice calls made by the reference's PTWorkDispatcher /
 * PTRenderer for the path-tracing compute dispatch.
 *
 * Every entry point names the reference call site it replaces (paths relative
 * to the reference checkout).  All pointers are host pointers; the caller owns
 * them and the library copies before returning (PackedByteArray semantics).
 * Handles are opaque non-zero 64-bit values (the role Godot's RID plays); 0
 * means failure.  Functions returning int return 0 on success and a negative
 * PTK_E_* code on failure; ptk_last_error() then holds a message.  Nothing
 * here aborts the process.  One context = one GPU + one HIP stream; calls on a
 * context are not re-entrant (Godot drives them from its main thread).
 */
#ifndef PTK_H
#define PTK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ptk_ctx ptk_ctx;

enum {
    PTK_OK = 0,
    PTK_E_INVALID = -1,     /* bad argument / unknown handle */
    PTK_E_HIP = -2,         /* a HIP runtime call failed */
    PTK_E_UNSUPPORTED = -3, /* scene outside what the kernels accept (e.g. node wider than 8) */
    PTK_E_STATE = -4        /* a required buffer is not bound */
};

/* (set, binding) contract of shaders/ray_tracer.comp:364-414 and
 * proprietary_dataclasses/pt_work_dispatcher.gd:76-109. */
enum {
    PTK_SET_NONE = -1,    /* ptk_buffer_create: not bound yet (ptk_buffer_bind places it) */
    PTK_SET_IMAGE = 0,
    PTK_SET_OBJECT = 2,   /* bindings: 0 materials, 1 spheres, 2 planes, 3 triangles */
    PTK_SET_BVH = 3,      /* bindings: 0 nodes, 1 object ids */
    PTK_SET_TRIANGLE = 4, /* bindings: 0 vertex, 1 uv, 2 index (bound, never read by the kernel) */
    PTK_SET_TEXTURE = 5,
    PTK_SET_TRANSFORM = 6 /* binding 0: world->mesh matrices */
};

/* PTBVHTree.BVHType, proprietary_dataclasses/bvh_trees/pt_bvh_tree.gd:40-46 */
enum { PTK_BVH_EMPTY = 0, PTK_BVH_X_SORTED = 1, PTK_BVH_Y_SORTED = 2, PTK_BVH_Z_SORTED = 3, PTK_BVH_XYZ_SORTED = 4,
       /* Not a reference type: the slot the reference keeps for a surface-area-heuristic builder and leaves empty
        * (bvh_trees/pt_bvh_sweep_sah.gd:1-21, pt_bvh_tree.gd:40-55).  Binned SAH, 8-wide nodes (order must be 8), same
        * node bytes, id table and indexing rules as the other types, so everything downstream works unchanged. */
       PTK_BVH_SAH = 5 };

/* Replaces RenderingServer.create_local_rendering_device() /
 * get_rendering_device(), pt_work_dispatcher.gd:146-160. */
ptk_ctx* ptk_create(int device_ordinal);
/* Replaces PTWorkDispatcher.free_rids() for the whole device, pt_work_dispatcher.gd:448-454. */
void ptk_destroy(ptk_ctx* ctx);
/* Error text of the last failed call on ctx (ctx may be NULL after a failed ptk_create). */
const char* ptk_last_error(ptk_ctx* ctx);

/* Replaces shader_compile_spirv_from_source + shader_create_from_spirv +
 * compute_pipeline_create (pt_work_dispatcher.gd:436-445) fed by
 * PTUtils.load_shader (pt_utils.gd:140-195): selects the precompiled kernel
 * variant.  bvh_order = text-replaced `order` constant; refraction_mode,
 * use_emission, early_t_break = the shader's compile-time switches
 * (ray_tracer.comp:84,92,96).  proc_tex_paths[i] is the texture_path of the
 * scene's procedural texture with function index i+1; only the two shipped
 * files (checker_board.comp, checker_board_spatial.comp) are accepted. */
int ptk_configure(ptk_ctx* ctx, int bvh_order, int refraction_mode, int use_emission, int early_t_break,
                  const char* const* proc_tex_paths, int n_paths);
/* The same selection for a host that keeps PTWorkDispatcher.load_shader as it is (pt_work_dispatcher.gd:436-445):
 * replaces rd.shader_compile_spirv_from_source + shader_create_from_spirv + compute_pipeline_create, taking the
 * GLSL text of RDShaderSource.source_compute as PTUtils.load_shader edited it (pt_utils.gd:140-195).  Nothing is
 * compiled: the `order` constant, the #define switches use_refraction_tracking / USE_EMISSION / use_early_t_break
 * and the spliced _procedural_texture_function<i> definitions are read from the text, and each spliced function
 * is matched, by a digest of its source without comments and white space, to one of the built-in functions (the
 * two files the plugin ships, plus whatever ptk_register_procedural_texture added).  A function that matches
 * none is PTK_E_UNSUPPORTED; a text without the `order` line is PTK_E_INVALID. */
int ptk_configure_from_source(ptk_ctx* ctx, const char* source, size_t nbytes);
/* What ptk_configure_from_source reads from a source text, without applying it (ctx may be NULL: then only the
 * shipped files are recognised): info21 = {order, refraction_mode, use_emission, early_t_break, n_procedural,
 * built-in function of procedural texture 1..16 (1 checker_board, 2 checker_board_spatial, 0 unknown)}.
 * No counterpart in the reference. */
int ptk_shader_source_info(ptk_ctx* ctx, const char* source, size_t nbytes, int32_t* info21);
/* Lets ptk_configure_from_source recognise another procedural-texture file as one of the built-in functions:
 * text = the file as written (a GLSL function `vec4 procedural_texture(vec3 pos)` with whatever it declares in
 * front), builtin_path = the shipped file whose function it computes ("checker_board.comp" or
 * "checker_board_spatial.comp").  For scenes whose texture files are copies or re-formattings of the shipped
 * ones; the library does not check that the two functions agree.  No counterpart in the reference, where the
 * driver compiles whatever the file holds.  ptk_procedural_texture_digest returns the digest used for the match. */
int ptk_register_procedural_texture(ptk_ctx* ctx, const char* text, size_t nbytes, const char* builtin_path);
int ptk_procedural_texture_digest(const char* text, size_t nbytes, uint64_t* digest);

/* Replaces rd.storage_buffer_create + RDUniform binding + uniform_set_create,
 * pt_work_dispatcher.gd:642-653 and :408-433.  Creating a buffer for a
 * (set, binding) that already has one rebinds the slot (the old handle stays
 * valid until freed), as PTWorkDispatcher does when it grows a buffer
 * (:511-637). */
uint64_t ptk_buffer_create(ptk_ctx* ctx, int set, int binding, const void* bytes, size_t nbytes);
/* Replaces rd.uniform_set_create + compute_list_bind_uniform_set for one uniform (pt_work_dispatcher.gd:408-433,
 * 482-487) when the buffer was created without a place (set = PTK_SET_NONE, as rd.storage_buffer_create(size, bytes)
 * creates it, :643): from now on the kernels read `handle` at (set, binding).  Also takes an image handle for
 * (0,0).  Binding what is bound there already changes nothing (the plugin binds every set before every
 * dispatch); a buffer sits at one place at a time. */
int ptk_buffer_bind(ptk_ctx* ctx, uint64_t handle, int set, int binding);
/* Replaces rd.buffer_update, proprietary_dataclasses/pt_renderer.gd:664-697,771-788. */
int ptk_buffer_update(ptk_ctx* ctx, uint64_t buffer, size_t offset, const void* bytes, size_t nbytes);
/* Replaces rd.buffer_get_data (used by the reference only while debugging); reads a buffer back. */
int ptk_buffer_read(ptk_ctx* ctx, uint64_t buffer, size_t offset, void* dst, size_t nbytes);

/* Replaces PTWorkDispatcher._create_image_buffer, pt_work_dispatcher.gd:374-405:
 * an rgba32f storage image bound at (0,0).  Contents start as zeros. */
uint64_t ptk_image_create(ptk_ctx* ctx, int width, int height);
/* Replaces rd.texture_get_data(image, 0), pt_renderer.gd:327: width*height*16 bytes. */
int ptk_image_read(ptk_ctx* ctx, uint64_t image, void* dst, size_t nbytes);
/* The display side of the image (shaders/canvas.gdshader:22-36, shown through the Texture2DRD of
 * pt_work_dispatcher.gd:192-196 and pt_renderer.gd:519-540): width*height*4 bytes of RGBA8, ready for a Godot
 * ImageTexture, converted on the GPU so that a quarter of the bytes cross PCIe.  transfer selects how much of
 * the canvas shader's work is done here: PTK_DISPLAY_STORED = clamp and quantise the stored values (the canvas
 * shader then applies its own pow when it samples; also what a screenshot of the floats clamps to,
 * pt_renderer.gd:320-350); PTK_DISPLAY_CANVAS = the shader's pow(rgb, 2.2) first (canvas.gdshader:33);
 * PTK_DISPLAY_CANVAS_SRGB = that, then the linear -> sRGB encoding of Godot's viewport output. */
enum { PTK_DISPLAY_STORED = 0, PTK_DISPLAY_CANVAS = 1, PTK_DISPLAY_CANVAS_SRGB = 2 };
int ptk_image_read_display(ptk_ctx* ctx, uint64_t image, void* dst_rgba8, size_t nbytes, int transfer);
/* Replaces rd.texture_update on the image (created with CAN_UPDATE, pt_work_dispatcher.gd:375-382). */
int ptk_image_update(ptk_ctx* ctx, uint64_t image, const void* src, size_t nbytes);

/* Replaces the per-texture rd.texture_create + sampler pair of
 * PTWorkDispatcher.create_texture_buffers, pt_work_dispatcher.gd:297-347:
 * RGBA32F level-0 data for slot `slot` of the 512-entry sampler array; rgba
 * NULL selects the 8x8 filler grid (:350-370). */
uint64_t ptk_texture_create(ptk_ctx* ctx, int slot, int width, int height, const float* rgba);
/* Places a texture created with slot = -1 into slot `slot` of the sampler array: the uniform.add_id(sampler),
 * uniform.add_id(texture) pairs of create_texture_buffers (pt_work_dispatcher.gd:314-345) once their uniform set
 * is bound.  One texture may fill many slots (the plugin's filler fills every unused one). */
int ptk_texture_bind(ptk_ctx* ctx, uint64_t texture, int slot);

/* Replaces compute_list_begin .. compute_list_dispatch .. compute_list_end,
 * pt_work_dispatcher.gd:470-497.  push128 is the 128-byte push constant block
 * of _push_constant_byte_array (:771-808); groups are 8x8x1 invocations
 * (ray_tracer.comp:55).  Asynchronous on the context's stream. */
int ptk_dispatch(ptk_ctx* ctx, const void* push128, int groups_x, int groups_y, int groups_z);
/* ptk_dispatch for callers that cannot allocate device memory (serving loops with a fixed memory budget, harnesses
 * that forbid hipMalloc while rendering).  No counterpart in the reference.  Same arguments, same image
 * bit for bit as ptk_dispatch; rendered by the persistent path kernel, whose pixel counter and
 * traversal-stack spill column live in a static 8 MiB device array of the library (one copy per GPU, held
 * until the process ends) instead of allocated buffers.  It calls no hipMalloc / hipFree.  Conditions,
 * each reported as an error rather than worked around by allocating:
 *   - the bound scene must be unchanged since the last successful ptk_dispatch (that call builds and
 *     allocates the derived tables): PTK_E_STATE otherwise;
 *   - plain render mode with the BVH flag, no tracing, no early-t variant, max_default_depth > 0, no mesh
 *     socket below another: PTK_E_UNSUPPORTED otherwise;
 *   - the bound BVH's deepest traversal stack must fit: 20 entries per ray are kept on chip, the rest in the
 *     static array, which at one wave per compute unit holds (8 MiB - 256) / (CUs * 256 B) further entries
 *     (127 on a 256-CU MI355X); a deeper tree is PTK_E_UNSUPPORTED.
 * All calls on one GPU share that GPU's array.  The library orders them: a call waits, on its context's stream, for
 * the end of the last launch on that GPU that used the array (an event recorded behind every such launch), so two
 * contexts may both use this entry point; their dispatches then run one after the other.  The launch is issued on
 * the context's own stream, not on one the caller could be capturing into a HIP graph. */
int ptk_dispatch_noalloc(ptk_ctx* ctx, const void* push128, int groups_x, int groups_y, int groups_z);
/* Replaces rd.submit() + rd.sync(), pt_work_dispatcher.gd:494-497. */
int ptk_sync(ptk_ctx* ctx);
/* Replaces rd.free_rid, pt_work_dispatcher.gd:457-465. */
int ptk_free(ptk_ctx* ctx, uint64_t handle);

/* Replaces PTBVHTree.create_bvh (pt_bvh_tree.gd:164-189) ->
 * PTBVHAxisSort.create_bvh_node_list (pt_bvh_axis_sort.gd:156-229) for one
 * tree.  aabbs6[i] = object.get_bvh_aabb() as position.xyz,size.xyz and
 * object_ids[i] = object.get_object_id(), in the order
 * create_disconnected_node_list visits objects (:62-84: spheres, then
 * triangles).  mesh_bvhs are handles of trees built earlier for the scene's
 * PTMeshes (adopted by the new tree), mesh_transforms12 their Transform3D as
 * basis rows (9) + origin (3), mesh_indices their index in the scene's mesh
 * list (the transform index). */
uint64_t ptk_bvh_build(ptk_ctx* ctx, int order, int type, int n_objects, const float* aabbs6,
                       const int32_t* object_ids, int n_meshes, const uint64_t* mesh_bvhs,
                       const float* mesh_transforms12, const int32_t* mesh_indices);
/* Node and object-id counts of a built tree (PTBVHTree.size(), object_ids.size()). */
int ptk_bvh_sizes(ptk_ctx* ctx, uint64_t bvh, int32_t* n_nodes, int32_t* n_object_ids);
/* Replaces PTWorkDispatcher._create_bvh_byte_array (index_tree + to_byte_array,
 * pt_work_dispatcher.gd:723-744) and _create_object_id_byte_array (:747-758):
 * 32 bytes per node in index_tree order, and the object id table. */
int ptk_bvh_read(ptk_ctx* ctx, uint64_t bvh, void* node_bytes, int32_t* object_ids);

/* Replaces the tree -> bytes -> buffer step of PTWorkDispatcher.create_buffers for a tree built here
 * (create_object_id_buffer + create_bvh_buffer, pt_work_dispatcher.gd:184-185,723-758): the node bytes and the id
 * table are written from the builder's device buffers straight into two new storage buffers bound at (3,0) and
 * (3,1), sized and padded as the plugin sizes them, without a trip through host memory.  The handles come back
 * for buffer_read / free_rid.  The bound tree stays refittable (ptk_bvh_refit) until the caller edits or
 * replaces these two buffers itself. */
int ptk_bvh_bind(ptk_ctx* ctx, uint64_t bvh, uint64_t* nodes_buffer, uint64_t* ids_buffer);

/* Replaces PTBVHTree.update_aabb / update_mesh_socket_aabb and the dirty-node uploads that follow them
 * (proprietary_dataclasses/bvh_trees/pt_bvh_tree.gd:254-271,873-882; pt_renderer.gd:771-788) for the tree bound
 * with ptk_bvh_bind.  Each event is one call of the reference: an object whose get_bvh_aabb() changed
 * (object_id = its id, v[0..5] = the new box as position.xyz, size.xyz) or a mesh whose transform changed
 * (object_id = mesh index | PTK_REFIT_MESH << 24, v[0..11] = the new Transform3D as basis rows + origin).
 * Events are applied in order with the reference's rule -- recompute the node's box, stop unless it differs from
 * the old one by more than AABB.is_equal_approx allows, otherwise queue the node and go on with the parent --
 * on the GPU, all events of the call at once.  The queued nodes' records are rewritten in the bound node buffer
 * (and in what the kernels derive from it); *n_nodes_updated = how many.  The primitive records of moved objects
 * are the caller's to update (ptk_buffer_update), as in the plugin. */
typedef struct ptk_refit_event {
    int32_t object_id;
    float v[12];
} ptk_refit_event;
enum { PTK_REFIT_MESH = 4 };
int ptk_bvh_refit(ptk_ctx* ctx, const ptk_refit_event* events, int n_events, int32_t* n_nodes_updated);

/* Replaces PTObjectContainer.mesh_to_pttriangles (proprietary_dataclasses/pt_object_container.gd:163-186: one
 * PTTriangle per index triplet of Mesh.surface_get_arrays(0)) together with what those triangles feed:
 * PTTriangle.to_byte_array (public/primitives/pt_triangle.gd:125-150) and get_bvh_aabb (pt_triangle.gd:107-112,
 * pt_object.gd:269-272).  The surface is given as Godot stores it (the dictionary of
 * RenderingServer.mesh_get_surface: format flags, vertex / attribute / index byte blocks, aabb, uv_scale);
 * positions are float32 or, under ARRAY_FLAG_COMPRESS_ATTRIBUTES, unorm16 relative to the aabb.  Surfaces without
 * an index array or with custom attribute channels are PTK_E_UNSUPPORTED.
 *   transform12    the meshlets' Transform3D as basis rows (9) + origin (3); NULL = identity, points copied as stored
 *   records80      out, index_count / 3 records of 80 bytes, ready for the (2,3) triangle buffer
 *   aabbs6         out or NULL, position.xyz,size.xyz per triangle in ptk_bvh_build's input form
 *   uvs6           out or NULL, (u,v) of the three corners per triangle (zeros when the surface has no UVs):
 *                  PTTriangle.set_uvs; the per-triangle table PTK_OPT_TRIANGLE_UVS reads at (4,1)
 * Returns the triangle count in *n_triangles. */
typedef struct ptk_mesh_surface {
    uint64_t format;                 /* Mesh.ArrayFormat flags of the surface */
    int32_t vertex_count, index_count;
    const void* vertex_data; size_t vertex_bytes;
    const void* attribute_data; size_t attribute_bytes;   /* may be NULL / 0 */
    const void* index_data; size_t index_bytes;
    float aabb[6];                   /* position.xyz, size.xyz */
    float uv_scale[4];
} ptk_mesh_surface;
int ptk_mesh_to_triangles(ptk_ctx* ctx, const ptk_mesh_surface* surface, const float* transform12, int32_t material_index,
                          uint32_t texture_id, void* records80, float* aabbs6, float* uvs6, int32_t* n_triangles);

/* Switches with no counterpart in the reference; every one defaults to the reference's behaviour.
 * PTK_OPT_TRIANGLE_UVS (default 0): with 1, the buffer at (4,1) -- bound but never read by the reference
 * (ray_tracer.comp:396-406) -- is read as six floats per triangle record, the (u,v) of point1, point2, point3
 * (ptk_mesh_to_triangles' uvs6), and a triangle hit's uv becomes their interpolation
 * (1-u-v)*uv1 + u*uv2 + v*uv3 instead of the barycentric (u,v) themselves: the conversion the shader leaves as
 * a TODO at ray_tracer.comp:1071-1072.  Triangles past the end of the table keep the barycentric uv.
 * PTK_OPT_LAUNCH_TIMING (default 0): with n > 0, every n-th dispatch of the WAVEFRONT pipeline (the first one
 * after the call included) records a HIP event before and after each of its traversal launches, which
 * ptk_dispatch_traverse_ms reads; each record costs about 3 us of stream time, so it is off unless somebody
 * measures. */
enum { PTK_OPT_TRIANGLE_UVS = 1, PTK_OPT_LAUNCH_TIMING = 2 };
int ptk_set_option(ptk_ctx* ctx, int option, int value);

/* Measurement hooks.  The reference counts box and object tests per invocation
 * in node_hit_count / object_hit_count (ray_tracer.comp:221-222,672,982) and
 * can only show them through the BVH-view render mode; these calls expose the
 * same counters, plus the first segment's hit object_id, as per-pixel buffers.
 * flags: bit 0 = primary hit id, bit 1 = counters.  With tracing on, a dispatch
 * also fills the buffers; image output is unchanged. */
enum { PTK_TRACE_PRIMARY_ID = 1, PTK_TRACE_COUNTERS = 2 };
enum {
    PTK_CNT_RAYS = 0, PTK_CNT_BOX_TESTS, PTK_CNT_OBJ_TESTS, PTK_CNT_SPHERE_TESTS, PTK_CNT_TRI_TESTS,
    PTK_CNT_SOCKETS, PTK_CNT_HIT_SPHERE, PTK_CNT_HIT_PLANE, PTK_CNT_HIT_TRIANGLE, PTK_CNT_MATERIAL_FETCH,
    PTK_NCOUNTERS
};
int ptk_trace_enable(ptk_ctx* ctx, int flags);
/* which = PTK_TRACE_PRIMARY_ID: width*height int32; PTK_TRACE_COUNTERS: width*height*PTK_NCOUNTERS int32. */
int ptk_trace_read(ptk_ctx* ctx, int which, void* dst, size_t nbytes);
/* Duration in milliseconds of the last ptk_dispatch's kernel, from HIP events
 * recorded on the context's stream around the launch (blocks until it ended). */
int ptk_last_dispatch_ms(ptk_ctx* ctx, float* ms);
/* Same for the last n dispatches (n <= 64), ms[0] the oldest. */
int ptk_dispatch_history_ms(ptk_ctx* ctx, float* ms, int n);

/* Schedule selection; no counterpart in the reference, where the driver's shader compiler owns the
 * schedule.  The library renders a dispatch in one of three ways, all producing the same image bit for
 * bit: SINGLE_KERNEL, one kernel in which a lane runs a pixel's whole bounce loop and a wave is an 8x8
 * tile (ray_tracer.comp:1553-1635 as written); WAVEFRONT, a pipeline with one traversal launch and one
 * shading launch per bounce over ray queues in device memory; PERSISTENT, one kernel of persistent
 * waves whose lanes take a new pixel as soon as their path ends.  WAVEFRONT and PERSISTENT cover plain
 * render mode with the BVH flag set, no tracing and no early-t variant (WAVEFRONT: at most 64 bounces).
 * Where several apply, AUTO times each on the first dispatches of a workload (scene contents, group
 * counts, depth limits) and keeps the fastest: the pipeline wins on heavy scenes, the persistent kernel
 * on light scenes whose paths differ in length, the tile kernel on light scenes whose paths do not.
 * WAVEFRONT / PERSISTENT make ptk_dispatch fail with PTK_E_UNSUPPORTED where they do not apply. */
enum { PTK_RENDERER_AUTO = 0, PTK_RENDERER_SINGLE_KERNEL = 1, PTK_RENDERER_WAVEFRONT = 2, PTK_RENDERER_PERSISTENT = 3 };
int ptk_set_renderer(ptk_ctx* ctx, int renderer);
/* Which of the three schedules rendered the last dispatch (one of PTK_RENDERER_SINGLE_KERNEL / _WAVEFRONT /
 * _PERSISTENT): what AUTO settled on, once its probing dispatches are over.  No counterpart in the reference. */
int ptk_last_dispatch_renderer(ptk_ctx* ctx, int* renderer);
/* Durations (ms) of the traversal-kernel launches of the dispatch `back` calls ago (0 = the last
 * one), from HIP events on the context's stream.  *n_launches = number of launches (0 when that
 * dispatch used another renderer); ms needs room for that many.  A dispatch issued with
 * PTK_OPT_LAUNCH_TIMING off has no events: its durations read -1. */
int ptk_dispatch_traverse_ms(ptk_ctx* ctx, int back, float* ms, int capacity, int* n_launches);

/* How often the library had to re-validate the bound scene and rebuild its derived structures
 * (full_validations), and how many ptk_buffer_update calls it absorbed by patching them in place
 * (patched_updates): the reference's incremental edits (pt_renderer.gd:664-788: single records, boxes
 * of dirty BVH nodes) are patched, and so are an object leaving or joining a leaf that has room
 * (pt_bvh_tree.gd:333-338, 375-385: the leaf's record with another object count over the same id
 * slots, and writes to those slots of the object-id table, up to 4096 slots a call); any other change
 * of a node's id or a change of its transform index makes the next dispatch rebuild.  No counterpart
 * in the reference. */
int ptk_update_stats(ptk_ctx* ctx, long long* full_validations, long long* patched_updates);

/* Multi-GPU glue; no counterpart in the reference, which renders on one device.  Copies the rgba32f
 * image to (to_image = 0) or from (to_image = 1) a device pointer owned by the caller on the same GPU
 * (e.g. the torch tensor handed to an RCCL reduce / gather), without a trip through host memory. */
int ptk_image_copy_device(ptk_ctx* ctx, uint64_t image, void* device_ptr, size_t nbytes, int to_image);

/* Combining the per-GPU images of one node over RCCL, for hosts without torch.distributed; no counterpart in the
 * reference, which renders on one device.  One context per GPU, one process (or thread) per context; RCCL is
 * loaded at the first ptk_comm_* call (librccl.so), so a single-GPU host never needs it.
 *   ptk_comm_unique_id   rank 0 makes the 128-byte id (ncclGetUniqueId) and hands it to the other ranks by any means
 *   ptk_comm_init        every rank joins (ncclCommInitRank on the context's GPU); collective
 *   ptk_image_reduce_mean   sample batches: every rank holds the running mean of its own samples (frame numbers
 *                        from 1, so that none is overwritten); afterwards root's image is the mean over ranks
 *                        (sum, then * 1/world); the other ranks' images are left as they were
 *   ptk_image_gather_bands  tile rows: rank r has rendered rows [first_row[r], first_row[r] + n_rows[r]) of the
 *                        image (multiples of 8 from the group counts and image_offset_y it dispatched with);
 *                        afterwards root's image holds every band, bit for bit
 *   ptk_comm_destroy     leaves the communicator (ptk_destroy does it too)
 * All collectives run on the context's stream, ordered after its dispatches, and return when they have ended. */
enum { PTK_COMM_ID_BYTES = 128 };
int ptk_comm_unique_id(void* id128);
int ptk_comm_init(ptk_ctx* ctx, int world, int rank, const void* id128);
int ptk_image_reduce_mean(ptk_ctx* ctx, uint64_t image, int root);
int ptk_image_gather_bands(ptk_ctx* ctx, uint64_t image, const int32_t* first_row, const int32_t* n_rows, int root);
int ptk_comm_destroy(ptk_ctx* ctx);
/* The band tables for ptk_image_gather_bands (and for the image_offset_y / group counts each rank dispatches with):
 * the image's 8-row tiles dealt out in rank order, contiguous, tile counts differing by at most one; a rank may get
 * no rows when there are more ranks than tile rows.  ptk_band_check is the test ptk_image_gather_bands applies to
 * tables made elsewhere: every band inside the image, no two overlapping (PTK_E_INVALID otherwise, message in
 * ptk_last_error(NULL)).  Both are plain arithmetic: no context, no GPU. */
int ptk_band_layout(int image_height, int world, int32_t* first_row, int32_t* n_rows);
int ptk_band_check(int image_height, int world, const int32_t* first_row, const int32_t* n_rows);

#ifdef __cplusplus
}
#endif
#endif /* PTK_H */
