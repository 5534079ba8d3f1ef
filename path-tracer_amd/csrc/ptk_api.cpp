// ptk_api.cpp -- host side of libptk: contexts, buffers, validation, dispatch.
//
// Serves the calls the reference's PTWorkDispatcher / PTRenderer make on
// Godot's RenderingDevice (proprietary_dataclasses/pt_work_dispatcher.gd,
// pt_renderer.gd); see include/ptk.h for the call-by-call mapping.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfenv>
#include <climits>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <chrono>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "ptk_bvh.h"
#include "ptk_device.h"
#include "ptk_source.h"

namespace {

enum class Kind { Buffer, Image, Texture, Bvh };

struct Object {
    Kind kind = Kind::Buffer;
    void* dev = nullptr;
    size_t nbytes = 0;
    int set = -1, binding = -1;
    int width = 0, height = 0, slot = -1;
    std::vector<uint8_t> shadow;  // host copy of storage buffers (validation, derived tables)
    std::shared_ptr<PtkBvh> bvh;  // shared with every scene tree that merged it as a mesh
};

// The tree bound with ptk_bvh_bind, kept refittable until its buffers are edited or rebound behind its back.
struct BoundTree {
    uint64_t nodes_handle = 0, ids_handle = 0;
    std::shared_ptr<PtkBvh> bvh;
    std::unique_ptr<PtkRefitState> state;
    std::vector<std::pair<int32_t, int32_t>> slot_of_object;  // (object id, slot in the id table), sorted; built at the first refit
};

struct Slot {
    uint64_t handle = 0;
};

}  // namespace

struct ptk_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    static constexpr int kTimingRing = 64;
    hipEvent_t ev0[kTimingRing] = {nullptr}, ev1[kTimingRing] = {nullptr};
    long long dispatch_count = 0;
    std::string error;
    std::map<uint64_t, std::unique_ptr<Object>> objects;
    uint64_t next_handle = 1;
    std::map<std::pair<int, int>, uint64_t> bound;  // (set, binding) -> buffer handle
    uint64_t image = 0;
    std::unique_ptr<BoundTree> bound_tree;
    void* comm = nullptr;  // ncclComm_t of ptk_comm_init
    int comm_world = 0, comm_rank = -1;
    bool in_refit = false;  // the library's own node updates do not invalidate the bound tree
    std::map<int, uint64_t> texture_slots;
    // configuration (ptk_configure)
    int order = 8, refraction_mode = 2, use_emission = 0, early_t_break = 0;
    int opt_triangle_uvs = 0;  // PTK_OPT_TRIANGLE_UVS
    int opt_launch_timing = 0; // PTK_OPT_LAUNCH_TIMING: 0 off, n > 0: every n-th pipeline dispatch
    long long wf_dispatches = 0;  // pipeline dispatches since the option was set
    int n_proc = 0;
    int proc_builtin[PTK_MAX_PROC] = {0};
    std::vector<std::pair<uint64_t, int>> proc_registered;  // ptk_register_procedural_texture: digest -> PTK_PROC_*
    // derived state, rebuilt when a bound buffer changes
    bool dirty = true;
    int stack_need = 1;
    int* d_live_planes = nullptr;
    int n_live_planes = 0;
    PtkTexDesc* d_textures = nullptr;
    int n_textures = 0;
    bool textures_dirty = true;
    int* d_spill = nullptr;
    int* d_spill_t = nullptr;
    size_t spill_ints = 0;
    // wavefront renderer
    int renderer = PTK_RENDERER_AUTO;
    int last_choice = -1;          // schedule of the last dispatch: 0 one-kernel renderer, 1 pipeline, 2 persistent kernel
    bool wf_ok = false;            // scene has no mesh socket below another mesh socket
    float4* d_blocks = nullptr;    // compiled inner nodes, 256 bytes each (see PtkWavefront)
    size_t d_blocks_bytes = 0, d_sockets_bytes = 0;  // allocated sizes of d_blocks / d_sockets
    float4* d_inv_xf = nullptr;    // inverse of every matrix of the transform buffer (PtkParams::inv_transforms)
    size_t d_inv_xf_bytes = 0;
    bool inv_xf_dirty = false;     // a matrix was edited: the table is made again with the next flush
    int2* d_sockets = nullptr;
    float root_box[6] = {0};
    uint32_t root_entry = 0;
    uint32_t n_blocks = 0, n_sockets = 0, n_records = 0;
    bool leaves_hold_triangles = false;  // any triangle in the id table (decides whether long leaves are split, see trav_leaf)
    // Whether any path can gain refraction bounces (ray_tracer.comp:1476, 1519: only a hit on a material with opacity < 1,
    // or on material 0, the medium a path starts in, extends the bounce loop).  mat_used marks the materials of the
    // primitives the id table and the live planes name; marks are only ever added between two validations.
    std::vector<uint8_t> mat_used;
    bool may_refract = true;
    // Validation and derivation on the device (ptk_derive.hip): -1 for trees of 4096 nodes or more, 0 never, 1 always
    // (PTK_DEVICE_DERIVE, read once at create).  The maps the patches of single edits need -- blocks, socket table,
    // node -> child slot, primitive -> id slot -- then stay on the device until the first such edit asks for them.
    int device_derive = -1;
    void* d_derive = nullptr;
    size_t derive_bytes = 0;
    bool mirrors_partial = false;  // h_blocks / h_sockets hold only their first rows, node_slot / tri_slot / sph_slot nothing
    const int32_t* d_node_slot = nullptr; const int32_t* d_tri_slot = nullptr; const int32_t* d_sph_slot = nullptr;
    int derive_levels = 0;         // levels of the tree the last device pass walked (0: the host walked)
    long long device_derivations = 0;
    double last_validate_ms = 0.0; // host time of the last validate_scene
    // host mirrors and maps of the derived structures, for patching them on buffer_update
    std::vector<float> h_blocks;
    std::unique_ptr<float[]> h_leaf_records;  // 12 floats per object-id slot; only slots patched since the last validation are set
    size_t h_leaf_records_slots = 0, d_leaf_records_slots = 0;
    std::vector<int32_t> h_sockets;  // {entry of the mesh's root, transform index} per world-level socket
    uint32_t tree_entry = 0;         // the root node's own entry (root_entry is what follows the chain)
    int n_chain = 0;
    PtkChainStep chain[PTK_MAX_CHAIN];
    std::vector<int32_t> node_slot;  // node index -> block * 8 + child slot holding its box, -1 if none
    std::vector<int32_t> tri_slot, sph_slot;  // primitive index -> object-id slot holding its record, -1 if none
    long long full_validations = 0, patched_updates = 0;
    // Updates waiting for the next flush (before a dispatch or a read): byte ranges of buffers whose shadow copy
    // already holds the new bytes, and the pieces of the derived structures their patches touched.
    struct PendingRange { uint64_t handle; size_t offset, nbytes; };
    std::vector<PendingRange> pending;
    std::vector<uint32_t> dirty_blocks, dirty_slots;   // block indices / object-id slots whose host mirror changed
    std::vector<uint8_t> block_is_dirty, slot_is_dirty;
    bool live_planes_dirty = false;
    std::vector<int> h_live_planes;
    size_t live_planes_capacity = 0;
    static constexpr size_t kStageBytes = (size_t)4 << 20;
    uint8_t* h_stage[2] = {nullptr, nullptr};  // pinned
    uint8_t* d_stage[2] = {nullptr, nullptr};
    hipEvent_t stage_ev[2] = {nullptr, nullptr};
    bool stage_used[2] = {false, false};
    int stage_next = 0;
    long long flushes = 0;
    // AUTO renderer choice: all schedules give the same image, so the first dispatches of a workload
    // are used to time each of them (twice) and the fastest one is kept until the workload changes
    struct AutoChoice {
        unsigned long long key = 0;       // what was timed: scene version, groups, depth limits, image
        float best_ms[3] = {-1.0f, -1.0f, -1.0f};  // [0] one-kernel renderer, [1] wavefront pipeline, [2] persistent path kernel
        int samples[3] = {0, 0, 0};
        long long probe_dispatch[3] = {-1, -1, -1};  // dispatch number of a timing still in flight
    } auto_choice;
    unsigned long long scene_version = 0;
    float4* d_leaf_records = nullptr;  // per object-id slot: the primitive by value (48 bytes)
    void* d_wf = nullptr;          // queues + path state, one allocation
    size_t wf_slots = 0;
    int* d_wf_spill = nullptr;
    uint32_t* d_pixel_counter = nullptr;  // persistent renderer: next pixel slot to hand out
    // one-kernel renderer: per-tile durations of the last dispatch and the order made from them (PtkParams::tile_order)
    uint32_t* d_tile_cost = nullptr;
    uint32_t* d_tile_order = nullptr;
    size_t tile_capacity = 0;
    int tile_gx = 0, tile_gy = 0;     // group counts the recorded costs belong to (0: none recorded)
    long long tile_order_age = 0;     // dispatches of this grid since its costs were first on record
    bool tile_ordering = true;        // (PTK_TILE_ORDER=0 turns it off; read once at create)
    void* d_display = nullptr;     // RGBA8 staging of ptk_image_read_display
    size_t display_bytes = 0;
    size_t wf_spill_ints = 0;
    int n_cus = 0;
    int lds_window = 20;           // traversal stack entries per lane kept in LDS, deeper ones spill to a global column
                                   // (PTK_LDS_WINDOW, read once at create; what the stacks leave of a compute unit's LDS holds
                                   // the top node blocks: 12 / 16 / 20 / 24 / 32 rows gave 8.53 / 7.76 / 7.62 / 7.63 / 7.69 ms per step)
    int wf_parts = 1;              // parts of a launch's queue (PTK_WF_PARTS=8: one per XCD, see PtkWavefront::n_parts; read once at create).
                                   // Measured on the 262 144-triangle scene: 8 parts 8.07 ms per sample against 7.50 with one head; the L2 hit
                                   // rate is the same (0.603 / 0.597), the launches are 8 % longer with the gather path less busy: the parts
                                   // differ in work and the workgroups end up on one another's parts late.
    bool wf_carry = true;          // the pipeline parks unfinished searches for the next launch (PTK_WF_CARRY=0 turns it off; read once at create)
    int wf_max_lag = 2, wf_park_below = 16;  // (PTK_WF_MAX_LAG, PTK_WF_PARK_BELOW: tuning, read once at create)
    int leaf_redo_all = 0;         // (PTK_LEAF_REDO_ALL=1: exercises trav_leaf's fallback on every helped leaf; for tests)
    int leaf_split = 4;            // leaves longer than this are shared with a helper lane (PTK_LEAF_SPLIT: tuning, read once at create)
    int wf_wg_claim = 0;           // queue entries per workgroup claim; 0: chosen per scene (PTK_WF_WG_CLAIM, read once at create)
    void* d_wf_carry = nullptr;
    size_t wf_carry_bytes = 0;
    int waves_per_cu_cap = 0;      // > 0: upper bound on persistent waves per CU (PTK_WF_WAVES_PER_CU, read once at create)
    int wf_wgs_per_cu = 0, pp_wgs_per_cu = 0, pp_wg_waves = 0;  // occupancy of the two persistent kernels for the bound scene's LDS need
    uint32_t n_top = 0;            // blocks [0, n_top): top of the tree, breadth-first, LDS-resident in the traversal kernel
    hipEvent_t* wf_ev[kTimingRing] = {nullptr};  // per dispatch slot: 2 events per traversal launch
    int wf_iters[kTimingRing] = {0};            // traversal launches of that dispatch (0: one-kernel renderer)
    bool wf_timed[kTimingRing] = {false};       // ... and whether the events were recorded around them
    // tracing
    int trace_flags = 0;
    int* d_trace_primary = nullptr;
    int* d_trace_counters = nullptr;
    size_t trace_pixels = 0;
};

static std::string g_create_error;

static int fail(ptk_ctx* ctx, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->error = buf;
    else g_create_error = buf;
    return code;
}

#define HIP_TRY(ctx, expr)                                                                       \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess) return fail(ctx, PTK_E_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

static Object* find(ptk_ctx* ctx, uint64_t h, Kind kind) {
    auto it = ctx->objects.find(h);
    if (it == ctx->objects.end() || it->second->kind != kind) return nullptr;
    return it->second.get();
}
static Object* bound_buffer(ptk_ctx* ctx, int set, int binding) {
    auto it = ctx->bound.find({set, binding});
    if (it == ctx->bound.end()) return nullptr;
    return find(ctx, it->second, Kind::Buffer);
}

static int patch_derived(ptk_ctx* ctx, Object* o, size_t offset, size_t nbytes, const std::vector<uint8_t>& before);
static int flush_updates(ptk_ctx* ctx);
static void derive_chain(ptk_ctx* ctx);

extern "C" {

ptk_ctx* ptk_create(int device_ordinal) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        fail(nullptr, PTK_E_HIP, "no HIP device available (%s)", e == hipSuccess ? "count is 0" : hipGetErrorString(e));
        return nullptr;
    }
    if (device_ordinal < 0 || device_ordinal >= n) {
        fail(nullptr, PTK_E_INVALID, "device ordinal %d out of range (0..%d)", device_ordinal, n - 1);
        return nullptr;
    }
    if (hipSetDevice(device_ordinal) != hipSuccess) {
        fail(nullptr, PTK_E_HIP, "hipSetDevice(%d) failed", device_ordinal);
        return nullptr;
    }
    // ptk_dispatch_noalloc's static scratch: resolve this device's copy of the array now (the device is current)
    if (hipError_t se = ptk_persistent_scratch_init(device_ordinal); se != hipSuccess) {
        fail(nullptr, PTK_E_HIP, "hipGetSymbolAddress of the static scratch array failed on device %d: %s", device_ordinal, hipGetErrorString(se));
        return nullptr;
    }
    auto* ctx = new ptk_ctx();
    ctx->device = device_ordinal;
    bool ok = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) == hipSuccess;
    for (int i = 0; ok && i < ptk_ctx::kTimingRing; i++)
        ok = hipEventCreate(&ctx->ev0[i]) == hipSuccess && hipEventCreate(&ctx->ev1[i]) == hipSuccess;
    for (int k = 0; ok && k < 2; k++)
        ok = hipHostMalloc((void**)&ctx->h_stage[k], ptk_ctx::kStageBytes) == hipSuccess &&
             hipMalloc((void**)&ctx->d_stage[k], ptk_ctx::kStageBytes) == hipSuccess && hipEventCreate(&ctx->stage_ev[k]) == hipSuccess;
    hipDeviceProp_t prop;
    ok = ok && hipGetDeviceProperties(&prop, device_ordinal) == hipSuccess;
    if (ok) ctx->n_cus = prop.multiProcessorCount;
    // development knobs are read here once: a variable that appears in the host's environment later changes nothing
    // (at most as many rows as sixteen waves' windows, their spare rows and the workgroup's words fit a compute unit's LDS)
    const int max_window = (int)((PTK_LDS_BYTES_PER_CU - PTK_WF_WG_FLAG_BYTES) / (PTK_WF_WG_WAVES * 256)) - 1;
    if (const char* w = getenv("PTK_LDS_WINDOW")) ctx->lds_window = std::max(8, std::min(max_window, atoi(w)));
    if (const char* w = getenv("PTK_WF_CARRY")) ctx->wf_carry = atoi(w) != 0;
    if (const char* w = getenv("PTK_WF_PARTS")) ctx->wf_parts = atoi(w) >= PTK_WF_MAX_PARTS ? PTK_WF_MAX_PARTS : 1;
    if (const char* w = getenv("PTK_DEVICE_DERIVE")) ctx->device_derive = atoi(w) != 0 ? 1 : 0;
    if (const char* w = getenv("PTK_WF_MAX_LAG")) ctx->wf_max_lag = std::max(1, std::min(PTK_WF_MAX_LAG, atoi(w)));
    if (const char* w = getenv("PTK_WF_PARK_BELOW")) ctx->wf_park_below = std::max(1, std::min(64, atoi(w)));
    if (const char* w = getenv("PTK_LEAF_REDO_ALL")) ctx->leaf_redo_all = atoi(w) != 0;
    if (const char* w = getenv("PTK_LEAF_SPLIT")) ctx->leaf_split = std::max(1, std::min(255, atoi(w)));
    if (const char* w = getenv("PTK_WF_WG_CLAIM")) ctx->wf_wg_claim = std::max(1, std::min(64, atoi(w) / 64)) * 64;
    if (const char* c = getenv("PTK_WF_WAVES_PER_CU")) ctx->waves_per_cu_cap = std::max(1, atoi(c));
    if (const char* c = getenv("PTK_TILE_ORDER")) ctx->tile_ordering = atoi(c) != 0;
    if (!ok) {
        fail(nullptr, PTK_E_HIP, "stream/event creation failed");
        delete ctx;
        return nullptr;
    }
    return ctx;
}

static void release(ptk_ctx* ctx, Object* o) {
    (void)ctx;
    if (o->dev) (void)hipFree(o->dev);
    o->dev = nullptr;
}

void ptk_destroy(ptk_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    (void)ptk_comm_destroy(ctx);
    for (auto& kv : ctx->objects) release(ctx, kv.second.get());
    ctx->objects.clear();
    if (ctx->d_live_planes) (void)hipFree(ctx->d_live_planes);
    if (ctx->d_textures) (void)hipFree(ctx->d_textures);
    if (ctx->d_spill) (void)hipFree(ctx->d_spill);
    if (ctx->d_spill_t) (void)hipFree(ctx->d_spill_t);
    if (ctx->d_trace_primary) (void)hipFree(ctx->d_trace_primary);
    if (ctx->d_trace_counters) (void)hipFree(ctx->d_trace_counters);
    if (ctx->d_blocks) (void)hipFree(ctx->d_blocks);
    if (ctx->d_sockets) (void)hipFree(ctx->d_sockets);
    if (ctx->d_derive) (void)hipFree(ctx->d_derive);
    if (ctx->d_inv_xf) (void)hipFree(ctx->d_inv_xf);
    if (ctx->d_leaf_records) (void)hipFree(ctx->d_leaf_records);
    if (ctx->d_wf) (void)hipFree(ctx->d_wf);
    if (ctx->d_wf_spill) (void)hipFree(ctx->d_wf_spill);
    if (ctx->d_wf_carry) (void)hipFree(ctx->d_wf_carry);
    if (ctx->d_pixel_counter) (void)hipFree(ctx->d_pixel_counter);
    if (ctx->d_tile_cost) (void)hipFree(ctx->d_tile_cost);
    if (ctx->d_tile_order) (void)hipFree(ctx->d_tile_order);
    if (ctx->d_display) (void)hipFree(ctx->d_display);
    for (int k = 0; k < 2; k++) {
        if (ctx->h_stage[k]) (void)hipHostFree(ctx->h_stage[k]);
        if (ctx->d_stage[k]) (void)hipFree(ctx->d_stage[k]);
        if (ctx->stage_ev[k]) (void)hipEventDestroy(ctx->stage_ev[k]);
    }
    for (int i = 0; i < ptk_ctx::kTimingRing; i++) {
        if (ctx->wf_ev[i]) {
            for (int k = 0; k < 2 * PTK_WF_MAX_ITER; k++) (void)hipEventDestroy(ctx->wf_ev[i][k]);
            delete[] ctx->wf_ev[i];
        }
        if (ctx->ev0[i]) (void)hipEventDestroy(ctx->ev0[i]);
        if (ctx->ev1[i]) (void)hipEventDestroy(ctx->ev1[i]);
    }
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char* ptk_last_error(ptk_ctx* ctx) { return ctx ? ctx->error.c_str() : g_create_error.c_str(); }

static bool ends_with(const std::string& s, const char* suffix) {
    size_t n = strlen(suffix);
    return s.size() >= n && s.compare(s.size() - n, n, suffix) == 0;
}

// PTK_PROC_* of a texture_path as ptk_configure takes it, 0 if it names neither shipped file.
static int builtin_of_path(const char* path) {
    const std::string p = path ? path : "";
    if (ends_with(p, "checker_board_spatial.comp")) return PTK_PROC_CHECKER_BOARD_SPATIAL;
    if (ends_with(p, "checker_board.comp")) return PTK_PROC_CHECKER_BOARD;
    return 0;
}

static int apply_configuration(ptk_ctx* ctx, int bvh_order, int refraction_mode, int use_emission, int early_t_break,
                               const int* builtin, int n_proc) {
    if (bvh_order < 2) return fail(ctx, PTK_E_INVALID, "bvh_order must be >= 2 (got %d)", bvh_order);
    if (bvh_order > PTK_MAX_WIDTH)
        return fail(ctx, PTK_E_UNSUPPORTED, "bvh_order %d is wider than the %d the kernels are built for", bvh_order, PTK_MAX_WIDTH);
    if (refraction_mode < 0 || refraction_mode > 2) return fail(ctx, PTK_E_INVALID, "refraction_mode must be 0, 1 or 2");
    ctx->order = bvh_order;
    ctx->refraction_mode = refraction_mode;
    ctx->use_emission = use_emission ? 1 : 0;
    ctx->early_t_break = early_t_break ? 1 : 0;
    ctx->n_proc = n_proc;
    memset(ctx->proc_builtin, 0, sizeof ctx->proc_builtin);
    for (int i = 0; i < n_proc; i++) ctx->proc_builtin[i] = builtin[i];
    ctx->dirty = true;
    return PTK_OK;
}

int ptk_configure(ptk_ctx* ctx, int bvh_order, int refraction_mode, int use_emission, int early_t_break,
                  const char* const* proc_tex_paths, int n_paths) {
    if (!ctx) return PTK_E_INVALID;
    if (n_paths < 0 || n_paths > PTK_MAX_PROC) return fail(ctx, PTK_E_UNSUPPORTED, "at most %d procedural textures", PTK_MAX_PROC);
    int builtin[PTK_MAX_PROC] = {0};
    for (int i = 0; i < n_paths; i++) {
        const char* p = proc_tex_paths && proc_tex_paths[i] ? proc_tex_paths[i] : "";
        builtin[i] = builtin_of_path(p);
        if (!builtin[i]) return fail(ctx, PTK_E_UNSUPPORTED, "procedural texture '%s' is not one of the built-in functions", p);
    }
    return apply_configuration(ctx, bvh_order, refraction_mode, use_emission, early_t_break, builtin, n_paths);
}

int ptk_procedural_texture_digest(const char* text, size_t nbytes, uint64_t* digest) {
    if (!digest) return PTK_E_INVALID;
    std::string err;
    if (!ptk_source::file_digest(text, nbytes, *digest, err)) return fail(nullptr, PTK_E_INVALID, "%s", err.c_str());
    return PTK_OK;
}

int ptk_register_procedural_texture(ptk_ctx* ctx, const char* text, size_t nbytes, const char* builtin_path) {
    if (!ctx) return PTK_E_INVALID;
    const int b = builtin_of_path(builtin_path);
    if (!b) return fail(ctx, PTK_E_UNSUPPORTED, "'%s' is not one of the built-in functions", builtin_path ? builtin_path : "");
    uint64_t d = 0;
    std::string err;
    if (!ptk_source::file_digest(text, nbytes, d, err)) return fail(ctx, PTK_E_INVALID, "%s", err.c_str());
    for (auto& kv : ctx->proc_registered) if (kv.first == d) { kv.second = b; return PTK_OK; }
    ctx->proc_registered.push_back({d, b});
    return PTK_OK;
}

// info = {order, refraction_mode, use_emission, early_t_break, n_proc, builtin[PTK_MAX_PROC]} (builtin 0: not recognised)
static int read_shader_source(ptk_ctx* ctx, const char* source, size_t nbytes, int32_t* info) {
    ptk_source::Info si;
    std::string err;
    if (!ptk_source::parse_shader(source, nbytes, si, err)) return fail(ctx, PTK_E_INVALID, "%s", err.c_str());
    info[0] = si.order; info[1] = si.refraction_mode; info[2] = si.use_emission; info[3] = si.early_t_break; info[4] = si.n_proc;
    for (int i = 0; i < PTK_MAX_PROC; i++) {
        int b = i < si.n_proc ? ptk_source::builtin_of_digest(si.digest[i]) : 0;
        if (i < si.n_proc && !b && ctx)
            for (auto& kv : ctx->proc_registered) if (kv.first == si.digest[i]) b = kv.second;
        info[5 + i] = b;
    }
    return PTK_OK;
}

int ptk_shader_source_info(ptk_ctx* ctx, const char* source, size_t nbytes, int32_t* info21) {
    if (!info21) return PTK_E_INVALID;
    return read_shader_source(ctx, source, nbytes, info21);
}

int ptk_configure_from_source(ptk_ctx* ctx, const char* source, size_t nbytes) {
    if (!ctx) return PTK_E_INVALID;
    int32_t info[5 + PTK_MAX_PROC];
    { int r = read_shader_source(ctx, source, nbytes, info); if (r != PTK_OK) return r; }
    for (int i = 0; i < info[4]; i++)
        if (!info[5 + i])
            return fail(ctx, PTK_E_UNSUPPORTED, "procedural texture function %d of the shader source is not one of the built-in functions "
                                                "(and was not registered with ptk_register_procedural_texture)", i + 1);
    return apply_configuration(ctx, info[0], info[1], info[2], info[3], info + 5, info[4]);
}

uint64_t ptk_buffer_create(ptk_ctx* ctx, int set, int binding, const void* bytes, size_t nbytes) {
    if (!ctx) return 0;
    if (nbytes == 0) { fail(ctx, PTK_E_INVALID, "buffer size must be > 0"); return 0; }
    if (hipSetDevice(ctx->device) != hipSuccess) { fail(ctx, PTK_E_HIP, "hipSetDevice failed"); return 0; }
    auto o = std::make_unique<Object>();
    o->kind = Kind::Buffer;
    o->nbytes = nbytes;
    o->set = set;
    o->binding = binding;
    o->shadow.assign(nbytes, 0);
    if (bytes) memcpy(o->shadow.data(), bytes, nbytes);
    size_t alloc = (nbytes + 255) & ~(size_t)255;  // keep 16-byte vector loads of the last record in bounds
    if (hipMalloc(&o->dev, alloc) != hipSuccess) { fail(ctx, PTK_E_HIP, "hipMalloc(%zu) failed", alloc); return 0; }
    if (hipMemsetAsync(o->dev, 0, alloc, ctx->stream) != hipSuccess ||
        hipMemcpyAsync(o->dev, o->shadow.data(), nbytes, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
        hipStreamSynchronize(ctx->stream) != hipSuccess) {
        (void)hipFree(o->dev);
        fail(ctx, PTK_E_HIP, "upload of %zu bytes failed", nbytes);
        return 0;
    }
    uint64_t h = ctx->next_handle++;
    ctx->objects[h] = std::move(o);
    if (set == PTK_SET_NONE) return h;  // bound later (ptk_buffer_bind): nothing the kernels read has changed
    ctx->bound[{set, binding}] = h;
    if (set == PTK_SET_BVH) ctx->bound_tree.reset();  // another tree is bound now
    ctx->dirty = true;
    return h;
}

int ptk_buffer_bind(ptk_ctx* ctx, uint64_t handle, int set, int binding) {
    if (!ctx) return PTK_E_INVALID;
    auto it = ctx->objects.find(handle);
    if (it == ctx->objects.end()) return fail(ctx, PTK_E_INVALID, "unknown handle %llu", (unsigned long long)handle);
    Object* o = it->second.get();
    if (o->kind == Kind::Image) {
        if (set != PTK_SET_IMAGE || binding != 0) return fail(ctx, PTK_E_INVALID, "an image binds at (0,0) only (got (%d,%d))", set, binding);
        ctx->image = handle;
        return PTK_OK;
    }
    if (o->kind != Kind::Buffer) return fail(ctx, PTK_E_INVALID, "handle %llu is neither a storage buffer nor an image", (unsigned long long)handle);
    const bool known = (set == PTK_SET_OBJECT && binding >= 0 && binding <= 3) || (set == PTK_SET_BVH && binding >= 0 && binding <= 1) ||
                       (set == PTK_SET_TRIANGLE && binding >= 0 && binding <= 2) || (set == PTK_SET_TRANSFORM && binding == 0);
    if (!known) return fail(ctx, PTK_E_INVALID, "the shader has no storage buffer at (%d,%d)", set, binding);
    auto cur = ctx->bound.find({set, binding});
    if (cur != ctx->bound.end() && cur->second == handle) return PTK_OK;  // as it was: derived data stays valid
    // a buffer sits at one (set, binding) at a time: binding it elsewhere takes it out of its old place
    for (auto b = ctx->bound.begin(); b != ctx->bound.end();) {
        if (b->second == handle) b = ctx->bound.erase(b);
        else ++b;
    }
    ctx->bound[{set, binding}] = handle;
    o->set = set;
    o->binding = binding;
    if (set == PTK_SET_BVH && ctx->bound_tree && handle != ctx->bound_tree->nodes_handle && handle != ctx->bound_tree->ids_handle)
        ctx->bound_tree.reset();  // another tree is bound now
    ctx->dirty = true;
    return PTK_OK;
}

int ptk_buffer_update(ptk_ctx* ctx, uint64_t buffer, size_t offset, const void* bytes, size_t nbytes) {
    if (!ctx) return PTK_E_INVALID;
    Object* o = find(ctx, buffer, Kind::Buffer);
    if (!o) return fail(ctx, PTK_E_INVALID, "unknown buffer handle %llu", (unsigned long long)buffer);
    if (!bytes || offset > o->nbytes || nbytes > o->nbytes - offset)
        return fail(ctx, PTK_E_INVALID, "update of %zu bytes at %zu exceeds buffer of %zu bytes", nbytes, offset, o->nbytes);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (nbytes == 0) return PTK_OK;
    if (ctx->bound_tree && !ctx->in_refit && (buffer == ctx->bound_tree->nodes_handle || buffer == ctx->bound_tree->ids_handle))
        ctx->bound_tree.reset();  // edited by the caller: the library no longer knows the boxes behind these bytes
    const std::vector<uint8_t> before(o->shadow.begin() + offset, o->shadow.begin() + offset + nbytes);
    memcpy(o->shadow.data() + offset, bytes, nbytes);
    // The new bytes live in the shadow copy; the device copy follows at the next flush (before the next dispatch
    // or read), all pending ranges in one transfer, stream-ordered after the dispatches already queued.  Nothing
    // here waits for the GPU.
    ctx->pending.push_back({buffer, offset, nbytes});
    const int patched = patch_derived(ctx, o, offset, nbytes, before);
    if (patched < 0) { ctx->dirty = true; return patched; }
    if (patched == 0) ctx->dirty = true;
    else ctx->patched_updates++;  // (the AUTO renderer choice stays: topology and sizes are as they were timed)
    return PTK_OK;
}

int ptk_buffer_read(ptk_ctx* ctx, uint64_t buffer, size_t offset, void* dst, size_t nbytes) {
    if (!ctx) return PTK_E_INVALID;
    Object* o = find(ctx, buffer, Kind::Buffer);
    if (!o) return fail(ctx, PTK_E_INVALID, "unknown buffer handle %llu", (unsigned long long)buffer);
    if (!dst || offset > o->nbytes || nbytes > o->nbytes - offset) return fail(ctx, PTK_E_INVALID, "read out of range");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    { int r = flush_updates(ctx); if (r != PTK_OK) return r; }
    HIP_TRY(ctx, hipMemcpyAsync(dst, (uint8_t*)o->dev + offset, nbytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PTK_OK;
}

uint64_t ptk_image_create(ptk_ctx* ctx, int width, int height) {
    if (!ctx) return 0;
    if (width <= 0 || height <= 0 || width > 65536 || height > 65536) { fail(ctx, PTK_E_INVALID, "bad image size %dx%d", width, height); return 0; }
    if (hipSetDevice(ctx->device) != hipSuccess) { fail(ctx, PTK_E_HIP, "hipSetDevice failed"); return 0; }
    auto o = std::make_unique<Object>();
    o->kind = Kind::Image;
    o->width = width;
    o->height = height;
    o->nbytes = (size_t)width * height * 16;
    if (hipMalloc(&o->dev, o->nbytes) != hipSuccess) { fail(ctx, PTK_E_HIP, "hipMalloc(%zu) failed", o->nbytes); return 0; }
    if (hipMemsetAsync(o->dev, 0, o->nbytes, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) {
        (void)hipFree(o->dev);
        fail(ctx, PTK_E_HIP, "image clear failed");
        return 0;
    }
    uint64_t h = ctx->next_handle++;
    ctx->objects[h] = std::move(o);
    ctx->image = h;
    return h;
}

int ptk_image_read(ptk_ctx* ctx, uint64_t image, void* dst, size_t nbytes) {
    if (!ctx) return PTK_E_INVALID;
    Object* o = find(ctx, image, Kind::Image);
    if (!o) return fail(ctx, PTK_E_INVALID, "unknown image handle");
    if (!dst || nbytes != o->nbytes) return fail(ctx, PTK_E_INVALID, "image read needs exactly %zu bytes", o->nbytes);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpyAsync(dst, o->dev, nbytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PTK_OK;
}

int ptk_image_read_display(ptk_ctx* ctx, uint64_t image, void* dst_rgba8, size_t nbytes, int transfer) {
    if (!ctx) return PTK_E_INVALID;
    Object* o = find(ctx, image, Kind::Image);
    if (!o) return fail(ctx, PTK_E_INVALID, "unknown image handle");
    const size_t n_pixels = (size_t)o->width * o->height;
    if (!dst_rgba8 || nbytes != n_pixels * 4) return fail(ctx, PTK_E_INVALID, "display read needs exactly %zu bytes", n_pixels * 4);
    if (transfer < PTK_DISPLAY_STORED || transfer > PTK_DISPLAY_CANVAS_SRGB) return fail(ctx, PTK_E_INVALID, "unknown display transfer %d", transfer);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t need = (nbytes + 255) & ~(size_t)255;
    if (need > ctx->display_bytes) {
        if (ctx->d_display) { (void)hipFree(ctx->d_display); ctx->d_display = nullptr; }
        ctx->display_bytes = 0;
        HIP_TRY(ctx, hipMalloc(&ctx->d_display, need));
        ctx->display_bytes = need;
    }
    HIP_TRY(ctx, ptk_launch_display((const float4*)o->dev, ctx->d_display, n_pixels, transfer, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(dst_rgba8, ctx->d_display, nbytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PTK_OK;
}

int ptk_image_update(ptk_ctx* ctx, uint64_t image, const void* src, size_t nbytes) {
    if (!ctx) return PTK_E_INVALID;
    Object* o = find(ctx, image, Kind::Image);
    if (!o) return fail(ctx, PTK_E_INVALID, "unknown image handle");
    if (!src || nbytes != o->nbytes) return fail(ctx, PTK_E_INVALID, "image update needs exactly %zu bytes", o->nbytes);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpyAsync(o->dev, src, nbytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PTK_OK;
}

uint64_t ptk_texture_create(ptk_ctx* ctx, int slot, int width, int height, const float* rgba) {
    if (!ctx) return 0;
    if (slot < -1 || slot >= 512) { fail(ctx, PTK_E_INVALID, "texture slot %d out of range (0..511, or -1 for none)", slot); return 0; }
    if (rgba && (width <= 0 || height <= 0)) { fail(ctx, PTK_E_INVALID, "bad texture size"); return 0; }
    if (hipSetDevice(ctx->device) != hipSuccess) { fail(ctx, PTK_E_HIP, "hipSetDevice failed"); return 0; }
    auto o = std::make_unique<Object>();
    o->kind = Kind::Texture;
    o->slot = slot;
    o->width = rgba ? width : 8;
    o->height = rgba ? height : 8;
    if (rgba) {
        o->nbytes = (size_t)width * height * 16;
        if (hipMalloc(&o->dev, o->nbytes) != hipSuccess) { fail(ctx, PTK_E_HIP, "hipMalloc failed"); return 0; }
        if (hipMemcpyAsync(o->dev, rgba, o->nbytes, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
            hipStreamSynchronize(ctx->stream) != hipSuccess) {
            (void)hipFree(o->dev);
            fail(ctx, PTK_E_HIP, "texture upload failed");
            return 0;
        }
    }
    uint64_t h = ctx->next_handle++;
    ctx->objects[h] = std::move(o);
    if (slot < 0) return h;  // placed later (ptk_texture_bind)
    ctx->texture_slots[slot] = h;
    ctx->textures_dirty = true;
    return h;
}

int ptk_texture_bind(ptk_ctx* ctx, uint64_t texture, int slot) {
    if (!ctx) return PTK_E_INVALID;
    if (slot < 0 || slot >= 512) return fail(ctx, PTK_E_INVALID, "texture slot %d out of range (0..511)", slot);
    Object* o = find(ctx, texture, Kind::Texture);
    if (!o) return fail(ctx, PTK_E_INVALID, "unknown texture handle %llu", (unsigned long long)texture);
    auto cur = ctx->texture_slots.find(slot);
    if (cur != ctx->texture_slots.end() && cur->second == texture) return PTK_OK;
    ctx->texture_slots[slot] = texture;  // (one texture may fill many slots, as the plugin's filler does)
    o->slot = slot;
    ctx->textures_dirty = true;
    return PTK_OK;
}

int ptk_free(ptk_ctx* ctx, uint64_t handle) {
    if (!ctx) return PTK_E_INVALID;
    auto it = ctx->objects.find(handle);
    if (it == ctx->objects.end()) return fail(ctx, PTK_E_INVALID, "unknown handle %llu", (unsigned long long)handle);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    { int r = flush_updates(ctx); if (r != PTK_OK) return r; }  // nothing may still point at this object's memory
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    Object* o = it->second.get();
    for (auto b = ctx->bound.begin(); b != ctx->bound.end();) {
        if (b->second == handle) b = ctx->bound.erase(b);
        else ++b;
    }
    for (auto t = ctx->texture_slots.begin(); t != ctx->texture_slots.end();) {
        if (t->second == handle) { t = ctx->texture_slots.erase(t); ctx->textures_dirty = true; }
        else ++t;
    }
    if (ctx->image == handle) ctx->image = 0;
    if (ctx->bound_tree && (handle == ctx->bound_tree->nodes_handle || handle == ctx->bound_tree->ids_handle)) ctx->bound_tree.reset();
    release(ctx, o);
    ctx->objects.erase(it);
    ctx->dirty = true;
    return PTK_OK;
}

int ptk_sync(ptk_ctx* ctx) {
    if (!ctx) return PTK_E_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PTK_OK;
}

}  // extern "C"

// ---- validation of the bound scene ------------------------------------------------
namespace {

struct NodeView {
    const uint8_t* p;
    int n;
    int node_id(int i) const { int32_t v; memcpy(&v, p + (size_t)i * 32 + 12, 4); return v; }
    int xform(int i) const { int32_t v; memcpy(&v, p + (size_t)i * 32 + 28, 4); return v; }
};
inline int sext(int v, int off, int bits) { return ((int)((uint32_t)v << (32 - off - bits))) >> (32 - bits); }

}  // namespace

// Planes with a non-zero normal, in index order (the others can never be hit, ray_tracer.comp:859).
static int refresh_live_planes(ptk_ctx* ctx, Object* pln) {
    const int n_pln = (int)(pln->nbytes / 32);
    std::vector<int>& live = ctx->h_live_planes;
    live.clear();
    for (int i = 0; i < n_pln; i++) {
        float nrm[3];
        memcpy(nrm, pln->shadow.data() + (size_t)i * 32, 12);
        if (!(nrm[0] == 0.0f && nrm[1] == 0.0f && nrm[2] == 0.0f)) live.push_back(i);
    }
    ctx->n_live_planes = (int)live.size();
    if ((size_t)n_pln > ctx->live_planes_capacity || !ctx->d_live_planes) {  // sized for every plane record: edits never reallocate
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->d_live_planes) { (void)hipFree(ctx->d_live_planes); ctx->d_live_planes = nullptr; }
        ctx->live_planes_capacity = 0;
        HIP_TRY(ctx, hipMalloc((void**)&ctx->d_live_planes, std::max<size_t>(16, (size_t)n_pln * 4)));
        ctx->live_planes_capacity = (size_t)n_pln;
    }
    ctx->live_planes_dirty = true;  // uploaded with the next flush
    return PTK_OK;
}

// Brings everything the pending updates touched to the GPU, stream-ordered: byte ranges of the bound buffers
// (merged per buffer, taken from the shadow copies), changed node blocks, leaf records and the live-plane
// list (from their host mirrors).  Small pieces travel together in a pinned staging block -- a table of
// destinations plus payloads, one transfer, one scatter kernel; a piece larger than a quarter of the block
// is copied on its own.  No allocation, no wait for the GPU except when both staging blocks are still in flight.
static int flush_updates(ptk_ctx* ctx) {
    if (ctx->pending.empty() && ctx->dirty_blocks.empty() && ctx->dirty_slots.empty() && !ctx->live_planes_dirty) return PTK_OK;
    struct Piece { uint8_t* dst; const uint8_t* src; size_t n; };
    std::vector<Piece> pieces;
    {   // byte ranges, merged per buffer
        std::sort(ctx->pending.begin(), ctx->pending.end(), [](const ptk_ctx::PendingRange& a, const ptk_ctx::PendingRange& b) {
            return a.handle != b.handle ? a.handle < b.handle : a.offset < b.offset;
        });
        for (size_t i = 0; i < ctx->pending.size();) {
            const uint64_t h = ctx->pending[i].handle;
            size_t lo = ctx->pending[i].offset, hi = lo + ctx->pending[i].nbytes;
            size_t j = i + 1;
            Object* o = find(ctx, h, Kind::Buffer);
            for (; j <= ctx->pending.size(); j++) {
                const bool more = j < ctx->pending.size() && ctx->pending[j].handle == h;
                if (more && ctx->pending[j].offset <= hi) { hi = std::max(hi, ctx->pending[j].offset + ctx->pending[j].nbytes); continue; }
                if (o) pieces.push_back({(uint8_t*)o->dev + lo, o->shadow.data() + lo, hi - lo});
                if (!more) break;
                lo = ctx->pending[j].offset;
                hi = lo + ctx->pending[j].nbytes;
            }
            i = j;
        }
        ctx->pending.clear();
    }
    auto runs = [&](std::vector<uint32_t>& idx, std::vector<uint8_t>& flag, uint8_t* dev, const uint8_t* host, size_t unit) {
        std::sort(idx.begin(), idx.end());
        for (size_t i = 0; i < idx.size();) {
            size_t j = i + 1;
            while (j < idx.size() && idx[j] == idx[j - 1] + 1) j++;
            if (dev) pieces.push_back({dev + (size_t)idx[i] * unit, host + (size_t)idx[i] * unit, (j - i) * unit});
            i = j;
        }
        for (uint32_t k : idx) flag[k] = 0;
        idx.clear();
    };
    runs(ctx->dirty_blocks, ctx->block_is_dirty, (uint8_t*)ctx->d_blocks, (const uint8_t*)ctx->h_blocks.data(), 256);
    runs(ctx->dirty_slots, ctx->slot_is_dirty, (uint8_t*)ctx->d_leaf_records, (const uint8_t*)ctx->h_leaf_records.get(), 48);
    if (ctx->live_planes_dirty) {
        if (!ctx->h_live_planes.empty() && ctx->d_live_planes)
            pieces.push_back({(uint8_t*)ctx->d_live_planes, (const uint8_t*)ctx->h_live_planes.data(), ctx->h_live_planes.size() * 4});
        ctx->live_planes_dirty = false;
    }
    const size_t cap = ptk_ctx::kStageBytes;
    size_t i = 0;
    while (i < pieces.size()) {
        if (pieces[i].n > cap / 4) {  // large piece: its own copy (the runtime stages pageable memory before it returns)
            HIP_TRY(ctx, hipMemcpyAsync(pieces[i].dst, pieces[i].src, pieces[i].n, hipMemcpyHostToDevice, ctx->stream));
            i++;
            continue;
        }
        // a batch: as many following small pieces as fit into one staging block
        size_t j = i, payload = 0;
        while (j < pieces.size() && pieces[j].n <= cap / 4) {
            const size_t need = (j - i + 1) * sizeof(PtkScatterEntry) + 16 + payload + ((pieces[j].n + 15) & ~(size_t)15);
            if (need > cap) break;
            payload += (pieces[j].n + 15) & ~(size_t)15;
            j++;
        }
        const int k = ctx->stage_next;
        ctx->stage_next ^= 1;
        if (ctx->stage_used[k]) HIP_TRY(ctx, hipEventSynchronize(ctx->stage_ev[k]));  // (the transfer two batches ago)
        uint8_t* hs = ctx->h_stage[k];
        const size_t n_ent = j - i;
        size_t off = (n_ent * sizeof(PtkScatterEntry) + 15) & ~(size_t)15;
        for (size_t e = 0; e < n_ent; e++) {
            PtkScatterEntry ent;
            ent.dst = (uint64_t)(uintptr_t)pieces[i + e].dst;
            ent.src_offset = (uint32_t)off;
            ent.nbytes = (uint32_t)pieces[i + e].n;
            memcpy(hs + e * sizeof(PtkScatterEntry), &ent, sizeof ent);
            memcpy(hs + off, pieces[i + e].src, pieces[i + e].n);
            off += (pieces[i + e].n + 15) & ~(size_t)15;
        }
        HIP_TRY(ctx, hipMemcpyAsync(ctx->d_stage[k], hs, off, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipEventRecord(ctx->stage_ev[k], ctx->stream));
        ctx->stage_used[k] = true;
        HIP_TRY(ctx, ptk_launch_scatter(ctx->d_stage[k], (uint32_t)n_ent, ctx->stream));
        i = j;
    }
    if (ctx->inv_xf_dirty) {  // (behind the scatter that brought the edited matrices to the device)
        Object* xfb = bound_buffer(ctx, PTK_SET_TRANSFORM, 0);
        if (xfb && ctx->d_inv_xf && xfb->nbytes <= ctx->d_inv_xf_bytes)
            HIP_TRY(ctx, ptk_launch_inverse_transforms(xfb->dev, (int)(xfb->nbytes / 64), ctx->d_inv_xf, ctx->stream));
        ctx->inv_xf_dirty = false;
    }
    ctx->flushes++;
    return PTK_OK;
}

// The host copies of what a device derivation left on the GPU, fetched when the first edit needs them.
static int ensure_mirrors(ptk_ctx* ctx) {
    if (!ctx->mirrors_partial) return PTK_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    Object* nodes = bound_buffer(ctx, PTK_SET_BVH, 0);
    Object* tri = bound_buffer(ctx, PTK_SET_OBJECT, 3);
    Object* sph = bound_buffer(ctx, PTK_SET_OBJECT, 1);
    if (!nodes || !tri || !sph) return fail(ctx, PTK_E_STATE, "the bound scene lost a buffer");
    const size_t n_nodes = nodes->nbytes / 32, n_tri = tri->nbytes / 80, n_sph = sph->nbytes / 32;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->h_blocks.assign((size_t)ctx->n_blocks * 64, 0.0f);
    ctx->h_sockets.assign((size_t)ctx->n_sockets * 2, 0);
    ctx->node_slot.assign(n_nodes, -1);
    ctx->tri_slot.assign(n_tri, -1);
    ctx->sph_slot.assign(n_sph, -1);
    if (ctx->n_blocks) HIP_TRY(ctx, hipMemcpy(ctx->h_blocks.data(), ctx->d_blocks, (size_t)ctx->n_blocks * 256, hipMemcpyDeviceToHost));
    if (ctx->n_sockets) HIP_TRY(ctx, hipMemcpy(ctx->h_sockets.data(), ctx->d_sockets, (size_t)ctx->n_sockets * 8, hipMemcpyDeviceToHost));
    if (n_nodes) HIP_TRY(ctx, hipMemcpy(ctx->node_slot.data(), ctx->d_node_slot, n_nodes * 4, hipMemcpyDeviceToHost));
    if (n_tri) HIP_TRY(ctx, hipMemcpy(ctx->tri_slot.data(), ctx->d_tri_slot, n_tri * 4, hipMemcpyDeviceToHost));
    if (n_sph) HIP_TRY(ctx, hipMemcpy(ctx->sph_slot.data(), ctx->d_sph_slot, n_sph * 4, hipMemcpyDeviceToHost));
    ctx->mirrors_partial = false;
    return PTK_OK;
}

// The words 0..2, 4..11 of a triangle's leaf record from its 80-byte record, as ptk_leaf_records_kernel makes them:
// A, E1 = B - A with N.x, E2 = C - A with N.y (N = E1 x E2), each by one rounded operation per step.  The only
// floating-point arithmetic the host does for an image: it runs in the default environment (round to nearest, denormals
// kept), whatever the calling thread has set, so that it gives what the device kernel gives.
static void triangle_leaf_words(float* r, const uint8_t* p80) {
    struct DefaultFenv {
        fenv_t saved;
        DefaultFenv() { fegetenv(&saved); fesetenv(FE_DFL_ENV); }
        ~DefaultFenv() { fesetenv(&saved); }
    } in_default_environment;
    float a[3], b[3], c[3];
    memcpy(a, p80, 12); memcpy(b, p80 + 16, 12); memcpy(c, p80 + 32, 12);
    const float e1x = b[0] - a[0], e1y = b[1] - a[1], e1z = b[2] - a[2];
    const float e2x = c[0] - a[0], e2y = c[1] - a[1], e2z = c[2] - a[2];
    r[0] = a[0]; r[1] = a[1]; r[2] = a[2];
    r[4] = e1x; r[5] = e1y; r[6] = e1z; r[7] = e1y * e2z - e2y * e1z;
    r[8] = e2x; r[9] = e2y; r[10] = e2z; r[11] = e1z * e2x - e2z * e1x;
}

static void mark_material_used(ptk_ctx* ctx, int32_t m) {
    if (m >= 0 && (size_t)m < ctx->mat_used.size()) ctx->mat_used[m] = 1;
    else ctx->may_refract = true;  // (an index validation will refuse; nothing to conclude from it)
}
static void refresh_may_refract(ptk_ctx* ctx) {
    Object* mats = bound_buffer(ctx, PTK_SET_OBJECT, 0);
    bool any = !mats || ctx->mat_used.empty() || ctx->mat_used[0] != 0;
    for (size_t m = 0; !any && m < ctx->mat_used.size(); m++) {
        if (!ctx->mat_used[m] || (m + 1) * 64 > mats->shadow.size()) continue;
        float opacity;
        memcpy(&opacity, mats->shadow.data() + m * 64 + 20, 4);
        any = opacity < 1.0f;
    }
    ctx->may_refract = any;
}

// The incremental edit path (pt_renderer.gd:664-788 updates single records and the boxes of dirty BVH
// nodes with rd.buffer_update).  When the derived structures are current and an update keeps the tree's
// topology, they are patched in place instead of being rebuilt: leaf records of the touched primitives,
// child boxes of the touched nodes, the live-plane list.  Returns 1 if the update was absorbed, 0 if
// the next dispatch has to validate and rebuild everything, negative on error.  `before` holds the
// bytes the update replaced.
static int patch_derived(ptk_ctx* ctx, Object* o, size_t offset, size_t nbytes, const std::vector<uint8_t>& before) {
    if (ctx->dirty) return 0;
    if (bound_buffer(ctx, o->set, o->binding) != o) return 1;  // not what the kernels read
    Object* mats = bound_buffer(ctx, PTK_SET_OBJECT, 0);
    const int n_mats = mats ? (int)(mats->nbytes / 64) : 0;
    const size_t kMaxPatched = 4096;  // beyond this many records a rebuild is the cheaper way
    auto records = [&](size_t rec_bytes, size_t& first, size_t& last) {
        first = offset / rec_bytes;
        last = (offset + nbytes + rec_bytes - 1) / rec_bytes;  // exclusive
        last = std::min(last, o->nbytes / rec_bytes);
    };
    if (o->set == PTK_SET_OBJECT && o->binding == 0) { refresh_may_refract(ctx); return 1; }  // materials: only whether paths can refract
    if (o->set == PTK_SET_TRANSFORM) { derive_chain(ctx); ctx->inv_xf_dirty = true; return 1; }  // (a chain step holds its matrix by value)
    if (o->set == PTK_SET_TRIANGLE || o->set == PTK_SET_TEXTURE) return 1;
    if (o->set == PTK_SET_OBJECT && o->binding == 2) {  // planes
        size_t a, b;
        records(32, a, b);
        for (size_t i = a; i < b; i++) {
            int32_t m; memcpy(&m, o->shadow.data() + i * 32 + 16, 4);
            if (m < 0 || m >= n_mats) return 0;  // let validation report it
            mark_material_used(ctx, m);
        }
        refresh_may_refract(ctx);
        int r = refresh_live_planes(ctx, o);
        return r == PTK_OK ? 1 : r;
    }
    if (o->set == PTK_SET_OBJECT && (o->binding == 1 || o->binding == 3)) {  // spheres, triangles
        const bool tri = o->binding == 3;
        const size_t rec = tri ? 80 : 32;
        size_t a, b;
        records(rec, a, b);
        if (b - a > kMaxPatched) return 0;
        { int r = ensure_mirrors(ctx); if (r != PTK_OK) return r; }
        const std::vector<int32_t>& slot_of = tri ? ctx->tri_slot : ctx->sph_slot;
        for (size_t i = a; i < b; i++) {
            int32_t m; memcpy(&m, o->shadow.data() + i * rec + (tri ? 64 : 16), 4);
            if (m < 0 || m >= n_mats) return 0;
            mark_material_used(ctx, m);
            if (i >= slot_of.size()) return 0;
            const int32_t slot = slot_of[i];
            if (slot == -1) continue;  // not in any leaf
            if (slot < 0) return 0;    // in several leaves: rebuild
            if ((size_t)slot >= ctx->h_leaf_records_slots) return 0;
            float* r = ctx->h_leaf_records.get() + (size_t)slot * 12;
            const uint8_t* p = o->shadow.data() + i * rec;
            int32_t oid = 0;  // the slot's word of the id table
            if (Object* idt = bound_buffer(ctx, PTK_SET_BVH, 1)) { if ((size_t)slot * 4 + 4 <= idt->shadow.size()) memcpy(&oid, idt->shadow.data() + (size_t)slot * 4, 4); }
            memset(r, 0, 48);
            if (tri) triangle_leaf_words(r, p);
            else { memcpy(r + 0, p, 12); memcpy(r + 4, p + 12, 4); }
            memcpy(r + 3, &oid, 4);
            if (!ctx->slot_is_dirty[slot]) { ctx->slot_is_dirty[slot] = 1; ctx->dirty_slots.push_back((uint32_t)slot); }
        }
        refresh_may_refract(ctx);
        return 1;
    }
    if (o->set == PTK_SET_BVH && o->binding == 0) {
        // nodes: boxes may move; a leaf may change the number of objects it lists (the plugin's remove_object / add_object
        // on a leaf with room, pt_bvh_tree.gd:333-338, 375-385: the record comes back with another count, the same
        // first id slot); anything else about an id, and any transform index, changes the topology
        size_t a, b;
        records(32, a, b);
        if (b - a > kMaxPatched || !ctx->wf_ok) return 0;
        { int r = ensure_mirrors(ctx); if (r != PTK_OK) return r; }
        Object* idt = bound_buffer(ctx, PTK_SET_BVH, 1);
        const int n_ids = idt ? (int)(idt->nbytes / 4) : 0;
        for (size_t i = a; i < b; i++) {
            const size_t lo = std::max(offset, i * 32), hi = std::min(offset + nbytes, i * 32 + 32);
            uint8_t old_rec[32];
            memcpy(old_rec, o->shadow.data() + i * 32, 32);
            memcpy(old_rec + (lo - i * 32), before.data() + (lo - offset), hi - lo);
            if (memcmp(old_rec + 28, o->shadow.data() + i * 32 + 28, 4) != 0) return 0;  // transform index
            if (memcmp(old_rec + 12, o->shadow.data() + i * 32 + 12, 4) != 0) {
                int32_t was, now;
                memcpy(&was, old_rec + 12, 4);
                memcpy(&now, o->shadow.data() + i * 32 + 12, 4);
                const int was_idx = sext(was, 0, 24), now_idx = sext(now, 0, 24), now_count = std::abs(sext(now, 24, 8));
                // a leaf before and after (negative words, as PTBVHTree writes them), same first slot, its slots inside the table
                if (!(was < 0 && now < 0 && was_idx < 0 && now_idx == was_idx)) return 0;
                const int base = -now_idx - 1;
                if (now_count > PTK_MAX_WIDTH * 16 || base < 0 || base + now_count > n_ids) return 0;
                if (i >= ctx->node_slot.size() || ctx->node_slot[i] < 0) return 0;  // (a root that is a leaf: rebuild)
                // the slots it lists now may never have been reachable: the same test validation makes of a leaf's ids
                Object* tri = bound_buffer(ctx, PTK_SET_OBJECT, 3);
                Object* sph = bound_buffer(ctx, PTK_SET_OBJECT, 1);
                if (!idt || !tri || !sph) return 0;
                const int n_tri = (int)(tri->nbytes / 80), n_sph = (int)(sph->nbytes / 32);
                for (int k = 0; k < now_count; k++) {
                    int32_t oid;
                    memcpy(&oid, idt->shadow.data() + (size_t)(base + k) * 4, 4);
                    const int type = sext(oid, 24, 8), oi = sext(oid, 0, 24);
                    if ((type == 1 && (oi < 0 || oi >= n_sph)) || (type == 3 && (oi < 0 || oi >= n_tri))) return 0;  // let validation report it
                }
            }
        }
        for (size_t i = a; i < b; i++) {
            float box[8];
            memcpy(box, o->shadow.data() + i * 32, 32);
            if (i == 0) { memcpy(ctx->root_box, box, 12); memcpy(ctx->root_box + 3, box + 4, 12); }
            if (i >= ctx->node_slot.size()) return 0;
            const int32_t ns = ctx->node_slot[i];
            if (ns < 0) continue;  // the root, a mesh's root, or a node the traversal cannot reach
            const size_t blk = (size_t)(ns >> 3) * 64, l = (size_t)(ns & 7);
            float* h = ctx->h_blocks.data() + blk;
            const int plane_of[6] = {0, 1, 2, 4, 5, 6};
            for (int k = 0; k < 6; k++) h[k * 8 + l] = box[plane_of[k]];
            int32_t id;
            memcpy(&id, o->shadow.data() + i * 32 + 12, 4);
            if (id < 0) memcpy(h + 48 + l, &id, 4);  // a leaf's entry in its parent's block is its id word
            const uint32_t bi = (uint32_t)(ns >> 3);
            if (!ctx->block_is_dirty[bi]) { ctx->block_is_dirty[bi] = 1; ctx->dirty_blocks.push_back(bi); }
        }
        derive_chain(ctx);  // a patched box may be one of the chain's
        return 1;
    }
    if (o->set == PTK_SET_BVH && o->binding == 1) {
        // id table: a slot names another primitive (the ids of a leaf that lost or gained an object are written again):
        // the slot's leaf record is made anew from that primitive, and the maps from primitives to slots follow
        if ((offset | nbytes) & 3) return 0;
        const size_t a = offset / 4, b = (offset + nbytes) / 4;
        if (b - a > kMaxPatched) return 0;
        { int r = ensure_mirrors(ctx); if (r != PTK_OK) return r; }
        Object* tri = bound_buffer(ctx, PTK_SET_OBJECT, 3);
        Object* sph = bound_buffer(ctx, PTK_SET_OBJECT, 1);
        if (!tri || !sph || b > ctx->h_leaf_records_slots || b > ctx->slot_is_dirty.size()) return 0;
        const int n_tri = (int)(tri->nbytes / 80), n_sph = (int)(sph->nbytes / 32);
        for (size_t i = a; i < b; i++) {
            int32_t was, now;
            memcpy(&was, before.data() + (i * 4 - offset), 4);
            memcpy(&now, o->shadow.data() + i * 4, 4);
            if (was == now) continue;
            const int type = sext(now, 24, 8), oi = sext(now, 0, 24);
            if (type == 3 ? (oi < 0 || oi >= n_tri) : type == 1 ? (oi < 0 || oi >= n_sph) : false) return 0;  // let validation report it
            const int wtype = sext(was, 24, 8), woi = sext(was, 0, 24);
            // the primitive that was listed here: no longer in this slot (one listed twice keeps its mark)
            if (wtype == 3 && woi >= 0 && (size_t)woi < ctx->tri_slot.size() && ctx->tri_slot[woi] == (int32_t)i) ctx->tri_slot[woi] = -1;
            if (wtype == 1 && woi >= 0 && (size_t)woi < ctx->sph_slot.size() && ctx->sph_slot[woi] == (int32_t)i) ctx->sph_slot[woi] = -1;
            float* r = ctx->h_leaf_records.get() + i * 12;
            memset(r, 0, 48);
            if (type == 3) {
                if ((size_t)oi >= ctx->tri_slot.size()) return 0;
                const uint8_t* p = tri->shadow.data() + (size_t)oi * 80;
                triangle_leaf_words(r, p);
                { int32_t m; memcpy(&m, p + 64, 4); mark_material_used(ctx, m); }
                ctx->tri_slot[oi] = ctx->tri_slot[oi] == -1 ? (int32_t)i : -2;
                ctx->leaves_hold_triangles = true;
            } else if (type == 1) {
                if ((size_t)oi >= ctx->sph_slot.size()) return 0;
                const uint8_t* p = sph->shadow.data() + (size_t)oi * 32;
                memcpy(r + 0, p, 12); memcpy(r + 4, p + 12, 4);
                { int32_t m; memcpy(&m, p + 16, 4); mark_material_used(ctx, m); }
                ctx->sph_slot[oi] = ctx->sph_slot[oi] == -1 ? (int32_t)i : -2;
            }
            if (type == 3 || type == 1) memcpy(r + 3, &now, 4);
            if (!ctx->slot_is_dirty[i]) { ctx->slot_is_dirty[i] = 1; ctx->dirty_slots.push_back((uint32_t)i); }
        }
        refresh_may_refract(ctx);
        return 1;
    }
    return 0;  // object ids and anything else: rebuild
}

// The steps every search takes at the top of the tree before it reaches a node with a choice
// (PtkWavefront::chain): from the root's entry, while the next node is an inner node with exactly one child
// (a box test) or a world-level mesh socket.  Reads the host mirrors of the compiled blocks, so it is run
// again whenever they are patched.
static void derive_chain(ptk_ctx* ctx) {
    uint32_t e = ctx->tree_entry;
    int n = 0;
    if (ctx->wf_ok) {
        while (n < PTK_MAX_CHAIN) {
            PtkChainStep& c = ctx->chain[n];
            if ((e & 0xFF000000u) == 0x7F000000u) {  // mesh socket
                const size_t si = e & 0x00FFFFFFu;
                if (si * 2 + 1 >= ctx->h_sockets.size()) break;
                memset(&c, 0, sizeof c);
                c.kind = 1;
                c.xf = ctx->h_sockets[si * 2 + 1];
                Object* xfb = bound_buffer(ctx, PTK_SET_TRANSFORM, 0);
                if (!xfb || c.xf < 0 || ((size_t)c.xf + 1) * 64 > xfb->shadow.size()) break;  // (validation has checked the index)
                memcpy(c.m, xfb->shadow.data() + (size_t)c.xf * 64, 64);
                e = (uint32_t)ctx->h_sockets[si * 2];
            } else if ((int32_t)e > 0 && (e >> 24) == 1u) {  // inner node with one child
                const size_t b = e & 0x00FFFFFFu;
                if ((b + 1) * 64 > ctx->h_blocks.size()) break;
                const float* blk = ctx->h_blocks.data() + b * 64;
                c.lo[0] = blk[0]; c.lo[1] = blk[8]; c.lo[2] = blk[16];
                c.hi[0] = blk[24]; c.hi[1] = blk[32]; c.hi[2] = blk[40];
                c.kind = 0;
                c.xf = -1;
                memcpy(&e, blk + 48, 4);
            } else {
                break;
            }
            n++;
        }
    }
    ctx->n_chain = n;
    ctx->root_entry = e;
}

// Device room for the compiled blocks and the socket table (kept while it is large enough: a scene change then frees
// and allocates nothing).
static int reserve_blocks(ptk_ctx* ctx, size_t n_blocks, size_t n_sockets) {
    const size_t bb = std::max<size_t>(256, n_blocks * 256), sb = std::max<size_t>(8, n_sockets * 8);
    if (bb > ctx->d_blocks_bytes) {
        if (ctx->d_blocks) { (void)hipFree(ctx->d_blocks); ctx->d_blocks = nullptr; }
        ctx->d_blocks_bytes = 0;
        HIP_TRY(ctx, hipMalloc((void**)&ctx->d_blocks, bb));
        ctx->d_blocks_bytes = bb;
    }
    if (sb > ctx->d_sockets_bytes) {
        if (ctx->d_sockets) { (void)hipFree(ctx->d_sockets); ctx->d_sockets = nullptr; }
        ctx->d_sockets_bytes = 0;
        HIP_TRY(ctx, hipMalloc((void**)&ctx->d_sockets, sb));
        ctx->d_sockets_bytes = sb;
    }
    return PTK_OK;
}

// The top blocks the traversal kernel's LDS has room for beside its stack windows, at this stack need.
static size_t top_block_limit(const ptk_ctx* ctx, int stack_need) {
    const size_t stacks = (size_t)PTK_WF_WG_WAVES * (std::min(stack_need, ctx->lds_window) + 1) * 256 + PTK_WF_WG_FLAG_BYTES;
    const size_t room = PTK_LDS_BYTES_PER_CU > stacks ? (PTK_LDS_BYTES_PER_CU - stacks) / PTK_TOP_BLOCK_STRIDE : 0;
    return std::min<size_t>(room, 480);
}

// validate_scene's checks and derivations on the device, from the bound buffers where they lie (ptk_derive.hip).
// Returns PTK_OK when the scene passed and its structures are in place, 1 when a check failed or the scene is one
// the device pass leaves to the host (whose walk over the shadow copies then names the fault), negative on a HIP error.
// Blocks are numbered in breadth-first order here (the host numbers the top so and the rest depth first); the socket
// table, stack need and maps are the same.
static int derive_on_device(ptk_ctx* ctx, Object* nodes, Object* ids, Object* sph, Object* pln, Object* tri, int n_xf, int n_mats) {
    const int n_nodes = (int)(nodes->nbytes / 32), n_ids = (int)(ids->nbytes / 4);
    const int n_sph = (int)(sph->nbytes / 32), n_pln = (int)(pln->nbytes / 32), n_tri = (int)(tri->nbytes / 80);
    if (n_nodes > 0x00FFFFFF) return 1;
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    const size_t o_info = take(256), o_ent_node = take((size_t)n_nodes * 4), o_ent_aux = take((size_t)n_nodes * 4);
    const size_t o_vis = take(((size_t)n_nodes + 31) / 32 * 4);
    const size_t o_blk = take((size_t)n_nodes * 4), o_sock = take((size_t)n_nodes * 4), o_slot = take((size_t)n_nodes * 4);
    const size_t o_tri = take((size_t)std::max(n_tri, 1) * 4), o_sph = take((size_t)std::max(n_sph, 1) * 4);
    const size_t o_mat = take((size_t)std::max(n_mats, 1));
    const size_t o_part = take((((size_t)n_nodes + 1023) / 1024 + 1) * 8);
    if (off > ctx->derive_bytes) {
        if (ctx->d_derive) { (void)hipFree(ctx->d_derive); ctx->d_derive = nullptr; }
        ctx->derive_bytes = 0;
        HIP_TRY(ctx, hipMalloc(&ctx->d_derive, off + off / 4));
        ctx->derive_bytes = off + off / 4;
    }
    uint8_t* const base = (uint8_t*)ctx->d_derive;
    PtkDerive D;
    memset(&D, 0, sizeof D);
    D.nodes = (const int32_t*)nodes->dev; D.ids = (const int32_t*)ids->dev;
    D.spheres = (const float4*)sph->dev; D.planes = (const float4*)pln->dev; D.triangles = (const float4*)tri->dev;
    D.n_nodes = n_nodes; D.n_ids = n_ids; D.n_sph = n_sph; D.n_pln = n_pln; D.n_tri = n_tri; D.n_xf = n_xf; D.n_mats = n_mats;
    D.info = (PtkDeriveInfo*)(base + o_info);
    D.ent_node = (uint32_t*)(base + o_ent_node); D.ent_aux = (uint32_t*)(base + o_ent_aux);
    D.visited = (uint32_t*)(base + o_vis);
    D.node_blk = (int32_t*)(base + o_blk); D.node_sock = (int32_t*)(base + o_sock); D.node_slot = (int32_t*)(base + o_slot);
    D.tri_slot = (int32_t*)(base + o_tri); D.sph_slot = (int32_t*)(base + o_sph);
    D.mat_used = base + o_mat;
    D.partial = (uint2*)(base + o_part);
    HIP_TRY(ctx, hipMemsetAsync(base + o_vis, 0, o_blk - o_vis, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(base + o_blk, 0xFF, o_mat - o_blk, ctx->stream));  // node_blk .. sph_slot: -1
    HIP_TRY(ctx, hipMemsetAsync(base + o_mat, 0, o_part - o_mat, ctx->stream));
    HIP_TRY(ctx, ptk_launch_derive_begin(&D, ctx->stream));
    PtkDeriveInfo info;
    memset(&info, 0, sizeof info);
    // the tree's depth is not known beforehand: levels go out in batches; a finished list turns the rest of a batch into no-ops
    for (int batch = 0; !info.finished; batch++) {
        if (batch > 40) return 1;
        HIP_TRY(ctx, ptk_launch_derive_levels(&D, 16, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(&info, D.info, sizeof info, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    if (info.error) return 1;
    const int stack_need = std::max(1, (int)info.stack_max + 1);
    if (stack_need > 1024) return 1;
    const uint32_t n_reached = info.end;
    bool wf_ok = info.nested_socket == 0 && info.n_blocks <= 0x00FFFFFFu;
    const uint32_t n_blocks = wf_ok ? info.n_blocks : 0u, n_sockets = wf_ok ? info.n_sockets : 0u;
    { int r = reserve_blocks(ctx, n_blocks, n_sockets); if (r != PTK_OK) return r; }
    // the first rows of both tables are all derive_chain reads: the chain's steps are the tree's first levels
    const uint32_t head_blocks = std::min<uint32_t>(n_blocks, PTK_MAX_CHAIN + 2), head_sockets = std::min<uint32_t>(n_sockets, PTK_MAX_CHAIN + 2);
    ctx->h_blocks.assign((size_t)head_blocks * 64, 0.0f);
    ctx->h_sockets.assign((size_t)head_sockets * 2, 0);
    ctx->mat_used.assign((size_t)n_mats, 0);
    if (wf_ok) {
        HIP_TRY(ctx, ptk_launch_derive_number(&D, n_reached, ctx->stream));
        HIP_TRY(ctx, ptk_launch_derive_fill(&D, n_reached, (float*)ctx->d_blocks, (int32_t*)ctx->d_sockets, ctx->stream));
        if (head_blocks) HIP_TRY(ctx, hipMemcpyAsync(ctx->h_blocks.data(), ctx->d_blocks, (size_t)head_blocks * 256, hipMemcpyDeviceToHost, ctx->stream));
        if (head_sockets) HIP_TRY(ctx, hipMemcpyAsync(ctx->h_sockets.data(), ctx->d_sockets, (size_t)head_sockets * 8, hipMemcpyDeviceToHost, ctx->stream));
    }
    if (n_mats) HIP_TRY(ctx, hipMemcpyAsync(ctx->mat_used.data(), D.mat_used, (size_t)n_mats, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stack_need = stack_need;
    ctx->wf_ok = wf_ok;
    ctx->n_top = (uint32_t)std::min<size_t>(n_blocks, top_block_limit(ctx, stack_need));
    ctx->n_blocks = n_blocks;
    ctx->n_sockets = n_sockets;
    ctx->leaves_hold_triangles = info.has_tri != 0;
    // the root is the first node of the breadth-first order: block 0, or socket 0
    {
        int32_t id, x;
        memcpy(&id, nodes->shadow.data() + 12, 4);
        memcpy(&x, nodes->shadow.data() + 28, 4);
        const int idx = sext(id, 0, 24), size = std::abs(sext(id, 24, 8));
        ctx->tree_entry = idx <= 0 ? (uint32_t)id : x >= 0 ? 0x7F000000u : ((uint32_t)size << 24);
        memcpy(ctx->root_box, nodes->shadow.data(), 12);
        memcpy(ctx->root_box + 3, nodes->shadow.data() + 16, 12);
    }
    ctx->node_slot.clear(); ctx->tri_slot.clear(); ctx->sph_slot.clear();
    ctx->d_node_slot = D.node_slot; ctx->d_tri_slot = D.tri_slot; ctx->d_sph_slot = D.sph_slot;
    ctx->mirrors_partial = true;
    ctx->derive_levels = (int)info.level;
    ctx->device_derivations++;
    return PTK_OK;
}

// Walks the node buffer exactly as the kernel will and checks every index it
// can reach, and computes the deepest the traversal stack can get.
static int validate_scene(ptk_ctx* ctx) {
    const auto t_begin = std::chrono::steady_clock::now();
    Object* nodes = bound_buffer(ctx, PTK_SET_BVH, 0);
    Object* ids = bound_buffer(ctx, PTK_SET_BVH, 1);
    Object* mats = bound_buffer(ctx, PTK_SET_OBJECT, 0);
    Object* sph = bound_buffer(ctx, PTK_SET_OBJECT, 1);
    Object* pln = bound_buffer(ctx, PTK_SET_OBJECT, 2);
    Object* tri = bound_buffer(ctx, PTK_SET_OBJECT, 3);
    Object* xf = bound_buffer(ctx, PTK_SET_TRANSFORM, 0);
    if (!nodes || !ids || !mats || !sph || !pln || !tri || !xf)
        return fail(ctx, PTK_E_STATE, "dispatch needs buffers at (2,0..3), (3,0..1) and (6,0)");
    if (nodes->nbytes < 32) return fail(ctx, PTK_E_INVALID, "BVH buffer smaller than one node");
    const int n_nodes = (int)(nodes->nbytes / 32);
    const int n_ids = (int)(ids->nbytes / 4);
    const int n_mats = (int)(mats->nbytes / 64);
    const int n_sph = (int)(sph->nbytes / 32), n_pln = (int)(pln->nbytes / 32), n_tri = (int)(tri->nbytes / 80);
    const int n_xf = (int)(xf->nbytes / 64);
    if (n_mats < 1) return fail(ctx, PTK_E_INVALID, "material buffer holds no record");

    // large trees are checked and compiled on the device, small ones by the walk below (a handful of launches and two
    // round trips cost more than walking a few thousand nodes); a scene the device pass refuses is walked here too,
    // which is what names the fault
    bool on_device = false;
    if (ctx->device_derive == 1 || (ctx->device_derive < 0 && n_nodes >= 4096)) {
        const int r = derive_on_device(ctx, nodes, ids, sph, pln, tri, n_xf, n_mats);
        if (r < 0) return r;
        on_device = r == PTK_OK;
    }
    const int32_t* idp = (const int32_t*)ids->shadow.data();
    if (!on_device) {
        ctx->mirrors_partial = false;
        ctx->derive_levels = 0;
        auto mat_ok = [&](int m) { return m >= 0 && m < n_mats; };
        // material indices of every primitive the kernel may resolve
        for (int i = 0; i < n_sph; i++) { int32_t m; memcpy(&m, sph->shadow.data() + (size_t)i * 32 + 16, 4); if (!mat_ok(m)) return fail(ctx, PTK_E_INVALID, "sphere %d: material index %d out of range", i, m); }
        for (int i = 0; i < n_pln; i++) { int32_t m; memcpy(&m, pln->shadow.data() + (size_t)i * 32 + 16, 4); if (!mat_ok(m)) return fail(ctx, PTK_E_INVALID, "plane %d: material index %d out of range", i, m); }
        for (int i = 0; i < n_tri; i++) { int32_t m; memcpy(&m, tri->shadow.data() + (size_t)i * 80 + 64, 4); if (!mat_ok(m)) return fail(ctx, PTK_E_INVALID, "triangle %d: material index %d out of range", i, m); }

        NodeView nv{nodes->shadow.data(), n_nodes};
        // depth-first walk; need(node) = how far the stack can grow above its level
        // while the subtree of a popped node is processed: all k children are pushed
        // at once, then each is popped with up to k-1 siblings still below it.
        std::vector<uint8_t> visited((size_t)n_nodes, 0);
        bool wf_ok = true;
        std::string walk_error;
        int walk_code = PTK_OK;
        struct Walker {
            const NodeView& nv; const int32_t* idp; std::vector<uint8_t>& visited;
            int n_nodes, n_ids, n_sph, n_tri, n_xf;
            std::string& err; int& code;
            bool bad(int c, const char* fmt, int a = 0, int b = 0, int d = 0) {
                char buf[256]; snprintf(buf, sizeof buf, fmt, a, b, d); err = buf; code = c; return false;
            }
            bool& wf_ok;
            bool need(int i, int depth, bool in_mesh, int& out) {
                if (i < 0 || i >= n_nodes) return bad(PTK_E_INVALID, "BVH references node %d outside the buffer (%d nodes)", i, n_nodes);
                if (visited[i]) return bad(PTK_E_INVALID, "BVH node %d is reachable twice (not a tree)", i);
                visited[i] = 1;
                if (depth > 512) return bad(PTK_E_UNSUPPORTED, "BVH deeper than 512 levels");
                int id = nv.node_id(i);
                int idx = sext(id, 0, 24), size = std::abs(sext(id, 24, 8));
                int x = nv.xform(i);
                // The kernels tell a leaf from an inner node by the sign of the whole word.  That equals the
                // shader's test on the 24-bit index (ray_tracer.comp:1137) for every id PTBVHTree writes
                // (pt_bvh_tree.gd:910-959: inner = index ^ count << 24 with both positive, leaf = negative
                // index ^ (count - 1) << 24); any other bit pattern is refused rather than misread.
                if ((idx > 0 && id <= 0) || (idx < 0 && id >= 0))
                    return bad(PTK_E_UNSUPPORTED, "node %d: id 0x%08x is not in the encoding PTBVHTree writes", i, id);
                if (x >= n_xf) return bad(PTK_E_INVALID, "node %d: transform index %d out of range (%d)", i, x, n_xf);
                // (an all-zero root, the empty tree, has transform index 0 and child index 0: harmless)
                if (x >= 0 && idx < 0) return bad(PTK_E_INVALID, "node %d: a mesh socket (transform index %d) must have a child node", i, x);
                if (x >= 0 && in_mesh) wf_ok = false;  // a mesh socket below another: one-kernel renderer only
                const bool child_in_mesh = in_mesh || x >= 0;
                if (idx <= 0) {
                    int base = std::abs(idx) - 1;
                    if (size > 0 && (base < 0 || base + size > n_ids)) return bad(PTK_E_INVALID, "leaf %d: object id slots from %d (count %d) outside the id buffer", i, base, size);
                    for (int k = 0; k < size; k++) {
                        int oid = idp[base + k];
                        int type = sext(oid, 24, 8), oi = sext(oid, 0, 24);
                        if (type == 1 && (oi < 0 || oi >= n_sph)) return bad(PTK_E_INVALID, "leaf %d: sphere index %d out of range", i, oi);
                        if (type == 3 && (oi < 0 || oi >= n_tri)) return bad(PTK_E_INVALID, "leaf %d: triangle index %d out of range", i, oi);
                    }
                    out = 0;
                    return true;
                }
                if (size > PTK_MAX_WIDTH) return bad(PTK_E_UNSUPPORTED, "node %d has %d children; the kernels handle at most %d", i, size, PTK_MAX_WIDTH);
                if (idx + size > n_nodes) return bad(PTK_E_INVALID, "node %d: children from %d (count %d) outside the buffer", i, idx, size);
                int best = size;
                for (int l = 0; l < size; l++) {
                    int c = 0;
                    if (!need(idx + l, depth + 1, child_in_mesh, c)) return false;
                    best = std::max(best, size - 1 + c);
                }
                out = best;
                return true;
            }
        } walker{nv, idp, visited, n_nodes, n_ids, n_sph, n_tri, n_xf, walk_error, walk_code, wf_ok};
        int result = 0;
        if (!walker.need(0, 0, false, result)) return fail(ctx, walk_code, "%s", walk_error.c_str());
        ctx->stack_need = std::max(1, result + 1);
        if (ctx->stack_need > 1024)
            return fail(ctx, PTK_E_UNSUPPORTED, "BVH can need %d stack entries; the reference's traversal stack holds 1024", ctx->stack_need);
        if (n_nodes > 0x00FFFFFF) wf_ok = false;
        ctx->wf_ok = wf_ok;

        // Compile the inner nodes for the traversal kernel (layout in PtkWavefront): block b holds the boxes
        // and entries of one inner node's children; blocks are numbered parent first, depth first.
        std::vector<float>& blocks = ctx->h_blocks;
        blocks.clear();
        std::vector<int32_t>& sockets = ctx->h_sockets;
        sockets.clear();
        ctx->node_slot.assign((size_t)n_nodes, -1);
        ctx->n_top = 0;
        if (wf_ok) {
            // Block numbers: the top of the tree first, breadth-first from the root (through world-level mesh
            // sockets into the meshes), as many blocks as the traversal kernel's LDS has room for beside its
            // stack windows; everything below in depth-first order, parent first.
            std::vector<int32_t> block_of((size_t)n_nodes, -1);
            uint32_t next_free = 0;
            {
                const size_t stacks = (size_t)PTK_WF_WG_WAVES * (std::min(ctx->stack_need, ctx->lds_window) + 1) * 256 + PTK_WF_WG_FLAG_BYTES;
                const size_t room = PTK_LDS_BYTES_PER_CU > stacks ? (PTK_LDS_BYTES_PER_CU - stacks) / PTK_TOP_BLOCK_STRIDE : 0;
                const size_t limit = std::min<size_t>(room, 480);
                std::vector<std::pair<int, bool>> queue = {{0, false}};  // node, inside a mesh
                for (size_t q = 0; q < queue.size() && next_free < limit; q++) {
                    const int i = queue[q].first;
                    const bool in_mesh = queue[q].second;
                    const int id = nv.node_id(i);
                    const int idx = sext(id, 0, 24), size = std::abs(sext(id, 24, 8));
                    if (idx <= 0) continue;  // leaf
                    if (nv.xform(i) >= 0 && !in_mesh) { queue.push_back({idx, true}); continue; }  // socket: its child is the mesh's root
                    block_of[i] = (int32_t)next_free++;
                    for (int l = 0; l < size; l++) queue.push_back({idx + l, in_mesh});
                }
                ctx->n_top = next_free;
            }
            struct Compiler {
                const NodeView& nv; std::vector<float>& blocks; std::vector<int32_t>& sockets; std::vector<int32_t>& node_slot;
                const std::vector<int32_t>& block_of; uint32_t& next_free;
                uint32_t compile(int i, bool in_mesh, int depth = 0) {
                    const int id = nv.node_id(i);
                    const int idx = sext(id, 0, 24), size = std::abs(sext(id, 24, 8));
                    const int x = nv.xform(i);
                    if (idx <= 0) return (uint32_t)id;  // leaf
                    if (x >= 0 && !in_mesh) {           // world-level socket: its child is the mesh's root
                        const uint32_t child = compile(idx, true, depth);
                        const int sidx = (int)(sockets.size() / 2);
                        sockets.push_back((int32_t)child);
                        sockets.push_back(x);
                        return 0x7F000000u | (uint32_t)sidx;
                    }
                    const size_t b = block_of[i] >= 0 ? (size_t)block_of[i] : (size_t)next_free++;
                    if (blocks.size() < (b + 1) * 64) blocks.resize((b + 1) * 64, 0.0f);
                    memcpy(blocks.data() + b * 64 + 56, &depth, 4);  // first spare word: the block's depth (measurement builds read it)
                    for (int l = 0; l < size; l++) {
                        float box[8];
                        memcpy(box, nv.p + (size_t)(idx + l) * 32, 32);
                        const uint32_t e = compile(idx + l, in_mesh || x >= 0, depth + 1);
                        float* blk = blocks.data() + b * 64;  // (re-read: compile() may have grown the vector)
                        blk[0 + l] = box[0]; blk[8 + l] = box[1]; blk[16 + l] = box[2];
                        blk[24 + l] = box[4]; blk[32 + l] = box[5]; blk[40 + l] = box[6];
                        memcpy(blk + 48 + l, &e, 4);
                        node_slot[(size_t)idx + l] = (int32_t)(b * 8 + l);
                    }
                    return (uint32_t)b | ((uint32_t)size << 24);
                }
            } compiler{nv, blocks, sockets, ctx->node_slot, block_of, next_free};
            blocks.resize((size_t)next_free * 64, 0.0f);
            ctx->tree_entry = compiler.compile(0, false);
            memcpy(ctx->root_box, nv.p, 12);
            memcpy(ctx->root_box + 3, nv.p + 16, 12);
            if (blocks.size() / 64 > 0x00FFFFFF) ctx->wf_ok = false;
        }
        ctx->n_blocks = (uint32_t)(blocks.size() / 64);
        ctx->n_sockets = (uint32_t)(sockets.size() / 2);
        ctx->n_records = (uint32_t)std::max(n_ids, 0);
        { int r = reserve_blocks(ctx, blocks.size() / 64, sockets.size() / 2); if (r != PTK_OK) return r; }
        if (!blocks.empty()) HIP_TRY(ctx, hipMemcpyAsync(ctx->d_blocks, blocks.data(), blocks.size() * 4, hipMemcpyHostToDevice, ctx->stream));
        if (!sockets.empty()) HIP_TRY(ctx, hipMemcpyAsync(ctx->d_sockets, sockets.data(), sockets.size() * 4, hipMemcpyHostToDevice, ctx->stream));

    }
    derive_chain(ctx);
    ctx->n_records = (uint32_t)std::max(n_ids, 0);

    // leaf records: what a leaf tests for object-id slot i, by value, so the traversal kernel reads a
    // leaf's primitives from consecutive addresses without going through the id table first.  They are made on
    // the device from the bound buffers (the caller flushed pending edits of those first); the host keeps which
    // slot a primitive sits in, and a mirror that only ever holds the records patched since (patch_derived).
    const size_t n_slots = (size_t)std::max(n_ids, 1);
    if (n_slots > ctx->h_leaf_records_slots) {  // (left unset: a patch writes a slot's whole record before it is read)
        ctx->h_leaf_records.reset(new float[n_slots * 12]);
        ctx->h_leaf_records_slots = n_slots;
    }
    if (!on_device) {  // (the device pass has made these maps, and keeps them until an edit needs them here)
        ctx->tri_slot.assign((size_t)n_tri, -1);
        ctx->sph_slot.assign((size_t)n_sph, -1);
        ctx->leaves_hold_triangles = false;
        ctx->mat_used.assign((size_t)n_mats, 0);
        for (int i = 0; i < n_ids; i++) {
            const int oid = idp[i];
            const int type = sext(oid, 24, 8), oi = sext(oid, 0, 24);
            if (type == 3 && oi >= 0 && oi < n_tri) {
                ctx->leaves_hold_triangles = true;
                ctx->tri_slot[oi] = ctx->tri_slot[oi] == -1 ? i : -2;  // -2: listed more than once
                int32_t m; memcpy(&m, tri->shadow.data() + (size_t)oi * 80 + 64, 4);
                ctx->mat_used[m] = 1;  // (in range: checked for every record above)
            } else if (type == 1 && oi >= 0 && oi < n_sph) {
                ctx->sph_slot[oi] = ctx->sph_slot[oi] == -1 ? i : -2;
                int32_t m; memcpy(&m, sph->shadow.data() + (size_t)oi * 32 + 16, 4);
                ctx->mat_used[m] = 1;
            }
        }
        for (int i = 0; i < n_pln; i++) {  // planes are tested outside the tree: every record that can be hit
            float nrm[3]; int32_t m;
            memcpy(nrm, pln->shadow.data() + (size_t)i * 32, 12);
            memcpy(&m, pln->shadow.data() + (size_t)i * 32 + 16, 4);
            if (!(nrm[0] == 0.0f && nrm[1] == 0.0f && nrm[2] == 0.0f)) ctx->mat_used[m] = 1;
        }
    }
    refresh_may_refract(ctx);
    if (n_slots > ctx->d_leaf_records_slots) {
        if (ctx->d_leaf_records) { (void)hipFree(ctx->d_leaf_records); ctx->d_leaf_records = nullptr; }
        ctx->d_leaf_records_slots = 0;
        HIP_TRY(ctx, hipMalloc((void**)&ctx->d_leaf_records, n_slots * 48));
        ctx->d_leaf_records_slots = n_slots;
    }
    if (n_ids == 0) HIP_TRY(ctx, hipMemsetAsync(ctx->d_leaf_records, 0, 48, ctx->stream));
    HIP_TRY(ctx, ptk_launch_leaf_records((const int32_t*)ids->dev, n_ids, tri->dev, n_tri, sph->dev, n_sph, ctx->d_leaf_records, ctx->stream));
    // the inverse of every mesh matrix (the kernels read it where the shader computes inverse(mat4) per hit)
    if ((size_t)std::max(n_xf, 1) * 64 > ctx->d_inv_xf_bytes) {
        if (ctx->d_inv_xf) { (void)hipFree(ctx->d_inv_xf); ctx->d_inv_xf = nullptr; }
        ctx->d_inv_xf_bytes = 0;
        HIP_TRY(ctx, hipMalloc((void**)&ctx->d_inv_xf, (size_t)std::max(n_xf, 1) * 64));
        ctx->d_inv_xf_bytes = (size_t)std::max(n_xf, 1) * 64;
    }
    HIP_TRY(ctx, ptk_launch_inverse_transforms(xf->dev, n_xf, ctx->d_inv_xf, ctx->stream));
    ctx->inv_xf_dirty = false;

    // the derived structures were uploaded whole: nothing of them is pending any more
    ctx->dirty_blocks.clear();
    ctx->dirty_slots.clear();
    ctx->block_is_dirty.assign((size_t)ctx->n_blocks + 1, 0);
    ctx->slot_is_dirty.assign((size_t)std::max(n_ids, 1), 0);
    { int r = refresh_live_planes(ctx, pln); if (r != PTK_OK) return r; }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    // persistent waves per CU of both persistent kernels at this scene's LDS need (window + one spare row); a kernel
    // that does not fit (or whose occupancy cannot be asked) is not eligible for this scene, the others still are
    {
        const size_t lds = (size_t)(std::min(ctx->stack_need, ctx->lds_window) + 1) * 64 * 4;
        if (ptk_wavefront_occupancy((size_t)ctx->n_top * PTK_TOP_BLOCK_STRIDE + PTK_WF_WG_WAVES * lds + PTK_WF_WG_FLAG_BYTES, &ctx->wf_wgs_per_cu) != hipSuccess) {
            (void)hipGetLastError();
            ctx->wf_wgs_per_cu = 0;
        }
        if (ptk_persistent_occupancy(lds, ctx->n_top, &ctx->pp_wg_waves, &ctx->pp_wgs_per_cu) != hipSuccess) {
            (void)hipGetLastError();
            ctx->pp_wgs_per_cu = 0;
        }
        if (ctx->waves_per_cu_cap > 0 && ctx->pp_wg_waves > 0 && ctx->pp_wgs_per_cu > 0)
            ctx->pp_wgs_per_cu = std::max(1, std::min(ctx->pp_wgs_per_cu, ctx->waves_per_cu_cap / ctx->pp_wg_waves));
    }
    ctx->dirty = false;
    ctx->full_validations++;
    ctx->scene_version++;
    ctx->tile_gx = ctx->tile_gy = 0;  // the tile kernel's order of tiles was measured on another scene: the next dispatch measures anew
    ctx->last_validate_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    return PTK_OK;
}

static int refresh_textures(ptk_ctx* ctx) {
    int n = 0;
    for (auto& kv : ctx->texture_slots) n = std::max(n, kv.first + 1);
    std::vector<PtkTexDesc> table((size_t)std::max(n, 1));
    for (auto& d : table) { d.rgba = nullptr; d.width = 8; d.height = 8; }
    for (auto& kv : ctx->texture_slots) {
        Object* o = find(ctx, kv.second, Kind::Texture);
        if (!o) continue;
        table[kv.first].rgba = (const float*)o->dev;
        table[kv.first].width = o->width;
        table[kv.first].height = o->height;
    }
    if (ctx->d_textures) { (void)hipFree(ctx->d_textures); ctx->d_textures = nullptr; }
    HIP_TRY(ctx, hipMalloc((void**)&ctx->d_textures, table.size() * sizeof(PtkTexDesc)));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_textures, table.data(), table.size() * sizeof(PtkTexDesc), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->n_textures = n;
    ctx->textures_dirty = false;
    return PTK_OK;
}

extern "C" {

int ptk_trace_enable(ptk_ctx* ctx, int flags) {
    if (!ctx) return PTK_E_INVALID;
    if (flags & ~(PTK_TRACE_PRIMARY_ID | PTK_TRACE_COUNTERS)) return fail(ctx, PTK_E_INVALID, "unknown trace flags %d", flags);
    ctx->trace_flags = flags;
    return PTK_OK;
}

int ptk_trace_read(ptk_ctx* ctx, int which, void* dst, size_t nbytes) {
    if (!ctx) return PTK_E_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int* src = nullptr;
    size_t need = 0;
    if (which == PTK_TRACE_PRIMARY_ID) { src = ctx->d_trace_primary; need = ctx->trace_pixels * 4; }
    else if (which == PTK_TRACE_COUNTERS) { src = ctx->d_trace_counters; need = ctx->trace_pixels * 4 * PTK_NCOUNTERS; }
    else return fail(ctx, PTK_E_INVALID, "unknown trace buffer %d", which);
    if (!src) return fail(ctx, PTK_E_STATE, "no traced dispatch has run for buffer %d", which);
    if (!dst || nbytes != need) return fail(ctx, PTK_E_INVALID, "trace read needs exactly %zu bytes", need);
    HIP_TRY(ctx, hipMemcpyAsync(dst, src, nbytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PTK_OK;
}

// Argument checks shared by the dispatch entry points.
static int check_dispatch_args(ptk_ctx* ctx, const void* push128, int groups_x, int groups_y, int groups_z) {
    if (!push128) return fail(ctx, PTK_E_INVALID, "push constants missing");
    if (groups_x <= 0 || groups_y <= 0 || groups_z != 1 || groups_x > 65535 || groups_y > 65535)
        return fail(ctx, PTK_E_INVALID, "bad group counts (%d,%d,%d); depth must be 1", groups_x, groups_y, groups_z);
    return PTK_OK;
}

// Kernel parameters from the push constants and the bound, validated scene (ctx->dirty must be false).
static int fill_params(ptk_ctx* ctx, const void* push128, PtkParams& P) {
    Object* img = find(ctx, ctx->image, Kind::Image);
    if (!img) return fail(ctx, PTK_E_STATE, "no image bound at (0,0)");
    memset(&P, 0, sizeof P);
    memcpy(&P.push, push128, 128);
    if (P.push.max_default_depth < 0 || P.push.max_default_depth > 4096 || P.push.max_refraction_bounces < 0 || P.push.max_refraction_bounces > 4096)
        return fail(ctx, PTK_E_INVALID, "ray depth limits out of range");
    Object* nodes = bound_buffer(ctx, PTK_SET_BVH, 0);
    Object* ids = bound_buffer(ctx, PTK_SET_BVH, 1);
    Object* mats = bound_buffer(ctx, PTK_SET_OBJECT, 0);
    Object* sph = bound_buffer(ctx, PTK_SET_OBJECT, 1);
    Object* pln = bound_buffer(ctx, PTK_SET_OBJECT, 2);
    Object* tri = bound_buffer(ctx, PTK_SET_OBJECT, 3);
    Object* xf = bound_buffer(ctx, PTK_SET_TRANSFORM, 0);
    if (!nodes || !ids || !mats || !sph || !pln || !tri || !xf)
        return fail(ctx, PTK_E_STATE, "dispatch needs buffers at (2,0..3), (3,0..1) and (6,0)");
    P.nodes = (const float4*)nodes->dev;
    P.object_ids = (const int*)ids->dev;
    P.materials = (const float4*)mats->dev;
    P.spheres = (const float4*)sph->dev;
    P.planes = (const float4*)pln->dev;
    P.triangles = (const float4*)tri->dev;
    P.transforms = (const float4*)xf->dev;
    P.inv_transforms = ctx->d_inv_xf;
    P.live_planes = ctx->d_live_planes;
    P.n_live_planes = ctx->n_live_planes;
    P.n_arg_planes = -1;
    if (ctx->h_live_planes.size() <= PTK_ARG_PLANES) {
        P.n_arg_planes = (int)ctx->h_live_planes.size();
        for (int i = 0; i < P.n_arg_planes; i++) {
            P.arg_plane_index[i] = ctx->h_live_planes[(size_t)i];
            memcpy(&P.arg_plane[i], pln->shadow.data() + (size_t)ctx->h_live_planes[(size_t)i] * 32, 16);
        }
    }
    P.n_spheres = (int)(sph->nbytes / 32);
    P.n_planes = (int)(pln->nbytes / 32);
    P.n_triangles = (int)(tri->nbytes / 80);
    P.image = (float4*)img->dev;
    P.img_w = img->width;
    P.img_h = img->height;
    P.textures = ctx->d_textures;
    P.n_textures = ctx->n_textures;
    P.n_proc = ctx->n_proc;
    memcpy(P.proc_builtin, ctx->proc_builtin, sizeof P.proc_builtin);
    P.refraction_mode = ctx->refraction_mode;
    P.use_emission = ctx->use_emission;
    if (ctx->opt_triangle_uvs) {
        Object* uv = bound_buffer(ctx, PTK_SET_TRIANGLE, 1);
        if (!uv) return fail(ctx, PTK_E_STATE, "PTK_OPT_TRIANGLE_UVS is on but no buffer is bound at (4,1)");
        P.triangle_uvs = (const float*)uv->dev;
        P.n_triangle_uvs = (int)(uv->nbytes / 24);  // triangles beyond the table keep their barycentric uv
    }
    return PTK_OK;
}

// The compiled node blocks and leaf records of the bound scene (PtkWavefront), for the block traversal.
static void fill_blocks(const ptk_ctx* ctx, PtkParams& P) {
    P.wf.blocks = ctx->d_blocks;
    P.wf.sockets = ctx->d_sockets;
    P.wf.root_min_x = ctx->root_box[0]; P.wf.root_min_y = ctx->root_box[1]; P.wf.root_min_z = ctx->root_box[2];
    P.wf.root_max_x = ctx->root_box[3]; P.wf.root_max_y = ctx->root_box[4]; P.wf.root_max_z = ctx->root_box[5];
    P.wf.root_entry = ctx->root_entry;
    P.wf.n_top = ctx->n_top;
    P.wf.n_chain = ctx->n_chain;
    memcpy(P.wf.chain, ctx->chain, sizeof P.wf.chain);
    P.wf.chain_xf = -1;
    for (int k = 0; k < ctx->n_chain; k++) if (ctx->chain[k].kind == 1) P.wf.chain_xf = ctx->chain[k].xf;
    P.wf.n_blocks = ctx->n_blocks; P.wf.n_sockets = ctx->n_sockets; P.wf.n_records = ctx->n_records;
    // leaves longer than this are shared with a helper lane: worth its set-up for triangle tests, not for sphere tests
    P.wf.leaf_split = ctx->leaves_hold_triangles ? ctx->leaf_split : 255;
    P.wf.leaf_redo_all = ctx->leaf_redo_all;
    P.wf.leaf_records = ctx->d_leaf_records;
}

// Whether the persistent path kernel (and, within its bounce limit, the wavefront pipeline) covers a
// dispatch: plain render mode over the BVH, untraced, no early-t variant, at least one bounce.
static bool persistent_covers(const ptk_ctx* ctx, const PtkPush& push, bool trace) {
    const int fl = push.flags;
    return ctx->wf_ok && ctx->pp_wgs_per_cu > 0 && fl >= 0 && (fl & 1) && push.render_mode != 1 && push.render_mode != 2 && !trace &&
           !ctx->early_t_break && push.max_default_depth > 0;
}

int ptk_dispatch(ptk_ctx* ctx, const void* push128, int groups_x, int groups_y, int groups_z) {
    if (!ctx) return PTK_E_INVALID;
    { int r = check_dispatch_args(ctx, push128, groups_x, groups_y, groups_z); if (r != PTK_OK) return r; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    Object* img = find(ctx, ctx->image, Kind::Image);
    if (!img) return fail(ctx, PTK_E_STATE, "no image bound at (0,0)");
    if (ctx->dirty) {
        // (validation makes the leaf records on the device, from the bound buffers as the pending edits leave them)
        { int r = flush_updates(ctx); if (r != PTK_OK) return r; }
        { int r = validate_scene(ctx); if (r != PTK_OK) return r; }
    }
    if (ctx->textures_dirty) { int r = refresh_textures(ctx); if (r != PTK_OK) return r; }
    { int r = flush_updates(ctx); if (r != PTK_OK) return r; }

    PtkParams P;
    { int r = fill_params(ctx, push128, P); if (r != PTK_OK) return r; }

    // traversal stack: an LDS window per lane plus a global spill column for deeper entries
    const int need = ctx->stack_need;
    P.lds_cap = std::min(need, ctx->lds_window);
    const size_t lanes = (size_t)groups_x * groups_y * 64;
    if (lanes > ((size_t)1 << 30)) return fail(ctx, PTK_E_UNSUPPORTED, "dispatch of %zu invocations; at most 2^30 per dispatch", lanes);
    P.spill_stride = lanes;
    const int spill_entries = need - P.lds_cap;
    if (spill_entries > 0) {
        size_t ints = (size_t)spill_entries * lanes;
        if (ints > ctx->spill_ints) {
            if (ctx->d_spill) { (void)hipFree(ctx->d_spill); ctx->d_spill = nullptr; }
            if (ctx->d_spill_t) { (void)hipFree(ctx->d_spill_t); ctx->d_spill_t = nullptr; }
            ctx->spill_ints = 0;
            HIP_TRY(ctx, hipMalloc((void**)&ctx->d_spill, ints * 4));
            if (ctx->early_t_break) HIP_TRY(ctx, hipMalloc((void**)&ctx->d_spill_t, ints * 4));
            ctx->spill_ints = ints;
        } else if (ctx->early_t_break && !ctx->d_spill_t) {
            HIP_TRY(ctx, hipMalloc((void**)&ctx->d_spill_t, ctx->spill_ints * 4));
        }
        P.spill = ctx->d_spill;
        P.spill_t = ctx->d_spill_t;
    }
    // the block traversal (untraced, no early-t) writes rejected children to one spare row past the window
    size_t lds_bytes = ctx->early_t_break ? (size_t)P.lds_cap * 64 * 4 * 2 : (size_t)(P.lds_cap + 1) * 64 * 4;

    const int trace = ctx->trace_flags != 0;
    if (trace) {
        size_t pixels = (size_t)img->width * img->height;
        if (pixels != ctx->trace_pixels) {
            if (ctx->d_trace_primary) { (void)hipFree(ctx->d_trace_primary); ctx->d_trace_primary = nullptr; }
            if (ctx->d_trace_counters) { (void)hipFree(ctx->d_trace_counters); ctx->d_trace_counters = nullptr; }
            ctx->trace_pixels = 0;
        }
        if ((ctx->trace_flags & PTK_TRACE_PRIMARY_ID) && !ctx->d_trace_primary) {
            HIP_TRY(ctx, hipMalloc((void**)&ctx->d_trace_primary, pixels * 4));
            HIP_TRY(ctx, hipMemsetAsync(ctx->d_trace_primary, 0xFF, pixels * 4, ctx->stream));
        }
        if ((ctx->trace_flags & PTK_TRACE_COUNTERS) && !ctx->d_trace_counters) {
            HIP_TRY(ctx, hipMalloc((void**)&ctx->d_trace_counters, pixels * 4 * PTK_NCOUNTERS));
            HIP_TRY(ctx, hipMemsetAsync(ctx->d_trace_counters, 0, pixels * 4 * PTK_NCOUNTERS, ctx->stream));
        }
        ctx->trace_pixels = pixels;
        P.trace_primary = (ctx->trace_flags & PTK_TRACE_PRIMARY_ID) ? ctx->d_trace_primary : nullptr;
        P.trace_counters = (ctx->trace_flags & PTK_TRACE_COUNTERS) ? ctx->d_trace_counters : nullptr;
    }

    // the compiled node blocks and leaf records: read by the wavefront pipeline and by the untraced
    // one-kernel renderer (null: that renderer walks the bound node buffer itself)
    if (ctx->wf_ok && !trace && !ctx->early_t_break) fill_blocks(ctx, P);

    const int slot = (int)(ctx->dispatch_count % ptk_ctx::kTimingRing);

    // Which schedule renders this dispatch.  The wavefront pipeline and the persistent path kernel cover
    // the plain render mode over the BVH; BVH/normal views, brute force, the early-t variant and traced
    // dispatches use the one-kernel renderer.  All three give the same image.
    // most bounces any path can take (ray_tracer.comp:1572): the depth limit plus the refraction bounces a path can
    // collect, which are none where no primitive that can be hit has a material that refracts or is passed through
    const int rb_max = ctx->may_refract ? std::max(P.push.max_refraction_bounces - 1, 0) : 0;
    const int n_iter = P.push.max_default_depth + rb_max;
    const bool pp_eligible = persistent_covers(ctx, P.push, trace != 0);
    const bool wf_eligible = pp_eligible && ctx->wf_wgs_per_cu > 0 && n_iter <= PTK_WF_MAX_ITER;  // the pipeline launches once per bounce
    int choice = 0;  // 0 one-kernel renderer, 1 wavefront pipeline, 2 persistent path kernel
    if (ctx->renderer == PTK_RENDERER_WAVEFRONT) {
        if (!wf_eligible)
            return fail(ctx, PTK_E_UNSUPPORTED, "the wavefront renderer does not cover this dispatch (render mode, flags, tracing, early-t, depth > %d or nested mesh sockets)", PTK_WF_MAX_ITER);
        choice = 1;
    } else if (ctx->renderer == PTK_RENDERER_PERSISTENT) {
        if (!pp_eligible)
            return fail(ctx, PTK_E_UNSUPPORTED, "the persistent renderer does not cover this dispatch (render mode, flags, tracing, early-t, depth 0 or nested mesh sockets)");
        choice = 2;
    } else if (ctx->renderer == PTK_RENDERER_AUTO && pp_eligible) {
        const bool eligible[3] = {true, wf_eligible, true};
        ptk_ctx::AutoChoice& ac = ctx->auto_choice;
        unsigned long long key = ctx->scene_version * 0x9E3779B97F4A7C15ull;
        for (unsigned long long v : {(unsigned long long)groups_x, (unsigned long long)groups_y, (unsigned long long)n_iter,
                                     (unsigned long long)P.push.max_default_depth, (unsigned long long)ctx->image,
                                     (unsigned long long)ctx->refraction_mode, (unsigned long long)ctx->use_emission})
            key = (key ^ v) * 0x100000001B3ull;
        if (key != ac.key) ac = ptk_ctx::AutoChoice(), ac.key = key;
        for (int k = 0; k < 3; k++) {  // collect timings that have completed (never blocks)
            if (ac.probe_dispatch[k] < 0) continue;
            if (ctx->dispatch_count - ac.probe_dispatch[k] >= ptk_ctx::kTimingRing) { ac.probe_dispatch[k] = -1; continue; }
            const int ps = (int)(ac.probe_dispatch[k] % ptk_ctx::kTimingRing);
            if (hipEventQuery(ctx->ev1[ps]) != hipSuccess) continue;
            float ms = 0.0f;
            if (hipEventElapsedTime(&ms, ctx->ev0[ps], ctx->ev1[ps]) == hipSuccess) {
                ac.best_ms[k] = ac.samples[k] == 0 ? ms : std::min(ac.best_ms[k], ms);
                ac.samples[k]++;
            }
            ac.probe_dispatch[k] = -1;
        }
        const int kProbes = 2;
        int probing = -1;  // this dispatch times renderer `probing`
        bool settled = true;
        for (int k = 0; k < 3; k++) {
            if (!eligible[k] || ac.samples[k] >= kProbes) continue;
            settled = false;
            if (probing < 0 && ac.probe_dispatch[k] < 0) probing = k;
        }
        if (probing >= 0) {
            choice = probing;
            ac.probe_dispatch[probing] = ctx->dispatch_count;
        } else if (settled) {
            for (int k = 1; k < 3; k++)
                if (eligible[k] && ac.best_ms[k] < ac.best_ms[choice]) choice = k;
        } else {
            choice = 2;  // timings still in flight: the persistent kernel meanwhile
        }
    }

    ctx->last_choice = choice;
    if (choice == 0) {
        ctx->wf_iters[slot] = 0;
        HIP_TRY(ctx, hipEventRecord(ctx->ev0[slot], ctx->stream));
        if (ctx->tile_ordering && !trace && (size_t)groups_x * groups_y >= 8192) {
            // long tiles first, by what the last dispatch of this grid measured (the order kernel is part of the dispatch's time)
            const size_t n_tiles = (size_t)groups_x * groups_y;
            if (n_tiles > ctx->tile_capacity) {
                if (ctx->d_tile_cost) { (void)hipFree(ctx->d_tile_cost); ctx->d_tile_cost = nullptr; }
                if (ctx->d_tile_order) { (void)hipFree(ctx->d_tile_order); ctx->d_tile_order = nullptr; }
                ctx->tile_capacity = 0;
                ctx->tile_gx = ctx->tile_gy = 0;
                HIP_TRY(ctx, hipMalloc((void**)&ctx->d_tile_cost, n_tiles * 4 * 3));  // (measurement builds keep two more words a tile)
                HIP_TRY(ctx, hipMalloc((void**)&ctx->d_tile_order, n_tiles * 4));
                ctx->tile_capacity = n_tiles;
            }
            if (ctx->tile_gx == groups_x && ctx->tile_gy == groups_y) {
                // the order is made anew from the last dispatch's costs every eighth dispatch of this grid (the sort is
                // one workgroup's work, 0.1 ms for 32 400 tiles; a progressive render's tiles keep their costs) and kept between
                if (ctx->tile_order_age % 8 == 0)
                    HIP_TRY(ctx, ptk_launch_tile_order(ctx->d_tile_cost, (uint32_t)n_tiles, ctx->d_tile_order, ctx->stream));
                ctx->tile_order_age++;
                P.tile_order = ctx->d_tile_order;
            } else {
                ctx->tile_order_age = 0;
            }
            P.tile_cost = ctx->d_tile_cost;
            ctx->tile_gx = groups_x;
            ctx->tile_gy = groups_y;
        }
        HIP_TRY(ctx, ptk_launch_render(&P, groups_x, groups_y, trace, ctx->early_t_break, lds_bytes, ctx->stream));
        HIP_TRY(ctx, hipEventRecord(ctx->ev1[slot], ctx->stream));
        ctx->dispatch_count++;
        return PTK_OK;
    }

    if (choice == 2) {
        const size_t pp_lds = (size_t)(P.lds_cap + 1) * 64 * 4;  // one spare row for rejected children
        if (ctx->pp_wgs_per_cu <= 0) return fail(ctx, PTK_E_HIP, "the path kernel does not fit on a compute unit");
        const size_t wg_lanes = (size_t)ctx->pp_wg_waves * 64;
        const int n_wgs = (int)std::min<size_t>((size_t)ctx->pp_wgs_per_cu * std::max(ctx->n_cus, 1), (lanes + wg_lanes - 1) / wg_lanes);
        const int n_waves = n_wgs * ctx->pp_wg_waves;
        if (!ctx->d_pixel_counter) HIP_TRY(ctx, hipMalloc((void**)&ctx->d_pixel_counter, 256));
        P.wf.heads = ctx->d_pixel_counter;
        P.wf.n_slots = (uint32_t)lanes;
        P.wf.tiles_x = groups_x;
        P.spill = nullptr;
        P.spill_t = nullptr;
        P.spill_stride = (size_t)n_waves * 64;
        if (spill_entries > 0) {
            size_t ints = (size_t)spill_entries * n_waves * 64;
            if (ints > ctx->wf_spill_ints) {
                if (ctx->d_wf_spill) { (void)hipFree(ctx->d_wf_spill); ctx->d_wf_spill = nullptr; }
                ctx->wf_spill_ints = 0;
                HIP_TRY(ctx, hipMalloc((void**)&ctx->d_wf_spill, ints * 4));
                ctx->wf_spill_ints = ints;
            }
            P.spill = ctx->d_wf_spill;
        }
        ctx->wf_iters[slot] = 0;
        HIP_TRY(ctx, hipEventRecord(ctx->ev0[slot], ctx->stream));
        HIP_TRY(ctx, ptk_launch_persistent(&P, pp_lds, n_wgs, ctx->pp_wg_waves, ctx->stream));
        HIP_TRY(ctx, hipEventRecord(ctx->ev1[slot], ctx->stream));
        ctx->dispatch_count++;
        return PTK_OK;
    }

    // queues and path state for groups_x * groups_y * 64 pixel slots
    const size_t n_slots = lanes;
    if (n_slots > ctx->wf_slots) {
        if (ctx->d_wf) { (void)hipFree(ctx->d_wf); ctx->d_wf = nullptr; }
        ctx->wf_slots = 0;
        const size_t per_slot = 2 * 32 + 2 * 32 + 16 + 8 + 40 + 2 * PTK_WF_PREP_PLANES * 4;  // rays x2, state x2, hits, hit uv, inside arrays, started searches x2
        HIP_TRY(ctx, hipMalloc(&ctx->d_wf, n_slots * per_slot + PTK_WF_HEADER_BYTES));
        ctx->wf_slots = n_slots;
    }
    {
        uint8_t* base = (uint8_t*)ctx->d_wf;
        const size_t cap = ctx->wf_slots;
        P.wf.counts = (uint32_t*)base;                       // counts, parked counts, heads: zeroed per dispatch
        P.wf.n_resume = P.wf.counts + (PTK_WF_MAX_ITER + 1);
        P.wf.heads = P.wf.counts + 192;                      // (PTK_WF_MAX_ITER + 1) * PTK_WF_MAX_PARTS words
        static_assert((192 + (PTK_WF_MAX_ITER + 1) * PTK_WF_MAX_PARTS) * 4 <= PTK_WF_HEADER_BYTES, "queue header too small");
        static_assert(2 * (PTK_WF_MAX_ITER + 1) + 1 <= 192, "counts and parked counts overlap the heads");
        P.wf.n_parts = (uint32_t)ctx->wf_parts;
        base += PTK_WF_HEADER_BYTES;
        P.wf.rays[0] = (float4*)base; base += cap * 32;
        P.wf.rays[1] = (float4*)base; base += cap * 32;
        P.wf.state[0] = (float4*)base; base += cap * 32;
        P.wf.state[1] = (float4*)base; base += cap * 32;
        P.wf.hits = (int4*)base; base += cap * 16;
        P.wf.hit_uv = (float2*)base; base += cap * 8;
        P.wf.inside = (int*)base; base += cap * 40;
        P.wf.prep[0] = (uint32_t*)base; base += cap * PTK_WF_PREP_PLANES * 4;
        P.wf.prep[1] = (uint32_t*)base;
        P.wf.prep_cap = (uint32_t)cap;
        P.wf.n_slots = (uint32_t)n_slots;
        P.wf.tiles_x = groups_x;
    }
    // persistent traversal workgroups: as many as fit on the device at once
    const size_t wf_lds = (size_t)ctx->n_top * PTK_TOP_BLOCK_STRIDE + (size_t)PTK_WF_WG_WAVES * (P.lds_cap + 1) * 64 * 4 + PTK_WF_WG_FLAG_BYTES;  // one spare row per wave
    if (ctx->wf_wgs_per_cu <= 0) return fail(ctx, PTK_E_HIP, "the traversal kernel does not fit on a compute unit");
    const size_t wg_rays = (size_t)PTK_WF_WG_WAVES * 64;
    const int n_wgs = (int)std::min<size_t>((size_t)ctx->wf_wgs_per_cu * std::max(ctx->n_cus, 1), (n_slots + wg_rays - 1) / wg_rays);
    const size_t n_waves = (size_t)n_wgs * PTK_WF_WG_WAVES;
    P.spill = nullptr;
    P.spill_t = nullptr;
    P.spill_stride = n_waves * 64;
    if (spill_entries > 0) {
        size_t ints = (size_t)spill_entries * n_waves * 64;
        if (ints > ctx->wf_spill_ints) {
            if (ctx->d_wf_spill) { (void)hipFree(ctx->d_wf_spill); ctx->d_wf_spill = nullptr; }
            ctx->wf_spill_ints = 0;
            HIP_TRY(ctx, hipMalloc((void**)&ctx->d_wf_spill, ints * 4));
            ctx->wf_spill_ints = ints;
        }
        P.spill = ctx->d_wf_spill;
    }
    // carry-over of unfinished searches between launches: room for every lane in flight, for both queue parities
    int n_launches = n_iter;
    P.wf.carry[0] = P.wf.carry[1] = nullptr;
    P.wf.carry_cap = 0;
    P.wf.max_lag = ctx->wf_max_lag;
    P.wf.park_below = ctx->wf_park_below;
    // What a workgroup has taken no other can take, so small claims even out the end of a launch; but the device-wide
    // counter serves one atomic at a time (~12 ns), which is what limits a launch over a tree that sits in LDS as a whole.
    P.wf.wg_claim = ctx->wf_wg_claim > 0 ? (uint32_t)ctx->wf_wg_claim : (ctx->n_blocks > ctx->n_top ? 256u : 512u);
    // (not for a tree that sits in LDS as a whole: its searches are short, and parking costs each wave a
    // device-wide atomic and the step two more launches)
    if (ctx->wf_carry && n_iter > 1 && n_iter + ctx->wf_max_lag <= PTK_WF_MAX_ITER && ctx->n_blocks > ctx->n_top) {
        const size_t cap = n_waves * 64;
        const size_t bytes = (size_t)(PTK_WF_CARRY_PLANES + P.lds_cap) * cap * 4;
        if (2 * bytes > ctx->wf_carry_bytes) {
            if (ctx->d_wf_carry) { (void)hipFree(ctx->d_wf_carry); ctx->d_wf_carry = nullptr; }
            ctx->wf_carry_bytes = 0;
            HIP_TRY(ctx, hipMalloc(&ctx->d_wf_carry, 2 * bytes));
            ctx->wf_carry_bytes = 2 * bytes;
        }
        P.wf.carry[0] = (uint32_t*)ctx->d_wf_carry;
        P.wf.carry[1] = (uint32_t*)((uint8_t*)ctx->d_wf_carry + bytes);
        P.wf.carry_cap = (uint32_t)cap;
        n_launches = n_iter + ctx->wf_max_lag;  // a parked path runs at most this many launches behind its bounce number
    }
    const bool timed = ctx->opt_launch_timing > 0 && (ctx->wf_dispatches++ % ctx->opt_launch_timing) == 0;
    if (timed && !ctx->wf_ev[slot]) {
        ctx->wf_ev[slot] = new hipEvent_t[2 * PTK_WF_MAX_ITER];
        for (int k = 0; k < 2 * PTK_WF_MAX_ITER; k++) HIP_TRY(ctx, hipEventCreate(&ctx->wf_ev[slot][k]));
    }
    ctx->wf_iters[slot] = n_launches;
    HIP_TRY(ctx, hipEventRecord(ctx->ev0[slot], ctx->stream));
    ctx->wf_timed[slot] = timed;
    HIP_TRY(ctx, ptk_launch_wavefront(&P, groups_x, groups_y, n_launches, wf_lds, n_wgs, ctx->wf_timed[slot] ? ctx->wf_ev[slot] : nullptr, ctx->stream));
    HIP_TRY(ctx, hipEventRecord(ctx->ev1[slot], ctx->stream));
    ctx->dispatch_count++;
    return PTK_OK;
}

// ptk_dispatch for callers that must not allocate: the persistent path kernel with its pixel counter and
// stack-spill column in the library's static device array instead of hipMalloc'ed buffers.  Calls no
// hipMalloc / hipFree / allocating routine: anything that would need one is an error instead.
int ptk_dispatch_noalloc(ptk_ctx* ctx, const void* push128, int groups_x, int groups_y, int groups_z) {
    if (!ctx) return PTK_E_INVALID;
    { int r = check_dispatch_args(ctx, push128, groups_x, groups_y, groups_z); if (r != PTK_OK) return r; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (ctx->dirty || ctx->textures_dirty)
        return fail(ctx, PTK_E_STATE, "the bound scene changed since the last ptk_dispatch: its derived tables are rebuilt (and allocated) "
                                      "there; call ptk_dispatch once, then ptk_dispatch_noalloc");
    { int r = flush_updates(ctx); if (r != PTK_OK) return r; }  // patched edits: staged through preallocated blocks
    PtkParams P;
    { int r = fill_params(ctx, push128, P); if (r != PTK_OK) return r; }
    if (ctx->trace_flags != 0 || !persistent_covers(ctx, P.push, false))
        return fail(ctx, PTK_E_UNSUPPORTED, "ptk_dispatch_noalloc covers the plain render mode over the BVH only (no tracing, debug views, "
                                            "brute force, early-t variant, depth 0 or nested mesh sockets)");
    const size_t lanes = (size_t)groups_x * groups_y * 64;
    if (lanes > ((size_t)1 << 30)) return fail(ctx, PTK_E_UNSUPPORTED, "dispatch of %zu invocations; at most 2^30 per dispatch", lanes);
    uint8_t* const scratch_ptr = (uint8_t*)ptk_persistent_scratch(ctx->device);
    if (!scratch_ptr) return fail(ctx, PTK_E_STATE, "the static scratch array was not resolved on device %d", ctx->device);
    fill_blocks(ctx, P);
    P.lds_cap = std::min(ctx->stack_need, ctx->lds_window);
    const size_t pp_lds = (size_t)(P.lds_cap + 1) * 64 * 4;
    if (ctx->pp_wgs_per_cu <= 0) return fail(ctx, PTK_E_HIP, "the path kernel does not fit on a compute unit");
    const size_t wg_waves = (size_t)ctx->pp_wg_waves;
    size_t n_wgs = std::min<size_t>((size_t)ctx->pp_wgs_per_cu * std::max(ctx->n_cus, 1), (lanes + wg_waves * 64 - 1) / (wg_waves * 64));
    size_t n_waves = n_wgs * wg_waves;
    // stack entries past the LDS window: one int per entry and lane, [entry][wave * 64 + lane], behind the counter line
    const int spill_entries = ctx->stack_need - P.lds_cap;
    const size_t spill_room = PTK_NOALLOC_SCRATCH_BYTES - 256;
    if (spill_entries > 0) {
        // fewer persistent waves need a smaller column (the image does not depend on their number), but
        // below one wave per compute unit the dispatch is refused rather than crawled through
        const size_t fit = spill_room / ((size_t)spill_entries * 64 * 4);
        const size_t floor_waves = std::max(wg_waves, std::min<size_t>((size_t)std::max(ctx->n_cus, 1), (lanes + 63) / 64));
        if (fit < floor_waves)
            return fail(ctx, PTK_E_UNSUPPORTED, "the bound BVH can need %d traversal-stack entries per ray; the static scratch of %zu bytes "
                                                "holds at most %zu beyond the %d kept on chip for this dispatch: use ptk_dispatch",
                        ctx->stack_need, (size_t)PTK_NOALLOC_SCRATCH_BYTES, spill_room / (floor_waves * 64 * 4), P.lds_cap);
        n_wgs = std::min(n_wgs, fit / wg_waves);  // whole workgroups
        n_waves = n_wgs * wg_waves;
        P.spill = (int*)(scratch_ptr + 256);
    }
    P.spill_stride = n_waves * 64;
    P.wf.heads = (uint32_t*)scratch_ptr;  // cleared by the launch below: the array keeps the last call's contents
    P.wf.n_slots = (uint32_t)lanes;
    P.wf.tiles_x = groups_x;
    const int slot = (int)(ctx->dispatch_count % ptk_ctx::kTimingRing);
    ctx->wf_iters[slot] = 0;
    ctx->last_choice = 2;
    HIP_TRY(ctx, hipEventRecord(ctx->ev0[slot], ctx->stream));
    HIP_TRY(ctx, ptk_launch_persistent_static(&P, pp_lds, (int)n_wgs, ctx->pp_wg_waves, ctx->device, ctx->stream));
    HIP_TRY(ctx, hipEventRecord(ctx->ev1[slot], ctx->stream));
    ctx->dispatch_count++;
    return PTK_OK;
}

#ifdef PTK_WF_STATS
// Measurement builds only (not declared in ptk.h): reads and clears the traversal kernel's round statistics.
extern "C" hipError_t ptk_wf_stats_fetch(unsigned long long* out, int n, hipStream_t stream);
int ptk_dev_wf_stats(ptk_ctx* ctx, unsigned long long* out, int n) {
    if (!ctx || !out) return PTK_E_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, ptk_wf_stats_fetch(out, n, ctx->stream));
    return PTK_OK;
}
#endif

#ifdef PTK_WF_STATS
extern "C" hipError_t ptk_pp_stats_fetch(unsigned long long* out, hipStream_t stream);
int ptk_dev_pp_stats(ptk_ctx* ctx, unsigned long long* out8) {
    if (!ctx || !out8) return PTK_E_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, ptk_pp_stats_fetch(out8, ctx->stream));
    return PTK_OK;
}
#endif

#ifdef PTK_WF_ENDS
extern "C" hipError_t ptk_wf_ends_fetch(unsigned long long* out, size_t n, hipStream_t stream);
int ptk_dev_wf_ends(ptk_ctx* ctx, unsigned long long* out, size_t n) {
    if (!ctx || !out) return PTK_E_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, ptk_wf_ends_fetch(out, n, ctx->stream));
    return PTK_OK;
}
#endif

// Development aid (not declared in ptk.h): the structures derived from the bound scene, for tests that compare the
// host's and the device's derivation.  kind 0: blocks (64 words each), 1: socket table (2 words a row), 2: node ->
// child slot, 3: triangle -> id slot, 4: sphere -> id slot, 5: {stack need, top blocks, blocks, sockets, block
// traversal covers the scene, levels the device pass walked (0: host), root entry, paths may refract, leaves hold
// triangles, device derivations so far, host microseconds of the last validation}, 6: leaf records (12 words per id slot).
// *n = words there are; at most `capacity` are copied.
extern "C" int ptk_dev_read_derived(ptk_ctx* ctx, int kind, uint32_t* out, size_t capacity, size_t* n) {
    if (!ctx || !n || ctx->dirty) return PTK_E_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    Object* nodes = bound_buffer(ctx, PTK_SET_BVH, 0);
    Object* tri = bound_buffer(ctx, PTK_SET_OBJECT, 3);
    Object* sph = bound_buffer(ctx, PTK_SET_OBJECT, 1);
    if (!nodes || !tri || !sph) return PTK_E_STATE;
    const void* dev = nullptr;
    const void* host = nullptr;
    size_t words = 0;
    uint32_t info[11];
    switch (kind) {
    case 0: words = (size_t)ctx->n_blocks * 64; dev = ctx->d_blocks; break;
    case 1: words = (size_t)ctx->n_sockets * 2; dev = ctx->d_sockets; break;
    case 2: words = nodes->nbytes / 32; if (ctx->mirrors_partial) dev = ctx->d_node_slot; else host = ctx->node_slot.data(); break;
    case 3: words = tri->nbytes / 80; if (ctx->mirrors_partial) dev = ctx->d_tri_slot; else host = ctx->tri_slot.data(); break;
    case 4: words = sph->nbytes / 32; if (ctx->mirrors_partial) dev = ctx->d_sph_slot; else host = ctx->sph_slot.data(); break;
    case 5:
        info[0] = (uint32_t)ctx->stack_need; info[1] = ctx->n_top; info[2] = ctx->n_blocks; info[3] = ctx->n_sockets;
        info[4] = ctx->wf_ok ? 1u : 0u; info[5] = (uint32_t)ctx->derive_levels; info[6] = ctx->tree_entry;
        info[7] = ctx->may_refract ? 1u : 0u; info[8] = ctx->leaves_hold_triangles ? 1u : 0u; info[9] = (uint32_t)ctx->device_derivations;
        info[10] = (uint32_t)(ctx->last_validate_ms * 1000.0);
        words = 11; host = info; break;
    case 6: {  // the leaf records, 12 words per object-id slot (pending patches are flushed first, as a dispatch would)
        { int r = flush_updates(ctx); if (r != PTK_OK) return r; }
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        Object* ids = bound_buffer(ctx, PTK_SET_BVH, 1);
        if (!ids || !ctx->d_leaf_records) return PTK_E_STATE;
        words = std::min(ids->nbytes / 4, ctx->d_leaf_records_slots) * 12; dev = ctx->d_leaf_records; break;
    }
    default: return PTK_E_INVALID;
    }
    *n = words;
    const size_t copy = std::min(words, capacity);
    if (copy && out) {
        if (dev) HIP_TRY(ctx, hipMemcpy(out, dev, copy * 4, hipMemcpyDeviceToHost));
        else memcpy(out, host, copy * 4);
    }
    return PTK_OK;
}

// Development aid (not declared in ptk.h): the pipeline's queue counters after its last dispatch: 65 words of counts
// (rays queued for launch i), 65 of heads (rays handed out), then n_resume (searches parked for launch i).
extern "C" int ptk_dev_wf_counts(ptk_ctx* ctx, uint32_t* out256) {
    if (!ctx || !out256 || !ctx->d_wf) return PTK_E_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    uint32_t h[PTK_WF_HEADER_BYTES / 4];
    HIP_TRY(ctx, hipMemcpy(h, ctx->d_wf, sizeof h, hipMemcpyDeviceToHost));
    memset(out256, 0, 1024);
    for (int i = 0; i <= PTK_WF_MAX_ITER; i++) {
        out256[i] = h[i];
        out256[130 + i] = h[PTK_WF_MAX_ITER + 1 + i];
        for (int p = 0; p < PTK_WF_MAX_PARTS; p++) out256[65 + i] += h[192 + i * PTK_WF_MAX_PARTS + p];
    }
    return PTK_OK;
}

// Development aid (not declared in ptk.h): the tile kernel's per-tile durations of its last dispatch, in ticks of
// the 100 MHz wall clock; *n = number of tiles (0 when no dispatch recorded any).
extern "C" int ptk_dev_tile_costs(ptk_ctx* ctx, uint32_t* out, size_t capacity, size_t* n) {
    if (!ctx || !n) return PTK_E_INVALID;
    const size_t tiles = ctx->d_tile_cost ? (size_t)ctx->tile_gx * ctx->tile_gy : 0;
    *n = tiles;
    if (!out || capacity < tiles || tiles == 0) return PTK_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(out, ctx->d_tile_cost, (capacity >= 3 * tiles ? 3 : 1) * tiles * 4, hipMemcpyDeviceToHost));
    return PTK_OK;
}

int ptk_update_stats(ptk_ctx* ctx, long long* full_validations, long long* patched_updates) {
    if (!ctx || !full_validations || !patched_updates) return PTK_E_INVALID;
    *full_validations = ctx->full_validations;
    *patched_updates = ctx->patched_updates;
    return PTK_OK;
}

int ptk_set_option(ptk_ctx* ctx, int option, int value) {
    if (!ctx) return PTK_E_INVALID;
    if (option == PTK_OPT_TRIANGLE_UVS) { ctx->opt_triangle_uvs = value ? 1 : 0; return PTK_OK; }
    if (option == PTK_OPT_LAUNCH_TIMING) { ctx->opt_launch_timing = value > 0 ? value : 0; ctx->wf_dispatches = 0; return PTK_OK; }
    return fail(ctx, PTK_E_INVALID, "unknown option %d", option);
}

int ptk_set_renderer(ptk_ctx* ctx, int renderer) {
    if (!ctx) return PTK_E_INVALID;
    if (renderer != PTK_RENDERER_AUTO && renderer != PTK_RENDERER_SINGLE_KERNEL && renderer != PTK_RENDERER_WAVEFRONT &&
        renderer != PTK_RENDERER_PERSISTENT)
        return fail(ctx, PTK_E_INVALID, "unknown renderer %d", renderer);
    ctx->renderer = renderer;
    return PTK_OK;
}

int ptk_last_dispatch_renderer(ptk_ctx* ctx, int* renderer) {
    if (!ctx || !renderer) return PTK_E_INVALID;
    if (ctx->last_choice < 0) return fail(ctx, PTK_E_STATE, "no dispatch has run");
    *renderer = ctx->last_choice == 0 ? PTK_RENDERER_SINGLE_KERNEL : ctx->last_choice == 1 ? PTK_RENDERER_WAVEFRONT : PTK_RENDERER_PERSISTENT;
    return PTK_OK;
}

int ptk_dispatch_traverse_ms(ptk_ctx* ctx, int back, float* ms, int capacity, int* n_launches) {
    if (!ctx || !n_launches || back < 0 || back >= ptk_ctx::kTimingRing) return PTK_E_INVALID;
    if (ctx->dispatch_count <= back) return fail(ctx, PTK_E_STATE, "only %lld dispatches have run", ctx->dispatch_count);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int slot = (int)((ctx->dispatch_count - 1 - back) % ptk_ctx::kTimingRing);
    const int n = ctx->wf_iters[slot];
    *n_launches = n;
    if (n > 0 && (!ms || capacity < n)) return fail(ctx, PTK_E_INVALID, "room for %d durations needed", n);
    for (int k = 0; k < n; k++) {
        if (!ctx->wf_timed[slot]) { ms[k] = -1.0f; continue; }
        HIP_TRY(ctx, hipEventSynchronize(ctx->wf_ev[slot][2 * k + 1]));
        HIP_TRY(ctx, hipEventElapsedTime(&ms[k], ctx->wf_ev[slot][2 * k], ctx->wf_ev[slot][2 * k + 1]));
    }
    return PTK_OK;
}

int ptk_dispatch_history_ms(ptk_ctx* ctx, float* ms, int n) {
    if (!ctx || !ms || n <= 0) return PTK_E_INVALID;
    if (n > ptk_ctx::kTimingRing) return fail(ctx, PTK_E_INVALID, "at most %d dispatches are remembered", ptk_ctx::kTimingRing);
    if (ctx->dispatch_count < n) return fail(ctx, PTK_E_STATE, "only %lld dispatches have run", ctx->dispatch_count);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    for (int k = 0; k < n; k++) {  // ms[0] = oldest of the last n
        int slot = (int)((ctx->dispatch_count - n + k) % ptk_ctx::kTimingRing);
        HIP_TRY(ctx, hipEventSynchronize(ctx->ev1[slot]));
        HIP_TRY(ctx, hipEventElapsedTime(&ms[k], ctx->ev0[slot], ctx->ev1[slot]));
    }
    return PTK_OK;
}

int ptk_last_dispatch_ms(ptk_ctx* ctx, float* ms) { return ptk_dispatch_history_ms(ctx, ms, 1); }

int ptk_image_copy_device(ptk_ctx* ctx, uint64_t image, void* device_ptr, size_t nbytes, int to_image) {
    if (!ctx) return PTK_E_INVALID;
    Object* o = find(ctx, image, Kind::Image);
    if (!o) return fail(ctx, PTK_E_INVALID, "unknown image handle");
    if (!device_ptr || nbytes != o->nbytes) return fail(ctx, PTK_E_INVALID, "image copy needs exactly %zu bytes", o->nbytes);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (to_image) HIP_TRY(ctx, hipMemcpyAsync(o->dev, device_ptr, nbytes, hipMemcpyDeviceToDevice, ctx->stream));
    else HIP_TRY(ctx, hipMemcpyAsync(device_ptr, o->dev, nbytes, hipMemcpyDeviceToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PTK_OK;
}

// ---- mesh ingestion (device work in ptk_mesh.hip) ---------------------------------------------------
int ptk_mesh_to_triangles(ptk_ctx* ctx, const ptk_mesh_surface* sf, const float* transform12, int32_t material_index,
                          uint32_t texture_id, void* records80, float* aabbs6, float* uvs6, int32_t* n_triangles) {
    if (!ctx) return PTK_E_INVALID;
    if (!sf || !records80 || !n_triangles) return fail(ctx, PTK_E_INVALID, "surface, record buffer and count pointer are required");
    // Mesh.ArrayFormat: VERTEX 1<<0, COLOR 1<<3, TEX_UV 1<<4, TEX_UV2 1<<5, CUSTOM0..3 1<<6..9, INDEX 1<<12,
    // ARRAY_FLAG_COMPRESS_ATTRIBUTES 1<<29
    const uint64_t f = sf->format;
    if (!(f & 1ull)) return fail(ctx, PTK_E_UNSUPPORTED, "surface has no vertex array");
    if (!(f & (1ull << 12)) || sf->index_count <= 0)
        return fail(ctx, PTK_E_UNSUPPORTED, "surface has no index array (mesh_to_pttriangles walks ARRAY_INDEX)");
    if (f & (0xFull << 6)) return fail(ctx, PTK_E_UNSUPPORTED, "surfaces with custom attribute channels are not decoded");
    if (sf->vertex_count <= 0 || sf->index_count % 3 != 0) return fail(ctx, PTK_E_INVALID, "bad vertex / index count (%d, %d)", sf->vertex_count, sf->index_count);
    const bool compressed = (f & (1ull << 29)) != 0;
    const size_t pos_stride = compressed ? 8 : 12;
    const bool index32 = sf->vertex_count > 65536;
    if (!sf->vertex_data || sf->vertex_bytes < (size_t)sf->vertex_count * pos_stride)
        return fail(ctx, PTK_E_INVALID, "vertex block of %zu bytes is smaller than %d positions of %zu bytes", sf->vertex_bytes, sf->vertex_count, pos_stride);
    if (!sf->index_data || sf->index_bytes < (size_t)sf->index_count * (index32 ? 4 : 2))
        return fail(ctx, PTK_E_INVALID, "index block of %zu bytes is smaller than %d indices", sf->index_bytes, sf->index_count);
    const bool has_uv = (f & (1ull << 4)) != 0;
    const size_t uv_size = compressed ? 4 : 8;
    const size_t attr_stride = ((f & (1ull << 3)) ? 4 : 0) + (has_uv ? uv_size : 0) + ((f & (1ull << 5)) ? uv_size : 0);
    if (has_uv && uvs6 && (!sf->attribute_data || sf->attribute_bytes < (size_t)sf->vertex_count * attr_stride))
        return fail(ctx, PTK_E_INVALID, "attribute block of %zu bytes is smaller than %d vertices of %zu bytes", sf->attribute_bytes, sf->vertex_count, attr_stride);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const uint32_t n_tri = (uint32_t)(sf->index_count / 3);
    const size_t vb = (size_t)sf->vertex_count * pos_stride, ib = (size_t)sf->index_count * (index32 ? 4 : 2);
    const size_t ab = (has_uv && uvs6) ? (size_t)sf->vertex_count * attr_stride : 0;
    auto pad = [](size_t n) { return (n + 255) & ~(size_t)255; };
    const size_t o_v = 0, o_a = o_v + pad(vb), o_i = o_a + pad(ab), o_r = o_i + pad(ib), o_b = o_r + pad((size_t)n_tri * 80),
                 o_u = o_b + pad(aabbs6 ? (size_t)n_tri * 24 : 0), o_e = o_u + pad(uvs6 ? (size_t)n_tri * 24 : 0), total = o_e + 256;
    uint8_t* d = nullptr;
    HIP_TRY(ctx, hipMalloc((void**)&d, total));
    struct Guard { uint8_t* p; ~Guard() { (void)hipFree(p); } } guard{d};
    HIP_TRY(ctx, hipMemcpyAsync(d + o_v, sf->vertex_data, vb, hipMemcpyHostToDevice, ctx->stream));
    if (ab) HIP_TRY(ctx, hipMemcpyAsync(d + o_a, sf->attribute_data, ab, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(d + o_i, sf->index_data, ib, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(d + o_e, 0, 256, ctx->stream));
    PtkMeshJob J;
    memset(&J, 0, sizeof J);
    J.vertex_data = d + o_v; J.attribute_data = d + o_a; J.index_data = d + o_i;
    J.vertex_count = (uint32_t)sf->vertex_count; J.n_triangles = n_tri;
    J.compressed = compressed; J.index32 = index32; J.has_uv = has_uv && ab;
    J.uv_rescale = !(sf->uv_scale[0] == 0.0f && sf->uv_scale[1] == 0.0f && sf->uv_scale[2] == 0.0f && sf->uv_scale[3] == 0.0f);
    J.has_transform = transform12 != nullptr;
    J.attribute_stride = (uint32_t)attr_stride; J.uv_offset = (f & (1ull << 3)) ? 4u : 0u;
    memcpy(J.aabb, sf->aabb, sizeof J.aabb);
    memcpy(J.uv_scale, sf->uv_scale, sizeof J.uv_scale);
    if (transform12) memcpy(J.transform, transform12, sizeof J.transform);
    J.material_index = material_index; J.texture_id = texture_id;
    J.records = d + o_r; J.aabbs = aabbs6 ? (float*)(d + o_b) : nullptr; J.uvs = uvs6 ? (float*)(d + o_u) : nullptr;
    J.error_count = (uint32_t*)(d + o_e);
    HIP_TRY(ctx, ptk_launch_mesh(&J, ctx->stream));
    uint32_t bad = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&bad, d + o_e, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(records80, d + o_r, (size_t)n_tri * 80, hipMemcpyDeviceToHost, ctx->stream));
    if (aabbs6) HIP_TRY(ctx, hipMemcpyAsync(aabbs6, d + o_b, (size_t)n_tri * 24, hipMemcpyDeviceToHost, ctx->stream));
    if (uvs6) HIP_TRY(ctx, hipMemcpyAsync(uvs6, d + o_u, (size_t)n_tri * 24, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (bad) return fail(ctx, PTK_E_INVALID, "%u triangles index past the %d vertices of the surface", bad, sf->vertex_count);
    *n_triangles = (int32_t)n_tri;
    return PTK_OK;
}

// ---- BVH builder entry points (device work in ptk_bvh.hip) ---------------------------
uint64_t ptk_bvh_build(ptk_ctx* ctx, int order, int type, int n_objects, const float* aabbs6,
                       const int32_t* object_ids, int n_meshes, const uint64_t* mesh_bvhs,
                       const float* mesh_transforms12, const int32_t* mesh_indices) {
    if (!ctx) return 0;
    if (hipSetDevice(ctx->device) != hipSuccess) { fail(ctx, PTK_E_HIP, "hipSetDevice failed"); return 0; }
    if (order < 2 || order > 128) { fail(ctx, PTK_E_INVALID, "BVH order has to be >= 2 (and <= 128)"); return 0; }
    if (type < PTK_BVH_X_SORTED || type > PTK_BVH_SAH) { fail(ctx, PTK_E_INVALID, "unknown BVH type %d", type); return 0; }
    if (n_objects < 0 || n_meshes < 0 || n_objects + n_meshes <= 0) { fail(ctx, PTK_E_INVALID, "cannot create bvh without objects or meshes"); return 0; }
    if (n_objects > 0 && (!aabbs6 || !object_ids)) { fail(ctx, PTK_E_INVALID, "object arrays missing"); return 0; }
    std::vector<std::shared_ptr<PtkBvh>> subs;
    for (int m = 0; m < n_meshes; m++) {
        Object* o = mesh_bvhs ? find(ctx, mesh_bvhs[m], Kind::Bvh) : nullptr;
        if (!o || !o->bvh) { fail(ctx, PTK_E_INVALID, "mesh %d: unknown BVH handle", m); return 0; }
        if (o->bvh->adopted) { fail(ctx, PTK_E_INVALID, "mesh %d: BVH already merged into another tree", m); return 0; }
        subs.push_back(o->bvh);
    }
    if (n_meshes > 0 && (!mesh_transforms12 || !mesh_indices)) { fail(ctx, PTK_E_INVALID, "mesh arrays missing"); return 0; }
    std::string err;
    std::shared_ptr<PtkBvh> bvh = ptk_bvh_build_device(ctx->stream, order, type, n_objects, aabbs6, object_ids, subs,
                                                      mesh_transforms12, mesh_indices, err);
    if (!bvh) { fail(ctx, PTK_E_UNSUPPORTED, "bvh build failed: %s", err.c_str()); return 0; }
    for (auto& s : subs) s->adopted = true;
    auto o = std::make_unique<Object>();
    o->kind = Kind::Bvh;
    o->bvh = std::move(bvh);
    uint64_t h = ctx->next_handle++;
    ctx->objects[h] = std::move(o);
    return h;
}

int ptk_bvh_bind(ptk_ctx* ctx, uint64_t bvh, uint64_t* nodes_buffer, uint64_t* ids_buffer) {
    if (!ctx) return PTK_E_INVALID;
    Object* t = find(ctx, bvh, Kind::Bvh);
    if (!t) return fail(ctx, PTK_E_INVALID, "unknown BVH handle");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int n_nodes = t->bvh->flat_node_count(), n_ids = t->bvh->flat_id_count();
    // the sizes PTWorkDispatcher gives these buffers: nodes padded to the next multiple of 64 strictly above the
    // count (pt_work_dispatcher.gd:13-19,163-164,723-744), ids as many as there are, 8 bytes when there are none (:756)
    const size_t node_bytes = (size_t)(n_nodes / 64 + 1) * 64 * 32, id_bytes = n_ids > 0 ? (size_t)n_ids * 4 : 8;
    auto make = [&](int binding, size_t nbytes) -> std::unique_ptr<Object> {
        auto o = std::make_unique<Object>();
        o->kind = Kind::Buffer; o->nbytes = nbytes; o->set = PTK_SET_BVH; o->binding = binding;
        o->shadow.assign(nbytes, 0);
        const size_t alloc = (nbytes + 255) & ~(size_t)255;
        if (hipMalloc(&o->dev, alloc) != hipSuccess || hipMemsetAsync(o->dev, 0, alloc, ctx->stream) != hipSuccess) return nullptr;
        return o;
    };
    std::unique_ptr<Object> nodes = make(0, node_bytes), ids = make(1, id_bytes);
    if (!nodes || !ids) {
        if (nodes) release(ctx, nodes.get());
        if (ids) release(ctx, ids.get());
        return fail(ctx, PTK_E_HIP, "device allocation of %zu + %zu bytes failed", node_bytes, id_bytes);
    }
    auto bt = std::make_unique<BoundTree>();
    bt->bvh = t->bvh;
    bt->state = std::make_unique<PtkRefitState>();
    std::string err;
    // node bytes and id table go from the builder's device buffers straight into the bound buffers; the host copies
    // (for buffer_read-style validation and later patches) come from the same pass
    if (!t->bvh->bind(ctx->stream, (uint8_t*)nodes->dev, n_ids > 0 ? (int32_t*)ids->dev : nullptr, nodes->shadow.data(),
                      n_ids > 0 ? (int32_t*)ids->shadow.data() : nullptr, bt->state.get(), err)) {
        release(ctx, nodes.get());
        release(ctx, ids.get());
        return fail(ctx, PTK_E_HIP, "bvh bind failed: %s", err.c_str());
    }
    const uint64_t hn = ctx->next_handle++, hi = ctx->next_handle++;
    ctx->objects[hn] = std::move(nodes);
    ctx->objects[hi] = std::move(ids);
    ctx->bound[{PTK_SET_BVH, 0}] = hn;
    ctx->bound[{PTK_SET_BVH, 1}] = hi;
    bt->nodes_handle = hn;
    bt->ids_handle = hi;
    ctx->bound_tree = std::move(bt);
    ctx->dirty = true;
    if (nodes_buffer) *nodes_buffer = hn;
    if (ids_buffer) *ids_buffer = hi;
    return PTK_OK;
}

// The records a refit queues (node index + 32 bytes each: new boxes, ids and transform indices as they were) reach the
// bound node buffer in one pass: shadow copy, pending range, and -- when the derived structures are current -- the box
// planes of the node's slot in its parent's block, with one look at the top-of-tree chain at the end.  A record that
// changes more than a box goes the general way (ptk_buffer_update).
static int apply_refit_records(ptk_ctx* ctx, uint64_t nodes_handle, const std::vector<int32_t>& nodes, const std::vector<uint8_t>& bytes) {
    Object* o = find(ctx, nodes_handle, Kind::Buffer);
    if (!o) return fail(ctx, PTK_E_STATE, "the bound tree's node buffer is gone");
    const bool patch = !ctx->dirty && ctx->wf_ok && bound_buffer(ctx, o->set, o->binding) == o && nodes.size() <= 65536;
    if (patch && !nodes.empty()) { int r = ensure_mirrors(ctx); if (r != PTK_OK) return r; }
    for (size_t k = 0; k < nodes.size(); k++) {
        const size_t i = (size_t)nodes[k];
        const uint8_t* rec = bytes.data() + k * 32;
        if ((i + 1) * 32 > o->nbytes) return fail(ctx, PTK_E_INVALID, "refit record for node %zu outside the buffer", i);
        uint8_t* sh = o->shadow.data() + i * 32;
        if (!patch || memcmp(sh + 12, rec + 12, 4) != 0 || memcmp(sh + 28, rec + 28, 4) != 0 || i >= ctx->node_slot.size()) {
            int r = ptk_buffer_update(ctx, nodes_handle, i * 32, rec, 32);
            if (r != PTK_OK) return r;
            continue;
        }
        memcpy(sh, rec, 32);
        ctx->pending.push_back({nodes_handle, i * 32, (size_t)32});
        ctx->patched_updates++;
        float box[8];
        memcpy(box, rec, 32);
        if (i == 0) { memcpy(ctx->root_box, box, 12); memcpy(ctx->root_box + 3, box + 4, 12); }
        const int32_t ns = ctx->node_slot[i];
        if (ns < 0) continue;  // the root, a mesh's root, or a node the traversal cannot reach
        const size_t blk = (size_t)(ns >> 3) * 64, l = (size_t)(ns & 7);
        float* h = ctx->h_blocks.data() + blk;
        const int plane_of[6] = {0, 1, 2, 4, 5, 6};
        for (int q = 0; q < 6; q++) h[q * 8 + l] = box[plane_of[q]];
        const uint32_t bi = (uint32_t)(ns >> 3);
        if (!ctx->block_is_dirty[bi]) { ctx->block_is_dirty[bi] = 1; ctx->dirty_blocks.push_back(bi); }
    }
    if (patch) derive_chain(ctx);  // (a patched box may be one of the chain's)
    return PTK_OK;
}

int ptk_bvh_refit(ptk_ctx* ctx, const ptk_refit_event* events, int n_events, int32_t* n_nodes_updated) {
    if (!ctx) return PTK_E_INVALID;
    if (n_nodes_updated) *n_nodes_updated = 0;
    if (n_events < 0 || (n_events > 0 && !events)) return fail(ctx, PTK_E_INVALID, "bad event list");
    BoundTree* bt = ctx->bound_tree.get();
    if (!bt) return fail(ctx, PTK_E_STATE, "no tree is bound with ptk_bvh_bind (or its buffers were edited or replaced since)");
    if (n_events == 0) return PTK_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (bt->slot_of_object.empty() && bt->state->n_ids > 0) {
        bt->slot_of_object.reserve((size_t)bt->state->n_ids);
        for (int i = 0; i < bt->state->n_ids; i++) bt->slot_of_object.push_back({bt->state->object_ids[i], i});
        std::sort(bt->slot_of_object.begin(), bt->slot_of_object.end());
    }
    std::vector<PtkRefitEvent> ev((size_t)n_events);
    for (int j = 0; j < n_events; j++) {
        const int32_t id = events[j].object_id;
        const int type = (int)((uint32_t)id >> 24);
        ev[j].slot = -1; ev[j].mesh = -1;
        memcpy(ev[j].v, events[j].v, 48);
        if (type == PTK_REFIT_MESH) {
            const int mi = id & 0x00FFFFFF;
            for (size_t m = 0; m < bt->state->mesh_index.size(); m++) if (bt->state->mesh_index[m] == mi) ev[j].mesh = (int)m;
            if (ev[j].mesh < 0) return fail(ctx, PTK_E_INVALID, "event %d: mesh %d is not part of the bound tree", j, mi);
        } else {
            auto it = std::lower_bound(bt->slot_of_object.begin(), bt->slot_of_object.end(), std::make_pair(id, (int32_t)INT32_MIN));
            if (it == bt->slot_of_object.end() || it->first != id) return fail(ctx, PTK_E_INVALID, "event %d: object id 0x%08x is not in the bound tree", j, (unsigned)id);
            ev[j].slot = it->second;
        }
    }
    std::vector<int32_t> nodes;
    std::vector<uint8_t> bytes;
    std::string err;
    if (!bt->state->refit(ctx->stream, ev, nodes, bytes, err)) { ctx->bound_tree.reset(); return fail(ctx, PTK_E_HIP, "refit failed: %s", err.c_str()); }
    // the queued nodes reach the bound node buffer the way the plugin's own dirty-node uploads do
    // (pt_renderer.gd:771-788): shadow copy, patched blocks, one staged transfer at the next flush
    ctx->in_refit = true;
    int rc = apply_refit_records(ctx, bt->nodes_handle, nodes, bytes);
    ctx->in_refit = false;
    if (rc != PTK_OK) return rc;
    if (n_nodes_updated) *n_nodes_updated = (int32_t)nodes.size();
    return PTK_OK;
}

int ptk_bvh_sizes(ptk_ctx* ctx, uint64_t bvh, int32_t* n_nodes, int32_t* n_object_ids) {
    if (!ctx) return PTK_E_INVALID;
    Object* o = find(ctx, bvh, Kind::Bvh);
    if (!o) return fail(ctx, PTK_E_INVALID, "unknown BVH handle");
    if (n_nodes) *n_nodes = o->bvh->flat_node_count();
    if (n_object_ids) *n_object_ids = o->bvh->flat_id_count();
    return PTK_OK;
}

int ptk_bvh_read(ptk_ctx* ctx, uint64_t bvh, void* node_bytes, int32_t* object_ids) {
    if (!ctx) return PTK_E_INVALID;
    Object* o = find(ctx, bvh, Kind::Bvh);
    if (!o) return fail(ctx, PTK_E_INVALID, "unknown BVH handle");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    std::string err;
    if (!o->bvh->flatten(ctx->stream, (uint8_t*)node_bytes, object_ids, err)) return fail(ctx, PTK_E_HIP, "bvh read failed: %s", err.c_str());
    return PTK_OK;
}

}  // extern "C"

// ---- RCCL glue ---------------------------------------------------------------------------------------
// librccl is opened at run time: a host that renders on one GPU never loads it, and a process that already
// holds a copy (PyTorch ships one) keeps using that copy.  Only the few entry points used here are bound;
// their signatures are those of rccl.h (NCCL 2 API): ncclResult_t 0 = success, ncclFloat32 = 7, ncclSum = 0.
namespace {
struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, const void* /* ncclUniqueId by value, see init */, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*Reduce)(const void*, void*, size_t, int, int, int, void*, hipStream_t) = nullptr;
    int (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string problem;
};
struct UniqueId { char bytes[PTK_COMM_ID_BYTES]; };  // ncclUniqueId: 128 opaque bytes, passed by value
typedef int (*CommInitRankFn)(void**, int, UniqueId, int);

Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"librccl.so", "librccl.so.1"};
        for (const char* n : names) if (!r.lib) r.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);  // a copy already in the process
        for (const char* n : names) if (!r.lib) r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (!r.lib) { r.problem = std::string("librccl.so not found: ") + dlerror(); return; }
        auto sym = [&](const char* name) { void* p = dlsym(r.lib, name); if (!p && r.problem.empty()) r.problem = std::string("librccl lacks ") + name; return p; };
        r.GetUniqueId = (int (*)(void*))sym("ncclGetUniqueId");
        r.CommInitRank = (int (*)(void**, int, const void*, int))sym("ncclCommInitRank");
        r.CommDestroy = (int (*)(void*))sym("ncclCommDestroy");
        r.Reduce = (int (*)(const void*, void*, size_t, int, int, int, void*, hipStream_t))sym("ncclReduce");
        r.Send = (int (*)(const void*, size_t, int, int, void*, hipStream_t))sym("ncclSend");
        r.Recv = (int (*)(void*, size_t, int, int, void*, hipStream_t))sym("ncclRecv");
        r.GroupStart = (int (*)())sym("ncclGroupStart");
        r.GroupEnd = (int (*)())sym("ncclGroupEnd");
        r.GetErrorString = (const char* (*)(int))sym("ncclGetErrorString");
    });
    return r;
}
int rccl_fail(ptk_ctx* ctx, const char* what, int rc) {
    Rccl& r = rccl();
    return fail(ctx, PTK_E_HIP, "%s: %s", what, r.GetErrorString ? r.GetErrorString(rc) : "RCCL error");
}
#define RCCL_TRY(ctx, expr)                                  \
    do {                                                     \
        const int rc_ = (expr);                              \
        if (rc_ != 0) return rccl_fail(ctx, #expr, rc_);     \
    } while (0)
constexpr int kNcclFloat32 = 7, kNcclSum = 0;
}  // namespace

int ptk_comm_unique_id(void* id128) {
    if (!id128) return PTK_E_INVALID;
    Rccl& r = rccl();
    if (!r.problem.empty()) return fail(nullptr, PTK_E_UNSUPPORTED, "%s", r.problem.c_str());
    const int rc = r.GetUniqueId(id128);
    if (rc != 0) return rccl_fail(nullptr, "ncclGetUniqueId", rc);
    return PTK_OK;
}

int ptk_comm_init(ptk_ctx* ctx, int world, int rank, const void* id128) {
    if (!ctx) return PTK_E_INVALID;
    if (!id128 || world < 1 || rank < 0 || rank >= world) return fail(ctx, PTK_E_INVALID, "bad communicator arguments (world %d, rank %d)", world, rank);
    if (ctx->comm) return fail(ctx, PTK_E_STATE, "the context already belongs to a communicator");
    Rccl& r = rccl();
    if (!r.problem.empty()) return fail(ctx, PTK_E_UNSUPPORTED, "%s", r.problem.c_str());
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    UniqueId id;
    memcpy(id.bytes, id128, sizeof id.bytes);
    void* comm = nullptr;
    RCCL_TRY(ctx, ((CommInitRankFn)(void*)r.CommInitRank)(&comm, world, id, rank));
    ctx->comm = comm;
    ctx->comm_world = world;
    ctx->comm_rank = rank;
    return PTK_OK;
}

int ptk_comm_destroy(ptk_ctx* ctx) {
    if (!ctx) return PTK_E_INVALID;
    if (!ctx->comm) return PTK_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    const int rc = rccl().CommDestroy(ctx->comm);
    ctx->comm = nullptr;
    ctx->comm_world = 0;
    ctx->comm_rank = -1;
    return rc == 0 ? PTK_OK : rccl_fail(ctx, "ncclCommDestroy", rc);
}

int ptk_image_reduce_mean(ptk_ctx* ctx, uint64_t image, int root) {
    if (!ctx) return PTK_E_INVALID;
    if (!ctx->comm) return fail(ctx, PTK_E_STATE, "ptk_comm_init has not been called");
    Object* o = find(ctx, image, Kind::Image);
    if (!o) return fail(ctx, PTK_E_INVALID, "unknown image handle");
    if (root < 0 || root >= ctx->comm_world) return fail(ctx, PTK_E_INVALID, "root %d outside the communicator", root);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (int rc = flush_updates(ctx)) return rc;
    Rccl& r = rccl();
    // in place on root (its image receives the sum); the other ranks only send
    RCCL_TRY(ctx, r.Reduce(o->dev, o->dev, o->nbytes / sizeof(float), kNcclFloat32, kNcclSum, root, ctx->comm, ctx->stream));
    if (ctx->comm_rank == root && ctx->comm_world > 1)
        HIP_TRY(ctx, ptk_launch_scale((float4*)o->dev, o->nbytes / 16, 1.0f / (float)ctx->comm_world, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PTK_OK;
}

int ptk_band_layout(int image_height, int world, int32_t* first_row, int32_t* n_rows) {
    if (image_height <= 0 || world <= 0 || !first_row || !n_rows) return PTK_E_INVALID;
    // contiguous bands of 8-row tiles whose tile counts differ by at most one; the last tile row may be cut by the image
    const int tiles = (image_height + 7) / 8, base = tiles / world, extra = tiles % world;
    int start = 0;
    for (int r = 0; r < world; r++) {
        const int n = base + (r < extra ? 1 : 0);
        first_row[r] = std::min(start * 8, image_height);
        n_rows[r] = std::max(0, std::min(n * 8, image_height - start * 8));
        start += n;
    }
    return PTK_OK;
}

// Bands as ptk_image_gather_bands accepts them: inside the image and disjoint (rank order is free).
static int check_bands(ptk_ctx* ctx, int height, int world, const int32_t* first_row, const int32_t* n_rows) {
    for (int k = 0; k < world; k++)
        if (first_row[k] < 0 || n_rows[k] < 0 || (long long)first_row[k] + n_rows[k] > height)
            return fail(ctx, PTK_E_INVALID, "band of rank %d (rows %d..%d) leaves the image (%d rows)", k, first_row[k], first_row[k] + n_rows[k], height);
    for (int a = 0; a < world; a++)
        for (int b = a + 1; b < world; b++)
            if (n_rows[a] > 0 && n_rows[b] > 0 && first_row[a] < first_row[b] + n_rows[b] && first_row[b] < first_row[a] + n_rows[a])
                return fail(ctx, PTK_E_INVALID, "bands of ranks %d and %d overlap", a, b);
    return PTK_OK;
}

int ptk_band_check(int image_height, int world, const int32_t* first_row, const int32_t* n_rows) {
    if (image_height <= 0 || world <= 0 || !first_row || !n_rows) return PTK_E_INVALID;
    return check_bands(nullptr, image_height, world, first_row, n_rows);
}

int ptk_image_gather_bands(ptk_ctx* ctx, uint64_t image, const int32_t* first_row, const int32_t* n_rows, int root) {
    if (!ctx) return PTK_E_INVALID;
    if (!ctx->comm) return fail(ctx, PTK_E_STATE, "ptk_comm_init has not been called");
    Object* o = find(ctx, image, Kind::Image);
    if (!o) return fail(ctx, PTK_E_INVALID, "unknown image handle");
    if (!first_row || !n_rows || root < 0 || root >= ctx->comm_world) return fail(ctx, PTK_E_INVALID, "band tables and a root inside the communicator are required");
    { int r = check_bands(ctx, o->height, ctx->comm_world, first_row, n_rows); if (r != PTK_OK) return r; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (int rc = flush_updates(ctx)) return rc;
    Rccl& r = rccl();
    const size_t row_floats = (size_t)o->width * 4;
    float* base = (float*)o->dev;
    RCCL_TRY(ctx, r.GroupStart());
    int rc_in_group = 0;  // (a failed call must not leave the group open)
    if (ctx->comm_rank == root) {
        for (int k = 0; k < ctx->comm_world && rc_in_group == 0; k++)
            if (k != root && n_rows[k] > 0)
                rc_in_group = r.Recv(base + (size_t)first_row[k] * row_floats, (size_t)n_rows[k] * row_floats, kNcclFloat32, k, ctx->comm, ctx->stream);
    } else if (n_rows[ctx->comm_rank] > 0) {
        rc_in_group = r.Send(base + (size_t)first_row[ctx->comm_rank] * row_floats, (size_t)n_rows[ctx->comm_rank] * row_floats, kNcclFloat32, root,
                             ctx->comm, ctx->stream);
    }
    if (rc_in_group != 0) {
        (void)r.GroupEnd();
        return rccl_fail(ctx, "ncclSend / ncclRecv", rc_in_group);
    }
    RCCL_TRY(ctx, r.GroupEnd());
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PTK_OK;
}
