// ptk_persistent.hip -- the render path as one persistent kernel for gfx950 (CDNA4, wave64).
//
// A lane carries one pixel's path, as an invocation of the reference shader does
// (shaders/ray_tracer.comp:1639-1705), but a wave is not tied to a tile: a lane whose path has ended
// stores its pixel and takes the next pixel from a device-wide counter, so no lane idles until the
// longest path of a tile is done, and nothing travels through queues in HBM between bounces.  A lane
// is in one of three states -- idle (needs a pixel), searching (closest-hit search in progress, the
// block traversal of ptk_render_core.h) or hit (search finished, the bounce of :1583-1621 is due) --
// and the wave runs one kind of round at a time: an inner-node or leaf round for the searching lanes,
// a bounce round once the waiting lanes outnumber the searching ones, a refill round once enough lanes are idle.
//
// Per path the arithmetic and its order are those of the other two renderers (same device functions),
// so the image has the same bits; only the order in which pixels are worked on differs.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>

#include "ptk_render_core.h"

namespace {

#ifndef PTK_PP_BOUNCE_NOW
// a bounce round runs once the lanes waiting for their bounce are as many as the lanes still searching
// (the same majority rule as between inner and leaf rounds); measured against fixed thresholds of 16..64
#define PTK_PP_BOUNCE_NOW(n_hit, n_search) ((n_hit) >= (n_search))
#endif
#ifndef PTK_PP_CLAIM_PER_WAVE
#define PTK_PP_CLAIM_PER_WAVE 16u  // pixel slots a workgroup takes from the device-wide counter at a time, per wave of it
                                   // (measured 64 / 32 / 16 on the 262 144-triangle scene: 14.9 / 14.5 / 14.4 ms; what a
                                   // workgroup holds nobody else can take, which lengthens the end of the launch)
#endif
#ifndef PTK_PP_REFILL
#define PTK_PP_REFILL 16           // idle lanes that make the wave run a refill round
#endif

PTK_DEV int pp_rank_in(unsigned long long mask) {  // set bits of mask below this lane
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

enum : int { PP_IDLE = 0, PP_SEARCH = 1, PP_HIT = 2 };

}  // namespace

#ifdef PTK_WF_STATS
// Measurement build: shader-clock ticks the waves spent in inner / leaf / refill / bounce rounds [0..3] and the
// numbers of those rounds [4..7].
__device__ unsigned long long g_pp_stats[8];
extern "C" hipError_t ptk_pp_stats_fetch(unsigned long long* out, hipStream_t stream) {
    hipError_t e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return e;
    unsigned long long h[8];
    e = hipMemcpyFromSymbol(h, HIP_SYMBOL(g_pp_stats), sizeof h);
    if (e != hipSuccess) return e;
    for (int i = 0; i < 8; i++) { out[i] = h[i]; h[i] = 0; }
    return hipMemcpyToSymbol(HIP_SYMBOL(g_pp_stats), h, sizeof h);
}
#endif

__global__ __launch_bounds__(PTK_PP_MAX_WG_WAVES * 64) void ptk_path_kernel(const PtkParams P) {
    // [top blocks: n_top * 272 bytes][per wave: stack window [entry][lane], P.lds_cap rows + one spare row][16 bytes: claim word]
    extern __shared__ __attribute__((aligned(16))) int lds_all[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wg_waves = (int)(blockDim.x >> 6);
    const PtkWavefront& W = P.wf;
    {
        // the waves of a workgroup share one thing: the LDS copy of the top of the block tree
        float4* top = (float4*)lds_all;
        for (uint32_t i = threadIdx.x; i < W.n_top * 16u; i += blockDim.x)
            top[(i >> 4) * (PTK_TOP_BLOCK_STRIDE / 16) + (i & 15u)] = W.blocks[i];
        if (threadIdx.x < PTK_WF_WG_FLAG_BYTES / 4) lds_all[W.n_top * (PTK_TOP_BLOCK_STRIDE / 4) + wg_waves * (P.lds_cap + 1) * 64 + threadIdx.x] = 0;
        __syncthreads();
    }
    // the workgroup's chunk of the pixel slots (see wg_claim), behind the stack windows
    unsigned long long* const wg_range = (unsigned long long*)(lds_all + W.n_top * (PTK_TOP_BLOCK_STRIDE / 4) + wg_waves * (P.lds_cap + 1) * 64);
    Stack S;
    S.top = (PtkLdsBlocks)lds_all; S.n_top = W.n_top;
    S.lds = lds_all + W.n_top * (PTK_TOP_BLOCK_STRIDE / 4) + wave * (P.lds_cap + 1) * 64;
    S.lane = lane; S.lds_cap = P.lds_cap; S.spill_stride = P.spill_stride;
    S.spill = P.spill ? P.spill + ((size_t)blockIdx.x * wg_waves + wave) * 64 + lane : nullptr;

    const uint32_t n_slots = W.n_slots;  // pixel slots, tile-major: slot >> 6 = 8x8 tile, slot & 63 = pixel in it
    uint32_t c_next = 0, c_end = 0;      // this wave's claimed range of slots
    const uint32_t wg_chunk = ((uint32_t)wg_waves * PTK_PP_CLAIM_PER_WAVE) / 64u > 0u ? ((uint32_t)wg_waves * PTK_PP_CLAIM_PER_WAVE) / 64u * 64u : 64u;  // a multiple of 64
    bool exhausted = n_slots == 0;

    int phase = PP_IDLE;
    uint32_t slot = 0;
    int bounce = 0;
    float4 color = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    Lane L;
    L.current_IOR = 1.0f; L.inside_of_count = 0; L.seed = 0.0f;
#pragma unroll
    for (int i = 0; i < MAX_OVERLAP; i++) { L.in_mat[i] = 0; L.in_pri[i] = INT_NEG_LIMIT; }
    L.is_emissive = 0; L.refraction_bounces = 0; L.ignore_attenuation = 0;
    L.node_hit_count = 0; L.object_hit_count = 0;
    Trav T;
    T.o = mk3(0, 0, 0); T.d = mk3(0, 0, 1); T.start = 0.0f; T.end = 0.0f;
    T.inv = mk3(0, 0, 0);
    T.h.hit = 0; T.h.object_id = 0; T.h.transform_index = -1; T.h.u = 0.0f; T.h.v = 0.0f;
    T.cur = ENTRY_NONE; T.sp = 0; T.sub_base = 0; T.cur_xf = -1;
    T.w_o = mk3(0, 0, 0); T.w_d = mk3(0, 0, 1); T.w_start = 0.0f; T.dir_len = 1.0f;

#ifdef PTK_WF_STATS
    long long st_clk[4] = {0, 0, 0, 0};  // shader-clock ticks this wave spent in inner / leaf / refill / bounce rounds
    long long st_n[4] = {0, 0, 0, 0};
#define PTK_PP_ROUND_BEGIN() const long long st_c0 = clock64()
#define PTK_PP_ROUND_END(k) (st_clk[k] += clock64() - st_c0, st_n[k]++)
#else
#define PTK_PP_ROUND_BEGIN()
#define PTK_PP_ROUND_END(k)
#endif
    // Two nested loops: bounce and refill rounds in the outer one, search rounds in the inner one until a round of
    // another kind is due.  (As branches of one loop, the rounds that replace a lane's whole path state made the
    // compiler carry a second copy of it through every search round.)
    for (;;) {
        {
            PTK_PP_ROUND_BEGIN();
            const unsigned long long m_search = __ballot(phase == PP_SEARCH);
            const bool none_searching = m_search == 0ull;
            const int n_hit = __popcll(__ballot(phase == PP_HIT));
            const int n_idle = __popcll(__ballot(phase == PP_IDLE));

            if (n_hit > 0 && (none_searching || PTK_PP_BOUNCE_NOW(n_hit, __popcll(m_search)))) {
                // ---- bounce round: the loop body of cast_ray after the search (:1583-1621) ----
                if (phase == PP_HIT) {
                    // the search ends at world level, so T.o / T.d are the segment's ray and T.end its hit distance
                    Ray nr;
                    nr.o = T.o; nr.d = T.d; nr.start = 0.0f; nr.end = T.end;
                    const Hit h = T.h;
                    bool done;
                    if (!h.hit) {
                        const float4 sky = P.use_emission ? make_float4(0.0f, 0.0f, 0.0f, 0.0f) : make_float4(0.7f, 0.7f, 0.9f, 1.0f);
                        color = make_float4(color.x * sky.x, color.y * sky.y, color.z * sky.z, color.w * sky.w);
                        done = true;
                    } else {
                        Surf s = determine_rayhit<false>(P, L, nr, h, nullptr);
                        if (P.use_emission && L.is_emissive) {
                            color = make_float4(color.x * s.albedo.x, color.y * s.albedo.y, color.z * s.albedo.z, color.w * 1.0f);
                            done = true;
                        } else {
                            const float4 tex_col = get_texture(P, s);
                            nr = bounce_ray(P, L, nr, s);
                            if (!L.ignore_attenuation) {
                                color.x *= s.albedo.x * tex_col.x;
                                color.y *= s.albedo.y * tex_col.y;
                                color.z *= s.albedo.z * tex_col.z;
                                color.w *= 1.0f * tex_col.w;
                            } else {
                                L.ignore_attenuation = 0;
                            }
                            bounce++;
                            done = !(bounce < P.push.max_default_depth + L.refraction_bounces);
                        }
                    }
                    if (done) {  // the end of main() (:1683-1704)
                        if (P.use_emission && !L.is_emissive) color = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                        const uint32_t tile = slot >> 6, l = slot & 63u;
                        const int gx = (int)(tile % (uint32_t)W.tiles_x) * 8 + (int)(l & 7u);
                        const int gy = (int)(tile / (uint32_t)W.tiles_x) * 8 + (int)(l >> 3);
                        store_pixel(P, gx + P.push.image_offset_x, gy + P.push.image_offset_y, color, false, color);
                        phase = PP_IDLE;
                    } else {
                        trav_begin(P, T, nr);
                        phase = T.cur == ENTRY_NONE ? PP_HIT : PP_SEARCH;  // (a ray that misses the root box has its result already)
                    }
                }
                PTK_PP_ROUND_END(3);
                continue;
            }

            if (!exhausted && n_idle > 0 && (none_searching || n_idle >= PTK_PP_REFILL)) {
                // ---- refill round: idle lanes take the next pixels and start their paths (:1645-1681) ----
#pragma unroll 1
                for (int pass = 0; pass < 2 && !exhausted; pass++) {
                    const unsigned long long need = __ballot(phase == PP_IDLE);
                    if (need == 0ull) break;
                    if (c_next == c_end) {
                        const int got = wg_claim(wg_range, W.heads, wg_chunk, n_slots, lane, c_next, c_end);
                        if (got == CLAIM_DRY) { exhausted = true; break; }
                        if (got == CLAIM_LATER) { __builtin_amdgcn_s_sleep(1); break; }  // another wave is fetching the workgroup's next chunk
                    }
                    const uint32_t avail = c_end - c_next;
                    const uint32_t my = (uint32_t)pp_rank_in(need);
                    if (phase == PP_IDLE && my < avail) {
                        slot = c_next + my;
                        const uint32_t tile = slot >> 6, l = slot & 63u;
                        const int gx = (int)(tile % (uint32_t)W.tiles_x) * 8 + (int)(l & 7u);
                        const int gy = (int)(tile / (uint32_t)W.tiles_x) * 8 + (int)(l >> 3);
                        const Primary pr = primary_ray(P.push, gx, gy);
                        L.current_IOR = 1.0f; L.inside_of_count = 0; L.seed = pr.seed;
#pragma unroll
                        for (int i = 0; i < MAX_OVERLAP; i++) { L.in_mat[i] = 0; L.in_pri[i] = INT_NEG_LIMIT; }
                        L.is_emissive = 0; L.refraction_bounces = 0; L.ignore_attenuation = 0;
                        color = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
                        bounce = 0;
                        trav_begin(P, T, pr.ray);
                        phase = T.cur == ENTRY_NONE ? PP_HIT : PP_SEARCH;
                    }
                    const uint32_t wanted = (uint32_t)__popcll(need);
                    c_next += wanted < avail ? wanted : avail;
                }
                PTK_PP_ROUND_END(2);
                continue;
            }

            if (none_searching) break;  // nothing to search, no bounce due, no pixel left to claim
        }

        // ---- search rounds: the larger group of searching lanes advances, the other waits ----
        for (;;) {
            PTK_PP_ROUND_BEGIN();
            const bool searching = phase == PP_SEARCH;
            const bool is_leaf = searching && trav_is_leaf(T);
            const bool is_inner = searching && trav_is_inner(T);
            const unsigned long long mi = __ballot(is_inner), ml = __ballot(is_leaf);
            const bool inner_round = PTK_VOTE_INNER(__popcll(mi), __popcll(ml));
            bool pop = false;
            if (inner_round) {
                if (is_inner) pop = trav_inner(P, S, T);
            } else {
                trav_leaf(P, S, T, is_leaf);
                pop = is_leaf;
            }
            if (pop) {
                trav_pop(S, T);
                if (T.cur == ENTRY_NONE) phase = PP_HIT;
            }
            PTK_PP_ROUND_END(inner_round ? 0 : 1);
            // is a round of another kind due (the tests of the outer loop)?
            const unsigned long long m_search = __ballot(phase == PP_SEARCH);
            const int n_hit = __popcll(__ballot(phase == PP_HIT));
            const int n_idle = 64 - n_hit - __popcll(m_search);
            if (m_search == 0ull) break;
            if (n_hit > 0 && PTK_PP_BOUNCE_NOW(n_hit, __popcll(m_search))) break;
            if (!exhausted && n_idle >= PTK_PP_REFILL) break;
        }
    }
#ifdef PTK_WF_STATS
    if (lane == 0) for (int k = 0; k < 4; k++) { atomicAdd(&g_pp_stats[k], (unsigned long long)st_clk[k]); atomicAdd(&g_pp_stats[4 + k], (unsigned long long)st_n[k]); }
#endif
}

// ---- static scratch for ptk_dispatch_noalloc --------------------------------------------------------
// The path kernel needs two pieces of device memory beyond the scene and the image: the pixel counter
// its waves claim work from, and the spill column for traversal-stack entries past the LDS window.
// ptk_dispatch takes both from hipMalloc; ptk_dispatch_noalloc, for callers that must not allocate,
// takes them from this array.  Every GPU the library runs on holds its own copy for the life of the
// process.  Layout: bytes [0, 256) the pixel counter, the rest the spill column.
static __device__ __align__(256) unsigned char scratch[PTK_NOALLOC_SCRATCH_BYTES];

namespace {
constexpr int kMaxDevices = 64;
void* g_scratch_ptr[kMaxDevices] = {nullptr};  // device address of `scratch` on each GPU (the array's name is not one)
// Every context on a GPU shares that GPU's array, each on a stream of its own: the end of the last launch that used
// the array is recorded here, and the next one waits for it on its stream before it touches the array.
hipEvent_t g_scratch_event[kMaxDevices] = {nullptr};
bool g_scratch_in_use[kMaxDevices] = {false};
std::mutex g_scratch_mutex;
}  // namespace

// Resolves this device's copy of the scratch array.  Call with `device` current; repeated calls are cheap.
extern "C" hipError_t ptk_persistent_scratch_init(int device) {
    if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (g_scratch_ptr[device]) return hipSuccess;
    void* ptr = nullptr;
    hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
    if (e != hipSuccess) return e;
    if (!ptr) return hipErrorInvalidSymbol;
    e = hipEventCreateWithFlags(&g_scratch_event[device], hipEventDisableTiming);  // (made here, so that the entry point itself creates nothing)
    if (e != hipSuccess) return e;
    g_scratch_ptr[device] = ptr;
    return hipSuccess;
}

// Device address of the scratch array on `device` (null before ptk_persistent_scratch_init succeeded there).
extern "C" void* ptk_persistent_scratch(int device) {
    if (device < 0 || device >= kMaxDevices) return nullptr;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    return g_scratch_ptr[device];
}

extern "C" hipError_t ptk_persistent_occupancy(size_t lds_per_wave, uint32_t n_top, int* wg_waves, int* wgs_per_cu) {
    // a workgroup may use up to the whole 160 KiB of a compute unit
    hipError_t e = hipFuncSetAttribute((const void*)ptk_path_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, PTK_LDS_BYTES_PER_CU);
    if (e != hipSuccess) return e;
    *wg_waves = 0;
    *wgs_per_cu = 0;
    for (int w : {PTK_PP_MAX_WG_WAVES, 8, 4, 2, 1}) {
        const size_t lds = (size_t)n_top * PTK_TOP_BLOCK_STRIDE + (size_t)w * lds_per_wave + PTK_WF_WG_FLAG_BYTES;
        if (lds > PTK_LDS_BYTES_PER_CU) continue;
        int n = 0;
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, ptk_path_kernel, w * 64, lds);
        if (e != hipSuccess) return e;
        if (n > 0) { *wg_waves = w; *wgs_per_cu = n; return hipSuccess; }
    }
    return hipSuccess;
}

// p->wf.heads points at the pixel counter (the first word of a 256-byte line, cleared here); p->wf.n_slots /
// tiles_x describe the dispatch; p->wf.blocks etc. the compiled scene.
extern "C" hipError_t ptk_launch_persistent(const PtkParams* p, size_t lds_per_wave, int n_wgs, int wg_waves, hipStream_t stream) {
    hipError_t e = hipMemsetAsync(p->wf.heads, 0, 256, stream);
    if (e != hipSuccess) return e;
    const size_t lds = (size_t)p->wf.n_top * PTK_TOP_BLOCK_STRIDE + (size_t)wg_waves * lds_per_wave + PTK_WF_WG_FLAG_BYTES;
    hipLaunchKernelGGL(ptk_path_kernel, dim3((unsigned)n_wgs), dim3((unsigned)wg_waves * 64), lds, stream, *p);
    return hipGetLastError();
}

// ptk_launch_persistent with the pixel counter and the spill column in the static array (p->wf.heads / p->spill point
// into it): ordered behind the last launch on this GPU that used the array, whichever context and stream issued it.
// The lock is held from the wait to the record, so that two host threads cannot both wait on the older event.
extern "C" hipError_t ptk_launch_persistent_static(const PtkParams* p, size_t lds_per_wave, int n_wgs, int wg_waves, int device, hipStream_t stream) {
    if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_event[device]) return hipErrorNotInitialized;
    hipError_t e = hipSuccess;
    if (g_scratch_in_use[device]) e = hipStreamWaitEvent(stream, g_scratch_event[device], 0);
    if (e != hipSuccess) return e;
    e = ptk_launch_persistent(p, lds_per_wave, n_wgs, wg_waves, stream);
    if (e != hipSuccess) return e;
    e = hipEventRecord(g_scratch_event[device], stream);
    g_scratch_in_use[device] = e == hipSuccess;
    return e;
}
