// ptk_kernels.hip -- the render kernel for gfx950 (CDNA4, wave64).
//
// One lane carries one pixel's path for one sample, like one invocation of the
// reference compute shader (shaders/ray_tracer.comp:1639-1705); one 64-thread
// workgroup is one 8x8 pixel tile (= exactly one wavefront).  The traversal
// stack lives in LDS, laid out [entry][lane] so that any mix of stack depths
// across the wave is bank-conflict free; entries deeper than the LDS window
// spill to a per-lane global column.  Stack entries are the child's packed
// node id (read together with its box), so popping a node needs no dependent
// load of the node record.
//
// Every arithmetic step follows the operation order written down in
// ptk_math.h; the file is compiled with -ffp-contract=off so results are
// independent of instruction scheduling.
#include <hip/hip_runtime.h>
#include <stdint.h>


#include "ptk_render_core.h"

// Four waves per SIMD for the untraced block-traversal variant (the one plain rendering runs): it needs some 150
// registers, three waves' worth; held to 128 it spills about forty of them and is still the faster (more tiles in
// flight per compute unit: 0.44 -> 0.41 ms on the cornell box, 2.08 -> 1.87 ms on the random spheres, 0.95 -> 0.94 ms on the dragon).
#ifndef PTK_TILE_WAVES
#define PTK_TILE_WAVES(blocks) __attribute__((amdgpu_waves_per_eu((blocks) ? 4 : 1, (blocks) ? 4 : 10)))
#endif
// One invocation of main(), :1639-1705, including cast_ray, :1553-1635.
// BLOCKS: the dispatch is known to search through the compiled node blocks only (BVH flag set, no BVH view, blocks
// present, untraced, no early-t): the kernel then holds that one search and needs fewer registers.
template <bool TRACE, bool EARLY_T, bool BLOCKS = false>
__global__ __launch_bounds__(64) PTK_TILE_WAVES(BLOCKS) void ptk_render_kernel(const PtkParams P) {
    extern __shared__ int lds_stack[];
    const int lane = threadIdx.x;
    const uint32_t wg = blockIdx.y * gridDim.x + blockIdx.x;
    const uint32_t tile = P.tile_order ? P.tile_order[wg] : wg;  // (a permutation of the tiles: every pixel is rendered once)
    const unsigned long long tile_t0 = P.tile_cost ? wall_clock64() : 0ull;
    const int gx = (int)(tile % gridDim.x) * 8 + (lane & 7);
    const int gy = (int)(tile / gridDim.x) * 8 + (lane >> 3);
    const PtkPush& push = P.push;

    int cnt[PTK_NCOUNTERS];
    if (TRACE) {
#pragma unroll
        for (int k = 0; k < PTK_NCOUNTERS; k++) cnt[k] = 0;
    }

    Stack S, ST;
    S.lds = lds_stack; S.lane = lane; S.lds_cap = P.lds_cap;
    const size_t glane = (size_t)wg * 64 + lane;
    S.spill_stride = P.spill_stride;
    S.spill = P.spill ? P.spill + glane : nullptr;
    ST = S;
    if (EARLY_T) {
        ST.lds = lds_stack + P.lds_cap * 64;
        ST.spill = P.spill_t ? P.spill_t + glane : nullptr;
    }

    Lane L;
    L.current_IOR = 1.0f; L.inside_of_count = 0;
#pragma unroll
    for (int i = 0; i < MAX_OVERLAP; i++) { L.in_mat[i] = 0; L.in_pri[i] = INT_NEG_LIMIT; }
    L.is_emissive = 0; L.refraction_bounces = 0; L.ignore_attenuation = 0;
    L.node_hit_count = 0; L.object_hit_count = 0;

    const Primary pr = primary_ray(push, gx, gy);
    const int ux = pr.ux, uy = pr.uy;
    L.seed = pr.seed;

    // parse_flags, :566-590
    const int fl = push.flags;
    const bool use_bvh = fl >= 0 && (fl & 1);
    const bool show_node_count = fl >= 0 && (fl & 4), show_object_count = fl >= 0 && (fl & 8);
    const bool show_bvh_view = push.render_mode == 1, show_normal_view = push.render_mode == 2;
    const bool single_sample = show_bvh_view || show_normal_view;

    Ray nr = pr.ray;

    // cast_ray, :1553-1635
    float4 new_color = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
    f3 last_normal = mk3(0.0f, 0.0f, 0.0f);
    int primary = -1;
    bool first = true;
    int i = 0;
    if (single_sample) {
        L.refraction_bounces = push.max_refraction_bounces;
        i = push.max_default_depth + push.max_refraction_bounces - 1;
    }
    for (; i < push.max_default_depth + L.refraction_bounces; i++) {
        if (TRACE) cnt[PTK_CNT_RAYS]++;
        Hit h;
        if (BLOCKS) h = check_ray_hit_blocks(P, nr, S);
        else if (!use_bvh) h = check_ray_hit_all<TRACE>(P, nr, cnt);
        else if (EARLY_T) h = check_ray_hit_bvh<TRACE, EARLY_T>(P, L, nr, S, ST, cnt);
        else if (!TRACE && P.wf.blocks != nullptr && !show_bvh_view) h = check_ray_hit_blocks(P, nr, S);  // no counters kept
        else h = check_ray_hit_bvh2<TRACE>(P, L, nr, S, cnt);
        if (first) { primary = h.hit ? h.object_id : -1; first = false; }
        if (!h.hit) {
            float4 sky = P.use_emission ? make_float4(0.0f, 0.0f, 0.0f, 0.0f) : make_float4(0.7f, 0.7f, 0.9f, 1.0f);
            new_color = make_float4(new_color.x * sky.x, new_color.y * sky.y, new_color.z * sky.z, new_color.w * sky.w);
            last_normal = mk3(0.0f, 0.0f, 0.0f);
            break;
        }
        Surf s = determine_rayhit<TRACE>(P, L, nr, h, cnt);
        last_normal = s.normal;
        if (P.use_emission && L.is_emissive) {
            new_color = make_float4(new_color.x * s.albedo.x, new_color.y * s.albedo.y, new_color.z * s.albedo.z, new_color.w * 1.0f);
            break;
        }
        float4 tex_col = get_texture(P, s);
        nr = bounce_ray(P, L, nr, s);
        if (!L.ignore_attenuation) {
            new_color.x *= s.albedo.x * tex_col.x;
            new_color.y *= s.albedo.y * tex_col.y;
            new_color.z *= s.albedo.z * tex_col.z;
            new_color.w *= 1.0f * tex_col.w;
        } else {
            L.ignore_attenuation = 0;
        }
    }
    if (P.use_emission && !L.is_emissive) new_color = make_float4(0.0f, 0.0f, 0.0f, 0.0f);

    bool override_color = false;
    float4 render_mode_color = make_float4(-1.0f, -1.0f, -1.0f, -1.0f);
    if (show_bvh_view) {  // :1624-1628
        float a = show_node_count ? (float)L.node_hit_count / (float)push.node_count_threshold : 0.0f;
        float b = show_object_count ? (float)L.object_hit_count / (float)push.object_count_threshold : 0.0f;
        render_mode_color = (pmax(a, b) > 1.0f) ? make_float4(1.0f, 1.0f, 1.0f, 1.0f) : make_float4(b, 0.0f, a, 0.0f);
    }
    if (show_normal_view) {  // :1630-1632
        render_mode_color = make_float4((last_normal.x + 1.0f) / 2.0f, (last_normal.y + 1.0f) / 2.0f, (last_normal.z + 1.0f) / 2.0f, 1.0f);
    }
    override_color = render_mode_color.x != -1.0f || render_mode_color.y != -1.0f || render_mode_color.z != -1.0f || render_mode_color.w != -1.0f;

    const bool in_image = store_pixel(P, ux, uy, new_color, override_color, render_mode_color);
    const size_t pix = (size_t)uy * P.img_w + ux;
    if (in_image) {
        if (TRACE) {
            if (P.trace_primary) P.trace_primary[pix] = primary;
            if (P.trace_counters) {
                cnt[PTK_CNT_BOX_TESTS] = L.node_hit_count;
                cnt[PTK_CNT_OBJ_TESTS] = L.object_hit_count;
#pragma unroll
                for (int k = 0; k < PTK_NCOUNTERS; k++) P.trace_counters[pix * PTK_NCOUNTERS + k] = cnt[k];
            }
        }
    }
    if (P.tile_cost && lane == 0) {
        const unsigned long long dt = wall_clock64() - tile_t0;
        P.tile_cost[tile] = dt > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)dt;
#ifdef PTK_TILE_HWID  // measurement builds: where and when the tile ran (the cost array has room for three words a tile)
        const uint32_t n_tiles = gridDim.x * gridDim.y;
        P.tile_cost[n_tiles + tile] = (__builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)) & 0xFFFFu) | (__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) << 16);
#ifdef PTK_TILE_WGID  // ... or which workgroup of the launch it was (its place in the order)
        P.tile_cost[2 * n_tiles + tile] = wg;
#else
        P.tile_cost[2 * n_tiles + tile] = (uint32_t)tile_t0;
#endif
#endif
    }
}

// Tiles in the order the next dispatch should start them: by the time they took in the last one, the longest first.
// A counting sort by one workgroup over classes an eighth of an octave wide (the exponent of the cost and its top three
// mantissa bits); a wave places 64 consecutive tiles at a time, so tiles of one class stay in runs of neighbours (which
// share rays' nodes in the caches), and when all tiles fall into two adjacent octaves the frame is even enough to be
// left in its own order.  Fine classes matter for where the longest tiles run: workgroups w, w + 1024, w + 2048 and
// w + 3072 of a launch start on the same SIMD (observed), so with the tiles in order of cost every SIMD begins with one
// of the 1024 longest, one of the next 1024 and so on, instead of several of the longest taking turns on one SIMD while
// others run dry.  The order changes when a tile runs, not what it computes.
constexpr int TILE_CLASSES = 33 * 8;
__device__ inline int tile_cost_class(uint32_t cost) {
    if (cost == 0u) return 0;
    const int lz = __clz(cost);
    return ((32 - lz) << 3) | (int)(((cost << lz) >> 28) & 7u);
}
__global__ __launch_bounds__(1024) void ptk_tile_order_kernel(const uint32_t* __restrict__ cost, uint32_t n, uint32_t* __restrict__ order) {
    __shared__ uint32_t count[TILE_CLASSES], start[TILE_CLASSES];
    __shared__ int even;
    const int lane = threadIdx.x & 63;
    if (threadIdx.x < TILE_CLASSES) count[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += 1024) atomicAdd(&count[tile_cost_class(cost[i])], 1u);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t at = 0;
        int lo = TILE_CLASSES, hi = -1;
        for (int c = TILE_CLASSES - 1; c >= 0; c--) {
            start[c] = at;
            at += count[c];
            if (count[c]) { lo = c; if (hi < 0) hi = c; }
        }
        even = (hi >> 3) - (lo >> 3) <= 1;
    }
    __syncthreads();
    if (even) {
        for (uint32_t i = threadIdx.x; i < n; i += 1024) order[i] = i;
        return;
    }
    for (uint32_t i0 = (threadIdx.x >> 6) * 64; i0 < n; i0 += 1024) {  // (uniform over the wave)
        const uint32_t i = i0 + lane;
        const int c = i < n ? tile_cost_class(cost[i]) : -1;
        unsigned long long todo = __ballot(c >= 0);
        while (todo) {
            const int cls = __builtin_amdgcn_readlane(c, __builtin_ctzll(todo));
            const unsigned long long m = __ballot(c == cls);
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&start[cls], (uint32_t)__popcll(m));
            base = __builtin_amdgcn_readfirstlane(base);
            if (c == cls) order[base + (uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = i;
            todo &= ~m;
        }
    }
}

extern "C" hipError_t ptk_launch_tile_order(const uint32_t* cost, uint32_t n, uint32_t* order, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(ptk_tile_order_kernel, dim3(1), dim3(1024), 0, stream, cost, n, order);
    return hipGetLastError();
}

extern "C" hipError_t ptk_launch_render(const PtkParams* p, int gx, int gy, int trace, int early_t, size_t lds_bytes, hipStream_t stream) {
    dim3 grid(gx, gy, 1), block(64, 1, 1);
    if (trace) {
        if (early_t) hipLaunchKernelGGL((ptk_render_kernel<true, true>), grid, block, lds_bytes, stream, *p);
        else hipLaunchKernelGGL((ptk_render_kernel<true, false>), grid, block, lds_bytes, stream, *p);
    } else {
        const bool blocks_only = !early_t && p->push.flags >= 0 && (p->push.flags & 1) && p->push.render_mode != 1 && p->wf.blocks != nullptr;
        if (early_t) hipLaunchKernelGGL((ptk_render_kernel<false, true>), grid, block, lds_bytes, stream, *p);
        else if (blocks_only) hipLaunchKernelGGL((ptk_render_kernel<false, false, true>), grid, block, lds_bytes, stream, *p);
        else hipLaunchKernelGGL((ptk_render_kernel<false, false>), grid, block, lds_bytes, stream, *p);
    }
    return hipGetLastError();
}
