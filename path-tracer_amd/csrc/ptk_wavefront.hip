// ptk_wavefront.hip -- the render path as a wavefront pipeline for gfx950 (CDNA4, wave64).
//
// The reference runs one invocation per pixel through the whole bounce loop
// (shaders/ray_tracer.comp:1553-1635).  Every pixel's path is independent of every other's, so the
// same per-pixel sequence of closest-hit searches and bounces can be run bounce by bounce over queues
// of rays kept in HBM:
//
//   generate : one lane per pixel, camera ray + initial path state            (main(), :1639-1681)
//   traverse : persistent waves; a lane carries one ray through planes + BVH  (check_ray_hit_BVH, :1200-1320)
//              and takes the next ray from the queue as soon as it is done
//   shade    : one lane per ray of the bounce: hit resolution, texture, bounce, attenuation
//              (:1583-1621); paths that go on append their next ray, paths that end store the pixel
//
// The arithmetic of each path is the code in ptk_render_core.h, shared with the one-kernel renderer
// in ptk_kernels.hip, so both produce the same bits.  What changes is the schedule: lanes of the
// traversal kernel never wait for the longest path of their tile, and shading runs on compacted rays.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "ptk_render_core.h"

namespace {

constexpr int SHADE_BLOCK = 512;    // threads per shading block (one queue reservation per block)
#ifndef PTK_WF_VOTE_INNER
// which kind of round runs next: an inner round unless the lanes at a leaf outnumber those at an inner node by half
// (a leaf round costs more than an inner round but puts waiting lanes to work as helpers; ratios 1 / 1.5 / 2: 8.93, 8.88, 9.0 ms)
#define PTK_WF_VOTE_INNER(ni, nl) (2 * (ni) >= 3 * (nl))
#endif
#ifndef PTK_WF_REFILL
#define PTK_WF_REFILL 12           // lanes without work that make the wave run a refill round
#endif

PTK_DEV void slot_to_pixel(const PtkParams& P, uint32_t slot, int& gx, int& gy) {
    const uint32_t tile = slot >> 6, l = slot & 63u;
    gx = (int)(tile % (uint32_t)P.wf.tiles_x) * 8 + (int)(l & 7u);
    gy = (int)(tile / (uint32_t)P.wf.tiles_x) * 8 + (int)(l >> 3);
}

PTK_DEV int lane_rank_in(unsigned long long mask) {  // set bits of mask below this lane
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

}  // namespace

#ifdef PTK_WF_STATS
// Measurement build: per inner round of the traversal kernel, how many lanes run it, how many distinct
// node blocks they fetch, how large the groups of lanes on one block are, and at which tree depth.
//   [0, 65)    inner rounds by number of lanes on an inner node (sockets excluded)
//   [65, 130)  inner rounds by number of distinct blocks among those lanes
//   [130, 195) lane visits by size of the group of lanes that share the block
//   [200, 232) lane visits by block depth          [232, 264) distinct block fetches by block depth
//   [264]      lane visits of mesh sockets         [265] leaf rounds   [266] lanes on a leaf in leaf rounds
//   [272..274] shader-clock ticks spent in inner / leaf / refill rounds, summed over the waves; [276..278] their numbers
//   (-DPTK_WF_CLOCKS_ONLY: only these, so that the counting itself does not weigh on the rounds it times)
#define PTK_WF_NSTATS 280
__device__ unsigned long long g_wf_stats[PTK_WF_NSTATS];
namespace {
PTK_DEV void wf_stats_inner_round(const PtkParams& P, const Trav& T, bool is_inner, unsigned long long mi, int lane) {
    const bool blk = is_inner && (T.cur & 0xFF000000u) != SOCKET_MARK;
    const unsigned long long m = __ballot(blk);
    int depth = 0;
    if (blk) depth = (int)fbits(P.wf.blocks[(size_t)(T.cur & 0x00FFFFFFu) * 16 + 14].x);
    depth = depth < 0 ? 0 : (depth > 31 ? 31 : depth);
    int distinct = 0;
    unsigned long long rest = m;
    while (rest) {
        const int leader = __builtin_ctzll(rest);
        const uint32_t v = (uint32_t)__builtin_amdgcn_readlane((int)T.cur, leader);
        const int d = __builtin_amdgcn_readlane(depth, leader);
        const unsigned long long same = __ballot(blk && T.cur == v);
        const int g = __popcll(same);
        rest &= ~same;
        distinct++;
        if (lane == 0) {
            atomicAdd(&g_wf_stats[130 + g], (unsigned long long)g);
            atomicAdd(&g_wf_stats[200 + d], (unsigned long long)g);
            atomicAdd(&g_wf_stats[232 + d], 1ull);
        }
    }
    if (lane == 0) {
        atomicAdd(&g_wf_stats[__popcll(m)], 1ull);
        atomicAdd(&g_wf_stats[65 + distinct], 1ull);
        atomicAdd(&g_wf_stats[264], (unsigned long long)(__popcll(mi) - __popcll(m)));
    }
}
}  // namespace
extern "C" hipError_t ptk_wf_stats_fetch(unsigned long long* out, int n, hipStream_t stream) {
    hipError_t e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return e;
    unsigned long long h[PTK_WF_NSTATS];
    e = hipMemcpyFromSymbol(h, HIP_SYMBOL(g_wf_stats), sizeof h);
    if (e != hipSuccess) return e;
    for (int i = 0; i < n && i < PTK_WF_NSTATS; i++) out[i] = h[i];
    memset(h, 0, sizeof h);
    return hipMemcpyToSymbol(HIP_SYMBOL(g_wf_stats), h, sizeof h);
}
#endif

#ifdef PTK_WF_ENDS
// Measurement build: when every wave of a traversal launch started and ended (100 MHz wall clock), per bounce.
#define PTK_WF_ENDS_ITERS 16
#define PTK_WF_ENDS_WAVES 8192
__device__ unsigned long long g_wf_ends[PTK_WF_ENDS_ITERS * PTK_WF_ENDS_WAVES * 2];
extern "C" hipError_t ptk_wf_ends_fetch(unsigned long long* out, size_t n, hipStream_t stream) {
    hipError_t e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return e;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wf_ends), sizeof(unsigned long long) * n);
}
#endif

// The start of the search of queue entry `at` (world ray o, d over [0, inf)): planes, root box and the chain at the top
// of the tree (trav_begin), run where the ray is queued and left in PtkWavefront::prep for the traversal kernel.
// Ten words: the ray as the search goes on with it (mesh-local once the chain has entered a mesh), its end, |M d|,
// the plane hit so far (object id, 0 = none: a plane's id has its type in the top byte) and two flags: bit 0 the search
// goes on at PtkWavefront::root_entry (else it is over already), bit 1 it is inside the chain's mesh.  What is left of
// trav_begin's outcome follows from these: start = 0 (times |M d| in a mesh), 1 / d, the world ray = the queue entry's.
PTK_DEV void wf_prepare(const PtkParams& P, int parity, uint32_t at, f3 o, f3 d) {
    Ray wr;
    wr.o = o; wr.d = d; wr.start = 0.0f; wr.end = PT_INF;
    Trav T;
    T.w_o = o; T.w_d = d; T.w_start = 0.0f;
    trav_begin(P, T, wr);
    uint32_t* c = P.wf.prep[parity] + at;
    const size_t cap = P.wf.prep_cap;
    const uint32_t w[PTK_WF_PREP_PLANES] = {
        fbits(T.o.x), fbits(T.o.y), fbits(T.o.z), fbits(T.d.x), fbits(T.d.y), fbits(T.d.z), fbits(T.end), fbits(T.dir_len),
        T.h.hit ? (uint32_t)T.h.object_id : 0u, (T.cur != ENTRY_NONE ? 1u : 0u) | (T.cur_xf >= 0 ? 2u : 0u)};
#pragma unroll
    for (int k = 0; k < PTK_WF_PREP_PLANES; k++) c[(size_t)k * cap] = w[k];
}
// ... and taken up by a lane of the traversal kernel (a, b: the queue entry, whose ray is the search's world ray).
PTK_DEV void wf_take_prepared(const PtkWavefront& W, Trav& T, int parity, uint32_t at, float4 a, float4 b) {
    const uint32_t* c = W.prep[parity] + at;
    const size_t cap = W.prep_cap;
    uint32_t w[PTK_WF_PREP_PLANES];
#pragma unroll
    for (int k = 0; k < PTK_WF_PREP_PLANES; k++) w[k] = c[(size_t)k * cap];
    T.o = mk3(bitsf(w[0]), bitsf(w[1]), bitsf(w[2]));
    T.d = mk3(bitsf(w[3]), bitsf(w[4]), bitsf(w[5]));
    T.inv = mk3(1.0f / T.d.x, 1.0f / T.d.y, 1.0f / T.d.z);
    T.end = bitsf(w[6]);
    T.dir_len = bitsf(w[7]);
    const bool in_mesh = (w[9] & 2u) != 0;
    T.start = in_mesh ? 0.0f * T.dir_len : 0.0f;  // (trav_enter_mesh_m: the world ray's start, 0, times |M d|)
    T.h.hit = w[8] != 0u ? 1 : 0; T.h.object_id = (int)w[8]; T.h.transform_index = -1; T.h.u = 0.0f; T.h.v = 0.0f;
    T.cur = (w[9] & 1u) ? W.root_entry : ENTRY_NONE;
    T.sp = 0; T.sub_base = 0;
    T.cur_xf = in_mesh ? W.chain_xf : -1;
    T.w_o = mk3(a.x, a.y, a.z); T.w_d = mk3(b.x, b.y, b.z);
    T.w_start = 0.0f;
}

__global__ __launch_bounds__(64) void ptk_wf_generate(const PtkParams P) {
    const PtkWavefront& W = P.wf;
    const uint32_t slot = blockIdx.x * 64u + threadIdx.x;
    if (slot == 0) W.counts[0] = W.n_slots;  // (n_resume[0] stays 0)
    int gx, gy;
    slot_to_pixel(P, slot, gx, gy);
    const Primary pr = primary_ray(P.push, gx, gy);
    W.rays[0][(size_t)slot * 2] = make_float4(pr.ray.o.x, pr.ray.o.y, pr.ray.o.z, bitsf(slot));
    W.rays[0][(size_t)slot * 2 + 1] = make_float4(pr.ray.d.x, pr.ray.d.y, pr.ray.d.z, 0.0f);
    W.state[0][(size_t)slot * 2] = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
    W.state[0][(size_t)slot * 2 + 1] = make_float4(pr.seed, 1.0f, bitsf(0u), bitsf(0u));
    wf_prepare(P, 0, slot, pr.ray.o, pr.ray.d);
}

// Parking a search that is under way (see PtkWavefront::carry) and taking it up again.  Everything a Trav holds
// goes along except inv, which is 1 / d again; the stack rows below sp (all inside the LDS window, the caller
// checks) follow the PTK_WF_CARRY_PLANES state words.  Planes are [word][entry]: the lanes of a wave park to and
// resume from consecutive entries.
PTK_DEV void wf_park(const PtkWavefront& W, const Stack& S, const Trav& T, int iter, uint32_t ray_idx, uint32_t at) {
    const int from = iter & 1, to = (iter + 1) & 1;
    // the queue entry moves to the front of the next queue as it is: world ray, pixel slot, bounce number, path state
    W.rays[to][(size_t)at * 2] = W.rays[from][(size_t)ray_idx * 2];
    W.rays[to][(size_t)at * 2 + 1] = W.rays[from][(size_t)ray_idx * 2 + 1];
    W.state[to][(size_t)at * 2] = W.state[from][(size_t)ray_idx * 2];
    W.state[to][(size_t)at * 2 + 1] = W.state[from][(size_t)ray_idx * 2 + 1];
    W.hits[ray_idx] = make_int4(-2, 0, 0, 0);  // this launch's shading passes the entry over
    uint32_t* c = W.carry[to] + at;
    const size_t cap = W.carry_cap;
    const uint32_t w[PTK_WF_CARRY_PLANES] = {
        fbits(T.o.x), fbits(T.o.y), fbits(T.o.z), fbits(T.d.x), fbits(T.d.y), fbits(T.d.z), fbits(T.start), fbits(T.end),
        (uint32_t)T.h.hit, (uint32_t)T.h.object_id, (uint32_t)T.h.transform_index, fbits(T.h.u), fbits(T.h.v),
        T.cur, (uint32_t)T.sp, (uint32_t)T.sub_base, (uint32_t)T.cur_xf,
        fbits(T.w_o.x), fbits(T.w_o.y), fbits(T.w_o.z), fbits(T.w_d.x), fbits(T.w_d.y), fbits(T.w_d.z), fbits(T.w_start), fbits(T.dir_len)};
#pragma unroll
    for (int k = 0; k < PTK_WF_CARRY_PLANES; k++) c[(size_t)k * cap] = w[k];
    for (int r = 0; r < T.sp; r++) c[(size_t)(PTK_WF_CARRY_PLANES + r) * cap] = (uint32_t)S.lds[r * 64 + S.lane];
}
PTK_DEV void wf_unpark(const PtkWavefront& W, const Stack& S, Trav& T, int parity, uint32_t at) {
    const uint32_t* c = W.carry[parity] + at;
    const size_t cap = W.carry_cap;
    uint32_t w[PTK_WF_CARRY_PLANES];
#pragma unroll
    for (int k = 0; k < PTK_WF_CARRY_PLANES; k++) w[k] = c[(size_t)k * cap];
    T.o = mk3(bitsf(w[0]), bitsf(w[1]), bitsf(w[2]));
    T.d = mk3(bitsf(w[3]), bitsf(w[4]), bitsf(w[5]));
    T.start = bitsf(w[6]); T.end = bitsf(w[7]);
    T.inv = mk3(1.0f / T.d.x, 1.0f / T.d.y, 1.0f / T.d.z);
    T.h.hit = (int)w[8]; T.h.object_id = (int)w[9]; T.h.transform_index = (int)w[10]; T.h.u = bitsf(w[11]); T.h.v = bitsf(w[12]);
    T.cur = w[13]; T.sp = (int)w[14]; T.sub_base = (int)w[15]; T.cur_xf = (int)w[16];
    T.w_o = mk3(bitsf(w[17]), bitsf(w[18]), bitsf(w[19]));
    T.w_d = mk3(bitsf(w[20]), bitsf(w[21]), bitsf(w[22]));
    T.w_start = bitsf(w[23]); T.dir_len = bitsf(w[24]);
    for (int r = 0; r < T.sp; r++) S.lds[r * 64 + S.lane] = (int)c[(size_t)(PTK_WF_CARRY_PLANES + r) * cap];
}

// Closest hit of every ray of bounce `iter`.  Per lane this is the block traversal of
// ptk_render_core.h (trav_*: the reference's order of box and object tests); around it, lanes whose ray
// is finished write the hit record and claim the next queue entry.
//
// A workgroup is PTK_WF_WG_WAVES independent waves that share one thing: an LDS copy of the top of the
// block tree (blocks [0, n_top), breadth-first from the root), loaded once per launch.  About half of all
// node visits are to those few blocks; reading them from LDS takes them off the vector cache's gather path,
// which handles one lane's address per clock and is what this kernel saturates otherwise.
__global__ __launch_bounds__(PTK_WF_WG_WAVES * 64) void ptk_wf_traverse(const PtkParams P, int iter) {
    // [top blocks: n_top * 272 bytes][per wave: stack window [entry][lane], P.lds_cap rows + one spare row]
    extern __shared__ __attribute__((aligned(16))) int lds_all[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const PtkWavefront& W = P.wf;
#ifdef PTK_WF_ENDS
    const size_t ends_at = ((size_t)iter * PTK_WF_ENDS_WAVES + blockIdx.x * PTK_WF_WG_WAVES + wave) * 2;
    if (lane == 0 && iter < PTK_WF_ENDS_ITERS) g_wf_ends[ends_at] = wall_clock64();
#endif
    const uint32_t n_res = __builtin_amdgcn_readfirstlane(W.n_resume[iter]);  // parked searches at the queue's front
    const uint32_t count = n_res + __builtin_amdgcn_readfirstlane(W.counts[iter]);
    // A queue with no more than one chunk for every workgroup is dealt out by position, without the device-wide counter
    const bool dealt = count <= gridDim.x * W.wg_claim;
    // ... and evenly: every wave of the launch gets its share (a short queue holds the searches that are left over, long ones;
    // what a wave runs side by side takes turns in its rounds, so a search among few is through sooner than among 64)
    const uint32_t grab = dealt ? (count + gridDim.x * PTK_WF_WG_WAVES - 1) / (gridDim.x * PTK_WF_WG_WAVES) : 64u;
    const uint32_t chunk = dealt ? grab * PTK_WF_WG_WAVES : W.wg_claim;
    if (count == 0 || (dealt && blockIdx.x * chunk >= count)) {  // nothing for this workgroup
#ifdef PTK_WF_ENDS
        if (lane == 0 && iter < PTK_WF_ENDS_ITERS) g_wf_ends[ends_at + 1] = wall_clock64();
#endif
        return;
    }
    {
        float4* top = (float4*)lds_all;
        for (uint32_t i = threadIdx.x; i < W.n_top * 16u; i += PTK_WF_WG_WAVES * 64)
            top[(i >> 4) * (PTK_TOP_BLOCK_STRIDE / 16) + (i & 15u)] = W.blocks[i];
        if (threadIdx.x < PTK_WF_WG_FLAG_BYTES / 4) {
            int v = 0;
            if (dealt && (threadIdx.x == 2 || threadIdx.x == 3)) v = (int)((blockIdx.x + (threadIdx.x - 2)) * chunk);  // [2] next = the chunk's start, [3] its end
            if (threadIdx.x == 4) v = (int)(blockIdx.x % W.n_parts);  // [4] the part of the queue this workgroup draws from
            lds_all[W.n_top * (PTK_TOP_BLOCK_STRIDE / 4) + PTK_WF_WG_WAVES * (P.lds_cap + 1) * 64 + threadIdx.x] = v;
        }
        __syncthreads();  // the only workgroup-wide step: from here on every wave runs on its own
    }
    // words the waves of the workgroup tell each other through: [0] one of them found the queue dry,
    // [1] how many have parked their searches or left, [2..3] the workgroup's chunk of the queue
    // (an LDS-address-space pointer: through a generic pointer these reads, made in every round, would be flat loads
    // that take the vector-memory path and wait for every load in flight)
    typedef volatile int __attribute__((address_space(3))) * LdsFlags;
    const LdsFlags wg_flags = (LdsFlags)(lds_all + W.n_top * (PTK_TOP_BLOCK_STRIDE / 4) + PTK_WF_WG_WAVES * (P.lds_cap + 1) * 64);
    unsigned long long* const wg_range = (unsigned long long*)(lds_all + W.n_top * (PTK_TOP_BLOCK_STRIDE / 4) + PTK_WF_WG_WAVES * (P.lds_cap + 1) * 64 + 2);  // see wg_claim
    // parts of the queue (PtkWavefront::n_parts): equal stretches of whole chunks
    const uint32_t part_len = (count + W.n_parts * chunk - 1u) / (W.n_parts * chunk) * chunk;
    Stack S;
    S.top = (PtkLdsBlocks)lds_all; S.n_top = W.n_top;
    S.lds = lds_all + W.n_top * (PTK_TOP_BLOCK_STRIDE / 4) + wave * (P.lds_cap + 1) * 64;
    S.lane = lane; S.lds_cap = P.lds_cap; S.spill_stride = P.spill_stride;
    S.spill = P.spill ? P.spill + ((size_t)blockIdx.x * PTK_WF_WG_WAVES + wave) * 64 + lane : nullptr;

    const float4* __restrict__ rq = W.rays[iter & 1];
    const bool carry_on = W.carry[0] != nullptr;
    uint32_t c_next = 0, c_end = 0;  // this wave's claimed queue range
    bool exhausted = false;

    bool have = false;  // the lane holds a ray
    uint32_t ray_idx = 0;
    int ray_bounce = 0;  // the bounce number of the lane's path (<= iter)
    Trav T;
    T.o = mk3(0, 0, 0); T.d = mk3(0, 0, 1); T.start = 0.0f; T.end = 0.0f;
    T.inv = mk3(0, 0, 0);
    T.h.hit = 0; T.h.object_id = 0; T.h.transform_index = -1; T.h.u = 0.0f; T.h.v = 0.0f;
    T.cur = ENTRY_NONE; T.sp = 0; T.sub_base = 0; T.cur_xf = -1;
    T.w_o = mk3(0, 0, 0); T.w_d = mk3(0, 0, 1); T.w_start = 0.0f; T.dir_len = 1.0f;

    bool parked = false;  // this wave has parked its searches (what it could not park it finishes itself)
#ifdef PTK_WF_STATS
    long long st_clk[3] = {0, 0, 0};  // shader-clock ticks this wave spent in inner / leaf / refill rounds
    long long st_n[3] = {0, 0, 0};    // ... and how many of each it ran
#define PTK_ST_ROUND_BEGIN() const long long st_c0 = clock64()
#define PTK_ST_ROUND_END(k) (st_clk[k] += clock64() - st_c0, st_n[k]++)
#else
#define PTK_ST_ROUND_BEGIN()
#define PTK_ST_ROUND_END(k)
#endif
    // Two nested loops: a refill round, then search rounds until the next refill is due.  (Written as one loop with
    // the refill as a branch, every search round paid some seventy register moves: the refill replaces a lane's whole
    // search state, and the compiler kept a second copy of that state alive through all the other rounds.)
    for (;;) {
        {
            // ---- refill round: retire finished rays, claim new ones ----
            PTK_ST_ROUND_BEGIN();
#if defined(PTK_WF_STATS) && !defined(PTK_WF_CLOCKS_ONLY)
            if (lane == 0) atomicAdd(&g_wf_stats[271], 1ull);
#endif
            if (have && T.cur == ENTRY_NONE) {
                W.hits[ray_idx] = make_int4(T.h.hit, T.h.object_id, T.h.transform_index, (int)fbits(T.end));
                W.hit_uv[ray_idx] = make_float2(T.h.u, T.h.v);
                have = false;
            }
#pragma unroll 1
            for (int pass = 0; pass < 2 && !exhausted; pass++) {
                const unsigned long long need = __ballot(!have);
                if (need == 0ull) break;
                if (c_next == c_end) {
                    const int got = wg_claim(wg_range, dealt ? nullptr : W.heads + iter * PTK_WF_MAX_PARTS, chunk, count, lane, c_next, c_end, grab,
                                             W.n_parts, part_len, (volatile int __attribute__((address_space(3)))*)wg_flags + 4);
                    if (got == CLAIM_DRY) { exhausted = true; if (lane == 0) wg_flags[0] = 1; break; }
                    if (got == CLAIM_LATER) { __builtin_amdgcn_s_sleep(1); break; }  // another wave is fetching the workgroup's next chunk
                }
                const uint32_t avail = c_end - c_next;
                const uint32_t my = (uint32_t)lane_rank_in(need);
                if (!have && my < avail) {
                    ray_idx = c_next + my;
                    const float4 a = rq[(size_t)ray_idx * 2], b = rq[(size_t)ray_idx * 2 + 1];
                    ray_bounce = (int)fbits(b.w);
                    if (ray_idx < n_res) wf_unpark(W, S, T, iter & 1, ray_idx);
                    else wf_take_prepared(W, T, iter & 1, ray_idx, a, b);
                    have = true;
                }
                const uint32_t wanted = (uint32_t)__popcll(need);
                c_next += wanted < avail ? wanted : avail;
            }
            if (exhausted && carry_on && !parked && (__popcll(__ballot(have)) <= W.park_below || wg_flags[1] >= PTK_WF_WG_WAVES / 2)) {
                // the queue is dry: park what is still searching, unless the path is too far behind already or
                // part of its stack lives in the spill column
                parked = true;
                if (lane == 0) __hip_atomic_fetch_add((__attribute__((address_space(3))) int*)wg_flags + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const bool park = have && T.cur != ENTRY_NONE && T.sp <= S.lds_cap && iter - ray_bounce < W.max_lag;
                const unsigned long long pm = __ballot(park);
                if (pm != 0ull) {
                    uint32_t base = 0;
                    if (lane == 0) base = atomicAdd(W.n_resume + iter + 1, (uint32_t)__popcll(pm));
                    base = __builtin_amdgcn_readfirstlane(base);
                    if (park) {
                        const uint32_t at = base + (uint32_t)lane_rank_in(pm);
                        if (at < W.carry_cap) {
                            wf_park(W, S, T, iter, ray_idx, at);
                            have = false;
                        }
                    }
                }
            }
            PTK_ST_ROUND_END(2);
            if (exhausted && __ballot(have) == 0ull) {
                if (!parked && lane == 0) __hip_atomic_fetch_add((__attribute__((address_space(3))) int*)wg_flags + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                break;
            }
        }
        // ---- search rounds until lanes have run out of work (or the wave is to park what it has) ----
        for (;;) {
            PTK_ST_ROUND_BEGIN();
            const bool is_leaf = have && trav_is_leaf(T);
            const bool is_inner = have && trav_is_inner(T);
            const unsigned long long mi = __ballot(is_inner), ml = __ballot(is_leaf);
            const int n_idle = 64 - __popcll(mi | ml);
            // The queue is dry (this wave found out at a refill, or another wave of the workgroup did) and few searches
            // are left here, or half of the workgroup's waves are through: time to park what is left
            // (a wave with part of a claimed chunk left still has rays to hand to its lanes)
            if (carry_on && !exhausted && c_next == c_end && wg_flags[0] != 0) exhausted = true;
            const bool park_now = exhausted && carry_on && !parked &&
                                  (64 - n_idle <= W.park_below || wg_flags[1] >= PTK_WF_WG_WAVES / 2);
            if ((mi | ml) == 0ull || park_now || (!exhausted && n_idle >= PTK_WF_REFILL)) break;

            const bool inner_round = PTK_WF_VOTE_INNER(__popcll(mi), __popcll(ml));
#if defined(PTK_WF_STATS) && !defined(PTK_WF_CLOCKS_ONLY)
            if (inner_round) {
                wf_stats_inner_round(P, T, is_inner, mi, lane);
                if (lane == 0) { atomicAdd(&g_wf_stats[267], (unsigned long long)__popcll(ml)); atomicAdd(&g_wf_stats[268], (unsigned long long)n_idle); }
            } else if (lane == 0) {
                atomicAdd(&g_wf_stats[265], 1ull); atomicAdd(&g_wf_stats[266], (unsigned long long)__popcll(ml));
                atomicAdd(&g_wf_stats[269], (unsigned long long)__popcll(mi)); atomicAdd(&g_wf_stats[270], (unsigned long long)n_idle);
            }
#endif
            bool pop = false;
            if (inner_round) {
                if (is_inner) pop = trav_inner(P, S, T);  // ---- inner round ----
            } else {
                trav_leaf(P, S, T, is_leaf);                 // ---- leaf round ----
                pop = is_leaf;
            }
            if (pop) trav_pop(S, T);
            PTK_ST_ROUND_END(inner_round ? 0 : 1);
        }
    }
#ifdef PTK_WF_STATS
    if (lane == 0) for (int k = 0; k < 3; k++) { atomicAdd(&g_wf_stats[272 + k], (unsigned long long)st_clk[k]); atomicAdd(&g_wf_stats[276 + k], (unsigned long long)st_n[k]); }
#endif
#ifdef PTK_WF_ENDS
    if (lane == 0 && iter < PTK_WF_ENDS_ITERS) g_wf_ends[ends_at + 1] = wall_clock64();
#endif
}

// The bounce-loop body after the closest-hit search (:1583-1621) for every ray of bounce `iter`, and
// the end of main() (:1683-1704) for the paths that stop here.
// (held to the registers of six waves per SIMD: the kernel waits on dependent gathers -- hit, primitive, material,
// texture -- so more waves in flight pay; measured 5 / 6 / 7 / 8 waves: 9.03, 8.93, 9.00, 9.15 ms per step)
__global__ __launch_bounds__(SHADE_BLOCK) __attribute__((amdgpu_waves_per_eu(6))) void ptk_wf_shade(const PtkParams P, int iter) {
    __shared__ uint32_t s_wave_total[SHADE_BLOCK / 64];
    __shared__ uint32_t s_base;
    const PtkWavefront& W = P.wf;
    const uint32_t count = W.n_resume[iter] + W.counts[iter];
    if (blockIdx.x * (uint32_t)SHADE_BLOCK >= count) return;  // the whole block is past the queue's end
    const uint32_t idx = blockIdx.x * (uint32_t)SHADE_BLOCK + threadIdx.x;
    const bool track_inside = P.refraction_mode == 2;
    uint32_t slot = 0;
    Ray nr;
    nr.o = mk3(0, 0, 0); nr.d = mk3(0, 0, 1); nr.start = 0.0f; nr.end = 0.0f;
    Lane L;
    L.seed = 0.0f; L.current_IOR = 1.0f; L.inside_of_count = 0; L.refraction_bounces = 0;
    float4 new_color = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    bool inside_stored = false;
    bool going = false;
    int bounce = 0;  // the path's own bounce number (a parked path is behind the launch number)
    // (hit == -2: the search was parked for the next launch, nothing to shade yet)
    if (idx < count && W.hits[idx].x != -2) {
        const float4 qa = W.rays[iter & 1][(size_t)idx * 2], qb = W.rays[iter & 1][(size_t)idx * 2 + 1];
        slot = fbits(qa.w);
        bounce = (int)fbits(qb.w);
        const int4 hr = W.hits[idx];
        const float2 uv = W.hit_uv[idx];
        nr.o = mk3(qa.x, qa.y, qa.z);
        nr.start = 0.0f;
        nr.d = mk3(qb.x, qb.y, qb.z);
        nr.end = bitsf((uint32_t)hr.w);
        Hit h;
        h.hit = hr.x; h.object_id = hr.y; h.transform_index = hr.z; h.u = uv.x; h.v = uv.y;

        const float4 st0 = W.state[iter & 1][(size_t)idx * 2], st1 = W.state[iter & 1][(size_t)idx * 2 + 1];
        L.seed = st1.x;
        L.current_IOR = st1.y;
        L.inside_of_count = (int)fbits(st1.z);
        const int misc = (int)fbits(st1.w);
        L.refraction_bounces = misc & 0x3FFFFFFF;
        inside_stored = ((misc >> 30) & 1) != 0;
        L.is_emissive = 0; L.ignore_attenuation = 0; L.node_hit_count = 0; L.object_hit_count = 0;
        const bool track_inside = P.refraction_mode == 2;
    #pragma unroll
        for (int i = 0; i < MAX_OVERLAP; i++) {
            L.in_mat[i] = 0;
            L.in_pri[i] = INT_NEG_LIMIT;
            if (track_inside && inside_stored) {
                L.in_mat[i] = W.inside[(size_t)i * W.n_slots + slot];
                L.in_pri[i] = W.inside[(size_t)(MAX_OVERLAP + i) * W.n_slots + slot];
            }
        }

        new_color = st0;
        bool done = false;
        if (!h.hit) {
            const float4 sky = P.use_emission ? make_float4(0.0f, 0.0f, 0.0f, 0.0f) : make_float4(0.7f, 0.7f, 0.9f, 1.0f);
            new_color = make_float4(new_color.x * sky.x, new_color.y * sky.y, new_color.z * sky.z, new_color.w * sky.w);
            done = true;
        } else {
            Surf s = determine_rayhit<false>(P, L, nr, h, nullptr);
            if (P.use_emission && L.is_emissive) {
                new_color = make_float4(new_color.x * s.albedo.x, new_color.y * s.albedo.y, new_color.z * s.albedo.z, new_color.w * 1.0f);
                done = true;
            } else {
                const float4 tex_col = get_texture(P, s);
                nr = bounce_ray(P, L, nr, s);
                if (!L.ignore_attenuation) {
                    new_color.x *= s.albedo.x * tex_col.x;
                    new_color.y *= s.albedo.y * tex_col.y;
                    new_color.z *= s.albedo.z * tex_col.z;
                    new_color.w *= 1.0f * tex_col.w;
                }
                done = !(bounce + 1 < P.push.max_default_depth + L.refraction_bounces);
            }
        }

        if (done) {
            if (P.use_emission && !L.is_emissive) new_color = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            int gx, gy;
            slot_to_pixel(P, slot, gx, gy);
            store_pixel(P, gx + P.push.image_offset_x, gy + P.push.image_offset_y, new_color, false, new_color);
        } else {
            going = true;
        }
    }

    // Paths that go on queue their next ray together with the path state.  The block reserves its
    // range of the next queue with one atomic (one per wave would serialise on the counter's address),
    // and keeps the order of its rays.
    const unsigned long long mask = __ballot(going);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) s_wave_total[wave] = (uint32_t)__popcll(mask);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
#pragma unroll
        for (int w = 0; w < SHADE_BLOCK / 64; w++) total += s_wave_total[w];
        s_base = total ? atomicAdd(W.counts + iter + 1, total) : 0u;
    }
    __syncthreads();
    if (!going) return;
    uint32_t pos = W.n_resume[iter + 1] + s_base + (uint32_t)lane_rank_in(mask);  // behind what the traversal parked there
    for (int w = 0; w < wave; w++) pos += s_wave_total[w];

    bool inside_default = true;
#pragma unroll
    for (int i = 0; i < MAX_OVERLAP; i++) inside_default = inside_default && L.in_mat[i] == 0 && L.in_pri[i] == INT_NEG_LIMIT;
    const bool store_inside = track_inside && (inside_stored || !inside_default);
    if (store_inside) {
#pragma unroll
        for (int i = 0; i < MAX_OVERLAP; i++) {
            W.inside[(size_t)i * W.n_slots + slot] = L.in_mat[i];
            W.inside[(size_t)(MAX_OVERLAP + i) * W.n_slots + slot] = L.in_pri[i];
        }
    }
    float4* out = W.rays[(iter + 1) & 1];
    out[(size_t)pos * 2] = make_float4(nr.o.x, nr.o.y, nr.o.z, bitsf(slot));
    out[(size_t)pos * 2 + 1] = make_float4(nr.d.x, nr.d.y, nr.d.z, bitsf((uint32_t)(bounce + 1)));
    float4* sout = W.state[(iter + 1) & 1];
    sout[(size_t)pos * 2] = new_color;
    sout[(size_t)pos * 2 + 1] = make_float4(L.seed, L.current_IOR, bitsf((uint32_t)L.inside_of_count),
                                            bitsf((uint32_t)((L.refraction_bounces & 0x3FFFFFFF) | (store_inside ? (1 << 30) : 0))));
    wf_prepare(P, (iter + 1) & 1, pos, nr.o, nr.d);
}

extern "C" hipError_t ptk_wavefront_occupancy(size_t lds_bytes, int* wgs_per_cu) {
    int n = 0;
    // a workgroup may use up to the whole 160 KiB of a compute unit
    hipError_t e = hipFuncSetAttribute((const void*)ptk_wf_traverse, hipFuncAttributeMaxDynamicSharedMemorySize, PTK_LDS_BYTES_PER_CU);
    if (e != hipSuccess) return e;
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, ptk_wf_traverse, PTK_WF_WG_WAVES * 64, lds_bytes);
    if (e != hipSuccess) return e;
    *wgs_per_cu = n;
    return hipSuccess;
}

extern "C" hipError_t ptk_launch_wavefront(const PtkParams* p, int gx, int gy, int n_iter, size_t lds_bytes,
                                           int n_wgs, hipEvent_t* ev, hipStream_t stream) {
    const unsigned blocks = (unsigned)gx * (unsigned)gy;
    hipError_t e = hipMemsetAsync(p->wf.counts, 0, PTK_WF_HEADER_BYTES, stream);  // counts, parked counts, heads
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(ptk_wf_generate, dim3(blocks), dim3(64), 0, stream, *p);
    for (int it = 0; it < n_iter; it++) {
        if (ev) { e = hipEventRecord(ev[2 * it], stream); if (e != hipSuccess) return e; }
        hipLaunchKernelGGL(ptk_wf_traverse, dim3((unsigned)n_wgs), dim3(PTK_WF_WG_WAVES * 64), lds_bytes, stream, *p, it);
        if (ev) { e = hipEventRecord(ev[2 * it + 1], stream); if (e != hipSuccess) return e; }
        hipLaunchKernelGGL(ptk_wf_shade, dim3((blocks * 64u + SHADE_BLOCK - 1) / SHADE_BLOCK), dim3(SHADE_BLOCK), 0, stream, *p, it);
    }
    return hipGetLastError();
}
