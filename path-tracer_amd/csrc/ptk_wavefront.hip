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

#include "ptk_render_core.h"

namespace {

typedef float v2f __attribute__((ext_vector_type(2)));

constexpr int SHADE_BLOCK = 512;    // threads per shading block (one queue reservation per block)
constexpr uint32_t WF_CHUNK = 64;  // rays a wave claims from the queue per atomic
#ifndef PTK_WF_VOTE_INNER
#define PTK_WF_VOTE_INNER(ni, nl) ((ni) >= (nl))  // which kind of round runs next: the larger group's
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

__global__ __launch_bounds__(64) void ptk_wf_generate(const PtkParams P) {
    const PtkWavefront& W = P.wf;
    const uint32_t slot = blockIdx.x * 64u + threadIdx.x;
    if (slot == 0) W.counts[0] = W.n_slots;
    int gx, gy;
    slot_to_pixel(P, slot, gx, gy);
    const Primary pr = primary_ray(P.push, gx, gy);
    W.rays[0][(size_t)slot * 2] = make_float4(pr.ray.o.x, pr.ray.o.y, pr.ray.o.z, bitsf(slot));
    W.rays[0][(size_t)slot * 2 + 1] = make_float4(pr.ray.d.x, pr.ray.d.y, pr.ray.d.z, 0.0f);
    W.state[0][(size_t)slot * 2] = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
    W.state[0][(size_t)slot * 2 + 1] = make_float4(pr.seed, 1.0f, bitsf(0u), bitsf(0u));
}

// Closest hit of every ray of bounce `iter`.  Per lane this is check_ray_hit_bvh2 of
// ptk_render_core.h (same order of box and object tests); around it, lanes whose ray is finished
// write the hit record and claim the next queue entry.
__global__ __launch_bounds__(64) void ptk_wf_traverse(const PtkParams P, int iter) {
    extern __shared__ int lds_stack[];
    const int lane = threadIdx.x;
    const PtkWavefront& W = P.wf;
    Stack S;
    S.lds = lds_stack; S.lane = lane; S.lds_cap = P.lds_cap; S.spill_stride = P.spill_stride;
    S.spill = P.spill ? P.spill + (size_t)blockIdx.x * 64 + lane : nullptr;
    int* const lds_lane = lds_stack + lane;

    const uint32_t count = __builtin_amdgcn_readfirstlane(W.counts[iter]);
    const float4* __restrict__ rq = W.rays[iter & 1];
    uint32_t c_next = 0, c_end = 0;  // this wave's claimed queue range
    bool exhausted = count == 0;

    bool have = false;
    uint32_t ray_idx = 0;
    Ray ray;
    ray.o = mk3(0, 0, 0); ray.d = mk3(0, 0, 1); ray.start = 0.0f; ray.end = 0.0f;
    Hit h;
    h.hit = 0; h.object_id = 0; h.transform_index = -1; h.u = 0.0f; h.v = 0.0f;
    f3 inv = mk3(0, 0, 0);
    int neg = 0;
    uint32_t cur = ENTRY_NONE;
    int sp = 0;
    bool in_mesh = false;
    int sub_base = 0, cur_xf = -1;
    f3 w_o = mk3(0, 0, 0), w_d = mk3(0, 0, 1);
    float w_start = 0.0f, dir_len = 1.0f;

    for (;;) {
        const bool has = have && cur != ENTRY_NONE;
        const bool is_leaf = has && ((int)cur <= 0);
        const bool is_inner = has && !is_leaf;
        const unsigned long long mi = __ballot(is_inner), ml = __ballot(is_leaf);
        const int n_idle = 64 - __popcll(mi | ml);

#ifdef PTK_ROUND_STATS
#define PTK_STAT(k, v) do { if (lane == 0) atomicAdd((unsigned long long*)(W.counts + 128) + (k), (unsigned long long)(v)); } while (0)
#else
#define PTK_STAT(k, v) do { } while (0)
#endif
        if ((mi | ml) == 0ull || (!exhausted && n_idle >= PTK_WF_REFILL)) {
            PTK_STAT(0, 1); PTK_STAT(1, n_idle);
            // ---- refill round: retire finished rays, claim new ones ----
            if (have && !has) {
                W.hits[ray_idx] = make_int4(h.hit, h.object_id, h.transform_index, (int)fbits(ray.end));
                W.hit_uv[ray_idx] = make_float2(h.u, h.v);
                have = false;
            }
#pragma unroll 1
            for (int pass = 0; pass < 2 && !exhausted; pass++) {
                const unsigned long long need = __ballot(!have);
                if (need == 0ull) break;
                if (c_next == c_end) {
                    uint32_t base = 0;
                    if (lane == 0) base = atomicAdd(W.heads + iter, WF_CHUNK);
                    base = __builtin_amdgcn_readfirstlane(base);
                    c_next = base < count ? base : count;
                    c_end = (base + WF_CHUNK) < count ? (base + WF_CHUNK) : count;
                    if (c_next >= c_end) { exhausted = true; break; }
                }
                const uint32_t avail = c_end - c_next;
                const uint32_t my = (uint32_t)lane_rank_in(need);
                if (!have && my < avail) {
                    ray_idx = c_next + my;
                    const float4 a = rq[(size_t)ray_idx * 2], b = rq[(size_t)ray_idx * 2 + 1];
                    ray.o = mk3(a.x, a.y, a.z);
                    ray.d = mk3(b.x, b.y, b.z);
                    ray.start = 0.0f;
                    ray.end = PT_INF;
                    h.hit = 0; h.object_id = 0; h.transform_index = -1; h.u = 0.0f; h.v = 0.0f;
                    // planes first, outside the BVH (:1204); only records with a non-zero normal can hit
                    for (int i = 0; i < P.n_live_planes; ++i) hit_plane(P, ray, h, P.live_planes[i], -1);
                    inv = mk3(1.0f / ray.d.x, 1.0f / ray.d.y, 1.0f / ray.d.z);
                    neg = (inv.x < 0.0f ? 1 : 0) | (inv.y < 0.0f ? 2 : 0) | (inv.z < 0.0f ? 4 : 0);
                    const float4 rmin = make_float4(W.root_min[0], W.root_min[1], W.root_min[2], 0.0f);
                    const float4 rmax = make_float4(W.root_max[0], W.root_max[1], W.root_max[2], 0.0f);
                    const float root_hit = hit_aabb_bl(rmin, rmax, ray.o, inv, neg, ray.start, ray.end);
                    cur = in_range(root_hit, ray.start, ray.end) ? W.root_entry : ENTRY_NONE;
                    sp = 0;
                    in_mesh = false;
                    cur_xf = -1;
                    have = true;
                }
                const uint32_t wanted = (uint32_t)__popcll(need);
                c_next += wanted < avail ? wanted : avail;
            }
            if (exhausted && __ballot(have) == 0ull) break;
            continue;
        }

        bool refill = false;
        if (PTK_WF_VOTE_INNER(__popcll(mi), __popcll(ml))) {
            // ---- inner round ----
            PTK_STAT(2, 1); PTK_STAT(3, __popcll(mi)); PTK_STAT(4, __popcll(__ballot(is_inner && (cur & 0xFF000000u) == SOCKET_MARK)));
            if (is_inner) {
#ifdef PTK_CHECKED
                {
                    const bool is_sock = (cur & 0xFF000000u) == SOCKET_MARK;
                    const uint32_t ix = cur & 0x00FFFFFFu;
                    const bool bad = is_sock ? ix >= W.n_sockets : (ix >= W.n_blocks || (cur >> 24) > 8u || sp < 0 || sp > 1024);
                    if (bad) { atomicOr(W.counts + 200, 1u); W.counts[201] = cur; W.counts[202] = (uint32_t)sp; }
                }
                if (((cur & 0xFF000000u) == SOCKET_MARK) ? (cur & 0x00FFFFFFu) >= W.n_sockets : (cur & 0x00FFFFFFu) >= W.n_blocks) {
                    cur = ENTRY_NONE; sp = 0; in_mesh = false;
                } else
#endif
                if ((cur & 0xFF000000u) == SOCKET_MARK) {
                    // world level: enter the mesh (:1286-1313); its root is the next node
                    const int2 sock = W.sockets[cur & 0x00FFFFFFu];
                    const int xf = sock.y;
                    const m44 M = load_xform(P, xf);
                    w_o = ray.o; w_d = ray.d; w_start = ray.start;
                    const f3 unnorm = mul_mv(M, ray.d, 0.0f);
                    dir_len = length3(unnorm);
                    const f3 no = mul_mv(M, ray.o, 1.0f);
                    ray.d = div3(unnorm, dir_len);
                    ray.o = no;
                    ray.end = ray.end * dir_len;
                    ray.start = ray.start * dir_len;
                    inv = mk3(1.0f / ray.d.x, 1.0f / ray.d.y, 1.0f / ray.d.z);
                    neg = (inv.x < 0.0f ? 1 : 0) | (inv.y < 0.0f ? 2 : 0) | (inv.z < 0.0f ? 4 : 0);
                    sub_base = sp;
                    cur_xf = xf;
                    in_mesh = true;
                    cur = (uint32_t)sock.x;
                } else {
                    const int node_size = (int)(cur >> 24);
                    // the node's block: eight child slots as planes (minx[8] .. maxz[8], entry[8]), all
                    // fetched before any is used; slots past node_size hold zeros and are ignored
                    const float4* __restrict__ cb = W.blocks + (size_t)(cur & 0x00FFFFFFu) * 16;
                    float4 q[14];
#pragma unroll
                    for (int k = 0; k < 14; k++) q[k] = cb[k];
                    __builtin_amdgcn_sched_barrier(0);
                    const bool nx = (neg & 1) != 0, ny = (neg & 2) != 0, nz = (neg & 4) != 0;
                    // a NaN in the ray's own range (singular mesh transform) fails in_range for every box
                    const bool range_ok = ray.start == ray.start && ray.end == ray.end;
                    // key = bits of the entry distance t (t >= +0, so unsigned integer order is float
                    // order); rejected children get the bits of +inf, which no accepted t reaches
                    uint32_t key[PTK_MAX_WIDTH];
                    uint32_t ent[PTK_MAX_WIDTH];
                    int added = 0;
#pragma unroll
                    for (int c = 0; c < PTK_MAX_WIDTH / 2; c++) {
                        // hit_AABB (:670-689) for children 2c and 2c+1 side by side in packed fp32:
                        // per axis t0 = (min - o) * inv, t1 = (max - o) * inv, the arithmetic swap when
                        // inv < 0, then start = max, end = min.  start and end are never NaN and only
                        // move inwards, so testing end <= start once after the last axis equals the
                        // shader's three early returns, and a box that passes has its t inside
                        // [ray.start, ray.end] already.
                        const int h4 = c >> 1;         // which float4 of the plane
                        const bool hi2 = (c & 1) != 0;  // .zw or .xy
#define PTK_PAIR(v) (hi2 ? v2f{(v).z, (v).w} : v2f{(v).x, (v).y})
                        const v2f lox = PTK_PAIR(q[0 + h4]), loy = PTK_PAIR(q[2 + h4]), loz = PTK_PAIR(q[4 + h4]);
                        const v2f hix = PTK_PAIR(q[6 + h4]), hiy = PTK_PAIR(q[8 + h4]), hiz = PTK_PAIR(q[10 + h4]);
                        const v2f en = PTK_PAIR(q[12 + h4]);
#undef PTK_PAIR
                        const v2f t0x = (lox - ray.o.x) * inv.x, t1x = (hix - ray.o.x) * inv.x;
                        const v2f t0y = (loy - ray.o.y) * inv.y, t1y = (hiy - ray.o.y) * inv.y;
                        const v2f t0z = (loz - ray.o.z) * inv.z, t1z = (hiz - ray.o.z) * inv.z;
                        const v2f swx = t0x + t1x, sbx = swx - t1x, scx = swx - sbx;
                        const v2f swy = t0y + t1y, sby = swy - t1y, scy = swy - sby;
                        const v2f swz = t0z + t1z, sbz = swz - t1z, scz = swz - sbz;
#pragma unroll
                        for (int e = 0; e < 2; e++) {
                            const int l = 2 * c + e;
                            float start = ray.start, end = ray.end;
                            start = __builtin_fmaxf(start, nx ? scx[e] : t0x[e]);
                            end = __builtin_fminf(end, nx ? sbx[e] : t1x[e]);
                            start = __builtin_fmaxf(start, ny ? scy[e] : t0y[e]);
                            end = __builtin_fminf(end, ny ? sby[e] : t1y[e]);
                            start = __builtin_fmaxf(start, nz ? scz[e] : t0z[e]);
                            end = __builtin_fminf(end, nz ? sbz[e] : t1z[e]);
                            const bool acc = (l < node_size) && (end > start) && range_ok;
                            key[l] = acc ? (fbits(start) & 0x7FFFFFFFu) : 0x7F800000u;
                            ent[l] = fbits(en[e]);
                            added += acc ? 1 : 0;
                        }
                    }
                    // rank = number of children the shader's selection loop (:1157-1178) would take
                    // earlier: smaller t, or equal t in a lower slot.  For j < l, u = 1 iff child l is
                    // strictly nearer than child j (sign bit of key[l] - key[j]; both keys < 2^31):
                    // rank[l] = l - sum_{j<l} u(j,l) + sum_{m>l} u(l,m).
                    int up[PTK_MAX_WIDTH], dn[PTK_MAX_WIDTH];
#pragma unroll
                    for (int l = 0; l < PTK_MAX_WIDTH; l++) { up[l] = 0; dn[l] = 0; }
#pragma unroll
                    for (int l = 1; l < PTK_MAX_WIDTH; l++) {
#pragma unroll
                        for (int j = 0; j < l; j++) {
                            const int u = (int)((key[l] - key[j]) >> 31);
                            up[j] += u;
                            dn[l] += u;
                        }
                    }
                    // rank r goes to stack slot sp + added - 1 - r, so the nearest child lands in the free
                    // slot above the new top and is read back as `cur`; rejected children (rank >= added)
                    // are sent to a spare row.
                    cur = ENTRY_NONE;
                    const int top = sp + added - 1;
                    if (sp + PTK_MAX_WIDTH <= S.lds_cap) {
#pragma unroll
                        for (int l = 0; l < PTK_MAX_WIDTH; l++) {
                            const int rank = l - dn[l] + up[l];
                            lds_lane[((key[l] != 0x7F800000u) ? top - rank : S.lds_cap) * 64] = (int)ent[l];
                        }
                        if (added > 0) cur = (uint32_t)lds_lane[top * 64];
                    } else {
#pragma unroll
                        for (int l = 0; l < PTK_MAX_WIDTH; l++) {
                            const int rank = l - dn[l] + up[l];
                            if (key[l] != 0x7F800000u) {
                                if (rank == 0) cur = ent[l];
                                else st_write(S, top - rank, (int)ent[l]);
                            }
                        }
                    }
                    sp += (added > 0) ? added - 1 : 0;
                    refill = added == 0;
#ifdef PTK_ROUND_STATS
                    {
                        const int c0 = __popcll(__ballot(added == 0)), c1 = __popcll(__ballot(added == 1)), c2 = __popcll(__ballot(added == 2));
                        const int c3 = __popcll(__ballot(added == 3)), c4 = __popcll(__ballot(added >= 4));
                        const unsigned long long act = __ballot(true);
                        if (lane == __ffsll((long long)act) - 1) {
                            unsigned long long* st = (unsigned long long*)(W.counts + 128);
                            atomicAdd(st + 9, (unsigned long long)c0); atomicAdd(st + 10, (unsigned long long)c1); atomicAdd(st + 11, (unsigned long long)c2);
                            atomicAdd(st + 12, (unsigned long long)c3); atomicAdd(st + 13, (unsigned long long)c4);
                            atomicAdd(st + 14, (unsigned long long)((c3 + c4) > 0)); atomicAdd(st + 15, 1ull);
                        }
                    }
#endif
                }
            }
        } else {
            // ---- leaf round: the objects of the leaf, in order (:1137-1142) ----
            PTK_STAT(5, 1); PTK_STAT(6, __popcll(ml));
            int node_size = 0, base = 0;
            if (is_leaf) {
                node_size = iabs(bfe_i((int)cur, 24, 8));
                base = iabs(bfe_i((int)cur, 0, 24)) - 1;
            }
            // leaf records: {A.xyz | centre.xyz, object id}, {B.xyz | radius, -}, {C.xyz, -}, stored in
            // object-id-slot order, so a leaf's objects are contiguous and need no id lookup first
#ifdef PTK_CHECKED
            if (is_leaf && node_size > 0 && (base < 0 || (uint32_t)(base + node_size) > W.n_records)) {
                atomicOr(W.counts + 200, 2u); W.counts[203] = cur;
                node_size = 0;
            }
#endif
            const float4* __restrict__ rec = W.leaf_records + (size_t)base * 3;
            for (int k = 0;; k++) {
                const bool go = k < node_size;
                if (__ballot(go) == 0ull) break;
                if (go) {
                    const float4 r0 = rec[k * 3], r1 = rec[k * 3 + 1], r2 = rec[k * 3 + 2];
                    const int object_id = (int)fbits(r0.w);
                    const int type = bfe_i(object_id, 24, 8);
                    if (type == IS_TRIANGLE) hit_triangle_rec(ray, h, object_id, cur_xf, ld3(r0), ld3(r1), ld3(r2));
                    else if (type == IS_SPHERE) hit_sphere_rec(ray, h, object_id, cur_xf, ld3(r0), r1.x);
                }
            }
            refill = is_leaf;
        }
        if (refill) {
            // next node of this lane: pop; an empty mesh sub-stack first maps the ray back (:1309-1310)
            cur = ENTRY_NONE;
            if (in_mesh && sp <= sub_base) {
                const float local_end = ray.end;
                ray.o = w_o; ray.d = w_d; ray.start = w_start;
                ray.end = local_end / dir_len;
                inv = mk3(1.0f / ray.d.x, 1.0f / ray.d.y, 1.0f / ray.d.z);
                neg = (inv.x < 0.0f ? 1 : 0) | (inv.y < 0.0f ? 2 : 0) | (inv.z < 0.0f ? 4 : 0);
                in_mesh = false;
                cur_xf = -1;
            }
            if (sp > 0) cur = (uint32_t)st_read(S, --sp);
        }
    }
}

// The bounce-loop body after the closest-hit search (:1583-1621) for every ray of bounce `iter`, and
// the end of main() (:1683-1704) for the paths that stop here.
__global__ __launch_bounds__(SHADE_BLOCK) void ptk_wf_shade(const PtkParams P, int iter) {
    __shared__ uint32_t s_wave_total[SHADE_BLOCK / 64];
    __shared__ uint32_t s_base;
    const PtkWavefront& W = P.wf;
    const uint32_t count = W.counts[iter];
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
    if (idx < count) {
        const float4 qa = W.rays[iter & 1][(size_t)idx * 2], qb = W.rays[iter & 1][(size_t)idx * 2 + 1];
        slot = fbits(qa.w);
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
                done = !(iter + 1 < P.push.max_default_depth + L.refraction_bounces);
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
    uint32_t pos = s_base + (uint32_t)lane_rank_in(mask);
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
    out[(size_t)pos * 2 + 1] = make_float4(nr.d.x, nr.d.y, nr.d.z, 0.0f);
    float4* sout = W.state[(iter + 1) & 1];
    sout[(size_t)pos * 2] = new_color;
    sout[(size_t)pos * 2 + 1] = make_float4(L.seed, L.current_IOR, bitsf((uint32_t)L.inside_of_count),
                                            bitsf((uint32_t)((L.refraction_bounces & 0x3FFFFFFF) | (store_inside ? (1 << 30) : 0))));
}

extern "C" hipError_t ptk_wavefront_occupancy(size_t lds_bytes, int* waves_per_cu) {
    int n = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, ptk_wf_traverse, 64, lds_bytes);
    if (e != hipSuccess) return e;
    *waves_per_cu = n;
    return hipSuccess;
}

extern "C" hipError_t ptk_launch_wavefront(const PtkParams* p, int gx, int gy, int n_iter, size_t lds_bytes,
                                           int n_waves, hipEvent_t* ev, hipStream_t stream) {
    const unsigned blocks = (unsigned)gx * (unsigned)gy;
    hipError_t e = hipMemsetAsync(p->wf.counts, 0, 1024, stream);  // counts, heads (and development counters)
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(ptk_wf_generate, dim3(blocks), dim3(64), 0, stream, *p);
    for (int it = 0; it < n_iter; it++) {
        if (ev) { e = hipEventRecord(ev[2 * it], stream); if (e != hipSuccess) return e; }
        hipLaunchKernelGGL(ptk_wf_traverse, dim3((unsigned)n_waves), dim3(64), lds_bytes, stream, *p, it);
        if (ev) { e = hipEventRecord(ev[2 * it + 1], stream); if (e != hipSuccess) return e; }
        hipLaunchKernelGGL(ptk_wf_shade, dim3((blocks * 64u + SHADE_BLOCK - 1) / SHADE_BLOCK), dim3(SHADE_BLOCK), 0, stream, *p, it);
    }
    return hipGetLastError();
}
