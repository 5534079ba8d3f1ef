// ptk_bvh.hip -- BVH builder on the GPU.
//
// Reproduces, bit for bit, the tree the reference builds on the CPU:
//   proprietary_dataclasses/bvh_trees/pt_bvh_axis_sort.gd:62-84,156-229,269-376  (one node per object,
//       recursive equal-width binning along a fixed or the longest axis)
//   proprietary_dataclasses/bvh_trees/pt_bvh_tree.gd:457-524,780-791,856-870,910-959 (indexing, ids, node bytes)
//   proprietary_dataclasses/pt_utils.gd:27-43,71-97 (AABB validity / merge / byte encoding)
// including the float32 arithmetic of Godot 4.3's AABB / Transform3D and the element order produced by
// Array.sort_custom (core/templates/sort_array.h: introsort, median-of-3 pivot, threshold 16, heapsort
// fallback, final insertion sort), whose tie order is observable in the leaves.
//
// The recursion is run level by level.  All per-element work of a level is done on the device, one
// wavefront per list ("call"):
//   * the list's box is the reference's sequential merge chain (not associative in float32, so it is
//     evaluated in list order, 64 boxes per coalesced load),
//   * the sort is an exact emulation of the introsort: every quicksort level is one launch with one wave
//     per segment; a wave replays the Hoare partition with 64-wide scans from both ends (the swap pairs are
//     the i-th stop from the left with the i-th stop from the right, exactly as the two-pointer loop meets
//     them), small segments end in the same stable insertion sort,
//   * bin indices use the reference's float32-difference / float64-divide, and elements are moved with a
//     stable per-bin ranking (ballot + popcount), which keeps the list order inside a bin.
// The host only keeps the tree's topology (which bin became which node) and launches the next level.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <initializer_list>

#include "ptk_bvh.h"

namespace {

constexpr int MAX_DEPTH_LIMIT = 32;  // pt_bvh_axis_sort.gd:10
constexpr int MAXB = 8;              // widest order the device builder bins for
constexpr int SORT_THRESHOLD = 16;   // SortArray::INTROSORT_THRESHOLD
constexpr int PTK_BVH_SAH_TYPE = 5;   // PTK_BVH_SAH of include/ptk.h

struct A6 {
    float p[3], s[3];
};

#define DEV __device__ __forceinline__

// ---- Godot AABB arithmetic (core/math/aabb.h), float32 ---------------------------------------
DEV A6 a6_zero() { A6 a; for (int i = 0; i < 3; i++) { a.p[i] = 0.0f; a.s[i] = 0.0f; } return a; }
DEV float gmin(float a, float b) { return a < b ? a : b; }
DEV bool a6_valid(const A6& b) {  // pt_utils.gd:27-28
    bool nonzero = !(b.s[0] == 0.0f && b.s[1] == 0.0f && b.s[2] == 0.0f);
    bool same = true, fin = true;
    for (int i = 0; i < 3; i++) {
        float ap = b.p[i] + gmin(b.s[i], 0.0f);
        float as = fabsf(b.s[i]);
        same = same && (ap == b.p[i]) && (as == b.s[i]);
        fin = fin && (fabsf(b.p[i]) < __builtin_inff()) && (fabsf(b.s[i]) < __builtin_inff());
    }
    return nonzero && same && fin;
}
DEV A6 a6_merge(const A6& a, const A6& b) {  // AABB::merge_with
    A6 r;
    for (int i = 0; i < 3; i++) {
        float e1 = a.p[i] + a.s[i], e2 = b.p[i] + b.s[i];
        float mn = a.p[i] < b.p[i] ? a.p[i] : b.p[i];
        float mx = e1 > e2 ? e1 : e2;
        r.p[i] = mn;
        r.s[i] = mx - mn;
    }
    return r;
}
DEV A6 merge_aabb(const A6& a, const A6& b) {  // pt_utils.gd:34-43
    bool va = a6_valid(a), vb = a6_valid(b);
    if (!(va || vb)) return a6_zero();
    if (!va) return b;
    if (!vb) return a;
    return a6_merge(a, b);
}
DEV A6 a6_abs(const A6& a) {
    A6 r;
    for (int i = 0; i < 3; i++) { r.p[i] = a.p[i] + gmin(a.s[i], 0.0f); r.s[i] = fabsf(a.s[i]); }
    return r;
}
DEV A6 a6_grow(A6 a, float by) {
    for (int i = 0; i < 3; i++) { a.p[i] -= by; a.s[i] += 2.0f * by; }
    return a;
}
DEV float a6_center(const A6& a, int axis) { return a.p[axis] + a.s[axis] * 0.5f; }
DEV int a6_longest_axis(const A6& a) {
    int axis = 0;
    float m = a.s[0];
    if (a.s[1] > m) { axis = 1; m = a.s[1]; }
    if (a.s[2] > m) { axis = 2; }
    return axis;
}
DEV A6 a6_xform(const float* t, const A6& a) {  // Transform3D::xform(AABB); t = basis rows + origin
    A6 r;
    for (int i = 0; i < 3; i++) {
        float lo = t[9 + i], hi = t[9 + i];
        for (int j = 0; j < 3; j++) {
            float e = t[i * 3 + j] * a.p[j];
            float f = t[i * 3 + j] * (a.p[j] + a.s[j]);
            if (e < f) { lo += e; hi += f; }
            else { lo += f; hi += e; }
        }
        r.p[i] = lo;
        r.s[i] = hi - lo;
    }
    return r;
}
DEV A6 a6_load(const float* base, int i) {
    A6 a;
    const float* q = base + (size_t)i * 6;
    a.p[0] = q[0]; a.p[1] = q[1]; a.p[2] = q[2]; a.s[0] = q[3]; a.s[1] = q[4]; a.s[2] = q[5];
    return a;
}
DEV void a6_store(float* base, int i, const A6& a) {
    float* q = base + (size_t)i * 6;
    q[0] = a.p[0]; q[1] = a.p[1]; q[2] = a.p[2]; q[3] = a.s[0]; q[4] = a.s[1]; q[5] = a.s[2];
}
DEV float rdlane(float v, int lane) {  // value of v in a lane chosen uniformly for the whole wave
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
DEV A6 a6_shfl(const A6& a, int lane) {
    A6 r;
    for (int i = 0; i < 3; i++) { r.p[i] = __shfl(a.p[i], lane); r.s[i] = __shfl(a.s[i], lane); }
    return r;
}

struct Call {
    int begin, end;  // range in perm
};
struct CallOut {
    float aabb[6];   // list box after abs().grow(1e-6)
    int axis;
    int mesh_count;
    int counts[MAXB];
    int error;
    int pad[3];
};

// ---- queues of the sort (an emulation of Godot's SortArray, further below) ---------------------------
struct Seg {
    int first, last;  // range of the list
    int depth;        // introsort's remaining depth budget
    int axis;         // the keys are the boxes' centres on this axis
};
constexpr int SORT_BIG = 512;          // segments longer than this are partitioned by a whole 1024-thread workgroup
constexpr int SORT_RANK_MAX = 8192;    // longest segment whose heapsort fallback is done by counting
constexpr int SORT_LEVEL_GROUP = 8;    // long-segment levels launched per host round trip
constexpr int SORT_MAX_LEVELS = 72;    // more than the largest depth budget, 2 * bitlog(2^31)
constexpr int SORT_COUNTERS = 2 + SORT_MAX_LEVELS + SORT_LEVEL_GROUP;  // [0] short segments, [2 + l] long segments of level l

// a segment that still needs work goes to the queue of the long or of the short segments
DEV void sort_emit(Seg part, Seg* small, int* small_count, Seg* big, int* big_count) {
    const int len = part.last - part.first;
    if (len > SORT_BIG) big[atomicAdd(big_count, 1)] = part;
    else if (len > 1) small[atomicAdd(small_count, 1)] = part;
}
// SortArray::sort's depth budget: 2 * bitlog(length)
DEV int sort_depth_budget(int len) { return 2 * (31 - __clz(len)); }

// ---- summaries of a long list's batches ------------------------------------------------------------
// The merge chain of a list runs in list order on one wave (k_list_box below), 64 elements to a batch, and in a long
// unordered list almost every batch moves neither end of the box: it is passed over once a few comparisons have shown
// that -- every element's position above the box's, every far end below the box's, every size non-negative and not
// all three zero.  Those comparisons only need the batch's smallest position and largest far end per axis, which do
// not depend on the chain: they are made here, by as many waves as the chip has, so that the one wave of the chain
// reads 32 bytes per batch instead of gathering its 64 boxes.  summary[b] = {min position xyz, max far end xyz,
// shape ok (every size >= 0, not all three zero, nothing NaN), -}; mesh sockets among the full batches are counted
// into call_meshes[c].  Only full batches have a summary; a list's last, partial batch is always worked on in full.
constexpr int LIST_SUMMARY_MIN = 2048;  // lists at least this long get summaries
__global__ __launch_bounds__(256) void k_list_summary(const Call* calls, int n_calls, const int* summ_off, const int* perm, const float* in_aabb,
                                                      const int* in_kind, float* summary, int* call_meshes) {
    const int c = blockIdx.x;
    if (c >= n_calls) return;
    const int off = summ_off[c];
    if (off < 0) return;
    const Call call = calls[c];
    const int n_full = (call.end - call.begin) / 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int meshes = 0;
    for (int b = blockIdx.y * 4 + wave; b < n_full; b += gridDim.y * 4) {
        const int id = perm[call.begin + b * 64 + lane];
        const A6 m = a6_load(in_aabb, id);
        meshes += __popcll(__ballot(in_kind[id] == 1));
        float lo[3], hi[3];
        bool ok = !(m.s[0] == 0.0f && m.s[1] == 0.0f && m.s[2] == 0.0f);
        for (int a = 0; a < 3; a++) {
            lo[a] = m.p[a];
            hi[a] = m.p[a] + m.s[a];
            ok = ok && m.s[a] >= 0.0f && lo[a] == lo[a] && hi[a] == hi[a];
        }
        for (int o = 32; o > 0; o >>= 1) {
            for (int a = 0; a < 3; a++) {
                const float l2 = __shfl_xor(lo[a], o), h2 = __shfl_xor(hi[a], o);
                lo[a] = l2 < lo[a] ? l2 : lo[a];
                hi[a] = h2 > hi[a] ? h2 : hi[a];
            }
        }
        const bool all_ok = __all(ok);
        if (lane == 0) {
            float* q = summary + (size_t)(off + b) * 8;
            q[0] = lo[0]; q[1] = lo[1]; q[2] = lo[2]; q[3] = hi[0]; q[4] = hi[1]; q[5] = hi[2];
            q[6] = all_ok ? 1.0f : 0.0f; q[7] = 0.0f;
        }
    }
    if (lane == 0 && meshes) atomicAdd(call_meshes + c, meshes);
}

// ---- per-call kernels (one wave per call) ------------------------------------------------------
// list box: sequential merge chain in list order (get_array_aabb, pt_bvh_axis_sort.gd:512-516), then
// .abs().grow(1e-6) (:277); also counts mesh sockets and picks the axis (:279-280).
// With sort_small given (longest-axis trees, which sort every list): also queues the call's range for the sort
// (counters sort_now) and clears the counters the next level will use (sort_next).
__global__ __launch_bounds__(64) void k_list_box(const Call* calls, int n_calls, const int* perm, const float* in_aabb,
                                                 const int* in_kind, int type, int fixed_axis, CallOut* out, Seg* sort_small,
                                                 Seg* sort_big, int* sort_now, int* sort_next,
                                                 const int* summ_off, const float* summary, const int* call_meshes) {
    __shared__ __align__(16) float tr[6][64];
    int c = blockIdx.x;
    if (c >= n_calls) return;
    int lane = threadIdx.x;
    Call call = calls[c];
    A6 acc = a6_zero();
    bool acc_valid = false;  // a6_valid(acc), kept alongside (uniform over the wave)
    int meshes = 0;
    // The boxes are gathered through perm (two dependent loads); the loads of the next LIST_UNROLL batches are in flight
    // while the current ones are worked on, and the indices of the LIST_UNROLL after them.  (Taking 2 or 4 batches per
    // turn of the loop measured no faster than 1: a batch that changes nothing costs a handful of comparisons, one that
    // does, the chain below.)
#ifndef PTK_LIST_UNROLL
#define PTK_LIST_UNROLL 1
#endif
    constexpr int LIST_UNROLL = PTK_LIST_UNROLL;
    A6 ahead_box[LIST_UNROLL];
    int ahead_kind[LIST_UNROLL], ahead_id[LIST_UNROLL];
#pragma unroll
    for (int d = 0; d < LIST_UNROLL; d++) {
        const int i = call.begin + d * 64 + lane;
        ahead_box[d] = a6_zero();
        ahead_kind[d] = 0;
        if (i < call.end) { const int id = perm[i]; ahead_box[d] = a6_load(in_aabb, id); ahead_kind[d] = in_kind[id]; }
        ahead_id[d] = i + LIST_UNROLL * 64 < call.end ? perm[i + LIST_UNROLL * 64] : -1;
    }
    // The far end of the box as it stands and whether its size has settled (see the skipped batches below); kept
    // beside acc so that a batch which changes nothing is recognised with a handful of comparisons.
    float far_x = 0.0f, far_y = 0.0f, far_z = 0.0f;
    bool settled = false;
    auto note_box = [&]() {
        far_x = acc.p[0] + acc.s[0]; far_y = acc.p[1] + acc.s[1]; far_z = acc.p[2] + acc.s[2];
        settled = acc_valid && __float_as_uint(far_x - acc.p[0]) == __float_as_uint(acc.s[0]) &&
                  __float_as_uint(far_y - acc.p[1]) == __float_as_uint(acc.s[1]) && __float_as_uint(far_z - acc.p[2]) == __float_as_uint(acc.s[2]);
    };
    // one batch of 64 list elements, `mine` this lane's
    auto batch = [&](const int base, const A6& mine, const int kind) {
        const int i = base + lane;
        const bool in = i < call.end;
        meshes += __popcll(__ballot(in && kind == 1));
        if (settled && base + 64 <= call.end) {
            // the common case first: a full batch of valid boxes strictly inside the box (valid: finite, no negative
            // size, not all sizes zero -- a6_valid, given that the comparisons with the box's finite ends hold)
            const bool quiet = mine.s[0] >= 0.0f && mine.s[1] >= 0.0f && mine.s[2] >= 0.0f &&
                               !(mine.s[0] == 0.0f && mine.s[1] == 0.0f && mine.s[2] == 0.0f) &&
                               mine.p[0] > acc.p[0] && mine.p[1] > acc.p[1] && mine.p[2] > acc.p[2] &&
                               mine.p[0] + mine.s[0] < far_x && mine.p[1] + mine.s[1] < far_y && mine.p[2] + mine.s[2] < far_z;
            if (__all(quiet)) return;
        }
        const int cnt = min(64, call.end - base);
        const unsigned long long vmask = __ballot(in && a6_valid(mine));
        int j0 = 0;
        if (!acc_valid) {
            // invalid merged with invalid is the zero box; the first valid element replaces it (pt_utils.gd:34-43)
            if (vmask == 0ull) { acc = a6_zero(); note_box(); return; }
            j0 = __ffsll((long long)vmask) - 1;
            acc = a6_shfl(mine, j0);
            acc_valid = true;
            j0++;
        } else if (vmask == 0ull) {
            return;  // a valid box merged with invalid ones stays as it is
        }
        // Fast chain: while the accumulated box stays valid, a step is  e1 = p + s;  p' = min(p, bp);
        // s' = max(e1, bp + bs) - p'  per axis, in list order (the float merge is not associative, so the
        // order is kept; what is saved is the validity test of both operands inside every step).  A valid
        // box has s >= 0 and stays so; it can only turn invalid by becoming non-finite (which then sticks
        // until the end of the batch) or by all three sizes rounding to zero at one step.
        // If either happened, the batch is redone with the full merge.
        float e2x = mine.p[0] + mine.s[0], e2y = mine.p[1] + mine.s[1], e2z = mine.p[2] + mine.s[2];
        float px = acc.p[0], py = acc.p[1], pz = acc.p[2], sx = acc.s[0], sy = acc.s[1], sz = acc.s[2];
        bool collapsed = false;
        bool fast = false;
        float mx = mine.p[0], my = mine.p[1], mz = mine.p[2];
        if (j0 == 0 && vmask == ~0ull) {
            // A batch that moves neither end of the box -- every element's position strictly above the box's and
            // its far end strictly below the box's, on all three axes -- leaves p as it is and turns s into
            // (p + s) - p at every step: nothing at all if s is that already.  In a long unordered list almost
            // every batch is of this kind (the running extremes of n elements change about ln n times).
            {
                const float e1x = px + sx, e1y = py + sy, e1z = pz + sz;
                const bool inside = mx > px && my > py && mz > pz && e2x < e1x && e2y < e1y && e2z < e1z;
                const bool settled_now = __float_as_uint(e1x - px) == __float_as_uint(sx) && __float_as_uint(e1y - py) == __float_as_uint(sy) &&
                                         __float_as_uint(e1z - pz) == __float_as_uint(sz);
                if (settled_now && __all(inside)) return;
            }
            // A full batch of valid boxes.  The position after element j is the running "p < bp ? p : bp" over the
            // elements up to j, an associative pick (smallest value, the later one among equals) that does not
            // depend on the sizes: all 64 are found at once by a scan.
            for (int o = 1; o < 64; o <<= 1) {
                const float ux = __shfl_up(mx, o), uy = __shfl_up(my, o), uz = __shfl_up(mz, o);
                if (lane >= o) { mx = ux < mx ? ux : mx; my = uy < my ? uy : my; mz = uz < mz ? uz : mz; }
            }
            mx = px < mx ? px : mx; my = py < my ? py : my; mz = pz < mz ? pz : mz;
            // A size is max(e1, far end) - position with far end >= position, so it is zero only where an
            // element's far end equals the position after it; all three sizes zero at one element (the box turning
            // invalid) needs that on all three axes.  Batches where this can happen take the general loop below.
            fast = !__any(e2x == mx && e2y == my && e2z == mz);
        }
        if (fast) {
            // what is left is the chain of sizes, three dependent operations per element and axis:
            // lanes 0..2 run one axis each over the batch transposed through LDS
            tr[0][lane] = mx; tr[1][lane] = my; tr[2][lane] = mz;
            tr[3][lane] = e2x; tr[4][lane] = e2y; tr[5][lane] = e2z;
            __builtin_amdgcn_wave_barrier();  // one wave: its LDS accesses execute in program order
            const int a = lane < 3 ? lane : 0;
            float pprev = a == 0 ? px : (a == 1 ? py : pz);
            float sa = a == 0 ? sx : (a == 1 ? sy : sz);
            const float4* pos4 = reinterpret_cast<const float4*>(&tr[a][0]);
            const float4* far4 = reinterpret_cast<const float4*>(&tr[3 + a][0]);
#pragma unroll
            for (int q = 0; q < 16; q++) {
                const float4 pj = pos4[q], fe = far4[q];
                float e1;
                e1 = pprev + sa; sa = (e1 > fe.x ? e1 : fe.x) - pj.x; pprev = pj.x;
                e1 = pprev + sa; sa = (e1 > fe.y ? e1 : fe.y) - pj.y; pprev = pj.y;
                e1 = pprev + sa; sa = (e1 > fe.z ? e1 : fe.z) - pj.z; pprev = pj.z;
                e1 = pprev + sa; sa = (e1 > fe.w ? e1 : fe.w) - pj.w; pprev = pj.w;
            }
            __builtin_amdgcn_wave_barrier();
            px = rdlane(mx, 63); py = rdlane(my, 63); pz = rdlane(mz, 63);
            sx = rdlane(sa, 0); sy = rdlane(sa, 1); sz = rdlane(sa, 2);
        } else {
            for (int j = j0; j < cnt; j++) {
                if (!((vmask >> j) & 1ull)) continue;
                const float bx = rdlane(mine.p[0], j), by = rdlane(mine.p[1], j), bz = rdlane(mine.p[2], j);
                const float fx = rdlane(e2x, j), fy = rdlane(e2y, j), fz = rdlane(e2z, j);
                const float e1x = px + sx, e1y = py + sy, e1z = pz + sz;
                px = px < bx ? px : bx; py = py < by ? py : by; pz = pz < bz ? pz : bz;
                sx = (e1x > fx ? e1x : fx) - px;
                sy = (e1y > fy ? e1y : fy) - py;
                sz = (e1z > fz ? e1z : fz) - pz;
                collapsed = collapsed || (sx == 0.0f && sy == 0.0f && sz == 0.0f);
            }
        }
        const float inf = __builtin_inff();
        const bool finite = fabsf(px) < inf && fabsf(py) < inf && fabsf(pz) < inf && fabsf(sx) < inf && fabsf(sy) < inf && fabsf(sz) < inf;
        if (finite && !collapsed) {
            acc.p[0] = px; acc.p[1] = py; acc.p[2] = pz; acc.s[0] = sx; acc.s[1] = sy; acc.s[2] = sz;
        } else {
            for (int j = j0; j < cnt; j++) {
                A6 b = a6_shfl(mine, j);
                acc = merge_aabb(acc, b);
            }
            acc_valid = a6_valid(acc);
        }
        note_box();
    };
    const int s_off = summ_off ? summ_off[c] : -1;
    if (s_off >= 0) {
        // A long list: its full batches by their summaries (k_list_summary).  A batch is passed over exactly when the
        // test at the head of `batch` would pass it over -- the box settled, the batch's shape ok, its smallest
        // position above the box's and its largest far end below the box's on every axis (what holds for the
        // extremes holds for all 64) -- and is gathered and worked on otherwise.  Sockets were counted with the summaries.
        const int n_full = (call.end - call.begin) / 64;
        meshes = call_meshes[c];
        for (int b0 = 0; b0 < n_full; b0 += 64) {
            const int bj = b0 + lane;
            const bool has = bj < n_full;
            float4 sa = make_float4(0.0f, 0.0f, 0.0f, 0.0f), sb = sa;
            if (has) { const float4* q = reinterpret_cast<const float4*>(summary + (size_t)(s_off + bj) * 8); sa = q[0]; sb = q[1]; }
            const int n_here = min(64, n_full - b0);
            int w = 0;
            while (w < n_here) {
                if (settled) {
                    const bool quiet = has && sb.z != 0.0f && sa.x > acc.p[0] && sa.y > acc.p[1] && sa.z > acc.p[2] &&
                                       sa.w < far_x && sb.x < far_y && sb.y < far_z;
                    // the run of quiet batches from w on
                    const unsigned long long loud = ~__ballot(quiet) >> w;
                    w += loud == 0ull ? 64 - w : __builtin_ctzll(loud);
                    if (w >= n_here) break;
                }
                // (gathering the next such batch while this one's chain runs measured no faster: the chain is what is left)
                const int base = call.begin + (b0 + w) * 64;
                batch(base, a6_load(in_aabb, perm[base + lane]), 0);
                w++;
            }
        }
        const int tail = call.begin + n_full * 64;
        if (tail < call.end) {
            const int i = tail + lane;
            A6 mine = a6_zero();
            int kind = 0;
            if (i < call.end) { const int id = perm[i]; mine = a6_load(in_aabb, id); kind = in_kind[id]; }
            batch(tail, mine, kind);
        }
    } else
    for (int base0 = call.begin; base0 < call.end; base0 += 64 * LIST_UNROLL) {
        A6 cur[LIST_UNROLL];
        int cur_kind[LIST_UNROLL];
#pragma unroll
        for (int d = 0; d < LIST_UNROLL; d++) { cur[d] = ahead_box[d]; cur_kind[d] = ahead_kind[d]; }
#pragma unroll
        for (int d = 0; d < LIST_UNROLL; d++) {
            ahead_box[d] = a6_zero();
            ahead_kind[d] = 0;
            if (ahead_id[d] >= 0) { ahead_box[d] = a6_load(in_aabb, ahead_id[d]); ahead_kind[d] = in_kind[ahead_id[d]]; }
            const int i2 = base0 + (2 * LIST_UNROLL + d) * 64 + lane;
            ahead_id[d] = i2 < call.end ? perm[i2] : -1;
        }
#pragma unroll
        for (int d = 0; d < LIST_UNROLL; d++)
            if (base0 + d * 64 < call.end) batch(base0 + d * 64, cur[d], cur_kind[d]);
    }
    if (lane == 0) {
        A6 g = a6_grow(a6_abs(acc), 0.000001f);
        CallOut o;
        for (int i = 0; i < 3; i++) { o.aabb[i] = g.p[i]; o.aabb[3 + i] = g.s[i]; }
        o.axis = (type == 4) ? a6_longest_axis(g) : fixed_axis;
        o.mesh_count = meshes;
        for (int b = 0; b < MAXB; b++) o.counts[b] = 0;
        o.error = 0;
        o.pad[0] = o.pad[1] = o.pad[2] = 0;
        out[c] = o;
        if (sort_small && call.end - call.begin > 1) {
            Seg r = {call.begin, call.end, sort_depth_budget(call.end - call.begin), o.axis};
            sort_emit(r, sort_small, sort_now, sort_big, sort_now + 2);
        }
    }
    if (sort_small && c == 0)
        for (int i = lane; i < SORT_COUNTERS; i += 64) sort_next[i] = 0;
}

// sort key of every element of a call: centre of its box on the call's axis (pt_bvh_axis_sort.gd:180-181)
__global__ __launch_bounds__(256) void k_keys(const Call* calls, int n_calls, const int* perm, const float* in_aabb,
                                              const CallOut* co, float* keys) {
    int c = blockIdx.x;
    if (c >= n_calls) return;
    Call call = calls[c];
    int axis = co[c].axis;
    for (int i = call.begin + blockIdx.y * 256 + threadIdx.x; i < call.end; i += gridDim.y * 256) {
        A6 a = a6_load(in_aabb, perm[i]);
        keys[i] = a6_center(a, axis);
    }
}

// segment_count of a call (pt_bvh_axis_sort.gd:276,288-296): the order, less the sockets of a list at the depth limit
DEV int call_segment_count(int order, bool at_depth_limit, int mesh_count) {
    if (at_depth_limit && mesh_count > 0 && mesh_count < order) return order - mesh_count;
    return order;
}

// k_bins and k_partition cut a call's range into ny * 4 contiguous slices, one per wave of a (call, y) block of
// 256 threads; a slice holds a multiple of 64 elements.
DEV int call_slice_len(int len, int ny) {
    const int n_slices = ny * 4;
    return (((len + n_slices - 1) / n_slices) + 63) & ~63;
}

// bin index of every element (pt_bvh_axis_sort.gd:304-309); per-bin counts of every slice (slice_counts) and of
// the call (added to co[c].counts, which k_list_box left at zero)
__global__ __launch_bounds__(256) void k_bins(const Call* calls, int n_calls, const int* perm, const float* in_aabb,
                                              int order, int at_depth_limit, CallOut* co, unsigned char* bins, int* slice_counts) {
    int c = blockIdx.x;
    if (c >= n_calls) return;
    const int lane = threadIdx.x & 63;
    Call call = calls[c];
    const int slice = blockIdx.y * 4 + (threadIdx.x >> 6), slice_len = call_slice_len(call.end - call.begin, gridDim.y);
    const int begin = call.begin + slice * slice_len, end = min(call.end, begin + slice_len);
    if (begin >= call.end) return;
    const int axis = co[c].axis;
    const float lo = co[c].aabb[axis], size = co[c].aabb[3 + axis];
    const int segs = call_segment_count(order, at_depth_limit != 0, co[c].mesh_count);
    // (size / segment_count)[axis]: float32 divide, widened to double
    double segment = (double)(size / (float)segs);
    int counts[MAXB];
    for (int b = 0; b < MAXB; b++) counts[b] = 0;
    int err = 0;
    for (int base = begin; base < end; base += 64) {
        int i = base + lane;
        bool in = i < end;
        int bin = -1;
        if (in) {
            A6 a = a6_load(in_aabb, perm[i]);
            float rel = a6_center(a, axis) - lo;
            double q = floor((double)rel / segment);
            if (!(q >= 0.0 && q < (double)segs)) { err = 1; q = 0.0; }
            bin = (int)q;
            bins[i] = (unsigned char)bin;
        }
        for (int b = 0; b < MAXB; b++) counts[b] += __popcll(__ballot(in && bin == b));
    }
    err = __any(err);
    if (lane == 0) {
        int* mine = slice_counts + ((size_t)c * gridDim.y * 4 + slice) * MAXB;
        for (int b = 0; b < MAXB; b++) { mine[b] = counts[b]; if (counts[b]) atomicAdd(&co[c].counts[b], counts[b]); }
        if (err) atomicOr(&co[c].error, 1);
    }
}

// stable partition of a call's range by bin (bins keep list order), for calls with more than one occupied bin
// (pt_bvh_axis_sort.gd:314-325): every wave places its slice into tmp behind the earlier slices' elements of the
// same bin; k_partition_copy brings the result back.
__global__ __launch_bounds__(256) void k_partition(const Call* calls, int n_calls, const int* perm, int* tmp, const CallOut* co,
                                                   const unsigned char* bins, const int* slice_counts) {
    int c = blockIdx.x;
    if (c >= n_calls) return;
    const int lane = threadIdx.x & 63;
    Call call = calls[c];
    const int slice = blockIdx.y * 4 + (threadIdx.x >> 6), slice_len = call_slice_len(call.end - call.begin, gridDim.y);
    const int begin = call.begin + slice * slice_len, end = min(call.end, begin + slice_len);
    if (begin >= call.end) return;
    int offs[MAXB];
    int run = call.begin, occupied = 0;
    for (int b = 0; b < MAXB; b++) { offs[b] = run; run += co[c].counts[b]; occupied += co[c].counts[b] > 0; }
    if (occupied <= 1) return;
    // elements of each bin in the slices before this one
    const int* all = slice_counts + (size_t)c * gridDim.y * 4 * MAXB;
    int before[MAXB];
    for (int b = 0; b < MAXB; b++) before[b] = 0;
    for (int w = lane; w < slice; w += 64)
        for (int b = 0; b < MAXB; b++) before[b] += all[w * MAXB + b];
    for (int b = 0; b < MAXB; b++) {
        int v = before[b];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        offs[b] += v;
    }
    unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    for (int base = begin; base < end; base += 64) {
        int i = base + lane;
        bool in = i < end;
        int bin = in ? (int)bins[i] : -1;
        int v = in ? perm[i] : 0;
        for (int b = 0; b < MAXB; b++) {
            unsigned long long m = __ballot(in && bin == b);
            if (in && bin == b) tmp[offs[b] + __popcll(m & lt)] = v;
            offs[b] += __popcll(m);
        }
    }
}
// k_bins and k_partition in one for levels whose calls all have at most 1024 elements: the block keeps its
// elements in registers, so the list is rewritten in place.
__global__ __launch_bounds__(256) void k_bin_partition(const Call* calls, int n_calls, int* perm, const float* in_aabb, int order,
                                                       int at_depth_limit, CallOut* co) {
    __shared__ int s_cnt[4][MAXB];
    __shared__ int s_err;
    int c = blockIdx.x;
    if (c >= n_calls) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    Call call = calls[c];
    const int slice_len = call_slice_len(call.end - call.begin, 1);  // at most 256
    const int begin = call.begin + wave * slice_len, end = min(call.end, begin + slice_len);
    const int axis = co[c].axis;
    const float lo = co[c].aabb[axis], size = co[c].aabb[3 + axis];
    const int segs = call_segment_count(order, at_depth_limit != 0, co[c].mesh_count);
    double segment = (double)(size / (float)segs);
    int v[4], bin[4], counts[MAXB];
    for (int b = 0; b < MAXB; b++) counts[b] = 0;
    int err = 0;
    if (tid == 0) s_err = 0;
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const int i = begin + 64 * t + lane;
        const bool in = i < end;
        bin[t] = -1;
        v[t] = 0;
        if (in) {
            v[t] = perm[i];
            A6 a = a6_load(in_aabb, v[t]);
            float rel = a6_center(a, axis) - lo;
            double q = floor((double)rel / segment);
            if (!(q >= 0.0 && q < (double)segs)) { err = 1; q = 0.0; }
            bin[t] = (int)q;
        }
        for (int b = 0; b < MAXB; b++) counts[b] += __popcll(__ballot(in && bin[t] == b));
    }
    if (lane == 0) for (int b = 0; b < MAXB; b++) s_cnt[wave][b] = counts[b];
    __syncthreads();  // every element of the call is in a register by now
    if (__any(err) && lane == 0) s_err = 1;
    int offs[MAXB];
    int run = call.begin, occupied = 0;
    for (int b = 0; b < MAXB; b++) {
        int total = 0, before = 0;
        for (int w = 0; w < 4; w++) { total += s_cnt[w][b]; before += w < wave ? s_cnt[w][b] : 0; }
        offs[b] = run + before;
        run += total;
        occupied += total > 0;
        counts[b] = total;
    }
    __syncthreads();
    if (tid == 0) {
        for (int b = 0; b < MAXB; b++) co[c].counts[b] = counts[b];
        co[c].error = s_err;
    }
    if (occupied <= 1) return;
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
    for (int t = 0; t < 4; t++) {
        for (int b = 0; b < MAXB; b++) {
            const unsigned long long m = __ballot(bin[t] == b);
            if (bin[t] == b) perm[offs[b] + __popcll(m & lt)] = v[t];
            offs[b] += __popcll(m);
        }
    }
}
__global__ __launch_bounds__(256) void k_partition_copy(const Call* calls, int n_calls, int* perm, const int* tmp, const CallOut* co) {
    int c = blockIdx.x;
    if (c >= n_calls) return;
    Call call = calls[c];
    int occupied = 0;
    for (int b = 0; b < MAXB; b++) occupied += co[c].counts[b] > 0;
    if (occupied <= 1) return;
    for (int i = call.begin + blockIdx.y * 256 + threadIdx.x; i < call.end; i += gridDim.y * 256) perm[i] = tmp[i];
}

// ---- exact emulation of Godot's SortArray on (key descending, payload) ------------------------------
DEV bool cmp_less(float a, float b) { return a > b; }  // comparator of pt_bvh_axis_sort.gd:180-181

DEV float median_of_3(float a, float b, float c) {
    if (cmp_less(a, b)) {
        if (cmp_less(b, c)) return b;
        else if (cmp_less(a, c)) return c;
        else return a;
    } else if (cmp_less(a, c)) return a;
    else if (cmp_less(b, c)) return c;
    else return b;
}

// heapsort fallback (SortArray::partial_sort with middle == last), run by one lane
DEV void heap_adjust(float* k, int* v, int first, int hole, int len, float kv, int vv) {
    int top = hole;
    int second = 2 * hole + 2;
    while (second < len) {
        if (cmp_less(k[first + second], k[first + (second - 1)])) second--;
        k[first + hole] = k[first + second]; v[first + hole] = v[first + second];
        hole = second;
        second = 2 * (second + 1);
    }
    if (second == len) {
        k[first + hole] = k[first + (second - 1)]; v[first + hole] = v[first + (second - 1)];
        hole = second - 1;
    }
    int parent = (hole - 1) / 2;
    while (hole > top && cmp_less(k[first + parent], kv)) {
        k[first + hole] = k[first + parent]; v[first + hole] = v[first + parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    k[first + hole] = kv; v[first + hole] = vv;
}
DEV void heap_sort(float* k, int* v, int first, int last) {
    int len = last - first;
    if (len >= 2) {
        int parent = (len - 2) / 2;
        for (;;) {
            heap_adjust(k, v, first, parent, len, k[first + parent], v[first + parent]);
            if (parent == 0) break;
            parent--;
        }
    }
    int middle = last;
    while (middle - first > 1) {
        float kv = k[middle - 1]; int vv = v[middle - 1];
        k[middle - 1] = k[first]; v[middle - 1] = v[first];
        heap_adjust(k, v, first, 0, (middle - 1) - first, kv, vv);
        middle--;
    }
}

// Sorts one segment of at most SORT_BIG elements completely: one wave brings its keys (the boxes' centres on the
// segment's axis) and payloads to LDS and replays SortArray::introsort over it with an explicit stack (the two parts
// of a partition are independent, so the order in which they are taken does not matter), then writes the payloads
// back.  A partition lowers the depth budget by one and a segment with no budget left is heap-sorted without
// children, so the stack never holds more than budget + 1 entries.  The partition is the one of k_sort_level_big
// below, with the stop lists in LDS too.  The pieces of at most SORT_THRESHOLD elements get their share of the
// final insertion pass before the copy back.
__global__ __launch_bounds__(64) void k_sort_small(const Seg* segs, const int* n_segs, const float* in_aabb, int* vals) {
    __shared__ Seg stack[SORT_MAX_LEVELS + 4];
    __shared__ float sk[SORT_BIG];
    __shared__ int sv[SORT_BIG];
    __shared__ unsigned short stops_l[SORT_BIG], stops_r[SORT_BIG];
    __shared__ unsigned int done[SORT_BIG / 2];  // finished pieces of 2..SORT_THRESHOLD elements: first | length << 16
    int s = blockIdx.x;
    if (s >= *n_segs) return;
    const int lane = threadIdx.x;
    int n_done = 0;
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const Seg whole = segs[s];
    const int base = whole.first, n_whole = whole.last - whole.first;  // positions below are relative to base
    for (int i = lane; i < n_whole; i += 64) {
        const int v = vals[base + i];
        sv[i] = v;
        sk[i] = a6_center(a6_load(in_aabb, v), whole.axis);
    }
    if (lane == 0) { Seg r = {0, n_whole, whole.depth, whole.axis}; stack[0] = r; }
    int sp = 1;
    while (sp > 0) {
        __builtin_amdgcn_wave_barrier();  // one wave: its LDS accesses execute in program order
        const Seg seg = stack[--sp];
        __builtin_amdgcn_wave_barrier();
        const int first = seg.first, last = seg.last;
        const int n = last - first;
        if (n <= SORT_THRESHOLD) {  // introsort leaves such a piece to the final insertion pass
            if (n > 1) {
                if (lane == 0) done[n_done] = (unsigned)first | ((unsigned)n << 16);
                n_done++;
            }
            continue;
        }
        if (seg.depth == 0) {
            // SortArray::partial_sort over the whole segment, a heapsort.  With no two keys equal (and none a NaN) its
            // result is simply the sorted order, found here by counting, for every element, the elements before it;
            // otherwise one lane runs the heapsort itself, whose order among equal keys is its own.
            float rk[SORT_BIG / 64];
            int rv[SORT_BIG / 64], rank[SORT_BIG / 64];
            bool plain = true;
#pragma unroll
            for (int t = 0; t < SORT_BIG / 64; t++) {
                const int i = first + lane + 64 * t;
                rank[t] = -1;
                if (i < last) {
                    rk[t] = sk[i]; rv[t] = sv[i];
                    int before = 0, same = 0;
                    for (int j = first; j < last; j++) { const float kj = sk[j]; before += cmp_less(kj, rk[t]) ? 1 : 0; same += kj == rk[t] ? 1 : 0; }
                    rank[t] = before;
                    plain = plain && same == 1;
                }
            }
            plain = !__any(!plain);
            __builtin_amdgcn_wave_barrier();
            if (plain) {
#pragma unroll
                for (int t = 0; t < SORT_BIG / 64; t++) if (rank[t] >= 0) { sk[first + rank[t]] = rk[t]; sv[first + rank[t]] = rv[t]; }
            } else if (lane == 0) {
                heap_sort(sk, sv, first, last);
            }
            continue;  // fully sorted; no final insertion pass needed
        }
        const float P = median_of_3(sk[first], sk[first + (last - first) / 2], sk[last - 1]);
        int nl = 0, nr = 0;
        for (int chunk = 0; chunk < n; chunk += 64) {
            const int off = chunk + lane;
            const bool in = off < n;
            const bool stop_l = in && !cmp_less(sk[first + off], P);
            const bool stop_r = in && !cmp_less(P, sk[last - 1 - off]);
            const unsigned long long ml = __ballot(stop_l), mr = __ballot(stop_r);
            if (stop_l) stops_l[nl + __popcll(ml & lt)] = (unsigned short)(first + off);
            if (stop_r) stops_r[nr + __popcll(mr & lt)] = (unsigned short)(last - 1 - off);
            nl += __popcll(ml);
            nr += __popcll(mr);
        }
        __builtin_amdgcn_wave_barrier();
        const int n_pairs = min(nl, nr);
        int K = 0;  // pairs with L_k < R_k: a prefix of the pairs
        for (int k0 = 0; k0 < n_pairs; k0 += 64) {
            const int k = k0 + lane;
            const int c = __popcll(__ballot(k < n_pairs && stops_l[k] < stops_r[k]));
            K += c;
            if (c < 64) break;
        }
        for (int k = lane; k < K; k += 64) {  // every position takes part in at most one swap
            const int L = stops_l[k], R = stops_r[k];
            const float ka = sk[L], kb = sk[R];
            const int va = sv[L], vb = sv[R];
            sk[L] = kb; sv[L] = vb; sk[R] = ka; sv[R] = va;
        }
        const int Rk = K > 0 ? (int)stops_r[K - 1] : last;
        const int cut = K < nl ? min((int)stops_l[K], Rk) : Rk;
        // introsort(cut, last, depth-1); last = cut (same depth budget)
        if (lane == 0) {
            Seg a = {cut, last, seg.depth - 1, seg.axis}, b = {first, cut, seg.depth - 1, seg.axis};
            stack[sp] = a;
            stack[sp + 1] = b;
        }
        sp += 2;
    }
    __builtin_amdgcn_wave_barrier();
    // final_insertion_sort, restricted to each finished piece (elements never cross partition cuts): one lane each
    for (int f = lane; f < n_done; f += 64) {
        const int first = (int)(done[f] & 0xffffu), last = first + (int)(done[f] >> 16);
        for (int i = first + 1; i < last; i++) {
            const float kv = sk[i];
            const int vv = sv[i];
            int j = i;
            while (j > first && cmp_less(kv, sk[j - 1])) { sk[j] = sk[j - 1]; sv[j] = sv[j - 1]; j--; }
            sk[j] = kv; sv[j] = vv;
        }
    }
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < n_whole; i += 64) vals[base + i] = sv[i];
}

// The same partition for a long segment, by a 1024-thread workgroup.  The two-pointer loop of
// SortArray::partitioner swaps the k-th element from the left that is not before the pivot with the k-th from
// the right that is not after it, for as long as the former lies left of the latter; which elements those are
// depends on the segment's contents before the call only.  So: list the "left stops" in ascending and the
// "right stops" in descending position (block-wide prefix sums, 1024 elements per step), count the K pairs with
// L_k < R_k (the stops move towards each other, so these are the first K), swap them all at once, and cut where
// the loop would have stopped: min(L_K+1, R_K).
__global__ __launch_bounds__(1024) void k_sort_level_big(const Seg* segs, const int* n_segs, float* keys, int* vals, int* stops_l, int* stops_r,
                                                         Seg* next, int* next_count, Seg* next_big, int* big_count) {
    __shared__ int s_wave[16];
    __shared__ int s_total;
    __shared__ float s_keys[SORT_RANK_MAX];
    const int sidx = blockIdx.x;
    if (sidx >= *n_segs) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const Seg seg = segs[sidx];
    const int first = seg.first, last = seg.last;
    if (seg.depth == 0) {
        // the heapsort fallback, by counting when no two keys are equal (see k_sort_small)
        const int n = last - first;
        bool plain = n <= SORT_RANK_MAX;
        float rk[SORT_RANK_MAX / 1024];
        int rv[SORT_RANK_MAX / 1024], rank[SORT_RANK_MAX / 1024];
        if (plain) {
            for (int i = tid; i < n; i += 1024) s_keys[i] = keys[first + i];
            __syncthreads();
#pragma unroll
            for (int t = 0; t < SORT_RANK_MAX / 1024; t++) {
                const int i = tid + 1024 * t;
                rank[t] = -1;
                if (i < n) {
                    rk[t] = s_keys[i]; rv[t] = vals[first + i];
                    int before = 0, same = 0;
                    for (int j = 0; j < n; j++) { const float kj = s_keys[j]; before += cmp_less(kj, rk[t]) ? 1 : 0; same += kj == rk[t] ? 1 : 0; }
                    rank[t] = before;
                    plain = plain && same == 1;
                }
            }
        }
        if (tid == 0) s_total = 0;
        __syncthreads();
        if (!plain) s_total = 1;
        __syncthreads();
        if (s_total == 0) {
#pragma unroll
            for (int t = 0; t < SORT_RANK_MAX / 1024; t++) if (rank[t] >= 0) { keys[first + rank[t]] = rk[t]; vals[first + rank[t]] = rv[t]; }
        } else if (tid == 0) {
            heap_sort(keys, vals, first, last);
        }
        return;
    }
    const float P = median_of_3(keys[first], keys[first + (last - first) / 2], keys[last - 1]);
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    int n_stops[2];
    for (int side = 0; side < 2; side++) {  // 0: left stops, ascending; 1: right stops, descending
        int* out = side == 0 ? stops_l : stops_r;
        int base = 0;
        for (int chunk = 0; chunk < last - first; chunk += 1024) {
            const int off = chunk + tid;
            const bool in = off < last - first;
            const int i = side == 0 ? first + off : last - 1 - off;
            bool stop = false;
            if (in) { const float k = keys[i]; stop = side == 0 ? !cmp_less(k, P) : !cmp_less(P, k); }
            const unsigned long long m = __ballot(stop);
            if (lane == 0) s_wave[wave] = __popcll(m);
            __syncthreads();
            int before = 0, total = 0;
            for (int w = 0; w < 16; w++) { const int c = s_wave[w]; total += c; before += w < wave ? c : 0; }
            if (stop) out[first + base + before + __popcll(m & lt)] = i;
            base += total;
            __syncthreads();
        }
        n_stops[side] = base;
    }
    __threadfence_block();
    __syncthreads();
    const int n_pairs = min(n_stops[0], n_stops[1]);
    // K = number of k with L_k < R_k (a prefix of the pairs)
    int mine = 0;
    for (int k = tid; k < n_pairs; k += 1024) mine += stops_l[first + k] < stops_r[first + k] ? 1 : 0;
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_down(mine, o);
    if (lane == 0) s_wave[wave] = mine;
    __syncthreads();
    if (tid == 0) { int t = 0; for (int w = 0; w < 16; w++) t += s_wave[w]; s_total = t; }
    __syncthreads();
    const int K = s_total;
    for (int k = tid; k < K; k += 1024) {  // every position takes part in at most one swap
        const int L = stops_l[first + k], R = stops_r[first + k];
        const float ka = keys[L], kb = keys[R];
        const int va = vals[L], vb = vals[R];
        keys[L] = kb; vals[L] = vb; keys[R] = ka; vals[R] = va;
    }
    if (tid == 0) {
        const int Rk = K > 0 ? stops_r[first + K - 1] : last;
        const int cut = K < n_stops[0] ? min(stops_l[first + K], Rk) : Rk;
        Seg a = {cut, last, seg.depth - 1, seg.axis}, b = {first, cut, seg.depth - 1, seg.axis};
        sort_emit(a, next, next_count, next_big, big_count);
        sort_emit(b, next, next_count, next_big, big_count);
    }
}

// the queue's first entry for a fixed-axis tree: the whole list
__global__ void k_sort_seed(int n, int axis, Seg* small, int* small_count, Seg* big, int* big_count) {
    if (blockIdx.x != 0 || threadIdx.x != 0 || n < 2) return;
    Seg r = {0, n, sort_depth_budget(n), axis};
    sort_emit(r, small, small_count, big, big_count);
}

// ---- tree finishing ---------------------------------------------------------------------------------
// leaf boxes: merge chain over the leaf's objects in order (BVHNode.set_aabb, pt_bvh_tree.gd:862-865)
__global__ void k_leaf_boxes(int n, const int* leaf_node, const int* obj_begin, const int* obj_count, const int* perm,
                             const float* in_aabb, float* node_aabb) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    A6 acc = a6_zero();
    int b = obj_begin[t], cnt = obj_count[t];
    for (int k = 0; k < cnt; k++) acc = merge_aabb(acc, a6_load(in_aabb, perm[b + k]));
    a6_store(node_aabb, leaf_node[t], acc);
}
// socket boxes: mesh.transform * mesh_root.aabb (pt_bvh_tree.gd:858-860)
__global__ void k_socket_box(const float* transform12, const float* mesh_root_aabb, float* dst6) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float t[12];
    for (int i = 0; i < 12; i++) t[i] = transform12[i];
    A6 a = a6_load(mesh_root_aabb, 0);
    A6 r = a6_xform(t, a);
    a6_store(dst6, 0, r);
}
// inner boxes of one tree level: merge chain over the children in order (:867-870)
__global__ void k_inner_boxes(int n, const int* nodes, const int* child_begin, const int* child_count, const int* children,
                              float* node_aabb) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    A6 acc = a6_zero();
    int b = child_begin[t], cnt = child_count[t];
    for (int k = 0; k < cnt; k++) acc = merge_aabb(acc, a6_load(node_aabb, children[b + k]));
    a6_store(node_aabb, nodes[t], acc);
}
__global__ void k_copy_box(const float* src, int si, float* dst, int di) {
    if (threadIdx.x == 0 && blockIdx.x == 0) a6_store(dst, di, a6_load(src, si));
}
// node bytes (pt_utils.gd:71-97 + pt_bvh_tree.gd:910-959)
__global__ void k_encode(int n, const int* box_index, const int* node_id, const int* xform, const unsigned char* is_empty_root,
                         const float* boxes, float* out) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    float* o = out + (size_t)t * 8;
    if (is_empty_root[t]) { for (int i = 0; i < 8; i++) o[i] = 0.0f; return; }
    A6 raw = a6_load(boxes, box_index[t]);
    A6 a = a6_abs(raw);
    bool zero = raw.s[0] == 0.0f && raw.s[1] == 0.0f && raw.s[2] == 0.0f;
    float pad = zero ? 0.0f : 1e-6f;
    o[0] = a.p[0] - pad; o[1] = a.p[1] - pad; o[2] = a.p[2] - pad;
    o[3] = __int_as_float(node_id[t]);
    o[4] = (a.p[0] + a.s[0]) + pad; o[5] = (a.p[1] + a.s[1]) + pad; o[6] = (a.p[2] + a.s[2]) + pad;
    o[7] = __int_as_float(xform[t]);
}
// A node created for one object holds merge_aabb(AABB(), object box): an invalid (e.g. zero-size) object
// box leaves the node with the zero box (pt_bvh_axis_sort.gd:70-73 + pt_bvh_tree.gd:862-865).
__global__ void k_canonical_boxes(int n, float* in_aabb) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    A6 a = a6_load(in_aabb, t);
    a6_store(in_aabb, t, merge_aabb(a6_zero(), a));
}
__global__ void k_iota(int n, int* p) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) p[t] = t;
}
// perm position p -> its object-id slot dest[p] (-1: a socket): the id table and, optionally, the object boxes by slot
__global__ void k_scatter_ids(int n, const int* perm, const int* in_ref, const float* in_aabb, const int* dest, int* ids, float* obj_box) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n || dest[t] < 0) return;
    const int e = perm[t];
    ids[dest[t]] = in_ref[e];
    if (obj_box) a6_store(obj_box, dest[t], a6_load(in_aabb, e));
}
__global__ void k_gather_boxes(int n, const int* box_index, const float* boxes, float* out) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) a6_store(out, t, a6_load(boxes, box_index[t]));
}

// ---- binned SAH (PTK_BVH_SAH) ---------------------------------------------------------------------------
// The reference reserves a slot for a SAH builder and leaves it empty (bvh_trees/pt_bvh_sweep_sah.gd:1-21).
// This one emits the same node format through the same flattening, so the renderers and the refit work on
// its trees unchanged.  Rule: a wide node's children are the parts left after up to three rounds of binary
// splits of its list; a part of more than `order` elements is always split, a smaller one only when the best
// plane's area(L) * nL + area(R) * nR is below area(all) * (n - 1).  A binary split bins the elements' box
// centres c = (lo + hi) * 0.5 into 16 bins along the longest side of the centres' bounds (bin =
// trunc((c - cmin) * (16 / extent)), clamped), keeps count and box bounds per bin, and takes the plane with
// the smallest cost (first on ties; area = 2 * ((dx*dy + dy*dz) + dz*dx)); without spread, or with all
// elements in one bin, a list larger than `order` is cut in the middle.  Elements keep their order on both
// sides.  Minima and maxima are taken in the total order of the float bit patterns, so they are exact
// whatever the order of evaluation: one wave per part reduces with LDS atomics.
struct SahSet {
    int begin, end;
};
DEV unsigned ord_enc(float f) { unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
DEV float ord_dec(unsigned e) { return __uint_as_float((e & 0x80000000u) ? (e & 0x7FFFFFFFu) : ~e); }
DEV float sah_area(const float* lo, const float* hi) {
    const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    return 2.0f * ((dx * dy + dy * dz) + dz * dx);
}

__global__ __launch_bounds__(64) void k_sah_round(const SahSet* sets, int n_sets, int* perm, int* tmp, const float* in_aabb,
                                                  unsigned char* bin_of, int order, int* n_left_out) {
    __shared__ unsigned s_cmin[3], s_cmax[3];
    __shared__ unsigned s_lo[16][3], s_hi[16][3];
    __shared__ int s_n[16];
    __shared__ int s_best, s_nleft;
    const int sidx = blockIdx.x;
    if (sidx >= n_sets) return;
    const int lane = threadIdx.x;
    const SahSet set = sets[sidx];
    const int n = set.end - set.begin;
    if (lane < 3) { s_cmin[lane] = 0xFFFFFFFFu; s_cmax[lane] = 0u; }
    if (lane < 16) { s_n[lane] = 0; for (int a = 0; a < 3; a++) { s_lo[lane][a] = 0xFFFFFFFFu; s_hi[lane][a] = 0u; } }
    __syncthreads();
    // centre bounds
    {
        unsigned mn[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, mx[3] = {0u, 0u, 0u};
        for (int i = set.begin + lane; i < set.end; i += 64) {
            const A6 b = a6_load(in_aabb, perm[i]);
            for (int a = 0; a < 3; a++) {
                const unsigned e = ord_enc((b.p[a] + (b.p[a] + b.s[a])) * 0.5f);
                mn[a] = e < mn[a] ? e : mn[a];
                mx[a] = e > mx[a] ? e : mx[a];
            }
        }
        for (int a = 0; a < 3; a++) { atomicMin(&s_cmin[a], mn[a]); atomicMax(&s_cmax[a], mx[a]); }
    }
    __syncthreads();
    float cmin[3], ext[3];
    for (int a = 0; a < 3; a++) { cmin[a] = ord_dec(s_cmin[a]); ext[a] = ord_dec(s_cmax[a]) - cmin[a]; }
    int axis = 0;
    float m = ext[0];
    if (ext[1] > m) { axis = 1; m = ext[1]; }
    if (ext[2] > m) { axis = 2; m = ext[2]; }
    const float k = 16.0f / m;
    const bool spread = m > 0.0f && k < __builtin_inff();
    if (spread) {
        for (int i = set.begin + lane; i < set.end; i += 64) {
            const A6 b = a6_load(in_aabb, perm[i]);
            const float q = ((b.p[axis] + (b.p[axis] + b.s[axis])) * 0.5f - cmin[axis]) * k;
            const int bi = q >= 15.0f ? 15 : (q > 0.0f ? (int)q : 0);
            bin_of[i] = (unsigned char)bi;
            atomicAdd(&s_n[bi], 1);
            for (int a = 0; a < 3; a++) { atomicMin(&s_lo[bi][a], ord_enc(b.p[a])); atomicMax(&s_hi[bi][a], ord_enc(b.p[a] + b.s[a])); }
        }
    }
    __syncthreads();
    if (lane == 0) {
        int best = -1, n_left = -1;
        float best_cost = 0.0f;
        if (spread) {
            for (int p = 0; p < 15; p++) {
                unsigned llo[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, lhi[3] = {0u, 0u, 0u}, rlo[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, rhi[3] = {0u, 0u, 0u};
                int nl = 0, nr = 0;
                for (int b = 0; b < 16; b++) {
                    if (!s_n[b]) continue;
                    if (b <= p) { nl += s_n[b]; for (int a = 0; a < 3; a++) { llo[a] = min(llo[a], s_lo[b][a]); lhi[a] = max(lhi[a], s_hi[b][a]); } }
                    else { nr += s_n[b]; for (int a = 0; a < 3; a++) { rlo[a] = min(rlo[a], s_lo[b][a]); rhi[a] = max(rhi[a], s_hi[b][a]); } }
                }
                if (!nl || !nr) continue;
                float Llo[3], Lhi[3], Rlo[3], Rhi[3];
                for (int a = 0; a < 3; a++) { Llo[a] = ord_dec(llo[a]); Lhi[a] = ord_dec(lhi[a]); Rlo[a] = ord_dec(rlo[a]); Rhi[a] = ord_dec(rhi[a]); }
                const float cost = sah_area(Llo, Lhi) * (float)nl + sah_area(Rlo, Rhi) * (float)nr;
                if (best < 0 || cost < best_cost) { best = p; best_cost = cost; }
            }
            if (best >= 0 && n <= order) {
                unsigned plo[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, phi[3] = {0u, 0u, 0u};
                for (int b = 0; b < 16; b++) {
                    if (!s_n[b]) continue;
                    for (int a = 0; a < 3; a++) { plo[a] = min(plo[a], s_lo[b][a]); phi[a] = max(phi[a], s_hi[b][a]); }
                }
                float Plo[3], Phi[3];
                for (int a = 0; a < 3; a++) { Plo[a] = ord_dec(plo[a]); Phi[a] = ord_dec(phi[a]); }
                if (!(best_cost < sah_area(Plo, Phi) * (float)(n - 1))) { best = -2; n_left = 0; }
            }
            if (best >= 0) { n_left = 0; for (int b = 0; b <= best; b++) n_left += s_n[b]; }
        }
        if (n_left < 0) { best = -1; n_left = n <= order ? 0 : n / 2; }  // no spread: cut in the middle, or leave a small list whole
        s_best = best;
        s_nleft = n_left;
        n_left_out[sidx] = n_left;
    }
    __syncthreads();
    const int best = s_best, n_left = s_nleft;
    if (n_left == 0) return;
    if (best < 0) return;  // the middle cut moves nothing
    // stable partition: bins 0..best keep their order on the left, the others on the right
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    int run_l = set.begin, run_r = set.begin + n_left;
    for (int base = set.begin; base < set.end; base += 64) {
        const int i = base + lane;
        const bool in = i < set.end;
        const bool right = in && (int)bin_of[i] > best;
        const int v = in ? perm[i] : 0;
        const unsigned long long mr = __ballot(right), ml = __ballot(in && !right);
        if (in) tmp[right ? run_r + __popcll(mr & lt) : run_l + __popcll(ml & lt)] = v;
        run_l += __popcll(ml);
        run_r += __popcll(mr);
    }
    __threadfence_block();
    __syncthreads();
    for (int i = set.begin + lane; i < set.end; i += 64) perm[i] = tmp[i];
}

// ---- host side -------------------------------------------------------------------------------------------
template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    ~DevBuf() { if (p) (void)hipFree(p); }
    size_t cap = 0;
    bool alloc(size_t count) {  // contents are not kept; an allocation that is large enough is reused
        n = count;
        if (p && cap >= std::max<size_t>(count, 1)) return true;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        cap = std::max<size_t>(count, 1);
        if (hipMalloc((void**)&p, cap * sizeof(T)) != hipSuccess) { p = nullptr; cap = 0; return false; }
        return true;
    }
};

enum { N_INNER = 0, N_LEAF = 1, N_SOCKET = 2 };

struct HNode {
    int kind = N_INNER;
    std::vector<int> children;  // inner / socket (socket: the mesh root is not a local node)
    int obj_begin = 0, obj_count = 0;  // leaf: range of perm
    int mesh_slot = -1;                // socket
    bool recursive = false;            // created for a bin that was split further
    bool terminal_mesh = false;        // terminal bin of a list that contains sockets
};

struct BvhImpl : PtkBvh {
    int order = 8, type = 1;
    int n_objects = 0, n_in = 0;
    std::vector<HNode> nodes;          // node 0 = root
    std::vector<int> leaf_order;       // index_node order of this tree's own leaves
    std::vector<std::shared_ptr<BvhImpl>> meshes;  // adopted sub-trees, shared with the handles they were built under
    std::vector<int> mesh_index;       // transform index of each mesh
    DevBuf<float> in_aabb;             // n_in boxes
    DevBuf<int> in_kind, in_ref;       // 0 object / 1 socket; object id
    DevBuf<int> perm;                  // final element order
    DevBuf<float> node_aabb;           // one box per node
    int own_object_count = 0;

    int own_nodes() const { return (int)nodes.size(); }
    int flat_node_count() const override {
        int n = own_nodes();
        for (auto& m : meshes) n += m->flat_node_count();
        return n;
    }
    int flat_id_count() const override {
        int n = own_object_count;
        for (auto& m : meshes) n += m->flat_id_count();
        return n;
    }
    int mesh_count() const override { return (int)meshes.size(); }
    std::vector<float> mesh_xf12;      // Transform3D of each merged mesh (basis rows, origin)
    bool flatten(hipStream_t stream, uint8_t* node_bytes, int32_t* object_ids, std::string& err) override;
    bool bind(hipStream_t stream, uint8_t* d_node_bytes, int32_t* d_object_ids, uint8_t* h_node_bytes, int32_t* h_object_ids,
              PtkRefitState* state, std::string& err) override;
    bool emit(hipStream_t stream, uint8_t* d_node_bytes, int32_t* d_object_ids, uint8_t* node_bytes, int32_t* object_ids,
              PtkRefitState* state, std::string& err);
};

#define CK(expr)                                                                   \
    do {                                                                           \
        hipError_t e_ = (expr);                                                    \
        if (e_ != hipSuccess) { err = std::string(#expr) + ": " + hipGetErrorString(e_); return false; } \
    } while (0)

// queues of device_sort, kept over the calls of one build
struct SortWorkspace {
    DevBuf<Seg> small, big_a, big_b;
    DevBuf<int> counters;  // two sets of SORT_COUNTERS, used by alternate levels of the build
    DevBuf<int> stops_l, stops_r;
    int h_counts[SORT_COUNTERS];
    bool reserve(size_t covered, int n_total, bool any_big) {
        // segments of one queue are disjoint and longer than 1 (small) or SORT_BIG (big)
        if (!small.alloc(covered / 2 + 16) || !big_a.alloc(covered / (SORT_BIG + 1) + 16) || !big_b.alloc(covered / (SORT_BIG + 1) + 16)) return false;
        return !any_big || (stops_l.alloc(n_total) && stops_r.alloc(n_total));
    }
};

// Exact sort of the queued ranges of vals by their boxes' centres, an emulation of SortArray::sort.  The queues of
// counter set `set` hold the ranges (k_list_box or k_sort_seed put them there).  Segments longer than SORT_BIG
// (keys: their centres, by k_keys) are partitioned level by level, SORT_LEVEL_GROUP levels per host round trip;
// everything shorter is finished by one launch of k_sort_small.  covered: the sum of the ranges' lengths.
bool device_sort(hipStream_t st, SortWorkspace& ws, int set, bool any_big, size_t covered, float* keys, const float* in_aabb, int* vals,
                 std::string& err) {
    int* counters = ws.counters.p + set * SORT_COUNTERS;
    const size_t cap_small = covered / 2 + 16, cap_big = covered / (SORT_BIG + 1) + 16;
    if (any_big) {
        int level = 0;
        for (bool more = true; more;) {
            if (level + SORT_LEVEL_GROUP > SORT_MAX_LEVELS) { err = "sort did not terminate"; return false; }
            for (int g = 0; g < SORT_LEVEL_GROUP; g++, level++) {
                Seg* in = (level & 1) ? ws.big_b.p : ws.big_a.p;
                Seg* out = (level & 1) ? ws.big_a.p : ws.big_b.p;
                // more blocks than segments (the count is on the device): the others leave at once
                hipLaunchKernelGGL(k_sort_level_big, dim3((unsigned)cap_big), dim3(1024), 0, st, in, counters + 2 + level, keys, vals, ws.stops_l.p,
                                   ws.stops_r.p, ws.small.p, counters, out, counters + 2 + level + 1);
                CK(hipGetLastError());
            }
            CK(hipMemcpyAsync(ws.h_counts, counters, sizeof(int) * (2 + level + 1), hipMemcpyDeviceToHost, st));
            CK(hipStreamSynchronize(st));
            more = ws.h_counts[2 + level] > 0;
        }
    }
    hipLaunchKernelGGL(k_sort_small, dim3((unsigned)cap_small), dim3(64), 0, st, ws.small.p, counters, in_aabb, vals);
    CK(hipGetLastError());
    return true;
}

int bitlog(int n) { int k = 0; for (; n != 1; n >>= 1) ++k; return k; }

struct PendingCall {
    int begin, end;
    int parent;  // host node that receives the new nodes as children
};

bool build_tree(hipStream_t st, BvhImpl& T, const float* aabbs6, const int32_t* object_ids,
                const float* mesh_transforms12, std::string& err) {
    const int N = T.n_objects, M = (int)T.meshes.size(), n_in = N + M;
    T.n_in = n_in;
    const bool timing = getenv("PTK_BUILD_TIMING") != nullptr;  // (development aid: where a build's host time goes, on stderr)
    const auto t_begin = std::chrono::steady_clock::now();
    double t_wait = 0.0;  // of which: waiting for the GPU at the levels' ends
    auto lap = [&](const char* what) {
        if (timing) fprintf(stderr, "build %-12s %.3f ms (waits so far %.3f)\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count(), t_wait);
    };
    if (!T.in_aabb.alloc((size_t)n_in * 6) || !T.in_kind.alloc(n_in) || !T.in_ref.alloc(n_in) || !T.perm.alloc(n_in)) { err = "out of device memory"; return false; }
    std::vector<int> kind(n_in, 0), ref(n_in, -1);
    for (int i = 0; i < N; i++) ref[i] = object_ids[i];
    for (int m = 0; m < M; m++) { kind[N + m] = 1; ref[N + m] = m; }
    if (N) CK(hipMemcpyAsync(T.in_aabb.p, aabbs6, sizeof(float) * 6 * N, hipMemcpyHostToDevice, st));
    CK(hipMemcpyAsync(T.in_kind.p, kind.data(), sizeof(int) * n_in, hipMemcpyHostToDevice, st));
    CK(hipMemcpyAsync(T.in_ref.p, ref.data(), sizeof(int) * n_in, hipMemcpyHostToDevice, st));
    DevBuf<float> xf;
    if (M) {
        if (!xf.alloc((size_t)M * 12)) { err = "out of device memory"; return false; }
        CK(hipMemcpyAsync(xf.p, mesh_transforms12, sizeof(float) * 12 * M, hipMemcpyHostToDevice, st));
        for (int m = 0; m < M; m++) {
            hipLaunchKernelGGL(k_socket_box, dim3(1), dim3(1), 0, st, xf.p + (size_t)m * 12, T.meshes[m]->node_aabb.p, T.in_aabb.p + (size_t)(N + m) * 6);
            CK(hipGetLastError());
        }
    }
    if (N) {
        hipLaunchKernelGGL(k_canonical_boxes, dim3((N + 255) / 256), dim3(256), 0, st, N, T.in_aabb.p);
        CK(hipGetLastError());
    }
    hipLaunchKernelGGL(k_iota, dim3((n_in + 255) / 256), dim3(256), 0, st, n_in, T.perm.p);
    CK(hipGetLastError());

    DevBuf<float> keys;
    DevBuf<int> tmp;
    DevBuf<unsigned char> bins;
    if (!keys.alloc(n_in) || !tmp.alloc(n_in) || !bins.alloc(n_in)) { err = "out of device memory"; return false; }

    if (T.type == PTK_BVH_SAH_TYPE) {
        // ---- SAH topology: up to three rounds of binary splits per wide node, level by level ----
        T.nodes.clear();
        T.nodes.emplace_back();  // root (inner)
        struct WideCall { int parent; std::vector<std::pair<int, int>> parts; };
        std::vector<WideCall> wide = {{0, {{0, n_in}}}};
        DevBuf<SahSet> d_sets;
        DevBuf<int> d_nleft;
        std::vector<int> hperm;  // only fetched when the tree has sockets (to count them per list)
        while (!wide.empty()) {
            for (int round = 0; round < 3; round++) {
                std::vector<SahSet> sets;
                std::vector<std::pair<int, int>> where;  // (wide call, part) of every set
                for (size_t c = 0; c < wide.size(); c++)
                    for (size_t q = 0; q < wide[c].parts.size(); q++)
                        if (wide[c].parts[q].second - wide[c].parts[q].first >= 2) { sets.push_back({wide[c].parts[q].first, wide[c].parts[q].second}); where.push_back({(int)c, (int)q}); }
                if (sets.empty()) break;
                const int ns = (int)sets.size();
                if (!d_sets.alloc(ns) || !d_nleft.alloc(ns)) { err = "out of device memory"; return false; }
                CK(hipMemcpyAsync(d_sets.p, sets.data(), sizeof(SahSet) * ns, hipMemcpyHostToDevice, st));
                hipLaunchKernelGGL(k_sah_round, dim3(ns), dim3(64), 0, st, d_sets.p, ns, T.perm.p, tmp.p, T.in_aabb.p, bins.p, T.order, d_nleft.p);
                CK(hipGetLastError());
                std::vector<int> nleft(ns);
                CK(hipMemcpyAsync(nleft.data(), d_nleft.p, sizeof(int) * ns, hipMemcpyDeviceToHost, st));
                CK(hipStreamSynchronize(st));
                // split the parts (back to front inside a call, so that earlier part indices stay valid)
                for (int k = ns - 1; k >= 0; k--) {
                    if (nleft[k] <= 0) continue;
                    auto& parts = wide[where[k].first].parts;
                    const int q = where[k].second;
                    const std::pair<int, int> whole = parts[q];
                    parts[q] = {whole.first, whole.first + nleft[k]};
                    parts.insert(parts.begin() + q + 1, {whole.first + nleft[k], whole.second});
                }
            }
            if (M > 0) { hperm.resize(n_in); CK(hipMemcpy(hperm.data(), T.perm.p, sizeof(int) * n_in, hipMemcpyDeviceToHost)); }
            std::vector<WideCall> next;
            for (auto& wc : wide) {
                int mesh_count = 0;  // sockets in this call's whole list, as the reference counts them per call
                if (M > 0) for (int i = wc.parts.front().first; i < wc.parts.back().second; i++) mesh_count += kind[hperm[i]] == 1;
                for (auto& r : wc.parts) {
                    const int cnt = r.second - r.first;
                    if (cnt == 0) continue;
                    const int id = (int)T.nodes.size();
                    T.nodes.emplace_back();
                    T.nodes[wc.parent].children.push_back(id);
                    if (cnt > T.order) {
                        T.nodes[id].recursive = true;
                        next.push_back({id, {{r.first, r.second}}});
                    } else if (mesh_count == 0) {
                        T.nodes[id].kind = N_LEAF;
                        T.nodes[id].obj_begin = r.first;
                        T.nodes[id].obj_count = cnt;
                    } else {
                        T.nodes[id].terminal_mesh = true;
                        T.nodes[id].obj_begin = r.first;
                        T.nodes[id].obj_count = cnt;
                    }
                }
            }
            wide.swap(next);
        }
    } else {
        const int fixed_axis = (T.type == 2) ? 1 : (T.type == 3) ? 2 : 0;
        T.nodes.clear();
        T.nodes.emplace_back();  // root (inner)
        std::vector<PendingCall> calls = {{0, n_in, 0}};
        DevBuf<Call> d_calls;
        DevBuf<CallOut> d_out;
        DevBuf<int> d_slices;  // per-slice bin counts of k_bins
        DevBuf<float> d_summary;           // k_list_summary: 8 words per full batch of the level's long lists
        DevBuf<int> d_summ_off, d_call_meshes;
        if (!d_summary.alloc(((size_t)n_in / 64 + 1) * 8)) { err = "out of device memory"; return false; }
        SortWorkspace ws;
        if (!ws.counters.alloc(2 * SORT_COUNTERS)) { err = "out of device memory"; return false; }
        CK(hipMemsetAsync(ws.counters.p, 0, sizeof(int) * 2 * SORT_COUNTERS, st));

        // fixed-axis trees sort the whole list once before the recursion (pt_bvh_axis_sort.gd:193-194)
        if (T.type != 4) {
            Call c0 = {0, n_in};
            CallOut o0;
            memset(&o0, 0, sizeof o0);
            o0.axis = fixed_axis;
            const bool any_big = n_in > SORT_BIG;
            if (!d_calls.alloc(1) || !d_out.alloc(1) || !ws.reserve((size_t)n_in, n_in, any_big)) { err = "out of device memory"; return false; }
            if (any_big) {
                CK(hipMemcpyAsync(d_calls.p, &c0, sizeof c0, hipMemcpyHostToDevice, st));
                CK(hipMemcpyAsync(d_out.p, &o0, sizeof o0, hipMemcpyHostToDevice, st));
                hipLaunchKernelGGL(k_keys, dim3(1, (n_in + 1023) / 1024), dim3(256), 0, st, d_calls.p, 1, T.perm.p, T.in_aabb.p, d_out.p, keys.p);
                CK(hipGetLastError());
            }
            hipLaunchKernelGGL(k_sort_seed, dim3(1), dim3(1), 0, st, n_in, fixed_axis, ws.small.p, ws.counters.p, ws.big_a.p, ws.counters.p + 2);
            CK(hipGetLastError());
            if (!device_sort(st, ws, 0, any_big, (size_t)n_in, keys.p, T.in_aabb.p, T.perm.p, err)) return false;
            CK(hipStreamSynchronize(st));  // c0 and o0 go out of scope
        }

        for (int depth = 0; !calls.empty(); depth++) {
            if (depth >= MAX_DEPTH_LIMIT + 5) { err = "too much depth (reference assert)"; return false; }
            const int nc = (int)calls.size();
            std::vector<Call> hc(nc);
            int longest = 0;
            size_t covered = 0;
            for (int i = 0; i < nc; i++) {
                hc[i] = {calls[i].begin, calls[i].end};
                longest = std::max(longest, calls[i].end - calls[i].begin);
                covered += (size_t)(calls[i].end - calls[i].begin);
            }
            const int ny = std::min(64, (longest + 1023) / 1024);  // blocks along a call's range in k_keys, k_bins, k_partition
            const bool sorts = T.type == 4, any_big = sorts && longest > SORT_BIG;
            const int set = depth & 1;
            if (!d_calls.alloc(nc) || !d_out.alloc(nc) || (sorts && !ws.reserve(covered, n_in, any_big))) { err = "out of device memory"; return false; }
            CK(hipMemcpyAsync(d_calls.p, hc.data(), sizeof(Call) * nc, hipMemcpyHostToDevice, st));
            // long lists: their batches' summaries first, by many waves (k_list_summary), for the one wave of the chain
            const bool summaries = longest >= LIST_SUMMARY_MIN;
            std::vector<int> soff;  // (lives until the level's wait, like hc)
            if (summaries) {
                soff.assign(nc, -1);
                int run = 0;
                for (int i = 0; i < nc; i++) {
                    const int len = calls[i].end - calls[i].begin;
                    if (len >= LIST_SUMMARY_MIN) { soff[i] = run; run += len / 64; }
                }
                if (!d_summ_off.alloc(nc) || !d_call_meshes.alloc(nc)) { err = "out of device memory"; return false; }
                CK(hipMemcpyAsync(d_summ_off.p, soff.data(), sizeof(int) * nc, hipMemcpyHostToDevice, st));
                CK(hipMemsetAsync(d_call_meshes.p, 0, sizeof(int) * nc, st));
                const int sy = std::max(1, std::min(256, (longest / 64 + 15) / 16));  // four waves a block, four batches a wave at least
                hipLaunchKernelGGL(k_list_summary, dim3(nc, sy), dim3(256), 0, st, d_calls.p, nc, d_summ_off.p, T.perm.p, T.in_aabb.p, T.in_kind.p,
                                   d_summary.p, d_call_meshes.p);
                CK(hipGetLastError());
            }
            hipLaunchKernelGGL(k_list_box, dim3(nc), dim3(64), 0, st, d_calls.p, nc, T.perm.p, T.in_aabb.p, T.in_kind.p, T.type, fixed_axis, d_out.p,
                               sorts ? ws.small.p : nullptr, ws.big_a.p, ws.counters.p + set * SORT_COUNTERS, ws.counters.p + (set ^ 1) * SORT_COUNTERS,
                               summaries ? d_summ_off.p : nullptr, d_summary.p, d_call_meshes.p);
            CK(hipGetLastError());
            std::vector<CallOut> ho(nc);
            if (sorts) {
                if (any_big) {
                    hipLaunchKernelGGL(k_keys, dim3(nc, ny), dim3(256), 0, st, d_calls.p, nc, T.perm.p, T.in_aabb.p, d_out.p, keys.p);
                    CK(hipGetLastError());
                }
                if (!device_sort(st, ws, set, any_big, covered, keys.p, T.in_aabb.p, T.perm.p, err)) return false;
            }
            // bins and the stable partition follow on the device; the host only needs the outcome
            const bool at_limit = depth >= MAX_DEPTH_LIMIT;
            if (ny == 1) {
                hipLaunchKernelGGL(k_bin_partition, dim3(nc), dim3(256), 0, st, d_calls.p, nc, T.perm.p, T.in_aabb.p, T.order, at_limit ? 1 : 0, d_out.p);
                CK(hipGetLastError());
            } else {
                if (!d_slices.alloc((size_t)nc * ny * 4 * MAXB)) { err = "out of device memory"; return false; }
                hipLaunchKernelGGL(k_bins, dim3(nc, ny), dim3(256), 0, st, d_calls.p, nc, T.perm.p, T.in_aabb.p, T.order, at_limit ? 1 : 0, d_out.p, bins.p, d_slices.p);
                CK(hipGetLastError());
                hipLaunchKernelGGL(k_partition, dim3(nc, ny), dim3(256), 0, st, d_calls.p, nc, T.perm.p, tmp.p, d_out.p, bins.p, d_slices.p);
                CK(hipGetLastError());
                hipLaunchKernelGGL(k_partition_copy, dim3(nc, ny), dim3(256), 0, st, d_calls.p, nc, T.perm.p, tmp.p, d_out.p);
                CK(hipGetLastError());
            }
            CK(hipMemcpyAsync(ho.data(), d_out.p, sizeof(CallOut) * nc, hipMemcpyDeviceToHost, st));
            {
                const auto w0 = std::chrono::steady_clock::now();
                CK(hipStreamSynchronize(st));
                t_wait += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w0).count();
            }
            std::vector<int> segc(nc, T.order);
            std::vector<unsigned char> mv(nc, 0);  // calls that moved elements: those with more than one non-empty bin (:314-325)
            for (int i = 0; i < nc; i++) {
                if (ho[i].error) { err = "a node's centre falls outside its list's bins (reference assert 0 <= index < segment_count)"; return false; }
                const int mc = ho[i].mesh_count;
                if (at_limit && mc > 0 && mc < T.order) segc[i] = T.order - mc;  // segment_count (:276,288-296)
                int nonempty = 0;
                for (int b = 0; b < segc[i]; b++) nonempty += ho[i].counts[b] > 0;
                mv[i] = nonempty > 1;
            }

            // host: one new node per non-empty bin, in bin order (:327-371)
            std::vector<PendingCall> next;
            next.reserve((size_t)nc * 4);
            T.nodes.reserve(T.nodes.size() + (size_t)nc * T.order);
            for (int i = 0; i < nc; i++) {
                const int n = calls[i].end - calls[i].begin;
                std::pair<int, int> ranges[MAXB];  // [begin,end) of each bin list, in order
                int n_ranges = 0;
                if (mv[i]) {
                    int run = calls[i].begin;
                    for (int b = 0; b < segc[i]; b++) { ranges[n_ranges++] = {run, run + ho[i].counts[b]}; run += ho[i].counts[b]; }
                } else {
                    int half = n / 2;  // a single occupied bin is cut in halves
                    ranges[n_ranges++] = {calls[i].begin, calls[i].begin + half};
                    ranges[n_ranges++] = {calls[i].begin + half, calls[i].end};
                }
                const int mesh_count = ho[i].mesh_count;
                T.nodes[calls[i].parent].children.reserve(n_ranges);
                for (int q = 0; q < n_ranges; q++) {
                    const std::pair<int, int>& r = ranges[q];
                    int cnt = r.second - r.first;
                    if (cnt == 0) continue;
                    int id = (int)T.nodes.size();
                    T.nodes.emplace_back();
                    T.nodes[calls[i].parent].children.push_back(id);
                    if (cnt > T.order && depth < MAX_DEPTH_LIMIT) {
                        T.nodes[id].recursive = true;
                        next.push_back({r.first, r.second, id});
                    } else if (mesh_count == 0) {
                        T.nodes[id].kind = N_LEAF;
                        T.nodes[id].obj_begin = r.first;
                        T.nodes[id].obj_count = cnt;
                    } else {
                        T.nodes[id].terminal_mesh = true;
                        T.nodes[id].obj_begin = r.first;  // resolved below
                        T.nodes[id].obj_count = cnt;
                    }
                }
            }
            calls.swap(next);
        }
    }
    CK(hipStreamSynchronize(st));
    lap("levels");

    // terminal bins of lists that contain sockets: sockets become children in list order, the objects
    // go to one leaf child that is added last (pt_bvh_axis_sort.gd:349-369).  These lists are tiny.
    {
        bool any = false;
        for (auto& nd : T.nodes) any = any || nd.terminal_mesh;
        if (any) {
            std::vector<int> hperm(n_in);
            CK(hipMemcpy(hperm.data(), T.perm.p, sizeof(int) * n_in, hipMemcpyDeviceToHost));
            const int n_nodes0 = (int)T.nodes.size();
            for (int id = 0; id < n_nodes0; id++) {
                if (!T.nodes[id].terminal_mesh) continue;
                int b = T.nodes[id].obj_begin, cnt = T.nodes[id].obj_count;
                std::vector<int> objs, socks;
                for (int k = 0; k < cnt; k++) (kind[hperm[b + k]] == 1 ? socks : objs).push_back(hperm[b + k]);
                for (size_t k = 0; k < objs.size(); k++) hperm[b + k] = objs[k];
                for (size_t k = 0; k < socks.size(); k++) hperm[b + objs.size() + k] = socks[k];
                if ((int)socks.size() + (objs.empty() ? 0 : 1) > T.order) { err = "a node would hold more children than the order allows (reference assert)"; return false; }
                for (int s : socks) {
                    int sid = (int)T.nodes.size();
                    T.nodes.emplace_back();
                    T.nodes[sid].kind = N_SOCKET;
                    T.nodes[sid].mesh_slot = s - N;
                    T.nodes[id].children.push_back(sid);
                }
                if (!objs.empty()) {
                    int lid = (int)T.nodes.size();
                    T.nodes.emplace_back();
                    T.nodes[lid].kind = N_LEAF;
                    T.nodes[lid].obj_begin = b;
                    T.nodes[lid].obj_count = (int)objs.size();
                    T.nodes[id].children.push_back(lid);
                }
                T.nodes[id].obj_count = 0;
            }
            CK(hipMemcpy(T.perm.p, hperm.data(), sizeof(int) * n_in, hipMemcpyHostToDevice));
        }
    }
    if ((int)T.nodes[0].children.size() > T.order) { err = "root would hold more children than the order allows"; return false; }

    // leaf order = order of index_node calls (pt_bvh_axis_sort.gd:364-374): inside one call, split bins
    // contribute their whole subtree first and socket-list leaves immediately; plain leaves come last.
    T.leaf_order.clear();
    {
        struct Frame { int node; size_t child; };
        std::vector<Frame> stack = {{0, 0}};
        while (!stack.empty()) {
            Frame& f = stack.back();
            HNode& nd = T.nodes[f.node];
            if (f.child < nd.children.size()) {
                int c = nd.children[f.child++];
                if (T.nodes[c].recursive) stack.push_back({c, 0});
                else if (T.nodes[c].terminal_mesh) {
                    for (int g : T.nodes[c].children) if (T.nodes[g].kind == N_LEAF) T.leaf_order.push_back(g);
                }
                continue;
            }
            for (int c : nd.children) if (T.nodes[c].kind == N_LEAF) T.leaf_order.push_back(c);
            stack.pop_back();
        }
    }
    T.own_object_count = 0;
    for (int l : T.leaf_order) T.own_object_count += T.nodes[l].obj_count;

    lap("leaf order");
    // boxes: leaves, sockets, then inner nodes bottom-up
    const int nn = (int)T.nodes.size();
    if (!T.node_aabb.alloc((size_t)nn * 6)) { err = "out of device memory"; return false; }
    CK(hipMemsetAsync(T.node_aabb.p, 0, sizeof(float) * 6 * nn, st));
    {
        // depth of every node, then inner nodes deepest first
        std::vector<int> depth(nn, 0), order_bfs;
        order_bfs.reserve(nn);
        order_bfs.push_back(0);
        int maxd = 0;
        for (size_t q = 0; q < order_bfs.size(); q++) {
            int x = order_bfs[q];
            for (int c : T.nodes[x].children) { depth[c] = depth[x] + 1; maxd = std::max(maxd, depth[c]); order_bfs.push_back(c); }
        }
        // one upload: [leaf node | first object | object count] per leaf, then [node | first child slot | child count]
        // per inner node grouped by depth (deepest first), then the child lists
        std::vector<int> ln, ob, oc;
        ln.reserve(nn); ob.reserve(nn); oc.reserve(nn);
        std::vector<int> per_depth(maxd + 2, 0);  // inner nodes of each depth
        size_t n_child_slots = 0;
        for (int i = 0; i < nn; i++) {
            if (T.nodes[i].kind == N_LEAF) { ln.push_back(i); ob.push_back(T.nodes[i].obj_begin); oc.push_back(T.nodes[i].obj_count); }
            else if (T.nodes[i].kind == N_INNER) { per_depth[depth[i]]++; n_child_slots += T.nodes[i].children.size(); }
        }
        const int nl = (int)ln.size();
        std::vector<int> group_first, group_count, place(maxd + 2, 0);
        {
            int at = 0;
            for (int d = maxd; d >= 0; d--) {
                place[d] = at;
                if (per_depth[d] == 0) continue;
                group_first.push_back(at);
                group_count.push_back(per_depth[d]);
                at += per_depth[d];
            }
        }
        int n_inner = 0;
        for (int d = 0; d <= maxd; d++) n_inner += per_depth[d];
        std::vector<int> nd(n_inner), cb(n_inner), cc(n_inner), ch;
        for (int i = 0; i < nn; i++) if (T.nodes[i].kind == N_INNER) nd[place[depth[i]]++] = i;  // (ascending node index within a depth)
        ch.reserve(n_child_slots);
        for (int k = 0; k < n_inner; k++) {
            const HNode& x = T.nodes[nd[k]];
            cb[k] = (int)ch.size();
            cc[k] = (int)x.children.size();
            ch.insert(ch.end(), x.children.begin(), x.children.end());
        }
        const int ni = (int)nd.size();
        std::vector<int> packed;
        packed.reserve((size_t)3 * nl + 3 * ni + ch.size());
        packed.insert(packed.end(), ln.begin(), ln.end());
        packed.insert(packed.end(), ob.begin(), ob.end());
        packed.insert(packed.end(), oc.begin(), oc.end());
        packed.insert(packed.end(), nd.begin(), nd.end());
        packed.insert(packed.end(), cb.begin(), cb.end());
        packed.insert(packed.end(), cc.begin(), cc.end());
        packed.insert(packed.end(), ch.begin(), ch.end());
        DevBuf<int> d_packed;
        if (!d_packed.alloc(packed.size())) { err = "out of device memory"; return false; }
        if (!packed.empty()) CK(hipMemcpyAsync(d_packed.p, packed.data(), sizeof(int) * packed.size(), hipMemcpyHostToDevice, st));
        const int *d_ln = d_packed.p, *d_ob = d_ln + nl, *d_oc = d_ob + nl, *d_nd = d_oc + nl, *d_cb = d_nd + ni, *d_cc = d_cb + ni, *d_ch = d_cc + ni;
        if (nl) {
            hipLaunchKernelGGL(k_leaf_boxes, dim3((nl + 127) / 128), dim3(128), 0, st, nl, d_ln, d_ob, d_oc, T.perm.p, T.in_aabb.p, T.node_aabb.p);
            CK(hipGetLastError());
        }
        for (int i = 0; i < nn; i++) if (T.nodes[i].kind == N_SOCKET) {
            hipLaunchKernelGGL(k_copy_box, dim3(1), dim3(1), 0, st, T.in_aabb.p, N + T.nodes[i].mesh_slot, T.node_aabb.p, i);
            CK(hipGetLastError());
        }
        for (size_t g = 0; g < group_first.size(); g++) {
            const int k = group_count[g], f = group_first[g];
            hipLaunchKernelGGL(k_inner_boxes, dim3((k + 127) / 128), dim3(128), 0, st, k, d_nd + f, d_cb + f, d_cc + f, d_ch, T.node_aabb.p);
            CK(hipGetLastError());
        }
        CK(hipStreamSynchronize(st));
    }
    lap("boxes");
    return true;
}

}  // namespace

// index_tree order + node bytes + object id table (pt_bvh_tree.gd:488-524,780-791,910-959), written to
// device and/or host memory; with `state`, also everything a later refit needs.
bool BvhImpl::emit(hipStream_t st, uint8_t* d_node_bytes, int32_t* d_object_ids, uint8_t* node_bytes, int32_t* object_ids,
                   PtkRefitState* state, std::string& err) {
    struct Ref { BvhImpl* tree; int node; };
    const int total = flat_node_count();
    std::vector<Ref> flat;
    flat.reserve(total);
    // per tree: offset of its boxes in the concatenated box buffer, offset of its ids in the id table
    std::vector<BvhImpl*> trees = {this};
    for (auto& m : meshes) trees.push_back(m.get());
    std::vector<int> box_off(trees.size()), id_off(trees.size());
    int bo = 0, io = 0;
    for (size_t t = 0; t < trees.size(); t++) { box_off[t] = bo; bo += trees[t]->own_nodes(); id_off[t] = io; io += trees[t]->own_object_count; }
    auto tree_slot = [&](BvhImpl* t) { for (size_t i = 0; i < trees.size(); i++) if (trees[i] == t) return (int)i; return 0; };
    // leaf -> first object id slot: own leaves in leaf_order, then each mesh's (create_object_ids)
    std::vector<std::vector<int>> oid(trees.size());
    for (size_t t = 0; t < trees.size(); t++) {
        oid[t].assign(trees[t]->own_nodes(), -1);
        int run = id_off[t];
        for (int l : trees[t]->leaf_order) { oid[t][l] = run; run += trees[t]->nodes[l].obj_count; }
    }
    // flatten: children of a node are consecutive; then each inner child's block, in order (index_subnodes)
    std::vector<std::vector<int>> pos(trees.size());
    for (size_t t = 0; t < trees.size(); t++) pos[t].assign(trees[t]->own_nodes(), -1);
    auto children_of = [&](Ref r, std::vector<Ref>& out) {
        out.clear();
        const HNode& nd = r.tree->nodes[r.node];
        if (nd.kind == N_SOCKET) out.push_back({meshes[nd.mesh_slot].get(), 0});
        else for (int c : nd.children) out.push_back({r.tree, c});
    };
    std::vector<int> parent_of(total, -1), depth_of(total, 0);
    {
        flat.push_back({this, 0});
        pos[0][0] = 0;
        struct Frame { Ref r; int self; std::vector<Ref> kids; size_t next; };
        std::vector<Frame> stack;
        auto open = [&](Ref r) {
            Frame f; f.r = r; f.next = 0; f.self = pos[tree_slot(r.tree)][r.node];
            children_of(r, f.kids);
            for (Ref k : f.kids) {
                const int at = (int)flat.size();
                pos[tree_slot(k.tree)][k.node] = at;
                if (at < total) { parent_of[at] = f.self; depth_of[at] = depth_of[f.self] + 1; }
                flat.push_back(k);
            }
            stack.push_back(std::move(f));
        };
        open({this, 0});
        while (!stack.empty()) {
            Frame& f = stack.back();
            if (f.next < f.kids.size()) {
                Ref k = f.kids[f.next++];
                if (k.tree->nodes[k.node].kind != N_LEAF) open(k);
                continue;
            }
            stack.pop_back();
        }
    }
    if ((int)flat.size() != total) { err = "flatten: node count mismatch"; return false; }
    std::vector<int> box_index(total), node_id(total), xform(total), first_child(total, 0), n_children(total, 0), kind(total, 0),
        slot_first(total, 0), slot_count(total, 0), mesh_of_socket(total, -1);
    std::vector<unsigned char> empty_root(total, 0);
    std::vector<Ref> kids;
    for (int i = 0; i < total; i++) {
        Ref r = flat[i];
        const HNode& nd = r.tree->nodes[r.node];
        int ts = tree_slot(r.tree);
        box_index[i] = box_off[ts] + r.node;
        kind[i] = nd.kind;
        long long idx = 0, size = 0;
        if (nd.kind == N_LEAF) {
            idx = -(long long)oid[ts][r.node] - 1; size = nd.obj_count - 1;
            slot_first[i] = oid[ts][r.node]; slot_count[i] = nd.obj_count;
        } else {
            children_of(r, kids);
            if (kids.empty()) { empty_root[i] = 1; }
            else {
                idx = pos[tree_slot(kids[0].tree)][kids[0].node]; size = (long long)kids.size();
                first_child[i] = (int)idx; n_children[i] = (int)kids.size();
            }
            if (nd.kind == N_SOCKET) mesh_of_socket[i] = nd.mesh_slot;
        }
        idx ^= size << 24;
        node_id[i] = (int32_t)(uint32_t)(unsigned long long)idx;
        xform[i] = nd.kind == N_SOCKET ? mesh_index[nd.mesh_slot] : -1;
    }
    DevBuf<float> boxes, out;
    DevBuf<int> d_box, d_id, d_xf, d_ids;
    DevBuf<unsigned char> d_empty;
    if (!boxes.alloc((size_t)bo * 6) || !out.alloc((size_t)total * 8) || !d_box.alloc(total) || !d_id.alloc(total) || !d_xf.alloc(total) ||
        !d_empty.alloc(total) || !d_ids.alloc(std::max(io, 1))) { err = "out of device memory"; return false; }
    for (size_t t = 0; t < trees.size(); t++)
        CK(hipMemcpyAsync(boxes.p + (size_t)box_off[t] * 6, trees[t]->node_aabb.p, sizeof(float) * 6 * trees[t]->own_nodes(), hipMemcpyDeviceToDevice, st));
    CK(hipMemcpyAsync(d_box.p, box_index.data(), sizeof(int) * total, hipMemcpyHostToDevice, st));
    CK(hipMemcpyAsync(d_id.p, node_id.data(), sizeof(int) * total, hipMemcpyHostToDevice, st));
    CK(hipMemcpyAsync(d_xf.p, xform.data(), sizeof(int) * total, hipMemcpyHostToDevice, st));
    CK(hipMemcpyAsync(d_empty.p, empty_root.data(), total, hipMemcpyHostToDevice, st));
    float* enc = d_node_bytes ? (float*)d_node_bytes : out.p;
    hipLaunchKernelGGL(k_encode, dim3((total + 127) / 128), dim3(128), 0, st, total, d_box.p, d_id.p, d_xf.p, d_empty.p, boxes.p, enc);
    CK(hipGetLastError());
    if (node_bytes) CK(hipMemcpyAsync(node_bytes, enc, (size_t)total * 32, hipMemcpyDeviceToHost, st));
    // object ids (and, for a refit, object boxes) by id slot: every leaf's objects at its slots, in leaf order;
    // a leaf's range in perm is contiguous, so the destination slot of every perm position is known here
    const bool want_ids = (object_ids || d_object_ids || state) && io > 0;
    int32_t* ids_dev = d_object_ids ? d_object_ids : d_ids.p;
    float* obj_box_dev = nullptr;
    if (state && io > 0) { CK(hipMalloc((void**)&state->d_obj_box, sizeof(float) * 6 * io)); obj_box_dev = state->d_obj_box; }
    if (want_ids) {
        for (size_t t = 0; t < trees.size(); t++) {
            BvhImpl* T = trees[t];
            if (T->n_in == 0) continue;
            std::vector<int> dest(T->n_in, -1);
            for (int l : T->leaf_order) {
                const HNode& nd = T->nodes[l];
                for (int k = 0; k < nd.obj_count; k++) dest[nd.obj_begin + k] = oid[t][l] + k;
            }
            DevBuf<int> d_dest;
            if (!d_dest.alloc(T->n_in)) { err = "out of device memory"; return false; }
            CK(hipMemcpyAsync(d_dest.p, dest.data(), sizeof(int) * T->n_in, hipMemcpyHostToDevice, st));
            hipLaunchKernelGGL(k_scatter_ids, dim3((T->n_in + 255) / 256), dim3(256), 0, st, T->n_in, T->perm.p, T->in_ref.p, T->in_aabb.p,
                               d_dest.p, ids_dev, obj_box_dev);
            CK(hipGetLastError());
            CK(hipStreamSynchronize(st));  // d_dest and dest go out of scope
        }
        if (object_ids) CK(hipMemcpyAsync(object_ids, ids_dev, sizeof(int) * io, hipMemcpyDeviceToHost, st));
    }
    if (state) {
        state->n_nodes = total; state->n_ids = io; state->n_meshes = (int)meshes.size();
        state->parent.assign(parent_of.begin(), parent_of.end());
        state->depth.assign(depth_of.begin(), depth_of.end());
        state->kind.assign(kind.begin(), kind.end());
        state->slot_leaf.assign((size_t)std::max(io, 1), -1);
        for (int i = 0; i < total; i++)
            for (int k = 0; k < slot_count[i]; k++) state->slot_leaf[slot_first[i] + k] = i;
        state->mesh_socket_node.assign(meshes.size(), -1);
        for (int i = 0; i < total; i++) if (mesh_of_socket[i] >= 0) state->mesh_socket_node[mesh_of_socket[i]] = i;
        state->mesh_index.assign(mesh_index.begin(), mesh_index.end());
        state->object_ids.assign((size_t)std::max(io, 0), 0);
        if (io > 0) CK(hipMemcpyAsync(state->object_ids.data(), ids_dev, sizeof(int) * io, hipMemcpyDeviceToHost, st));
        auto up = [&](int32_t** dst, const std::vector<int>& v) -> bool {
            if (hipMalloc((void**)dst, sizeof(int) * std::max<size_t>(v.size(), 1)) != hipSuccess) return false;
            return v.empty() || hipMemcpyAsync(*dst, v.data(), sizeof(int) * v.size(), hipMemcpyHostToDevice, st) == hipSuccess;
        };
        std::vector<int> zeros(total, 0);
        if (!up(&state->d_first_child, first_child) || !up(&state->d_n_children, n_children) || !up(&state->d_kind, kind) ||
            !up(&state->d_slot_first, slot_first) || !up(&state->d_slot_count, slot_count) || !up(&state->d_mesh_of_socket, mesh_of_socket) ||
            !up(&state->d_node_id, node_id) || !up(&state->d_xform, xform) || !up(&state->d_rec_off, zeros) || !up(&state->d_rec_cnt, zeros)) {
            err = "out of device memory"; return false;
        }
        // raw boxes in flat order; mesh transforms by mesh slot
        CK(hipMalloc((void**)&state->d_raw_box, sizeof(float) * 6 * total));
        hipLaunchKernelGGL(k_gather_boxes, dim3((total + 255) / 256), dim3(256), 0, st, total, d_box.p, boxes.p, state->d_raw_box);
        CK(hipGetLastError());
        CK(hipMalloc((void**)&state->d_xf12, sizeof(float) * 12 * std::max<size_t>(meshes.size(), 1)));
        if (!meshes.empty()) CK(hipMemcpyAsync(state->d_xf12, mesh_xf12.data(), sizeof(float) * 12 * meshes.size(), hipMemcpyHostToDevice, st));
    }
    CK(hipStreamSynchronize(st));
    return true;
}

bool BvhImpl::flatten(hipStream_t st, uint8_t* node_bytes, int32_t* object_ids, std::string& err) {
    return emit(st, nullptr, nullptr, node_bytes, object_ids, nullptr, err);
}
bool BvhImpl::bind(hipStream_t st, uint8_t* d_node_bytes, int32_t* d_object_ids, uint8_t* h_node_bytes, int32_t* h_object_ids,
                   PtkRefitState* state, std::string& err) {
    return emit(st, d_node_bytes, d_object_ids, h_node_bytes, h_object_ids, state, err);
}

// ---- refit: the reference's update_aabb chains, all edits of a call at once -------------------------------
// BVHNode.update_aabb (pt_bvh_tree.gd:873-882): recompute the node's box (set_aabb, :856-870); if it is not
// is_equal_approx to the old one, queue the node for upload and update the parent the same way.  PTScene calls
// it once per moved object, in order, so a node's final box -- and whether it was ever queued -- depends on the
// order of the edits wherever a change is smaller than the comparison's tolerance.  To keep that order and still
// work on many edits at once, every edit j leaves a record at each node of its path to the root.  Nodes are
// processed deepest level first, one lane per node, each lane running its node's records in edit order; the box
// of a child "as of edit j" is the box stored in the child's last record with index <= j (records carry the box
// forward when an edit did not reach them), or the child's box from before the call.
namespace {

DEV bool approx1(float a, float b) {  // Math::is_equal_approx(a, b), CMP_EPSILON = 1e-5
    if (a == b) return true;
    float tol = 0.00001f * fabsf(a);
    if (tol < 0.00001f) tol = 0.00001f;
    return fabsf(a - b) < tol;
}
DEV bool a6_equal_approx(const A6& a, const A6& b) {  // AABB::is_equal_approx: position and size, component-wise
    return approx1(a.p[0], b.p[0]) && approx1(a.p[1], b.p[1]) && approx1(a.p[2], b.p[2]) && approx1(a.s[0], b.s[0]) &&
           approx1(a.s[1], b.s[1]) && approx1(a.s[2], b.s[2]);
}

struct RefitArgs {
    const int *first_child, *n_children, *kind, *slot_first, *slot_count, *mesh_of_socket, *node_id, *xform;
    int *rec_off, *rec_cnt;
    float *raw_box, *obj_box, *xf12;
    const int *rec_node, *rec_j, *rec_child;  // per record
    int* rec_changed;
    float* rec_box;
    const int *seg_first, *seg_count;         // per node segment
    const int *ev_slot, *ev_mesh;
    const float* ev_v;
    int *out_count, *out_node;
    float* out_bytes;
};

// box of node c as edit j sees it
DEV A6 box_as_of(const RefitArgs& R, int c, int j) {
    const int cnt = R.rec_cnt[c];
    if (cnt > 0) {
        const int off = R.rec_off[c];
        int lo = 0, hi = cnt;  // first record with rec_j > j
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (R.rec_j[off + mid] <= j) lo = mid + 1; else hi = mid; }
        if (lo > 0) return a6_load(R.rec_box, off + lo - 1);
    }
    return a6_load(R.raw_box, c);
}

// Segments of this many records or more at an inner node are run by a workgroup of their own (k_refit_level_wide)
constexpr int REFIT_WIDE = 8;

__global__ void k_refit_level(RefitArgs R, int seg_begin, int n_segs) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_segs) return;
    const int first = R.seg_first[seg_begin + t], count = R.seg_count[seg_begin + t];
    const int node = R.rec_node[first];
    const int kind = R.kind[node];
    if (count >= REFIT_WIDE && kind != N_SOCKET && kind != N_LEAF) return;  // k_refit_level_wide's
    A6 cur = a6_load(R.raw_box, node);
    for (int r = first; r < first + count; r++) {
        const int j = R.rec_j[r];
        const int child_rec = R.rec_child[r];
        int changed = 0;
        if (child_rec < 0 || R.rec_changed[child_rec]) {  // the edit itself, or an update_aabb call from the child
            if (child_rec < 0) {
                if (R.ev_slot[j] >= 0) a6_store(R.obj_box, R.ev_slot[j], a6_load(R.ev_v, j * 2));  // (12 floats per event: the box is the first 6)
                else for (int k = 0; k < 12; k++) R.xf12[R.ev_mesh[j] * 12 + k] = R.ev_v[j * 12 + k];
            }
            A6 acc = a6_zero();  // set_aabb
            if (kind == N_SOCKET) {
                float tr[12];
                const int m = R.mesh_of_socket[node];
                for (int k = 0; k < 12; k++) tr[k] = R.xf12[m * 12 + k];
                acc = a6_xform(tr, box_as_of(R, R.first_child[node], j));
            } else if (kind == N_LEAF) {
                const int b = R.slot_first[node], n = R.slot_count[node];
                for (int k = 0; k < n; k++) acc = merge_aabb(acc, a6_load(R.obj_box, b + k));
            } else {
                const int b = R.first_child[node], n = R.n_children[node];
                for (int k = 0; k < n; k++) acc = merge_aabb(acc, box_as_of(R, b + k, j));
            }
            changed = a6_equal_approx(acc, cur) ? 0 : 1;
            cur = acc;
        }
        R.rec_changed[r] = changed;
        a6_store(R.rec_box, r, cur);
    }
}

// The records of one inner node with many edits on their way through it (near the root every edit of a batch passes),
// one workgroup per node.  What k_refit_level does record by record splits into steps that do not wait for each other:
// a record is live iff its edit's update_aabb call reached the node (the child's record says so); a live record's box
// is the merge of the children's boxes as of that edit, whatever the node's earlier records were; and it counts as a
// change iff it differs (is_equal_approx) from the box the live record before it left -- the node's own at the start.
// Records that are not live repeat that box.  seg_list: the segments (indices into seg_first / seg_count) to run.
__global__ __launch_bounds__(256) void k_refit_level_wide(RefitArgs R, const int* __restrict__ seg_list) {
    __shared__ int s_wave[4];
    __shared__ int s_carry;
    const int seg = seg_list[blockIdx.x];
    const int first = R.seg_first[seg], count = R.seg_count[seg];
    const int node = R.rec_node[first];
    const int cb = R.first_child[node], cn = R.n_children[node];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_carry = -1;  // last live record so far
    __syncthreads();
    for (int r0 = first; r0 < first + count; r0 += 256) {
        const int r = r0 + (int)threadIdx.x;
        const bool in = r < first + count;
        bool live = false;
        if (in) {
            const int child_rec = R.rec_child[r];
            live = child_rec < 0 || R.rec_changed[child_rec] != 0;
            if (live) {
                const int j = R.rec_j[r];
                A6 acc = a6_zero();
                for (int k = 0; k < cn; k++) acc = merge_aabb(acc, box_as_of(R, cb + k, j));
                a6_store(R.rec_box, r, acc);
            }
        }
        // last live record at or before r: a running maximum over the workgroup
        int la = live ? r : -1;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int y = __shfl_up(la, d, 64);
            if (lane >= d && y > la) la = y;
        }
        if (lane == 63) s_wave[wave] = la;
        __syncthreads();  // (also: every live box of the chunk is stored)
        int before = s_carry;  // ... before this thread's wave
        for (int w = 0; w < wave; w++) before = s_wave[w] > before ? s_wave[w] : before;
        if (before > la) la = before;
        // the live record before r
        int prev = __shfl_up(la, 1, 64);
        if (lane == 0) prev = before;
        if (in) {
            A6 was = prev >= 0 ? a6_load(R.rec_box, prev) : a6_load(R.raw_box, node);
            if (live) {
                R.rec_changed[r] = a6_equal_approx(a6_load(R.rec_box, r), was) ? 0 : 1;
            } else {
                R.rec_changed[r] = 0;
                a6_store(R.rec_box, r, was);
            }
        }
        __syncthreads();
        if (threadIdx.x == 255) s_carry = la;
        __syncthreads();
    }
}

// after the last level: the nodes' boxes become their final ones; queued nodes are encoded for upload
__global__ void k_refit_commit(RefitArgs R, int n_segs) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_segs) return;
    const int first = R.seg_first[t], count = R.seg_count[t];
    const int node = R.rec_node[first];
    int queued = 0;
    for (int r = first; r < first + count; r++) queued |= R.rec_changed[r];
    const A6 raw = a6_load(R.rec_box, first + count - 1);
    a6_store(R.raw_box, node, raw);
    R.rec_cnt[node] = 0;  // (leaves the lookup table clean for the next call)
    if (!queued) return;
    const int k = atomicAdd(R.out_count, 1);
    R.out_node[k] = node;
    float* o = R.out_bytes + (size_t)k * 8;
    A6 a = a6_abs(raw);
    const bool zero = raw.s[0] == 0.0f && raw.s[1] == 0.0f && raw.s[2] == 0.0f;
    const float pad = zero ? 0.0f : 1e-6f;
    o[0] = a.p[0] - pad; o[1] = a.p[1] - pad; o[2] = a.p[2] - pad;
    o[3] = __int_as_float(R.node_id[node]);
    o[4] = (a.p[0] + a.s[0]) + pad; o[5] = (a.p[1] + a.s[1]) + pad; o[6] = (a.p[2] + a.s[2]) + pad;
    o[7] = __int_as_float(R.xform[node]);
}

__global__ void k_refit_index(const int* seg_first, const int* seg_count, const int* rec_node, int n_segs, int* rec_off, int* rec_cnt) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_segs) return;
    const int node = rec_node[seg_first[t]];
    rec_off[node] = seg_first[t];
    rec_cnt[node] = seg_count[t];
}

}  // namespace

PtkRefitState::~PtkRefitState() {
    for (int32_t* p : {d_first_child, d_n_children, d_kind, d_slot_first, d_slot_count, d_mesh_of_socket, d_node_id, d_xform, d_rec_off, d_rec_cnt})
        if (p) (void)hipFree(p);
    for (float* p : {d_raw_box, d_obj_box, d_xf12})
        if (p) (void)hipFree(p);
    if (d_ws) (void)hipFree(d_ws);
    if (h_ws) (void)hipHostFree(h_ws);
}

bool PtkRefitState::refit(hipStream_t st, const std::vector<PtkRefitEvent>& events, std::vector<int32_t>& nodes_out,
                          std::vector<uint8_t>& bytes_out, std::string& err) {
    nodes_out.clear();
    bytes_out.clear();
    const int n_ev = (int)events.size();
    if (n_ev == 0) return true;
    const bool timing = getenv("PTK_REFIT_TIMING") != nullptr;  // (development aid: where a call's host time goes, on stderr)
    const auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (timing) fprintf(stderr, "refit %-10s %.3f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    };
    // records: one per (edit, node on its path), grouped by node, deepest level first, edits in order inside a node
    struct Rec { int depth, node, j, idx_on_path; };
    std::vector<Rec> recs;
    recs.reserve((size_t)n_ev * 12);
    for (int j = 0; j < n_ev; j++) {
        int node = events[j].slot >= 0 ? slot_leaf[events[j].slot] : mesh_socket_node[events[j].mesh];
        if (node < 0) { err = "refit: edit does not belong to the bound tree"; return false; }
        for (int k = 0; node >= 0; node = parent[node], k++) recs.push_back({depth[node], node, j, k});
    }
    std::stable_sort(recs.begin(), recs.end(), [](const Rec& a, const Rec& b) { return a.depth != b.depth ? a.depth > b.depth : a.node < b.node; });
    lap("sorted");
    const int R = (int)recs.size();
    std::vector<int> rec_node(R), rec_j(R), rec_child(R, -1), seg_first, seg_count, level_first_seg, level_n_segs;
    // the record of edit j at the child on its path: the one record of that edit one step lower on the path
    std::vector<int> last_of_edit(n_ev, -1);
    {
        int cur_depth = -1;
        for (int r = 0; r < R; r++) {
            rec_node[r] = recs[r].node; rec_j[r] = recs[r].j;
            if (r == 0 || recs[r].node != recs[r - 1].node || recs[r].depth != recs[r - 1].depth) {
                if (recs[r].depth != cur_depth) { cur_depth = recs[r].depth; level_first_seg.push_back((int)seg_first.size()); level_n_segs.push_back(0); }
                seg_first.push_back(r); seg_count.push_back(0);
                level_n_segs.back()++;
            }
            seg_count.back()++;
        }
        // a path has one node per level and the levels are laid out deepest first, so the last record seen for
        // an edit is the one at the path's next lower node
        for (int r = 0; r < R; r++) {
            rec_child[r] = recs[r].idx_on_path == 0 ? -1 : last_of_edit[recs[r].j];
            last_of_edit[recs[r].j] = r;
        }
    }
    const int S = (int)seg_first.size();
    // device scratch, one block: [inputs: rec_node R | rec_j R | rec_child R | seg_first S | seg_count S | ev_slot n | ev_mesh n |
    // ev_v 12 n | wide_list W] [outputs: out_count 1 | out_node S | out_bytes 8 S] [rec_changed R | rec_box 6 R]; the host block mirrors
    // the inputs for the upload and the outputs for the download
    // per level: its segments that go to k_refit_level_wide
    std::vector<int> wide_list, wide_first(level_first_seg.size() + 1, 0);
    for (size_t l = 0; l < level_first_seg.size(); l++) {
        for (int q = level_first_seg[l]; q < level_first_seg[l] + level_n_segs[l]; q++) {
            const int k = kind[rec_node[seg_first[q]]];
            if (seg_count[q] >= REFIT_WIDE && k != N_SOCKET && k != N_LEAF) wide_list.push_back(q);
        }
        wide_first[l + 1] = (int)wide_list.size();
    }
    const size_t W = wide_list.size();
    const size_t in_words = (size_t)3 * R + 2 * (size_t)S + 14 * (size_t)n_ev + W, out_words = 1 + 9 * (size_t)S, tmp_words = 7 * (size_t)R;
    const size_t d_need = (in_words + out_words + tmp_words) * 4, h_need = std::max(in_words, out_words) * 4;
    if (d_need > d_ws_bytes) {
        if (d_ws) { (void)hipFree(d_ws); d_ws = nullptr; }
        d_ws_bytes = 0;
        if (hipMalloc(&d_ws, d_need * 2) != hipSuccess) { err = "out of device memory"; return false; }
        d_ws_bytes = d_need * 2;
    }
    if (h_need > h_ws_bytes) {
        if (h_ws) { (void)hipHostFree(h_ws); h_ws = nullptr; }
        h_ws_bytes = 0;
        if (hipHostMalloc(&h_ws, h_need * 2) != hipSuccess) { err = "out of pinned host memory"; return false; }
        h_ws_bytes = h_need * 2;
    }
    int* const d_in = (int*)d_ws;
    int* const d_rec_node = d_in; int* const d_rec_j = d_rec_node + R; int* const d_rec_child = d_rec_j + R;
    int* const d_seg_first = d_rec_child + R; int* const d_seg_count = d_seg_first + S;
    int* const d_ev_slot = d_seg_count + S; int* const d_ev_mesh = d_ev_slot + n_ev;
    float* const d_ev_v = (float*)(d_ev_mesh + n_ev);
    int* const d_wide = (int*)(d_ev_v + (size_t)12 * n_ev);
    int* const d_out = d_in + in_words;
    int* const d_out_count = d_out; int* const d_out_node = d_out + 1; float* const d_out_bytes = (float*)(d_out + 1 + S);
    int* const d_rec_changed = d_out + out_words; float* const d_rec_box = (float*)(d_rec_changed + R);
    {
        int* h = (int*)h_ws;
        memcpy(h, rec_node.data(), sizeof(int) * R); h += R;
        memcpy(h, rec_j.data(), sizeof(int) * R); h += R;
        memcpy(h, rec_child.data(), sizeof(int) * R); h += R;
        memcpy(h, seg_first.data(), sizeof(int) * S); h += S;
        memcpy(h, seg_count.data(), sizeof(int) * S); h += S;
        for (int j = 0; j < n_ev; j++) h[j] = events[j].slot;
        h += n_ev;
        for (int j = 0; j < n_ev; j++) h[j] = events[j].mesh;
        h += n_ev;
        for (int j = 0; j < n_ev; j++) memcpy(h + (size_t)j * 12, events[j].v, 48);
        h += (size_t)12 * n_ev;
        if (W) memcpy(h, wide_list.data(), sizeof(int) * W);
    }
    lap("packed");
    CK(hipMemcpyAsync(d_in, h_ws, in_words * 4, hipMemcpyHostToDevice, st));
    CK(hipMemsetAsync(d_out_count, 0, sizeof(int), st));
    RefitArgs A;
    A.first_child = d_first_child; A.n_children = d_n_children; A.kind = d_kind; A.slot_first = d_slot_first; A.slot_count = d_slot_count;
    A.mesh_of_socket = d_mesh_of_socket; A.node_id = d_node_id; A.xform = d_xform; A.rec_off = d_rec_off; A.rec_cnt = d_rec_cnt;
    A.raw_box = d_raw_box; A.obj_box = d_obj_box; A.xf12 = d_xf12;
    A.rec_node = d_rec_node; A.rec_j = d_rec_j; A.rec_child = d_rec_child; A.rec_changed = d_rec_changed; A.rec_box = d_rec_box;
    A.seg_first = d_seg_first; A.seg_count = d_seg_count; A.ev_slot = d_ev_slot; A.ev_mesh = d_ev_mesh; A.ev_v = d_ev_v;
    A.out_count = d_out_count; A.out_node = d_out_node; A.out_bytes = d_out_bytes;
    hipLaunchKernelGGL(k_refit_index, dim3((S + 127) / 128), dim3(128), 0, st, d_seg_first, d_seg_count, d_rec_node, S, d_rec_off, d_rec_cnt);
    CK(hipGetLastError());
    for (size_t l = 0; l < level_first_seg.size(); l++) {  // deepest level first
        hipLaunchKernelGGL(k_refit_level, dim3((level_n_segs[l] + 63) / 64), dim3(64), 0, st, A, level_first_seg[l], level_n_segs[l]);
        CK(hipGetLastError());
        if (wide_first[l + 1] > wide_first[l]) {
            hipLaunchKernelGGL(k_refit_level_wide, dim3((unsigned)(wide_first[l + 1] - wide_first[l])), dim3(256), 0, st, A, d_wide + wide_first[l]);
            CK(hipGetLastError());
        }
    }
    hipLaunchKernelGGL(k_refit_commit, dim3((S + 127) / 128), dim3(128), 0, st, A, S);
    CK(hipGetLastError());
    // count, node list and records come back in one copy (at most one record per node that had an edit on its path)
    lap("launched");
    CK(hipMemcpyAsync(h_ws, d_out, out_words * 4, hipMemcpyDeviceToHost, st));
    CK(hipStreamSynchronize(st));
    lap("done");
    const int* h_out = (const int*)h_ws;
    const int n_out = h_out[0];
    if (n_out < 0 || n_out > S) { err = "refit: bad record count"; return false; }
    if (n_out > 0) {
        nodes_out.assign(h_out + 1, h_out + 1 + n_out);
        bytes_out.resize((size_t)n_out * 32);
        memcpy(bytes_out.data(), h_out + 1 + S, (size_t)n_out * 32);
    }
    return true;
}

std::shared_ptr<PtkBvh> ptk_bvh_build_device(hipStream_t stream, int order, int type, int n_objects,
                                             const float* aabbs6, const int32_t* object_ids,
                                             const std::vector<std::shared_ptr<PtkBvh>>& mesh_trees,
                                             const float* mesh_transforms12, const int32_t* mesh_indices,
                                             std::string& err) {
    if (order > MAXB) { err = "the device builder bins for order <= 8"; return nullptr; }
    if (type == PTK_BVH_SAH_TYPE && order != 8) { err = "the SAH builder makes 8-wide nodes (order 8)"; return nullptr; }
    auto T = std::make_shared<BvhImpl>();
    T->order = order;
    T->type = type;
    T->n_objects = n_objects;
    for (size_t m = 0; m < mesh_trees.size(); m++) {
        // flatten() resolves one level of sockets: a mesh tree with sockets of its own is refused
        if (!mesh_trees[m] || mesh_trees[m]->mesh_count() != 0) { err = "a mesh tree must not contain mesh sockets itself"; return nullptr; }
        T->meshes.push_back(std::static_pointer_cast<BvhImpl>(mesh_trees[m]));
        T->mesh_index.push_back(mesh_indices[m]);
        T->mesh_xf12.insert(T->mesh_xf12.end(), mesh_transforms12 + m * 12, mesh_transforms12 + m * 12 + 12);
    }
    if (!build_tree(stream, *T, aabbs6, object_ids, mesh_transforms12, err)) return nullptr;
    return T;
}
