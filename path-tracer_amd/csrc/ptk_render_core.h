// ptk_render_core.h -- device functions shared by the render kernels: ray/box/primitive tests,
// BVH traversal, hit resolution, textures, nested dielectrics and the bounce, each following the
// operation order of shaders/ray_tracer.comp (line numbers cited per function).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ptk_device.h"
#include "ptk_math.h"

namespace {

constexpr int IS_SPHERE = 1;   // ray_tracer.comp:127-130
constexpr int IS_PLANE = 2;
constexpr int IS_TRIANGLE = 3;
constexpr int TEX_IS_SAMPLED = 1;  // :133-134
constexpr int TEX_IS_SPATIAL = 2;
constexpr int MAX_OVERLAP = 5;     // :156
constexpr int32_t INT_NEG_LIMIT = (int32_t)0x80000000u;
constexpr uint32_t SOCKET_MARK = 0x7F000000u;  // stack entry: mesh socket, low 24 bits = its node index
#define PT_INF (__builtin_inff())

struct Ray {
    f3 o;
    float start;
    f3 d;
    float end;
};

struct Hit {
    int hit;
    int object_id;
    int transform_index;
    float u, v;
};

struct Lane {
    // shader globals that persist over the path (:159-237)
    float current_IOR;
    int inside_of_count;
    int in_mat[MAX_OVERLAP];
    int in_pri[MAX_OVERLAP];
    int is_emissive;
    int refraction_bounces;
    int ignore_attenuation;
    float seed;
    int node_hit_count, object_hit_count;
};

PTK_DEV int bfe_i(int v, int off, int bits) { return ((int)((uint32_t)v << (32 - off - bits))) >> (32 - bits); }
PTK_DEV int iabs(int v) { return v < 0 ? -v : v; }
PTK_DEV bool in_range(float value, float start, float end) { return start <= value && end >= value; }  // :548-551

// ---- random (:464-507) -----------------------------------------------------
PTK_DEV float pt_rand(float seed) { return pfract((seed * 23489.52364f) / 0.0836f); }
PTK_DEV float noise(Lane& L) {
    float s = L.seed;
    L.seed = L.seed + 1.0f;
    return pfract(psin(s) * 43758.5453123f);
}
PTK_DEV float pcg(Lane& L) {
    float s = L.seed;
    L.seed = L.seed + 1.0f;
    uint32_t state = fbits(s) * 747796405u + 2891336453u;
    uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
    return (float)((word >> 22u) ^ word) / 4294967295.0f;
}
PTK_DEV f3 random_point_on_sphere(Lane& L) {
    float theta1 = pcg(L) * (3.1415926535897932385f * 2.0f);
    float theta2 = pacos(2.0f * pcg(L) - 1.0f);
    float s1, c1, s2, c2;
    psincos(theta1, s1, c1);
    psincos(theta2, s2, c2);
    return mk3(c1 * s2, c2, s1 * s2);
}
PTK_DEV float hash3(float x, float y, float z) {
    uint32_t px = fbits(x), py = fbits(y), pz = fbits(z);
    uint32_t qx = 1103515245u * ((px >> 1u) ^ py);
    uint32_t qy = 1103515245u * ((py >> 1u) ^ pz);
    uint32_t qz = 1103515245u * ((pz >> 1u) ^ px);
    uint32_t h32 = 1103515245u * ((qx ^ qz) ^ (qy >> 3u));
    uint32_t n = h32 ^ (h32 >> 16u);
    return (float)n * (1.0f / 2147483648.0f);
}

// ---- small helpers ------------------------------------------------------------
PTK_DEV float offset_axis(float p, float n) {  // :532-546
    int32_t off = pf2i(n * 256.0f);
    int32_t bits = (int32_t)(fbits(p) + (uint32_t)((p < 0.0f) ? -off : off));
    float p_int = bitsf((uint32_t)bits);
    return (pabs(p) < (1.0f / 32.0f)) ? (p + (1.0f / 65536.0f) * n) : p_int;
}
PTK_DEV f3 offset_ray(f3 point, f3 normal) {
    return mk3(offset_axis(point.x, normal.x), offset_axis(point.y, normal.y), offset_axis(point.z, normal.z));
}
PTK_DEV Ray new_ray(f3 origin, f3 direction) {  // :444-446
    Ray r;
    r.o = origin;
    r.start = 0.0f;
    r.d = normalize3(direction);
    r.end = PT_INF;
    return r;
}
PTK_DEV bool near_zero1(float f) { return pabs(f) < 1e-6f; }
PTK_DEV bool near_zero3(f3 v) { return pabs(v.x) < 1e-6f && pabs(v.y) < 1e-6f && pabs(v.z) < 1e-6f; }
PTK_DEV float reflectance(float eta, float cos_theta) {  // :456-461
    float r0 = (1.0f - eta) / (1.0f + eta);
    r0 = r0 * r0;
    return r0 + (1.0f - r0) * ppow(1.0f - cos_theta, 5.0f);
}
PTK_DEV m44 load_xform(const PtkParams& P, int i) {
    m44 m;
    const float4* p = P.transforms + (size_t)i * 4;
    m.c0 = p[0]; m.c1 = p[1]; m.c2 = p[2]; m.c3 = p[3];
    return m;
}

// inverse(mat4) of transform i (:1021, 1065), from the table made when the matrices were bound or edited
PTK_DEV m44 load_inv_xform(const PtkParams& P, int i) {
    m44 m;
    const float4* p = P.inv_transforms + (size_t)i * 4;
    m.c0 = p[0]; m.c1 = p[1]; m.c2 = p[2]; m.c3 = p[3];
    return m;
}

// slab test, :670-689.  The swap is the shader's arithmetic swap (a=a+b; b=a-b; a=a-b).
PTK_DEV float hit_aabb(float4 bmin, float4 bmax, f3 o, f3 inv, int neg_mask, float start, float end) {
    float t0, t1;
#define PTK_SLAB(MINV, MAXV, O, INV, BIT)                         \
    t0 = ((MINV) - (O)) * (INV);                                  \
    t1 = ((MAXV) - (O)) * (INV);                                  \
    if (neg_mask & (BIT)) { t0 = t0 + t1; t1 = t0 - t1; t0 = t0 - t1; } \
    if (t0 > start) start = t0;                                   \
    if (t1 < end) end = t1;                                       \
    if (end <= start) return -1.0f;
    PTK_SLAB(bmin.x, bmax.x, o.x, inv.x, 1)
    PTK_SLAB(bmin.y, bmax.y, o.y, inv.y, 2)
    PTK_SLAB(bmin.z, bmax.z, o.z, inv.z, 4)
#undef PTK_SLAB
    return start;
}

// Ray-sphere test on a sphere given by value (:803-841).
// `either_root` (optional): set when the first root is taken although the second lies in the range too and is
// the smaller one -- the one case in which a smaller ray.end could have led to the other root (see trav_leaf).
PTK_DEV void hit_sphere_rec(Ray& ray, Hit& h, int object_id, int cur_xf, f3 center, float radius, bool* either_root = nullptr) {
    f3 oc = ray.o - center;
    float half_b = dot3(-oc, ray.d);
    float r_sqrd = radius * radius;
    f3 temp = oc + ray.d * half_b;
    float discriminant = r_sqrd - dot3(temp, temp);
    if (discriminant < 0.0f) return;
    float c = dot3(oc, oc) - r_sqrd;
    float sqrtd = psqrt(discriminant);
    float q = half_b + psign(half_b) * sqrtd;
    float root = c / q;
    if (!in_range(root, ray.start, ray.end)) {
        root = q;
        if (!in_range(root, ray.start, ray.end)) return;
    } else if (either_root && q < root && in_range(q, ray.start, ray.end)) {
        *either_root = true;
    }
    h.hit = 1; h.object_id = object_id; h.transform_index = cur_xf;
    ray.end = root;
}
template <bool TRACE>
PTK_DEV void hit_sphere_gems(const PtkParams& P, Ray& ray, Hit& h, int object_id, int cur_xf, int* cnt) {
    float4 s = P.spheres[(size_t)bfe_i(object_id, 0, 24) * 2];
    if (TRACE) cnt[PTK_CNT_SPHERE_TESTS]++;
    hit_sphere_rec(ray, h, object_id, cur_xf, ld3(s), s.w);
}

PTK_DEV void hit_sphere_plain(const PtkParams& P, Ray& ray, Hit& h, int object_id, int cur_xf) {  // :691-727
    float4 s = P.spheres[(size_t)bfe_i(object_id, 0, 24) * 2];
    f3 oc = ray.o - ld3(s);
    float a = dot3(ray.d, ray.d);
    float half_b = dot3(oc, ray.d);
    float c = dot3(oc, oc) - s.w * s.w;
    float discriminant = half_b * half_b - a * c;
    if (discriminant < 0.0f) return;
    float sqrtd = psqrt(discriminant);
    float root = (-half_b - sqrtd) / a;
    if (!in_range(root, ray.start, ray.end)) {
        root = (-half_b + sqrtd) / a;
        if (!in_range(root, ray.start, ray.end)) return;
    }
    h.hit = 1; h.object_id = object_id; h.transform_index = cur_xf;
    ray.end = root;
}

// Moeller-Trumbore on a triangle given by value (:926-969).
PTK_DEV void hit_triangle_rec(Ray& ray, Hit& h, int object_id, int cur_xf, f3 A, f3 B, f3 C) {
    f3 E1 = B - A, E2 = C - A;
    f3 N = cross3(E1, E2);
    float det = -dot3(ray.d, N);
    if (near_zero1(det)) return;
    float invdet = 1.0f / det;
    f3 AO = ray.o - A;
    float tt = dot3(AO, N) * invdet;
    if (!in_range(tt, ray.start, ray.end)) return;
    f3 DAO = cross3(AO, ray.d);
    float u = dot3(E2, DAO) * invdet;
    float v = -dot3(E1, DAO) * invdet;
    if (u >= 0.0f && v >= 0.0f && (u + v) <= 1.0f) {
        h.hit = 1; h.object_id = object_id; h.transform_index = cur_xf;
        h.u = u; h.v = v;
        ray.end = tt;
    }
}
// The same test on a leaf record (PtkWavefront::leaf_records): E1, E2 and the first two components of N = E1 x E2 come
// with the record, made by these very operations when the record was.
PTK_DEV void hit_triangle_pre(Ray& ray, Hit& h, int object_id, int cur_xf, f3 A, f3 E1, f3 E2, float Nx, float Ny) {
    f3 N = mk3(Nx, Ny, E1.x * E2.y - E2.x * E1.y);
    float det = -dot3(ray.d, N);
    if (near_zero1(det)) return;
    float invdet = 1.0f / det;
    f3 AO = ray.o - A;
    float tt = dot3(AO, N) * invdet;
    if (!in_range(tt, ray.start, ray.end)) return;
    f3 DAO = cross3(AO, ray.d);
    float u = dot3(E2, DAO) * invdet;
    float v = -dot3(E1, DAO) * invdet;
    if (u >= 0.0f && v >= 0.0f && (u + v) <= 1.0f) {
        h.hit = 1; h.object_id = object_id; h.transform_index = cur_xf;
        h.u = u; h.v = v;
        ray.end = tt;
    }
}
template <bool TRACE>
PTK_DEV void hit_triangle_mt(const PtkParams& P, Ray& ray, Hit& h, int object_id, int cur_xf, int* cnt) {
    const float4* t = P.triangles + (size_t)bfe_i(object_id, 0, 24) * 5;
    if (TRACE) cnt[PTK_CNT_TRI_TESTS]++;
    hit_triangle_rec(ray, h, object_id, cur_xf, ld3(t[0]), ld3(t[1]), ld3(t[2]));
}

PTK_DEV void hit_plane_rec(Ray& ray, Hit& h, int index, int cur_xf, float4 pl) {  // :851-875
    f3 n = ld3(pl);
    float t = (pl.w - dot3(ray.o, n)) / dot3(n, ray.d);
    if (t != t || pabs(t) == PT_INF) return;
    if (!in_range(t, ray.start, ray.end)) return;
    h.hit = 1; h.object_id = index + (IS_PLANE << 24); h.transform_index = cur_xf;
    ray.end = t;
}
PTK_DEV void hit_plane(const PtkParams& P, Ray& ray, Hit& h, int index, int cur_xf) {
    hit_plane_rec(ray, h, index, cur_xf, P.planes[(size_t)index * 2]);
}
// The planes that can be hit, in index order, before anything else of a search (:1204): from the kernel's arguments
// when they fit there, else through the list in memory.
PTK_DEV void hit_live_planes(const PtkParams& P, Ray& ray, Hit& h) {
    if (P.n_arg_planes >= 0) {
#pragma unroll
        for (int i = 0; i < PTK_ARG_PLANES; ++i)
            if (i < P.n_arg_planes) hit_plane_rec(ray, h, P.arg_plane_index[i], -1, P.arg_plane[i]);
    } else {
        for (int i = 0; i < P.n_live_planes; ++i) hit_plane(P, ray, h, P.live_planes[i], -1);
    }
}

template <bool TRACE>
PTK_DEV void hit_object(const PtkParams& P, Lane& L, Ray& ray, Hit& h, int object_id, int cur_xf, int* cnt) {  // :979-996
    int type = bfe_i(object_id, 24, 8);
    L.object_hit_count++;
    if (type == IS_SPHERE) hit_sphere_gems<TRACE>(P, ray, h, object_id, cur_xf, cnt);
    else if (type == IS_TRIANGLE) hit_triangle_mt<TRACE>(P, ray, h, object_id, cur_xf, cnt);
}

// ---- traversal stack ------------------------------------------------------------
// LDS pointers are kept in the LDS address space, so that reads through them are LDS instructions (a generic
// pointer that may be either LDS or global memory turns into flat loads, which take the vector-memory path)
typedef float PtkV4 __attribute__((ext_vector_type(4)));
typedef const PtkV4 __attribute__((address_space(3))) * PtkLdsBlocks;
struct Stack {
    PtkLdsBlocks top = nullptr;   // LDS copy of node blocks [0, n_top), PTK_TOP_BLOCK_STRIDE bytes apart (traversal kernel only)
    uint32_t n_top = 0;
    int* lds;        // this wave's LDS window, [entry][lane]
    int* spill;      // global column of this lane, stride = spill_stride ints
    size_t spill_stride;
    int lds_cap;
    int lane;
};
// Entries past the LDS window go to the global column.  The host sizes the column from the deepest
// stack the bound BVH can produce; without a column (null) deeper entries are dropped rather than
// written through a null pointer.
PTK_DEV void st_write(const Stack& s, int i, int v) {
    if (i < s.lds_cap) s.lds[i * 64 + s.lane] = v;
    else if (s.spill) s.spill[(size_t)(i - s.lds_cap) * s.spill_stride] = v;
}
PTK_DEV int st_read(const Stack& s, int i) {
    if (i < s.lds_cap) return s.lds[i * 64 + s.lane];
    return s.spill ? s.spill[(size_t)(i - s.lds_cap) * s.spill_stride] : 0;
}

// ---- work claims of a workgroup's waves ---------------------------------------------------------
// Queue entries (or pixel slots) are handed out in two steps: a workgroup takes `wg_chunk` entries (64 for each
// of its waves: the last workgroup to get any then has one piece per wave, as with claims by the wave) from the
// device-wide counter with one global atomic, its waves take 64 at a time from that with an LDS atomic.
// Device-wide atomics on one address complete one after the other, about 12 ns each on this part: at 64 entries
// per atomic that alone holds a launch to ~5 G entries/s, and every wave's last, futile claim adds another.
// `range` is one 64-bit LDS word, low half = next entry to hand out, high half = end of the workgroup's
// chunk, zero at the start (next == end).  Every add returns a different `next`, so exactly one wave sees
// next == end: that one fetches the workgroup's next chunk, keeps its first 64 entries and publishes the rest
// with one 64-bit store; waves that come past the end meanwhile try again at their next refill round.  The
// wave that finds the device-wide counter past `count` leaves that as a mark in the word.
enum : int { CLAIM_OK = 0, CLAIM_LATER = 1, CLAIM_DRY = 2 };
// `grab` (at most 64, a divisor of wg_chunk) is what a wave takes at a time.
// Parts: with n_parts > 1 the queue is [0, count) in n_parts stretches of part_len entries (a multiple of wg_chunk),
// head[p] counts what has been taken from stretch p, and *cur_part (LDS, one word per workgroup, set to the workgroup's
// own part at the start) is the stretch the workgroup draws from; a stretch that is used up sends it to the next.
PTK_DEV int wg_claim(unsigned long long* range, uint32_t* head, uint32_t wg_chunk, uint32_t count, int lane, uint32_t& c_next, uint32_t& c_end, uint32_t grab = 64u,
                     uint32_t n_parts = 1u, uint32_t part_len = 0u, volatile int __attribute__((address_space(3))) * cur_part = nullptr) {
    uint32_t nx = 0, en = 0;
    if (lane == 0) {
        const unsigned long long old = __hip_atomic_fetch_add(range, (unsigned long long)grab, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        nx = (uint32_t)old; en = (uint32_t)(old >> 32);
    }
    nx = __builtin_amdgcn_readfirstlane(nx);
    en = __builtin_amdgcn_readfirstlane(en);
    if (nx == en) {
        uint32_t base = count;  // (no counter: the workgroup's one chunk was set at the start, and this is its end)
        if (head && n_parts > 1u) {
            if (lane == 0) {
                uint32_t p = (uint32_t)*cur_part;
                for (uint32_t tries = 0; tries < n_parts; tries++) {
                    const uint32_t off = atomicAdd(head + p, wg_chunk);
                    if (off < part_len && p * part_len + off < count) { base = p * part_len + off; break; }
                    p = p + 1u < n_parts ? p + 1u : 0u;
                }
                *cur_part = (int)p;
            }
            base = __builtin_amdgcn_readfirstlane(base);
        } else if (head) {
            if (lane == 0) base = atomicAdd(head, wg_chunk);
            base = __builtin_amdgcn_readfirstlane(base);
        }
        if (base >= count) {  // leave word of it for the others: a `next` no queue reaches, past an end of zero
            if (lane == 0) __hip_atomic_store(range, 0x80000000ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            return CLAIM_DRY;
        }
        if (lane == 0) __hip_atomic_store(range, ((unsigned long long)(base + wg_chunk) << 32) | (unsigned long long)(base + grab),
                                          __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        nx = base;
    } else if (nx > en) {
        return nx >= 0x80000000u ? CLAIM_DRY : CLAIM_LATER;
    }
    if (nx >= count) return CLAIM_DRY;
    c_next = nx;
    c_end = nx + grab < count ? nx + grab : count;
    return CLAIM_OK;
}

// Closest hit over planes + BVH, :1200-1320 with :1115-1198 folded into one loop.
template <bool TRACE, bool EARLY_T>
PTK_DEV Hit check_ray_hit_bvh(const PtkParams& P, Lane& L, Ray& ray, const Stack& S, const Stack& ST, int* cnt) {
    Hit h;
    h.hit = 0; h.object_id = 0; h.transform_index = -1; h.u = 0.0f; h.v = 0.0f;

    // planes are tested first, outside the BVH (:1204).  Records whose normal is
    // exactly zero always yield NaN or Inf and are rejected at :859, so only the
    // others are listed in live_planes (ascending index).
    hit_live_planes(P, ray, h);

    f3 inv = mk3(1.0f / ray.d.x, 1.0f / ray.d.y, 1.0f / ray.d.z);
    int neg = (inv.x < 0.0f ? 1 : 0) | (inv.y < 0.0f ? 2 : 0) | (inv.z < 0.0f ? 4 : 0);

    float4 rmin = P.nodes[0], rmax = P.nodes[1];
    L.node_hit_count++;
    float root_hit = hit_aabb(rmin, rmax, ray.o, inv, neg, ray.start, ray.end);
    int sp = 0;
    if (in_range(root_hit, ray.start, ray.end)) {
        int xf0 = (int)fbits(rmax.w);
        st_write(S, 0, xf0 >= 0 ? (int)(SOCKET_MARK | 0u) : (int)fbits(rmin.w));
        if (EARLY_T) st_write(ST, 0, (int)fbits(root_hit));
        sp = 1;
    }

    // state of an instanced (mesh-socket) sub-traversal, :1286-1313
    bool in_mesh = false;
    int sub_base = 0;
    int cur_xf = -1;
    Ray wray = ray;        // saved world ray while in_mesh
    f3 winv = inv;
    int wneg = neg;
    float dir_len = 1.0f;

    for (;;) {
        if (in_mesh && sp <= sub_base) {
            // check_ray_hit_BVH_mesh returned: map the local end back (:1309-1310)
            float local_end = ray.end;
            ray = wray; inv = winv; neg = wneg;
            ray.end = local_end / dir_len;
            in_mesh = false;
            cur_xf = -1;
        }
        if (sp <= 0) break;
        --sp;
        int entry = st_read(S, sp);
        if (EARLY_T) {
            float te = bitsf((uint32_t)st_read(ST, sp));
            if (!in_range(te, ray.start, ray.end)) {
                if (in_mesh) continue;  // calculate_top_node returns (:1128-1132)
                break;                  // world loop breaks (:1280-1284)
            }
        }
        int node_index, node_size;
        if (((uint32_t)entry & 0xFF000000u) == SOCKET_MARK) {
            int self = entry & 0x00FFFFFF;
            float4 smin = P.nodes[(size_t)self * 2], smax = P.nodes[(size_t)self * 2 + 1];
            int node_id = (int)fbits(smin.w);
            node_index = bfe_i(node_id, 0, 24);
            node_size = iabs(bfe_i(node_id, 24, 8));
            if (!in_mesh) {
                int xf = (int)fbits(smax.w);
                m44 M = load_xform(P, xf);
                if (TRACE) cnt[PTK_CNT_SOCKETS]++;
                wray = ray; winv = inv; wneg = neg;
                f3 unnorm = mul_mv(M, ray.d, 0.0f);
                dir_len = length3(unnorm);
                Ray nr;
                nr.d = div3(unnorm, dir_len);
                nr.o = mul_mv(M, ray.o, 1.0f);
                nr.end = ray.end * dir_len;
                nr.start = ray.start * dir_len;
                float4 cmin = P.nodes[(size_t)node_index * 2];
                if (EARLY_T) {
                    float4 cmax = P.nodes[(size_t)node_index * 2 + 1];
                    L.node_hit_count++;
                    float aabb_t = hit_aabb(cmin, cmax, ray.o, inv, neg, ray.start, ray.end);
                    st_write(ST, sp, (int)fbits(aabb_t));
                }
                sub_base = sp;
                st_write(S, sp, (int)fbits(cmin.w));
                sp++;
                cur_xf = xf;
                ray = nr;
                inv = mk3(1.0f / ray.d.x, 1.0f / ray.d.y, 1.0f / ray.d.z);
                neg = (inv.x < 0.0f ? 1 : 0) | (inv.y < 0.0f ? 2 : 0) | (inv.z < 0.0f ? 4 : 0);
                in_mesh = true;
                continue;
            }
            // inside a mesh sub-traversal sockets are ordinary nodes (:1115-1181)
        } else {
            node_index = bfe_i(entry, 0, 24);
            node_size = iabs(bfe_i(entry, 24, 8));
        }

        if (node_index <= 0) {
            // leaf: test its objects in order (:1137-1142)
            int base = iabs(node_index) - 1;
            for (int k = 0; k < node_size; k++) {
                hit_object<TRACE>(P, L, ray, h, P.object_ids[base + k], cur_xf, cnt);
            }
            continue;
        }

        // inner: slab-test the children, then push them nearest-on-top (:1144-1180)
        float ts[PTK_MAX_WIDTH];
        int es[PTK_MAX_WIDTH];
        int added = 0;
#pragma unroll
        for (int l = 0; l < PTK_MAX_WIDTH; l++) {
            ts[l] = PT_INF;
            es[l] = 0;
            if (l < node_size) {
                float4 cmin = P.nodes[(size_t)(node_index + l) * 2];
                float4 cmax = P.nodes[(size_t)(node_index + l) * 2 + 1];
                L.node_hit_count++;
                float t = hit_aabb(cmin, cmax, ray.o, inv, neg, ray.start, ray.end);
                if (in_range(t, ray.start, ray.end)) {
                    ts[l] = t;
                    int cxf = (int)fbits(cmax.w);
                    es[l] = (!in_mesh && cxf >= 0) ? (int)(SOCKET_MARK | (uint32_t)(node_index + l)) : (int)fbits(cmin.w);
                    added++;
                }
            }
        }
        // rank = number of accepted children that the shader's selection loop
        // would pick earlier: smaller t, or equal t in a lower slot.
#pragma unroll
        for (int l = 0; l < PTK_MAX_WIDTH; l++) {
            if (ts[l] < PT_INF) {
                int rank = 0;
#pragma unroll
                for (int j = 0; j < PTK_MAX_WIDTH; j++) {
                    if (j < l) rank += (ts[j] <= ts[l]) ? 1 : 0;
                    else if (j > l) rank += (ts[j] < ts[l]) ? 1 : 0;
                }
                int pos = sp + added - 1 - rank;
                st_write(S, pos, es[l]);
                if (EARLY_T) st_write(ST, pos, (int)fbits(ts[l]));
            }
        }
        sp += added;
    }
    return h;
}

// ---- phase-coherent traversal ------------------------------------------------------------------
// Same per-lane order of node pops, box tests and object tests as the loop above, but the wave only
// runs one kind of step at a time: in every round the lanes whose next node is an inner node (or a
// mesh socket) and the lanes whose next node is a leaf are counted by ballot, and one group
// advances while the other waits (the inner-node lanes unless the leaf lanes outnumber them by half).  A lane's own sequence is untouched, so results are identical; the
// wave no longer pays the inner-node code and the leaf code in every iteration.  The next node to
// visit stays in a register (`cur`): the nearest accepted child is not pushed and popped again.
#ifndef PTK_VOTE_INNER
#ifndef PTK_VOTE_NUM
#define PTK_VOTE_NUM 3
#define PTK_VOTE_DEN 2
#endif
#define PTK_VOTE_INNER(ni, nl) (PTK_VOTE_DEN * (ni) >= PTK_VOTE_NUM * (nl))
#endif
constexpr uint32_t ENTRY_NONE = 0x7E000000u;  // "inner node with 126 children": never a real entry

// slab test without early returns; same value as hit_aabb (a failed axis makes the result -1).
PTK_DEV float hit_aabb_bl(float4 bmin, float4 bmax, f3 o, f3 inv, int neg_mask, float start, float end) {
    float t0, t1;
    bool fail = false;
#define PTK_SLAB2(MINV, MAXV, O, INV, BIT)                                         \
    t0 = ((MINV) - (O)) * (INV);                                                   \
    t1 = ((MAXV) - (O)) * (INV);                                                   \
    {                                                                              \
        float a_ = t0 + t1, b_ = a_ - t1, c_ = a_ - b_;                            \
        bool n_ = (neg_mask & (BIT)) != 0;                                         \
        t0 = n_ ? c_ : t0;                                                         \
        t1 = n_ ? b_ : t1;                                                         \
    }                                                                              \
    start = (t0 > start) ? t0 : start;                                             \
    end = (t1 < end) ? t1 : end;                                                   \
    fail = fail || (end <= start);
    PTK_SLAB2(bmin.x, bmax.x, o.x, inv.x, 1)
    PTK_SLAB2(bmin.y, bmax.y, o.y, inv.y, 2)
    PTK_SLAB2(bmin.z, bmax.z, o.z, inv.z, 4)
#undef PTK_SLAB2
    return fail ? -1.0f : start;
}

template <bool TRACE>
PTK_DEV Hit check_ray_hit_bvh2(const PtkParams& P, Lane& L, Ray& ray, const Stack& S, int* cnt) {
    Hit h;
    h.hit = 0; h.object_id = 0; h.transform_index = -1; h.u = 0.0f; h.v = 0.0f;
    hit_live_planes(P, ray, h);

    f3 inv = mk3(1.0f / ray.d.x, 1.0f / ray.d.y, 1.0f / ray.d.z);
    int neg = (inv.x < 0.0f ? 1 : 0) | (inv.y < 0.0f ? 2 : 0) | (inv.z < 0.0f ? 4 : 0);

    uint32_t cur = ENTRY_NONE;
    int sp = 0;
    {
        float4 rmin = P.nodes[0], rmax = P.nodes[1];
        L.node_hit_count++;
        float root_hit = hit_aabb_bl(rmin, rmax, ray.o, inv, neg, ray.start, ray.end);
        if (in_range(root_hit, ray.start, ray.end)) cur = ((int)fbits(rmax.w) >= 0) ? SOCKET_MARK : fbits(rmin.w);
    }
    bool in_mesh = false;
    int sub_base = 0;
    int cur_xf = -1;
    f3 w_o = ray.o, w_d = ray.d;  // world ray while inside a mesh
    float w_start = ray.start;
    float dir_len = 1.0f;

    for (;;) {
        const bool has = cur != ENTRY_NONE;
        const bool is_leaf = has && ((int)cur <= 0);
        const bool is_inner = has && !is_leaf;
        const unsigned long long mi = __ballot(is_inner), ml = __ballot(is_leaf);
        if ((mi | ml) == 0ull) break;
        bool refill = false;
#ifdef PTK_ROUND_STATS
        if (TRACE) { cnt[PTK_CNT_HIT_SPHERE]++; if (PTK_VOTE_INNER(__popcll(mi), __popcll(ml))) { if (is_inner) cnt[PTK_CNT_SPHERE_TESTS]++; } else { cnt[PTK_CNT_SOCKETS]++; if (is_leaf) cnt[PTK_CNT_HIT_PLANE]++; } }
#endif
        if (PTK_VOTE_INNER(__popcll(mi), __popcll(ml))) {
            if (is_inner) {
                if ((cur & 0xFF000000u) == SOCKET_MARK) {
                    // world level: enter the mesh (:1286-1313); its root is the next node
                    int self = (int)(cur & 0x00FFFFFFu);
                    float4 smin = P.nodes[(size_t)self * 2], smax = P.nodes[(size_t)self * 2 + 1];
                    int node_index = bfe_i((int)fbits(smin.w), 0, 24);
                    int xf = (int)fbits(smax.w);
                    m44 M = load_xform(P, xf);
                    if (TRACE) cnt[PTK_CNT_SOCKETS]++;
                    w_o = ray.o; w_d = ray.d; w_start = ray.start;
                    f3 unnorm = mul_mv(M, ray.d, 0.0f);
                    dir_len = length3(unnorm);
                    f3 no = mul_mv(M, ray.o, 1.0f);
                    ray.d = div3(unnorm, dir_len);
                    ray.o = no;
                    ray.end = ray.end * dir_len;
                    ray.start = ray.start * dir_len;
                    inv = mk3(1.0f / ray.d.x, 1.0f / ray.d.y, 1.0f / ray.d.z);
                    neg = (inv.x < 0.0f ? 1 : 0) | (inv.y < 0.0f ? 2 : 0) | (inv.z < 0.0f ? 4 : 0);
                    sub_base = sp;
                    cur_xf = xf;
                    in_mesh = true;
                    cur = fbits(P.nodes[(size_t)node_index * 2].w);
                } else {
                    const int node_index = bfe_i((int)cur, 0, 24);
                    const int node_size = iabs(bfe_i((int)cur, 24, 8));
                    float ts[PTK_MAX_WIDTH];
                    uint32_t es[PTK_MAX_WIDTH];
                    int added = 0;
#pragma unroll
                    for (int l = 0; l < PTK_MAX_WIDTH; l++) {
                        const bool v = l < node_size;
                        const size_t ci = (size_t)(node_index + (v ? l : 0)) * 2;
                        const float4 cmin = P.nodes[ci];
                        const float4 cmax = P.nodes[ci + 1];
                        const float t = hit_aabb_bl(cmin, cmax, ray.o, inv, neg, ray.start, ray.end);
                        const bool acc = v && in_range(t, ray.start, ray.end);
                        ts[l] = acc ? t : PT_INF;
                        const bool sock = !in_mesh && ((int)fbits(cmax.w) >= 0);
                        es[l] = sock ? (SOCKET_MARK | (uint32_t)(node_index + l)) : fbits(cmin.w);
                        added += acc ? 1 : 0;
                        L.node_hit_count += v ? 1 : 0;
                    }
                    // rank r goes to stack slot sp + added - 1 - r; rank 0 (nearest) stays in `cur`
                    cur = ENTRY_NONE;
#pragma unroll
                    for (int l = 0; l < PTK_MAX_WIDTH; l++) {
                        int rank = 0;
#pragma unroll
                        for (int j = 0; j < PTK_MAX_WIDTH; j++) {
                            if (j < l) rank += (ts[j] <= ts[l]) ? 1 : 0;
                            else if (j > l) rank += (ts[j] < ts[l]) ? 1 : 0;
                        }
                        if (ts[l] < PT_INF) {
                            if (rank == 0) cur = es[l];
                            else st_write(S, sp + added - 1 - rank, (int)es[l]);
                        }
                    }
                    sp += (added > 0) ? added - 1 : 0;
                    refill = added == 0;
                }
            }
        } else {
            if (is_leaf) {
                const int node_index = bfe_i((int)cur, 0, 24);
                const int node_size = iabs(bfe_i((int)cur, 24, 8));
                const int base = iabs(node_index) - 1;
                for (int k = 0; k < node_size; k++) {
                    hit_object<TRACE>(P, L, ray, h, P.object_ids[base + k], cur_xf, cnt);
                }
                refill = true;
            }
        }
        if (refill) {
            // next node of this lane: pop; an empty mesh sub-stack first maps the ray back (:1309-1310)
            cur = ENTRY_NONE;
            if (in_mesh && sp <= sub_base) {
                float local_end = ray.end;
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
    return h;
}

// ---- traversal over compiled node blocks (PtkWavefront::blocks, ::sockets, ::leaf_records) ---------
// The same search as check_ray_hit_bvh2 -- per lane the same order of box and object tests -- reading
// the structures the host derives from the bound buffers: an inner node's eight child slots come as one
// 256-byte block of planes, a leaf's primitives by value from consecutive records.  The state of one
// ray's search between steps is a Trav; the callers (the wavefront traversal kernel and the one-kernel
// renderer) own the loop that votes between inner and leaf rounds.
typedef float v2f __attribute__((ext_vector_type(2)));

struct Trav {
    f3 o, d;        // the ray; mesh-local while in a mesh
    float start, end;
    f3 inv;         // 1 / d
    Hit h;
    uint32_t cur;   // next node: entry word or ENTRY_NONE
    int sp;
    int sub_base;   // stack level at which the mesh in progress was entered
    int cur_xf;     // its transform index, -1 at world level
    f3 w_o, w_d;    // the world ray while in a mesh
    float w_start, dir_len;
};

PTK_DEV bool trav_is_leaf(const Trav& T) { return T.cur != ENTRY_NONE && (int)T.cur <= 0; }
PTK_DEV bool trav_is_inner(const Trav& T) { return T.cur != ENTRY_NONE && (int)T.cur > 0; }

// Entry into a mesh socket at world level (:1286-1313): the ray is taken into the mesh's space.
PTK_DEV void trav_enter_mesh_m(Trav& T, const m44& M, int xf) {
    T.w_o = T.o; T.w_d = T.d; T.w_start = T.start;
    const f3 unnorm = mul_mv(M, T.d, 0.0f);
    T.dir_len = length3(unnorm);
    const f3 no = mul_mv(M, T.o, 1.0f);
    T.d = div3(unnorm, T.dir_len);
    T.o = no;
    T.end = T.end * T.dir_len;
    T.start = T.start * T.dir_len;
    T.inv = mk3(1.0f / T.d.x, 1.0f / T.d.y, 1.0f / T.d.z);
    T.sub_base = T.sp;
    T.cur_xf = xf;
}
PTK_DEV void trav_enter_mesh(const PtkParams& P, Trav& T, int xf) { trav_enter_mesh_m(T, load_xform(P, xf), xf); }

// hit_AABB (:670-689) and the acceptance test of :1152 for the only child of a one-child node: the same
// operations, in the same order, as one slot of the packed test in trav_inner.
PTK_DEV bool trav_single_child_accepts(const Trav& T, const PtkChainStep& c) {
    const f3 o = T.o, inv = T.inv;
    const bool nx = inv.x < 0.0f, ny = inv.y < 0.0f, nz = inv.z < 0.0f;
    const bool range_ok = T.start == T.start && T.end == T.end;
    const float t0x = (c.lo[0] - o.x) * inv.x, t1x = (c.hi[0] - o.x) * inv.x;
    const float t0y = (c.lo[1] - o.y) * inv.y, t1y = (c.hi[1] - o.y) * inv.y;
    const float t0z = (c.lo[2] - o.z) * inv.z, t1z = (c.hi[2] - o.z) * inv.z;
    const float swx = t0x + t1x, sbx = swx - t1x, scx = swx - sbx;
    const float swy = t0y + t1y, sby = swy - t1y, scy = swy - sby;
    const float swz = t0z + t1z, sbz = swz - t1z, scz = swz - sbz;
    float start = T.start, end = T.end;
    start = __builtin_fmaxf(start, nx ? scx : t0x);
    end = __builtin_fminf(end, nx ? sbx : t1x);
    start = __builtin_fmaxf(start, ny ? scy : t0y);
    end = __builtin_fminf(end, ny ? sby : t1y);
    start = __builtin_fmaxf(start, nz ? scz : t0z);
    end = __builtin_fminf(end, nz ? sbz : t1z);
    return (end > start) && range_ok;
}

// planes first, outside the BVH (:1204; only records with a non-zero normal can hit), then the root box,
// then the chain of one-child nodes and mesh sockets at the top of the tree (PtkWavefront::chain): the
// same tests a search would make there node by node, from kernel arguments instead of fetched blocks.
PTK_DEV void trav_begin(const PtkParams& P, Trav& T, const Ray& world_ray) {
    const PtkWavefront& W = P.wf;
    Ray r = world_ray;
    T.h.hit = 0; T.h.object_id = 0; T.h.transform_index = -1; T.h.u = 0.0f; T.h.v = 0.0f;
    hit_live_planes(P, r, T.h);
    T.o = r.o; T.d = r.d; T.start = r.start; T.end = r.end;
    T.inv = mk3(1.0f / T.d.x, 1.0f / T.d.y, 1.0f / T.d.z);
    const int neg = (T.inv.x < 0.0f ? 1 : 0) | (T.inv.y < 0.0f ? 2 : 0) | (T.inv.z < 0.0f ? 4 : 0);
    const float4 rmin = make_float4(W.root_min_x, W.root_min_y, W.root_min_z, 0.0f);
    const float4 rmax = make_float4(W.root_max_x, W.root_max_y, W.root_max_z, 0.0f);
    const float root_hit = hit_aabb_bl(rmin, rmax, T.o, T.inv, neg, T.start, T.end);
    bool alive = in_range(root_hit, T.start, T.end);
    T.sp = 0;
    T.sub_base = 0;
    T.cur_xf = -1;
    T.dir_len = 1.0f;
    for (int s = 0; s < W.n_chain; s++) {  // (uniform over the wave: the steps are kernel arguments)
        const PtkChainStep& c = W.chain[s];
        if (c.kind == 1) {
            if (alive) {
                m44 M;
                M.c0 = make_float4(c.m[0], c.m[1], c.m[2], c.m[3]); M.c1 = make_float4(c.m[4], c.m[5], c.m[6], c.m[7]);
                M.c2 = make_float4(c.m[8], c.m[9], c.m[10], c.m[11]); M.c3 = make_float4(c.m[12], c.m[13], c.m[14], c.m[15]);
                trav_enter_mesh_m(T, M, c.xf);
            }
        }
        else if (alive) alive = trav_single_child_accepts(T, c);
    }
    T.cur = alive ? W.root_entry : ENTRY_NONE;
    if (!alive && T.cur_xf >= 0) {
        // rejected inside the mesh: its empty sub-stack maps the ray back to world space (:1309-1310)
        const float local_end = T.end;
        T.o = T.w_o; T.d = T.w_d; T.start = T.w_start;
        T.end = local_end / T.dir_len;
        T.inv = mk3(1.0f / T.d.x, 1.0f / T.d.y, 1.0f / T.d.z);
        T.cur_xf = -1;
    }
}

// One inner step of a lane whose T.cur is an inner entry: enter a mesh socket, or slab-test the node's
// children and push the accepted ones nearest-on-top (:1144-1180).  Returns true if the lane now needs its
// next node from the stack (no child accepted).  The stack's LDS window must have one spare row past
// S.lds_cap (rejected children are written there).
PTK_DEV bool trav_inner(const PtkParams& P, const Stack& S, Trav& T) {
    const PtkWavefront& W = P.wf;
    uint32_t cur = T.cur;
    if ((cur & 0xFF000000u) == SOCKET_MARK) {
        // world level: enter the mesh (:1286-1313); its root is the next node
        const int2 sock = W.sockets[cur & 0x00FFFFFFu];
        trav_enter_mesh(P, T, sock.y);
        T.cur = (uint32_t)sock.x;
        return false;
    }
    int* const lds_lane = S.lds + S.lane;
    const int sp = T.sp;
    const int node_size = (int)(cur >> 24);
    // the node's block: eight child slots as planes (minx[8] .. maxz[8], entry[8]), all fetched before
    // any is used; slots past node_size hold zeros and are ignored
    const uint32_t blk = cur & 0x00FFFFFFu;
    float4 q[14];
    if (blk < S.n_top) {
        // the top of the tree is resident in LDS: no trip through the vector cache for these
        PtkLdsBlocks lb = S.top + blk * (PTK_TOP_BLOCK_STRIDE / 16);
#pragma unroll
        for (int k = 0; k < 14; k++) { const PtkV4 v = lb[k]; q[k] = make_float4(v.x, v.y, v.z, v.w); }
    } else {
        const float4* __restrict__ cb = W.blocks + (size_t)blk * 16;
#pragma unroll
        for (int k = 0; k < 14; k++) q[k] = cb[k];
    }
    __builtin_amdgcn_sched_barrier(0);
    const f3 o = T.o, inv = T.inv;
    const float rstart = T.start, rend = T.end;
    const bool nx = inv.x < 0.0f, ny = inv.y < 0.0f, nz = inv.z < 0.0f;
    // a NaN in the ray's own range (singular mesh transform) fails in_range for every box
    const bool range_ok = rstart == rstart && rend == rend;
    // key = bits of the entry distance t (t >= +0, so unsigned integer order is float order); rejected
    // children get the bits of +inf, which no accepted t reaches
    uint32_t key[PTK_MAX_WIDTH];
    uint32_t ent[PTK_MAX_WIDTH];
    int added = 0;
#pragma unroll
    for (int c = 0; c < PTK_MAX_WIDTH / 2; c++) {
        // hit_AABB (:670-689) for children 2c and 2c+1 side by side in packed fp32: per axis
        // t0 = (min - o) * inv, t1 = (max - o) * inv, the arithmetic swap when inv < 0, then start = max,
        // end = min.  start and end are never NaN and only move inwards, so testing end <= start once
        // after the last axis equals the shader's three early returns, and a box that passes has its t
        // inside [ray.start, ray.end] already.
        const int h4 = c >> 1;          // which float4 of the plane
        const bool hi2 = (c & 1) != 0;  // .zw or .xy
#define PTK_PAIR(v) (hi2 ? v2f{(v).z, (v).w} : v2f{(v).x, (v).y})
        const v2f lox = PTK_PAIR(q[0 + h4]), loy = PTK_PAIR(q[2 + h4]), loz = PTK_PAIR(q[4 + h4]);
        const v2f hix = PTK_PAIR(q[6 + h4]), hiy = PTK_PAIR(q[8 + h4]), hiz = PTK_PAIR(q[10 + h4]);
        const v2f en = PTK_PAIR(q[12 + h4]);
#undef PTK_PAIR
        const v2f t0x = (lox - o.x) * inv.x, t1x = (hix - o.x) * inv.x;
        const v2f t0y = (loy - o.y) * inv.y, t1y = (hiy - o.y) * inv.y;
        const v2f t0z = (loz - o.z) * inv.z, t1z = (hiz - o.z) * inv.z;
        const v2f swx = t0x + t1x, sbx = swx - t1x, scx = swx - sbx;
        const v2f swy = t0y + t1y, sby = swy - t1y, scy = swy - sby;
        const v2f swz = t0z + t1z, sbz = swz - t1z, scz = swz - sbz;
#pragma unroll
        for (int e = 0; e < 2; e++) {
            const int l = 2 * c + e;
            float start = rstart, end = rend;
            start = __builtin_fmaxf(start, nx ? scx[e] : t0x[e]);
            end = __builtin_fminf(end, nx ? sbx[e] : t1x[e]);
            start = __builtin_fmaxf(start, ny ? scy[e] : t0y[e]);
            end = __builtin_fminf(end, ny ? sby[e] : t1y[e]);
            start = __builtin_fmaxf(start, nz ? scz[e] : t0z[e]);
            end = __builtin_fminf(end, nz ? sbz[e] : t1z[e]);
            const bool acc = (l < node_size) && (end > start) && range_ok;
#ifdef PTK_INNER_V1
            key[l] = acc ? (fbits(start) & 0x7FFFFFFFu) : 0x7F800000u;
#else
            // key of an accepted child: the bits of its entry distance t (t >= 0, so unsigned order is float order) with
            // the top bit set -- which also makes -0 and +0 one key; a rejected child's key is 0, below every accepted one
            key[l] = acc ? (fbits(start) | 0x80000000u) : 0u;
#endif
            ent[l] = fbits(en[e]);
            added += acc ? 1 : 0;
        }
    }
#ifdef PTK_INNER_V1
    // rank = number of children the shader's selection loop (:1157-1178) would take earlier: smaller t, or
    // equal t in a lower slot.  For j < l, u = 1 iff child l is strictly nearer than child j (sign bit of
    // key[l] - key[j]; both keys < 2^31): rank[l] = l - sum_{j<l} u(j,l) + sum_{m>l} u(l,m).
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
    // rank r goes to stack slot sp + added - 1 - r, so the nearest child lands in the free slot above the
    // new top and is read back as `cur`; rejected children (rank >= added) are sent to the spare row.
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
#else
    // The shader's selection loop (:1157-1178) takes the accepted children in order of t, equal t in slot order, and
    // pushes them so that the first taken ends on top.  Child l therefore sits `above[l]` entries over the old top,
    // above[l] = how many accepted children are taken after it = #{j < l : key[j] > key[l]} + #{m > l : key[m] >= key[l]}
    //          = sum_{j<l} u(j,l) + (7 - l) - sum_{m>l} u(l,m)        with u(a,b) = [key[a] > key[b]], a < b.
    // The same formula sends the rejected children (key 0) to the rows from `added` upwards, each to its own: past
    // the new top, inside the window when sp + 8 <= lds_cap, where nothing reads them.  So all eight stores go out
    // unconditionally, and the nearest child is read back from row sp + added - 1.
    int above[PTK_MAX_WIDTH];
#pragma unroll
    for (int l = 0; l < PTK_MAX_WIDTH; l++) above[l] = PTK_MAX_WIDTH - 1 - l;
#pragma unroll
    for (int l = 1; l < PTK_MAX_WIDTH; l++) {
#pragma unroll
        for (int j = 0; j < l; j++) {
            const int u = key[j] > key[l] ? 1 : 0;
            above[l] += u;
            above[j] -= u;
        }
    }
    cur = ENTRY_NONE;
    if (sp + PTK_MAX_WIDTH <= S.lds_cap) {
        int* const row0 = lds_lane + sp * 64;
#pragma unroll
        for (int l = 0; l < PTK_MAX_WIDTH; l++) row0[above[l] * 64] = (int)ent[l];
        if (added > 0) cur = (uint32_t)row0[(added - 1) * 64];
    } else {
#pragma unroll
        for (int l = 0; l < PTK_MAX_WIDTH; l++) {
            if (key[l] != 0u) {
                if (above[l] == added - 1) cur = ent[l];
                else st_write(S, sp + above[l], (int)ent[l]);
            }
        }
    }
#endif
    T.cur = cur;
    T.sp = sp + ((added > 0) ? added - 1 : 0);
    return added == 0;
}

// Objects [first, first + count) of a leaf against ray r, in order (:1137-1142), for every lane of the wave with
// its own span; the loop runs to the longest span among them.  Leaf records: {A.xyz | centre.xyz, object id},
// {E1.xyz, N.x | radius, -}, {E2.xyz, N.y | -}, stored in object-id-slot order.
PTK_DEV void leaf_span(const PtkParams& P, int first, int count, int cur_xf, Ray& r, Hit& h, bool* either_root) {
    const float4* __restrict__ rec = P.wf.leaf_records + (size_t)first * 3;
#ifdef PTK_LEAF_NO_PREFETCH
    for (int k = 0;; k++) {
        const bool go = k < count;
        if (__ballot(go) == 0ull) break;
        if (go) {
            const float4 r0 = rec[k * 3], r1 = rec[k * 3 + 1], r2 = rec[k * 3 + 2];
            const int object_id = (int)fbits(r0.w);
            const int type = bfe_i(object_id, 24, 8);
            if (type == IS_TRIANGLE) hit_triangle_pre(r, h, object_id, cur_xf, ld3(r0), ld3(r1), ld3(r2), r1.w, r2.w);
            else if (type == IS_SPHERE) hit_sphere_rec(r, h, object_id, cur_xf, ld3(r0), r1.x, either_root);
        }
    }
#else
    // Four records at a time: their loads go out together, so the round waits for memory once per four objects
    // instead of once per object (a split leaf has at most four).  The tests themselves stay in order.
    constexpr int LEAF_BATCH = 4;
    for (int k0 = 0;; k0 += LEAF_BATCH) {
        if (__ballot(k0 < count) == 0ull) break;
        float4 q[LEAF_BATCH][3];
#pragma unroll
        for (int i = 0; i < LEAF_BATCH; i++) {
            if (k0 + i < count) {
                q[i][0] = rec[(k0 + i) * 3]; q[i][1] = rec[(k0 + i) * 3 + 1]; q[i][2] = rec[(k0 + i) * 3 + 2];
            } else {
                q[i][0] = q[i][1] = q[i][2] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            }
        }
#pragma unroll
        for (int i = 0; i < LEAF_BATCH; i++) {
            const bool go = k0 + i < count;
            if (i > 0 && __ballot(go) == 0ull) break;
            if (go) {
                const int object_id = (int)fbits(q[i][0].w);
                const int type = bfe_i(object_id, 24, 8);
                if (type == IS_TRIANGLE) hit_triangle_pre(r, h, object_id, cur_xf, ld3(q[i][0]), ld3(q[i][1]), ld3(q[i][2]), q[i][1].w, q[i][2].w);
                else if (type == IS_SPHERE) hit_sphere_rec(r, h, object_id, cur_xf, ld3(q[i][0]), q[i][1].x, either_root);
            }
        }
    }
#endif
}

PTK_DEV int wave_rank_in(unsigned long long mask) {  // set bits of mask below this lane
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
PTK_DEV float lane_read_f(int src_lane, float v) { return bitsf((uint32_t)__builtin_amdgcn_ds_bpermute(src_lane << 2, (int)fbits(v))); }
PTK_DEV int lane_read_i(int src_lane, int v) { return __builtin_amdgcn_ds_bpermute(src_lane << 2, v); }

// One leaf round for the whole wave: lanes with is_leaf test their leaf's objects.
//
// The lanes that are not at a leaf help: a leaf of more than PtkWavefront::leaf_split objects (4 where there are
// triangles: measured 2..5; never in scenes of spheres only, whose tests are too short for it) is split in two, the owner
// takes the first half and a helper lane, with a copy of the owner's ray, the second, so the round lasts half as
// many tests (leaves hold up to eight objects, 4.3 on average on the 262 144-triangle scene, and the loop runs
// to the longest).  The result is the shader's: every test accepts a hit iff its distance t lies in
// [ray.start, ray.end] and then sets ray.end = t, so the objects in order leave the smallest t, the last one
// among equals, whatever ray.end the later ones saw on the way; the helper's half starts from the ray.end of the
// leaf's start, and its hit replaces the owner's iff its t <= the owner's final ray.end.  A test's t itself
// depends on ray.end in one corner only -- a sphere whose second root is the smaller one and lies in the range as
// well (hit_sphere_rec's either_root) -- and a helper that meets it hands the whole leaf back to its owner.
// (u and v are a triangle hit's; a sphere hit leaves them as they were, and nothing reads them after one -- textured
// spheres compute their own in determine_rayhit -- so which earlier triangle they stem from is not carried over.)
// The spare row of the wave's stack window is where owners and helpers find each other.
PTK_DEV void trav_leaf(const PtkParams& P, const Stack& S, Trav& T, bool is_leaf) {
    int n = 0, base = 0;
    if (is_leaf) {
        n = iabs(bfe_i((int)T.cur, 24, 8));
        base = iabs(bfe_i((int)T.cur, 0, 24)) - 1;
    }
    Ray r;
    r.o = T.o; r.d = T.d; r.start = T.start; r.end = T.end;
    const bool big = n > P.wf.leaf_split;
    const unsigned long long m_big = __ballot(big), m_free = __ballot(!is_leaf);
    if (m_big == 0ull || m_free == 0ull) {  // nothing to split, or nobody to help
        leaf_span(P, base, n, T.cur_xf, r, T.h, nullptr);
        T.end = r.end;
        return;
    }
    int* const spare = S.lds + S.lds_cap * 64;
    const int n_big = __popcll(m_big), n_free = __popcll(m_free);
    const int n_pairs = n_big < n_free ? n_big : n_free;
    const int r_big = wave_rank_in(m_big), r_free = wave_rank_in(m_free);
    const bool helped = big && r_big < n_pairs, helper = !is_leaf && r_free < n_pairs;
    if (helped) spare[r_big] = S.lane;                    // the r-th owner ...
    __builtin_amdgcn_wave_barrier();
    int partner = S.lane;
    if (helper) partner = spare[r_free];                  // ... and the r-th free lane are a pair
    __builtin_amdgcn_wave_barrier();
    if (helper) spare[partner] = S.lane;                  // (the owner learns its helper's lane)
    // the helper's copy of its owner's ray and leaf (every lane reads: its own values, or its partner's)
    r.o = mk3(lane_read_f(partner, r.o.x), lane_read_f(partner, r.o.y), lane_read_f(partner, r.o.z));
    r.d = mk3(lane_read_f(partner, r.d.x), lane_read_f(partner, r.d.y), lane_read_f(partner, r.d.z));
    r.start = lane_read_f(partner, r.start);
    r.end = lane_read_f(partner, r.end);
    const int pn = lane_read_i(partner, n), pbase = lane_read_i(partner, base);
    const int half = (pn + 1) >> 1;
    int first = base, count = n;
    if (helped) count = half;
    if (helper) { first = pbase + half; count = pn - half; }
    Hit h = T.h;
    if (helper) { h.hit = 0; h.object_id = 0; h.transform_index = -1; h.u = 0.0f; h.v = 0.0f; }
    bool either_root = false;
    leaf_span(P, first, count, T.cur_xf, r, h, helper ? &either_root : nullptr);
    __builtin_amdgcn_wave_barrier();
    int from = S.lane;
    if (helped) from = spare[S.lane];
    // the owner takes its helper's hit if that one is at least as near (it is the later one in the leaf's order)
    const int o_hit = lane_read_i(from, h.hit), o_id = lane_read_i(from, h.object_id);
    const float o_u = lane_read_f(from, h.u), o_v = lane_read_f(from, h.v), o_end = lane_read_f(from, r.end);
    const bool o_redo = lane_read_i(from, either_root ? 1 : 0) != 0;
    if (helped && o_hit && o_end <= r.end) {
        h.hit = 1; h.object_id = o_id; h.transform_index = T.cur_xf; h.u = o_u; h.v = o_v;
        r.end = o_end;
    }
    const bool redo = helped && (o_redo || P.wf.leaf_redo_all != 0);
    if (__ballot(redo) != 0ull) {  // (the sphere corner: these owners run their whole leaf alone)
        Ray r2;
        r2.o = T.o; r2.d = T.d; r2.start = T.start; r2.end = T.end;
        Hit h2 = T.h;
        leaf_span(P, base, redo ? n : 0, T.cur_xf, r2, h2, nullptr);
        if (redo) { h = h2; r.end = r2.end; }
    }
    if (is_leaf) { T.h = h; T.end = r.end; }
}

// Next node of the lane: pop; an empty mesh sub-stack first maps the ray back to world space (:1309-1310).
PTK_DEV void trav_pop(const Stack& S, Trav& T) {
    T.cur = ENTRY_NONE;
    if (T.cur_xf >= 0 && T.sp <= T.sub_base) {
        const float local_end = T.end;
        T.o = T.w_o; T.d = T.w_d; T.start = T.w_start;
        T.end = local_end / T.dir_len;
        T.inv = mk3(1.0f / T.d.x, 1.0f / T.d.y, 1.0f / T.d.z);
        T.cur_xf = -1;
    }
    if (T.sp > 0) T.cur = (uint32_t)st_read(S, --T.sp);
}

// The whole search for the one-kernel renderer: every lane of the wave runs its own ray to the end.
PTK_DEV Hit check_ray_hit_blocks(const PtkParams& P, Ray& ray, const Stack& S) {
    Trav T;
    T.w_o = ray.o; T.w_d = ray.d; T.w_start = ray.start;
    trav_begin(P, T, ray);
    for (;;) {
        const bool is_leaf = trav_is_leaf(T), is_inner = trav_is_inner(T);
        const unsigned long long mi = __ballot(is_inner), ml = __ballot(is_leaf);
        if ((mi | ml) == 0ull) break;
        bool pop = false;
        if (PTK_VOTE_INNER(__popcll(mi), __popcll(ml))) {
            if (is_inner) pop = trav_inner(P, S, T);
        } else {
            trav_leaf(P, S, T, is_leaf);
            pop = is_leaf;
        }
        if (pop) trav_pop(S, T);
    }
    ray.end = T.end;
    return T.h;
}

// Brute-force closest hit when USE_BVH is off, :1093-1101.
template <bool TRACE>
PTK_DEV Hit check_ray_hit_all(const PtkParams& P, Ray& ray, int* cnt) {
    Hit h;
    h.hit = 0; h.object_id = 0; h.transform_index = -1; h.u = 0.0f; h.v = 0.0f;
    for (int i = 0; i < P.n_spheres; ++i) hit_sphere_plain(P, ray, h, i + (IS_SPHERE << 24), -1);
    for (int i = 0; i < P.n_triangles; ++i) hit_triangle_mt<TRACE>(P, ray, h, i + (IS_TRIANGLE << 24), -1, cnt);
    for (int i = 0; i < P.n_planes; ++i) hit_plane(P, ray, h, i, -1);
    return h;
}

// ---- hit resolution (:1003-1091) ----------------------------------------------------
struct Surf {
    f3 point;
    f3 normal;
    float u, v;
    int mtl_index;
    uint32_t texture_id;
    f3 albedo;
    int dielectric_priority;
};

PTK_DEV void sphere_get_uv(f3 pos, float& u, float& v) {  // :620-627
    float theta = pacos(pos.y);
    float phi = patan2(-pos.z, pos.x) + 3.1415926535897932385f;
    u = phi / (3.1415926535897932385f * 2.0f);
    v = theta / 3.1415926535897932385f;
}

template <bool TRACE>
PTK_DEV Surf determine_rayhit(const PtkParams& P, Lane& L, const Ray& ray_in, const Hit& h, int* cnt) {
    Surf s;
    int type = bfe_i(h.object_id, 24, 8);
    int index = bfe_i(h.object_id, 0, 24);
    s.u = h.u; s.v = h.v;
    s.point = mk3(0, 0, 0); s.normal = mk3(0, 0, 0); s.mtl_index = 0; s.texture_id = 0;
    if (type == IS_SPHERE) {
        float4 a = P.spheres[(size_t)index * 2], b = P.spheres[(size_t)index * 2 + 1];
        f3 center = ld3(a);
        s.mtl_index = (int)fbits(b.x);
        s.texture_id = fbits(b.y);
        bool need_uv = (s.texture_id >> 30) != 0u || (s.texture_id & 0x3fffffffu) != 0u;
        if (h.transform_index >= 0) {
            m44 m = load_xform(P, h.transform_index);
            m44 matrix = load_inv_xform(P, h.transform_index);
            f3 td = mul_mv(m, ray_in.d, 0.0f);
            f3 to = mul_mv(m, ray_in.o, 1.0f);
            f3 usn = normalize3((to + td * ray_in.end) - center);
            if (need_uv) sphere_get_uv(usn, s.u, s.v);
            s.point = center + usn * a.w;
            s.point = mul_mv(matrix, s.point, 1.0f);
            s.normal = normalize3(mul_vm(usn, 0.0f, m));
        } else {
            s.normal = normalize3((ray_in.o + ray_in.d * ray_in.end) - center);
            if (need_uv) sphere_get_uv(s.normal, s.u, s.v);
            s.point = center + s.normal * a.w;
        }
        if (TRACE) cnt[PTK_CNT_HIT_SPHERE]++;
    } else if (type == IS_PLANE) {
        float4 a = P.planes[(size_t)index * 2], b = P.planes[(size_t)index * 2 + 1];
        s.point = ray_in.o + ray_in.d * ray_in.end;
        s.mtl_index = (int)fbits(b.x);
        s.texture_id = fbits(b.y);
        s.normal = ld3(a);
        if (TRACE) cnt[PTK_CNT_HIT_PLANE]++;
    } else if (type == IS_TRIANGLE) {
        const float4* t = P.triangles + (size_t)index * 5;
        f3 p1 = ld3(t[0]), p2 = ld3(t[1]), p3 = ld3(t[2]);
        float4 pr = t[4];
        s.mtl_index = (int)fbits(pr.x);
        s.texture_id = fbits(pr.y);
        s.normal = normalize3(cross3(p1 - p3, p1 - p2));
        float w = (1.0f - h.u) - h.v;
        s.point = (p2 * h.u + p3 * h.v) + p1 * w;
        if (h.transform_index >= 0) {
            m44 m = load_xform(P, h.transform_index);
            m44 matrix = load_inv_xform(P, h.transform_index);
            s.normal = normalize3(mul_vm(s.normal, 0.0f, m));
            s.point = mul_mv(matrix, s.point, 1.0f);
        }
        if (P.triangle_uvs != nullptr && index >= 0 && index < P.n_triangle_uvs) {
            // real texture coordinates (the shader's TODO at :1071-1072): corner UVs weighted like the point above
            const float* q = P.triangle_uvs + (size_t)index * 6;
            s.u = (q[2] * h.u + q[4] * h.v) + q[0] * w;
            s.v = (q[3] * h.u + q[5] * h.v) + q[1] * w;
        }
        if (dot3(s.normal, ray_in.d) > 0.0f) s.normal = s.normal * -1.0f;
        if (TRACE) cnt[PTK_CNT_HIT_TRIANGLE]++;
    }
    const float4* mt = P.materials + (size_t)s.mtl_index * 4;
    float4 m0 = mt[0];
    float4 m2 = mt[2];
    s.albedo = ld3(m0);
    s.dielectric_priority = (int)fbits(m2.x);
    if (P.use_emission) L.is_emissive = ((int)fbits(m2.y)) != 0;
    if (TRACE) cnt[PTK_CNT_MATERIAL_FETCH]++;
    return s;
}

// ---- textures (:632-659) ------------------------------------------------------------
PTK_DEV float4 checker(float sum) {
    return (pmod(sum, 2.0f) != 0.0f) ? make_float4(0.9f, 0.9f, 0.9f, 1.0f) : make_float4(0.2f, 0.3f, 0.1f, 1.0f);
}
PTK_DEV float4 get_texture(const PtkParams& P, const Surf& s) {
    uint32_t type = s.texture_id >> 30;
    uint32_t index = s.texture_id & 0x3fffffffu;
    if (type & TEX_IS_SAMPLED) {
        // texture() with Godot's default sampler: nearest, clamp to edge, level 0
        const float* rgba = nullptr;
        int w = 8, hgt = 8;
        if (index < (uint32_t)P.n_textures) { rgba = P.textures[index].rgba; w = P.textures[index].width; hgt = P.textures[index].height; }
        int i = pf2i(floorf(s.u * (float)w)), j = pf2i(floorf(s.v * (float)hgt));
        i = i < 0 ? 0 : (i > w - 1 ? w - 1 : i);
        j = j < 0 ? 0 : (j > hgt - 1 ? hgt - 1 : j);
        if (rgba == nullptr) {  // 8x8 pink/black filler grid (pt_utils.gd:122-137)
            bool pink = (((j % 2) + (i % 2)) % 2) == 0;
            return pink ? make_float4(56.0f / 255.0f, 16.0f / 255.0f, 56.0f / 255.0f, 56.0f / 255.0f)
                        : make_float4(0.0f, 0.0f, 0.0f, 56.0f / 255.0f);
        }
        const float* p = rgba + ((size_t)j * w + i) * 4;
        return make_float4(p[0], p[1], p[2], p[3]);
    }
    if (index == 0) return make_float4(1.0f, 1.0f, 1.0f, 1.0f);
    f3 pos = (type & TEX_IS_SPATIAL) ? s.point : mk3(s.u, s.v, 0.0f);
    if (index <= (uint32_t)P.n_proc) {
        int b = P.proc_builtin[index - 1];
        if (b == PTK_PROC_CHECKER_BOARD) {  // procedural_textures/checker_board.comp
            return checker(floorf(pos.x * 2.0f) + floorf(pos.z * 2.0f));
        }
        if (b == PTK_PROC_CHECKER_BOARD_SPATIAL) {  // procedural_textures/checker_board_spatial.comp
            return checker((floorf(pos.x * 2.0f) + floorf(pos.y * 2.0f)) + floorf(pos.z * 2.0f));
        }
    }
    return make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}

// ---- nested dielectrics (:1369-1419) ----------------------------------------------------
PTK_DEV void add_material_to_inside_of(Lane& L, int material_index, int priority) {
    int shift = 1, insert_index = 0;
#pragma unroll
    for (int i = MAX_OVERLAP - 1; i > 0; i--) {
        if (priority < L.in_pri[i - 1] && shift == 1) { insert_index = i; shift = 0; }
        int src = i - shift;
        int m = L.in_mat[i], p = L.in_pri[i];
        if (src == i - 1) { m = L.in_mat[i - 1]; p = L.in_pri[i - 1]; }
        L.in_mat[i] = m; L.in_pri[i] = p;
    }
#pragma unroll
    for (int i = 0; i < MAX_OVERLAP; i++) {
        if (i == insert_index) { L.in_mat[i] = material_index; L.in_pri[i] = priority; }
    }
    L.inside_of_count++;
}
PTK_DEV void remove_material_from_inside_of(Lane& L, int material_index) {
    int shift = 0;
#pragma unroll
    for (int i = 0; i < MAX_OVERLAP - 1; i++) {
        if (shift == 0 && material_index == L.in_mat[i]) shift = 1;
        if (shift) { L.in_mat[i] = L.in_mat[i + 1]; L.in_pri[i] = L.in_pri[i + 1]; }
    }
    L.in_mat[MAX_OVERLAP - 1] = 0;
    L.in_pri[MAX_OVERLAP - 1] = INT_NEG_LIMIT;
    L.inside_of_count--;
}
PTK_DEV int material_chooser(const Lane& L, bool is_inside, float dielectric_priority, int object_material) {
    int mx = L.in_mat[0], my = L.in_mat[1];
    bool same_first = mx == object_material;
    bool same_second = my == object_material;
    bool old_higher = dielectric_priority < (float)L.in_pri[0];
    if (is_inside && same_first && !same_second) return my;
    if (is_inside || same_first || old_higher) return mx;
    return object_material;
}

// ---- bounce (:1324-1359, :1422-1549) ----------------------------------------------------
PTK_DEV Ray bounce_ray(const PtkParams& P, Lane& L, const Ray& ray_in, Surf& s) {
    bool is_inside = dot3(s.normal, ray_in.d) > 0.0f;
    f3 normal = is_inside ? -s.normal : s.normal;
    int material = s.mtl_index;
    float eta_in, eta_out;
    float4 om0, om1;  // the material that decides the branch: albedo/roughness, metallic/opacity/reflectivity/IOR
    if (P.refraction_mode == 0) {
        om1 = P.materials[(size_t)s.mtl_index * 4 + 1];
        float material_ior = om1.w;
        eta_in = is_inside ? material_ior : 1.0f;
        eta_out = is_inside ? 1.0f : material_ior;
    } else if (P.refraction_mode == 1) {
        om1 = P.materials[(size_t)s.mtl_index * 4 + 1];
        eta_in = L.current_IOR;
        eta_out = is_inside ? 1.0f : om1.w;
    } else {
        eta_in = L.current_IOR;
        if (L.inside_of_count > 0) material = material_chooser(L, is_inside, (float)s.dielectric_priority, material);
        if (material == L.in_mat[0]) {
            // same medium on both sides: pass straight through (:1516-1528)
            L.ignore_attenuation = 1;
            if (L.refraction_bounces < P.push.max_refraction_bounces - 1) L.refraction_bounces++;
            if (is_inside) remove_material_from_inside_of(L, s.mtl_index);
            else add_material_to_inside_of(L, s.mtl_index, s.dielectric_priority);
            return new_ray(offset_ray(s.point, -normal), ray_in.d);
        }
        om0 = P.materials[(size_t)material * 4];
        om1 = P.materials[(size_t)material * 4 + 1];
        eta_out = om1.w;
        s.albedo = ld3(om0);
    }
    const float metallic = om1.x, opacity = om1.y;
    if (opacity < 1.0f) {
        // refract_ray, :1422-1479
        if (is_inside) L.ignore_attenuation = 1;
        float4 hm1 = P.materials[(size_t)s.mtl_index * 4 + 1];  // the hit object's own material
        float eta = eta_in / eta_out;
        float cos_theta = pmin(dot3(-ray_in.d, normal), 1.0f);
        float sin_theta = psqrt(1.0f - cos_theta * cos_theta);
        bool cannot_refract = eta * sin_theta > 1.0f;
        bool do_reflect = cannot_refract;
        if (!do_reflect) do_reflect = reflectance(eta, cos_theta) * hm1.z > noise(L);
        if (do_reflect) return new_ray(offset_ray(s.point, normal), reflect3(ray_in.d, normal));
        if (P.refraction_mode != 0) L.current_IOR = eta_out;
        if (P.refraction_mode == 2) {
            if (is_inside) remove_material_from_inside_of(L, s.mtl_index);
            else {
                float4 hm2 = P.materials[(size_t)s.mtl_index * 4 + 2];
                add_material_to_inside_of(L, s.mtl_index, (int)fbits(hm2.x));
            }
        }
        if (L.refraction_bounces < P.push.max_refraction_bounces - 1) L.refraction_bounces++;
        return new_ray(offset_ray(s.point, -normal), refract3(ray_in.d, normal, eta));
    }
    // reflect_ray (:1340-1359) and scatter_ray (:1324-1338) start alike -- the origin moved off the surface, one random
    // point on the sphere (each draws it once, as its first and only use of the generator) -- so the metallic and the
    // diffuse lanes of a wave run that part together instead of one kind after the other
    const f3 new_origin = offset_ray(s.point, normal);
    const f3 sphere_point = random_point_on_sphere(L);
    f3 ray_dir;
    if (metallic > 0.0f) {
        const f3 reflected_dir = reflect3(ray_in.d, normal) * metallic;
        const f3 dir_offset = sphere_point * (1.0f - metallic);
        ray_dir = reflected_dir + dir_offset;
        if (dot3(ray_dir, normal) < 0.0f) ray_dir = reflect3(ray_dir, normal);
    } else {
        ray_dir = sphere_point + normal;
        if (near_zero3(ray_dir)) ray_dir = normal;
        if (dot3(ray_dir, normal) < 0.0f) ray_dir = ray_dir * -1.0f;
    }
    return new_ray(new_origin, ray_dir);
}

// ---- camera ray and pixel store (main(), :1639-1705) -----------------------------------------------
struct Primary {
    Ray ray;
    float seed;
    int ux, uy;  // pixel in the image
};
PTK_DEV Primary primary_ray(const PtkPush& push, int gx, int gy) {  // :1645-1681
    Primary pr;
    f3 cam_pos = mk3(push.cam_pos[0], push.cam_pos[1], push.cam_pos[2]);
    f3 right = mk3(push.cam_right[0], push.cam_right[1], push.cam_right[2]);
    f3 up = mk3(push.cam_up[0], push.cam_up[1], push.cam_up[2]);
    f3 forward = mk3(push.cam_forward[0], push.cam_forward[1], push.cam_forward[2]);
    f3 viewport_u = right * push.viewport_width;
    f3 viewport_v = (-up) * push.viewport_height;
    f3 pixel_delta_u = div3(viewport_u, (float)push.width);
    f3 pixel_delta_v = div3(viewport_v, (float)push.height);
    f3 viewport_upper_left = (cam_pos - forward * push.focal_length) - div3(viewport_u + viewport_v, 2.0f);
    f3 pixel00_loc = viewport_upper_left + (pixel_delta_u + pixel_delta_v) * 0.5f;
    pr.ux = gx + push.image_offset_x;
    pr.uy = gy + push.image_offset_y;
    const float UVx = (float)pr.ux, UVy = (float)pr.uy;
    f3 pixel_center = (pixel00_loc + pixel_delta_u * UVx) + pixel_delta_v * UVy;
    pr.seed = hash3(UVx, UVy, push.time);
    float jitter = pt_rand(push.time) - 1.0f;
    f3 offset = div3(pixel_delta_u, 2.0f) * jitter + div3(pixel_delta_v, 2.0f) * jitter;
    f3 ray_direction = normalize3((pixel_center - cam_pos) + offset);
    pr.ray = new_ray(cam_pos, ray_direction);
    return pr;
}

// gamma, accumulate, store (:1683-1704).  Returns whether the pixel lies inside the image.
PTK_DEV bool store_pixel(const PtkParams& P, int ux, int uy, float4 new_color, bool override_color, float4 render_mode_color) {
    const PtkPush& push = P.push;
    const bool multisample = push.flags >= 0 && (push.flags & 2);
    new_color.x = ppow(new_color.x, push.gamma);
    new_color.y = ppow(new_color.y, push.gamma);
    new_color.z = ppow(new_color.z, push.gamma);
    const bool in_image = ux >= 0 && uy >= 0 && ux < P.img_w && uy < P.img_h;
    const size_t pix = (size_t)uy * P.img_w + ux;
    if (multisample) {
        float4 prev = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (in_image) prev = P.image[pix];
        float a = 1.0f / pmax(1.0f, push.frame);
        float ia = 1.0f - a;
        new_color = make_float4(prev.x * ia + new_color.x * a, prev.y * ia + new_color.y * a,
                                prev.z * ia + new_color.z * a, prev.w * ia + new_color.w * a);
    }
    if (override_color) new_color = render_mode_color;
    if (in_image) P.image[pix] = new_color;
    return in_image;
}

}  // namespace
