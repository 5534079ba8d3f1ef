/*
 * pt_oracle.c -- CPU restatement of the reference render kernel.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product;
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * build, load or call it.
 *
 * Follows /root/reference/shaders/ray_tracer.comp function by function (one
 * invocation = one pixel = one sample per dispatch); every function cites the
 * shader lines it restates.  Arithmetic is IEEE binary32 with one rounding per
 * operation (build with -ffp-contract=off); transcendental functions are the
 * ones fixed in pto_math.h.
 *
 * PARITY PINNING: the reference holds no golden image, hit-id or RNG vector
 * for this kernel, and neither Godot, a GLSL compiler nor a Vulkan driver
 * exists in the build image, so the shader itself cannot be run.  The render
 * oracle is therefore "parity unpinned" against the reference; it is pinned
 * only by hand-derived known answers (hash/pcg integer arithmetic, analytic
 * single-primitive scenes) in tests/.  The BVH/AABB side (pt_bvh_oracle.c) is
 * pinned by the reference's own unit-test vectors.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

#include "pto_math.h"
#include "pt_oracle.h"

/* ray_tracer.comp:109-123 */
#define INT_NEG_LIMIT ((int32_t)0x80000000u)
static const float PT_INFINITY = INFINITY;
static const float PT_PI = 3.1415926535897932385f;
static const float PT_PI_DOUBLE = 3.1415926535897932385f * 2.0f;
static const float PT_EPS = 1e-6f;
static const float OFS_ORIGIN = 1.0f / 32.0f;
static const float OFS_FLOAT_SCALE = 1.0f / 65536.0f;
static const float OFS_INT_SCALE = 256.0f;

enum { IS_NOT_OBJ = 0, IS_SPHERE = 1, IS_PLANE = 2, IS_TRIANGLE = 3 }; /* :127-130 */
enum { TEX_IS_SAMPLED = 1, TEX_IS_SPATIAL = 2 };                        /* :133-134 */
#define MAX_OVERLAP 5                                                    /* :156 */
#define MAX_BVH_DEPTH (64 * 16)                                          /* :188 */

typedef struct { int material_index; int dielectric_priority; } InsideOf; /* :144-147 */

typedef struct { v3 origin; float start; v3 direction; float end; } Ray;  /* :252-257 */

typedef struct {                                                          /* :325-339 */
    int hit;
    int object_id;
    int transform_index;
    float uv_x, uv_y;
    v3 point;
    int mtl_index;
    v3 normal;
    uint32_t texture_id;
    v4 color;
    int dielectric_priority;
} RayHit;

/* Buffer records, ray_tracer.comp:259-323,341-359 */
typedef struct { float albedo[3]; float roughness, metallic, opacity, reflectivity, ior;
                 int32_t dielectric_priority; int32_t is_emissive; int32_t pad[6]; } Material;
typedef struct { float normal[3]; float d; int32_t mtl_index; uint32_t texture_id; int32_t f0, f1; } Plane;
typedef struct { float center[3]; float radius; int32_t mtl_index; uint32_t texture_id; int32_t f0, f1; } Sphere;
typedef struct { float p1[3], f1; float p2[3], f2; float p3[3], f3; float n[3], f4;
                 int32_t mtl_index; uint32_t texture_id; int32_t f0, f5; } Triangle;
typedef struct { float minimum[4]; float maximum[4]; } BVHNode;

/* Push constants, ray_tracer.comp:416-434 */
typedef struct {
    float cam_pos[3]; float focal_length;
    float cam_right[3]; float viewport_width;
    float cam_up[3]; float viewport_height;
    float cam_forward[3]; float gamma;
    int32_t flags, render_mode, filler1, filler2, filler3;
    int32_t image_offset_x, image_offset_y;
    int32_t node_count_threshold, object_count_threshold;
    int32_t width, height, max_default_depth, max_refraction_bounces;
    float time, frame, max_samples;
} Push;

/* Per-invocation mutable globals of the shader (:159-237, :1105-1113). */
typedef struct {
    const pto_scene* S;
    const Push* P;
    int order;
    float current_IOR;
    int inside_of_count;
    InsideOf inside_of[MAX_OVERLAP];
    int is_emissive;
    int refraction_bounces;
    int current_transform_index;
    int ignore_attenuation;
    int use_bvh, multisample, show_node_count, show_object_count;
    int show_bvh_view, show_normal_view, render_mode_is_single_sample;
    v4 debug_color, render_mode_color;
    int node_hit_count, object_hit_count;
    float seed, value; /* Random */
    int to_visit[MAX_BVH_DEPTH + PTO_MAX_ORDER + 1];
    float to_visit_t[MAX_BVH_DEPTH + PTO_MAX_ORDER + 1];
    int to_visit_i;
    int temp_children[PTO_MAX_ORDER + 1];
    float temp_ts[PTO_MAX_ORDER + 1];
    /* statistics for the algorithmic byte count (not in the shader) */
    int64_t st[PTO_NSTAT];
} Ctx;

static inline v3 ld3(const float* p) { return V3(p[0], p[1], p[2]); }
static inline v4 V4(float x, float y, float z, float w) { v4 r = {x, y, z, w}; return r; }

static inline const Material* mat_at(const Ctx* c, int i) { return (const Material*)(c->S->materials) + i; }
static inline const Sphere* sph_at(const Ctx* c, int i) { return (const Sphere*)(c->S->spheres) + i; }
static inline const Plane* pln_at(const Ctx* c, int i) { return (const Plane*)(c->S->planes) + i; }
static inline const Triangle* tri_at(const Ctx* c, int i) { return (const Triangle*)(c->S->triangles) + i; }
static inline const BVHNode* node_at(const Ctx* c, int i) { return (const BVHNode*)(c->S->bvh) + i; }
static inline m4 xform_at(const Ctx* c, int i) { m4 m; memcpy(&m, c->S->transforms + (size_t)i * 64, 64); return m; }

/* bitfieldExtract on a signed int: sign-extends (GLSL spec). */
static inline int32_t bfe_i(int32_t v, int off, int bits) {
    uint32_t u = (uint32_t)v << (32 - off - bits);
    return (int32_t)u >> (32 - bits);
}
static inline uint32_t bfe_u(uint32_t v, int off, int bits) {
    return (v >> off) & ((bits >= 32) ? 0xffffffffu : ((1u << bits) - 1u));
}
static inline int iabs(int v) { return v < 0 ? -v : v; }

static inline int get_object_type(int id) { return bfe_i(id, 24, 8); }      /* :592-594 */
static inline int get_object_index(int id) { return bfe_i(id, 0, 24); }     /* :596-598 */
static inline int get_node_index(int id) { return bfe_i(id, 0, 24); }       /* :600-602 */
static inline int get_node_size(int id) { return iabs(bfe_i(id, 24, 8)); }  /* :604-606 */
static inline uint32_t get_texture_type(uint32_t id) { return bfe_u(id, 30, 2); }  /* :608-610 */
static inline uint32_t get_texture_index(uint32_t id) { return bfe_u(id, 0, 30); } /* :612-614 */

static RayHit empty_rayhit(void) { /* :439-441 */
    RayHit r;
    memset(&r, 0, sizeof r);
    r.transform_index = -1;
    return r;
}
static Ray new_ray(v3 origin, v3 direction) { /* :444-446 */
    Ray r; r.origin = origin; r.start = 0.0f; r.direction = v3_normalize(direction); r.end = PT_INFINITY;
    return r;
}
static float reflectance(float eta, float cos_theta) { /* :456-461 */
    float r0 = (1.0f - eta) / (1.0f + eta);
    r0 = r0 * r0;
    return r0 + (1.0f - r0) * ptm_pow(1.0f - cos_theta, 5.0f);
}
static float pt_rand(float seed) { /* :464-466 */
    return ptm_fract((seed * 23489.52364f) / 0.0836f);
}
static float noise(Ctx* c) { /* :468-472 */
    float s = c->seed; c->seed = c->seed + 1.0f;
    c->value = ptm_fract(ptm_sin(s) * 43758.5453123f);
    return c->value;
}
static float pcg(Ctx* c) { /* :476-481 */
    float s = c->seed; c->seed = c->seed + 1.0f;
    uint32_t state = f2u(s) * 747796405u + 2891336453u;
    uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
    c->value = (float)((word >> 22u) ^ word) / 4294967295.0f;
    return c->value;
}
static v3 random_point_on_sphere(Ctx* c) { /* :487-498 */
    float theta1 = pcg(c) * PT_PI_DOUBLE;
    float theta2 = ptm_acos(2.0f * pcg(c) - 1.0f);
    float s2 = ptm_sin(theta2);
    return V3(ptm_cos(theta1) * s2, ptm_cos(theta2), ptm_sin(theta1) * s2);
}
float pto_hash(float x, float y, float z) { /* :500-507 */
    uint32_t px = f2u(x), py = f2u(y), pz = f2u(z);
    uint32_t qx = 1103515245u * ((px >> 1u) ^ py);
    uint32_t qy = 1103515245u * ((py >> 1u) ^ pz);
    uint32_t qz = 1103515245u * ((pz >> 1u) ^ px);
    uint32_t h32 = 1103515245u * ((qx ^ qz) ^ (qy >> 3u));
    uint32_t n = h32 ^ (h32 >> 16u);
    return (float)n * (1.0f / (float)0x7fffffff);
}
/* swap by arithmetic, exactly as written (not an exact swap in binary32). :509-514 */
static inline void pt_swap(float* a, float* b) {
    *a = *a + *b;
    *b = *a - *b;
    *a = *a - *b;
}
static inline int near_zero3(v3 v) { /* :523-526 */
    return ptm_abs(v.x) < PT_EPS && ptm_abs(v.y) < PT_EPS && ptm_abs(v.z) < PT_EPS;
}
static inline int near_zero1(float f) { return ptm_abs(f) < PT_EPS; } /* :528-530 */

static inline float offset_axis(float p, float n) { /* :532-546, one component */
    int32_t off = ptm_f2i(n * OFS_INT_SCALE);
    int32_t bits = (int32_t)((uint32_t)f2i_bits(p) + (uint32_t)((p < 0.0f) ? -off : off));
    float p_int = i2f_bits(bits);
    return (ptm_abs(p) < OFS_ORIGIN) ? (p + OFS_FLOAT_SCALE * n) : p_int;
}
static v3 offset_ray(v3 point, v3 normal) { /* :532-546 */
    return V3(offset_axis(point.x, normal.x), offset_axis(point.y, normal.y), offset_axis(point.z, normal.z));
}
static inline int in_range(float value, const Ray* ray) { /* :548-551 */
    return ray->start <= value && ray->end >= value;
}
static inline v3 ray_at_end(const Ray* ray) { /* :558-560 */
    return v3_add(ray->origin, v3_scale(ray->direction, ray->end));
}

static void parse_flags(Ctx* c) { /* :566-590 */
    /* The 31-iteration subtract loop yields bit i of push.flags for
     * non-negative flags and all-false for negative flags. */
    int32_t f = c->P->flags;
    int b0 = 0, b1 = 0, b2 = 0, b3 = 0;
    if (f >= 0) { b0 = f & 1; b1 = (f >> 1) & 1; b2 = (f >> 2) & 1; b3 = (f >> 3) & 1; }
    c->use_bvh = b0; c->multisample = b1; c->show_node_count = b2; c->show_object_count = b3;
    c->show_bvh_view = c->P->render_mode == 1;
    c->show_normal_view = c->P->render_mode == 2;
    c->render_mode_is_single_sample = c->show_bvh_view || c->show_normal_view;
}

/* ---- textures ----------------------------------------------------------- */
static void sphere_get_uv(v3 pos, float* u, float* v) { /* :620-627 */
    float theta = ptm_acos(pos.y);
    float phi = ptm_atan2(-pos.z, pos.x) + PT_PI;
    *u = phi / PT_PI_DOUBLE;
    *v = theta / PT_PI;
}
/* shaders/procedural_textures/checker_board.comp:4-26 */
static v4 proc_checker_board(v3 pos) {
    const float divisor = 1.0f / 0.5f;
    float nx = floorf(pos.x * divisor), nz = floorf(pos.z * divisor);
    float sum = nx + nz;
    return (ptm_mod(sum, 2.0f) != 0.0f) ? V4(0.9f, 0.9f, 0.9f, 1.0f) : V4(0.2f, 0.3f, 0.1f, 1.0f);
}
/* shaders/procedural_textures/checker_board_spatial.comp:4-13 */
static v4 proc_checker_board_spatial(v3 pos) {
    const float divisor1 = 1.0f / 0.5f;
    float nx = floorf(pos.x * divisor1), ny = floorf(pos.y * divisor1), nz = floorf(pos.z * divisor1);
    float sum = (nx + ny) + nz;
    return (ptm_mod(sum, 2.0f) != 0.0f) ? V4(0.9f, 0.9f, 0.9f, 1.0f) : V4(0.2f, 0.3f, 0.1f, 1.0f);
}
static v4 procedural_texture(const Ctx* c, v3 pos, uint32_t function) { /* :632-646 + pt_utils.gd:150-189 */
    if (function == 0) return V4(1, 1, 1, 1);
    if (function >= 1 && function <= (uint32_t)c->S->n_proc) {
        int b = c->S->proc_builtin[function - 1];
        if (b == PTO_PROC_CHECKER_BOARD) return proc_checker_board(pos);
        if (b == PTO_PROC_CHECKER_BOARD_SPATIAL) return proc_checker_board_spatial(pos);
    }
    return V4(0, 0, 0, 0);
}
/* texture(sampler2D, uv) with Godot's default RDSamplerState (nearest filter,
 * clamp to edge) at LOD 0 (no derivatives in a compute shader).
 * pt_work_dispatcher.gd:297-347. */
static v4 sample_texture(const Ctx* c, uint32_t index, float u, float v) {
    if (index >= (uint32_t)c->S->n_textures || c->S->textures[index].rgba == NULL) {
        /* filler texture: 8x8 RGBA8 pink/black grid, pt_utils.gd:122-137 */
        int i = ptm_f2i(floorf(u * 8.0f)), j = ptm_f2i(floorf(v * 8.0f));
        if (i < 0) i = 0; if (i > 7) i = 7; if (j < 0) j = 0; if (j > 7) j = 7;
        int pink = (((j % 2) + (i % 2)) % 2) == 0;
        if (pink) return V4(56.0f / 255.0f, 16.0f / 255.0f, 56.0f / 255.0f, 56.0f / 255.0f);
        return V4(0.0f, 0.0f, 0.0f, 56.0f / 255.0f);
    }
    const pto_texture* t = &c->S->textures[index];
    int i = ptm_f2i(floorf(u * (float)t->width)), j = ptm_f2i(floorf(v * (float)t->height));
    if (i < 0) i = 0; if (i > t->width - 1) i = t->width - 1;
    if (j < 0) j = 0; if (j > t->height - 1) j = t->height - 1;
    const float* p = t->rgba + ((size_t)j * t->width + i) * 4;
    return V4(p[0], p[1], p[2], p[3]);
}
static v4 get_texture(const Ctx* c, const RayHit* rh) { /* :649-659 */
    uint32_t type = get_texture_type(rh->texture_id);
    uint32_t index = get_texture_index(rh->texture_id);
    v3 sampled_point = (type & TEX_IS_SPATIAL) ? rh->point : V3(rh->uv_x, rh->uv_y, 0.0f);
    if (type & TEX_IS_SAMPLED) return sample_texture(c, index, rh->uv_x, rh->uv_y);
    return procedural_texture(c, sampled_point, index);
}

/* ---- intersection -------------------------------------------------------- */
static float hit_AABB(Ctx* c, Ray ray, const BVHNode* box, v3 inv_dir, const int is_neg[3]) { /* :670-689 */
    c->node_hit_count++;
    const float o[3] = {ray.origin.x, ray.origin.y, ray.origin.z};
    const float inv[3] = {inv_dir.x, inv_dir.y, inv_dir.z};
    for (int i = 0; i < 3; i++) {
        float t0 = (box->minimum[i] - o[i]) * inv[i];
        float t1 = (box->maximum[i] - o[i]) * inv[i];
        if (is_neg[i]) pt_swap(&t0, &t1);
        if (t0 > ray.start) ray.start = t0;
        if (t1 < ray.end) ray.end = t1;
        if (ray.end <= ray.start) return -1.0f;
    }
    return ray.start;
}

static void set_hit(Ctx* c, Ray* ray, RayHit* rh, int object_id, float t) {
    rh->hit = 1;
    rh->object_id = object_id;
    rh->transform_index = c->current_transform_index;
    ray->end = t;
}

static void hit_sphere(Ctx* c, Ray* ray, int object_id, RayHit* rh) { /* :691-727 */
    const Sphere* s = sph_at(c, get_object_index(object_id));
    v3 oc = v3_sub(ray->origin, ld3(s->center));
    float a = v3_dot(ray->direction, ray->direction);
    float half_b = v3_dot(oc, ray->direction);
    float cc = v3_dot(oc, oc) - s->radius * s->radius;
    float discriminant = half_b * half_b - a * cc;
    if (discriminant < 0.0f) return;
    float sqrtd = sqrtf(discriminant);
    float root = (-half_b - sqrtd) / a;
    if (!in_range(root, ray)) {
        root = (-half_b + sqrtd) / a;
        if (!in_range(root, ray)) return;
    }
    set_hit(c, ray, rh, object_id, root);
}

static void hit_sphere_gems(Ctx* c, Ray* ray, int object_id, RayHit* rh) { /* :803-841 */
    const Sphere* s = sph_at(c, get_object_index(object_id));
    c->st[PTO_ST_SPHERE_TESTS]++;
    v3 oc = v3_sub(ray->origin, ld3(s->center));
    float half_b = v3_dot(v3_neg(oc), ray->direction);
    float r_sqrd = s->radius * s->radius;
    v3 temp = v3_add(oc, v3_scale(ray->direction, half_b));
    float discriminant = r_sqrd - v3_dot(temp, temp);
    if (discriminant < 0.0f) return;
    float cc = v3_dot(oc, oc) - r_sqrd;
    float sqrtd = sqrtf(discriminant);
    float q = half_b + ptm_sign(half_b) * sqrtd;
    float root = cc / q;
    if (!in_range(root, ray)) {
        root = q;
        if (!in_range(root, ray)) return;
    }
    set_hit(c, ray, rh, object_id, root);
}

static void hit_plane(Ctx* c, Ray* ray, int object_id, RayHit* rh) { /* :851-875 */
    const Plane* p = pln_at(c, get_object_index(object_id));
    v3 n = ld3(p->normal);
    float t = (p->d - v3_dot(ray->origin, n)) / v3_dot(n, ray->direction);
    if (t != t || ptm_abs(t) == PT_INFINITY) return;
    if (!in_range(t, ray)) return;
    set_hit(c, ray, rh, object_id, t);
}
static void hit_planes(Ctx* c, Ray* ray, RayHit* rh) { /* :879-885 */
    for (int i = 0; i < c->S->n_planes; ++i) hit_plane(c, ray, i + (IS_PLANE << 24), rh);
}

static int hit_triangle_MT(Ctx* c, Ray* ray, int object_id, RayHit* rh) { /* :926-969 */
    const Triangle* tr = tri_at(c, get_object_index(object_id));
    c->st[PTO_ST_TRI_TESTS]++;
    v3 A = ld3(tr->p1), B = ld3(tr->p2), C = ld3(tr->p3);
    v3 E1 = v3_sub(B, A), E2 = v3_sub(C, A);
    v3 N = v3_cross(E1, E2);
    float det = -v3_dot(ray->direction, N);
    if (near_zero1(det)) return 0;
    float invdet = 1.0f / det;
    v3 AO = v3_sub(ray->origin, A);
    float t = v3_dot(AO, N) * invdet;
    if (!in_range(t, ray)) return 0;
    v3 DAO = v3_cross(AO, ray->direction);
    float u = v3_dot(v3_sub(C, A), DAO) * invdet;
    float v = -v3_dot(v3_sub(B, A), DAO) * invdet;
    if (u >= 0.0f && v >= 0.0f && (u + v) <= 1.0f) {
        set_hit(c, ray, rh, object_id, t);
        rh->uv_x = u; rh->uv_y = v;
        return 1;
    }
    return 0;
}

static void hit_object(Ctx* c, Ray* ray, RayHit* rh, int object_id) { /* :979-996 */
    int type = get_object_type(object_id);
    c->object_hit_count++;
    if (type == IS_SPHERE) hit_sphere_gems(c, ray, object_id, rh);
    else if (type == IS_TRIANGLE) hit_triangle_MT(c, ray, object_id, rh);
}

static void determine_rayhit(Ctx* c, const Ray* ray_in, RayHit* rh) { /* :1003-1091 */
    int type = get_object_type(rh->object_id);
    int index = get_object_index(rh->object_id);
    if (type == IS_SPHERE) {
        const Sphere* s = sph_at(c, index);
        v3 center = ld3(s->center);
        rh->mtl_index = s->mtl_index;
        rh->texture_id = s->texture_id;
        if (rh->transform_index >= 0) {
            m4 m = xform_at(c, rh->transform_index);
            m4 matrix = m4_inverse(&m);
            Ray tr = *ray_in;
            tr.direction = m4_mul_v(&m, ray_in->direction, 0.0f);
            tr.origin = m4_mul_v(&m, ray_in->origin, 1.0f);
            v3 usn = v3_normalize(v3_sub(ray_at_end(&tr), center));
            sphere_get_uv(usn, &rh->uv_x, &rh->uv_y);
            rh->point = v3_add(center, v3_scale(usn, s->radius));
            rh->point = m4_mul_v(&matrix, rh->point, 1.0f);
            rh->normal = v3_normalize(v_mul_m4(usn, 0.0f, &m));
        } else {
            rh->normal = v3_normalize(v3_sub(ray_at_end(ray_in), center));
            sphere_get_uv(rh->normal, &rh->uv_x, &rh->uv_y);
            rh->point = v3_add(center, v3_scale(rh->normal, s->radius));
        }
        c->st[PTO_ST_HIT_SPHERE]++;
    } else if (type == IS_PLANE) {
        const Plane* p = pln_at(c, index);
        rh->point = ray_at_end(ray_in);
        rh->mtl_index = p->mtl_index;
        rh->texture_id = p->texture_id;
        rh->normal = ld3(p->normal);
        c->st[PTO_ST_HIT_PLANE]++;
    } else if (type == IS_TRIANGLE) {
        const Triangle* t = tri_at(c, index);
        v3 p1 = ld3(t->p1), p2 = ld3(t->p2), p3 = ld3(t->p3);
        rh->mtl_index = t->mtl_index;
        rh->texture_id = t->texture_id;
        rh->normal = v3_normalize(v3_cross(v3_sub(p1, p3), v3_sub(p1, p2)));
        /* uv.x*p2 + uv.y*p3 + (1 - uv.x - uv.y)*p1, left to right */
        float w = (1.0f - rh->uv_x) - rh->uv_y;
        rh->point = v3_add(v3_add(v3_scale(p2, rh->uv_x), v3_scale(p3, rh->uv_y)), v3_scale(p1, w));
        if (rh->transform_index >= 0) {
            m4 m = xform_at(c, rh->transform_index);
            m4 matrix = m4_inverse(&m);
            rh->normal = v3_normalize(v_mul_m4(rh->normal, 0.0f, &m));
            rh->point = m4_mul_v(&matrix, rh->point, 1.0f);
        }
        if (c->S->triangle_uvs && index >= 0 && index < c->S->n_triangle_uvs) {
            /* uv.x*uv2 + uv.y*uv3 + (1 - uv.x - uv.y)*uv1, in the order of the point above */
            const float* q = c->S->triangle_uvs + (size_t)index * 6;
            float bu = rh->uv_x, bv = rh->uv_y;
            rh->uv_x = (q[2] * bu + q[4] * bv) + q[0] * w;
            rh->uv_y = (q[3] * bu + q[5] * bv) + q[1] * w;
        }
        if (v3_dot(rh->normal, ray_in->direction) > 0.0f) rh->normal = v3_scale(rh->normal, -1.0f);
        c->st[PTO_ST_HIT_TRIANGLE]++;
    }
    const Material* mt = mat_at(c, rh->mtl_index);
    rh->color = V4(mt->albedo[0], mt->albedo[1], mt->albedo[2], 1.0f);
    rh->dielectric_priority = mt->dielectric_priority;
    if (c->S->use_emission) c->is_emissive = mt->is_emissive != 0;
}

static RayHit check_ray_hit(Ctx* c, Ray* ray) { /* :1093-1101, brute force */
    RayHit rh = empty_rayhit();
    for (int i = 0; i < c->S->n_spheres; ++i) hit_sphere(c, ray, i + (IS_SPHERE << 24), &rh);     /* :843-849 */
    for (int i = 0; i < c->S->n_triangles; ++i) hit_triangle_MT(c, ray, i + (IS_TRIANGLE << 24), &rh); /* :971-977 */
    hit_planes(c, ray, &rh);
    return rh;
}

static void calculate_top_node(Ctx* c, Ray* ray, RayHit* rh, v3 inv_dir, const int is_neg[3]) { /* :1115-1181 */
    int current_index = c->to_visit[--c->to_visit_i];
    const BVHNode* node = node_at(c, current_index);
    int node_id = f2i_bits(node->minimum[3]);
    int node_index = get_node_index(node_id);
    int node_size = get_node_size(node_id);
    const int order = c->order;

    if (c->S->early_t_break) {
        if (!in_range(c->to_visit_t[c->to_visit_i], ray)) return;
    }
    if (node_index <= 0) {
        for (int k = 0; k < node_size; k++) {
            hit_object(c, ray, rh, c->S->object_ids[iabs(node_index) + k - 1]);
        }
        return;
    }
    int added_children = 0;
    for (int l = 0; l < node_size; l++) {
        const BVHNode* child = node_at(c, node_index + l);
        float aabb_t = hit_AABB(c, *ray, child, inv_dir, is_neg);
        if (in_range(aabb_t, ray)) {
            c->temp_ts[l] = aabb_t;
            c->temp_children[l] = node_index + l;
            added_children++;
        }
    }
    c->to_visit_i += added_children;
    for (int i = 0; i < added_children; i++) {
        float smallest_t = PT_INFINITY;
        int smallest_index = order;
        for (int j = 0; j < order; j++) {
            if (c->temp_ts[j] < smallest_t) {
                smallest_t = c->temp_ts[j];
                smallest_index = j;
            }
        }
        c->temp_ts[smallest_index] = PT_INFINITY;
        c->to_visit[c->to_visit_i - i - 1] = c->temp_children[smallest_index];
        if (c->S->early_t_break) c->to_visit_t[c->to_visit_i - i - 1] = smallest_t;
    }
}

static float check_ray_hit_BVH_mesh(Ctx* c, Ray ray, RayHit* rh) { /* :1183-1198 */
    v3 inv = V3(1.0f / ray.direction.x, 1.0f / ray.direction.y, 1.0f / ray.direction.z);
    int is_neg[3] = {inv.x < 0.0f, inv.y < 0.0f, inv.z < 0.0f};
    int start_index = c->to_visit_i;
    while (c->to_visit_i >= start_index && c->to_visit_i <= MAX_BVH_DEPTH) {
        calculate_top_node(c, &ray, rh, inv, is_neg);
    }
    return ray.end;
}

static RayHit check_ray_hit_BVH(Ctx* c, Ray* ray) { /* :1200-1320 */
    RayHit rh = empty_rayhit();
    hit_planes(c, ray, &rh);
    v3 inv = V3(1.0f / ray->direction.x, 1.0f / ray->direction.y, 1.0f / ray->direction.z);
    int is_neg[3] = {inv.x < 0.0f, inv.y < 0.0f, inv.z < 0.0f};
    float root_hit = hit_AABB(c, *ray, node_at(c, 0), inv, is_neg);
    c->to_visit_i = in_range(root_hit, ray) ? 1 : 0;
    c->to_visit[0] = 0;
    if (c->S->early_t_break) c->to_visit_t[0] = root_hit;
    for (int h = 0; h < c->order; h++) c->temp_ts[h] = PT_INFINITY;

    while (c->to_visit_i > 0 && c->to_visit_i <= MAX_BVH_DEPTH) {
        int current_index = c->to_visit[--c->to_visit_i];
        const BVHNode* node = node_at(c, current_index);
        int node_id = f2i_bits(node->minimum[3]);
        int node_index = get_node_index(node_id);
        if (c->S->early_t_break) {
            if (!in_range(c->to_visit_t[c->to_visit_i], ray)) break;
        }
        int transform_index = f2i_bits(node->maximum[3]);
        if (transform_index >= 0) {
            m4 matrix = xform_at(c, transform_index);
            c->st[PTO_ST_SOCKETS]++;
            Ray nr = *ray;
            v3 unnormalized = m4_mul_v(&matrix, ray->direction, 0.0f);
            float direction_length = v3_length(unnormalized);
            nr.direction = v3_divs(unnormalized, direction_length);
            nr.origin = m4_mul_v(&matrix, ray->origin, 1.0f);
            nr.end = nr.end * direction_length;
            nr.start = nr.start * direction_length;
            if (c->S->early_t_break) {
                const BVHNode* child = node_at(c, node_index);
                float aabb_t = hit_AABB(c, *ray, child, inv, is_neg);
                c->to_visit_t[c->to_visit_i] = aabb_t;
            }
            c->to_visit[c->to_visit_i++] = node_index;
            c->current_transform_index = transform_index;
            ray->end = check_ray_hit_BVH_mesh(c, nr, &rh) / direction_length;
            c->current_transform_index = -1;
            continue;
        }
        c->to_visit_i++;
        calculate_top_node(c, ray, &rh, inv, is_neg);
    }
    return rh;
}

/* ---- bounce -------------------------------------------------------------- */
static Ray scatter_ray(Ctx* c, v3 point, v3 normal) { /* :1324-1338 */
    v3 new_origin = offset_ray(point, normal);
    v3 ray_dir = v3_add(random_point_on_sphere(c), normal);
    if (near_zero3(ray_dir)) ray_dir = normal;
    if (v3_dot(ray_dir, normal) < 0.0f) ray_dir = v3_scale(ray_dir, -1.0f);
    return new_ray(new_origin, ray_dir);
}
static Ray reflect_ray(Ctx* c, const Ray* ray_in, v3 point, v3 normal, int material) { /* :1340-1359 */
    v3 new_origin = offset_ray(point, normal);
    float metallic = mat_at(c, material)->metallic;
    v3 reflected_dir = v3_scale(v3_reflect(ray_in->direction, normal), metallic);
    v3 dir_offset = v3_scale(random_point_on_sphere(c), 1.0f - metallic);
    v3 ray_dir = v3_add(reflected_dir, dir_offset);
    if (v3_dot(ray_dir, normal) < 0.0f) ray_dir = v3_reflect(ray_dir, normal);
    return new_ray(new_origin, ray_dir);
}
static void add_material_to_inside_of(Ctx* c, int material_index, int priority) { /* :1369-1382 */
    int shift = 1, insert_index = 0;
    for (int i = MAX_OVERLAP - 1; i > 0; i--) {
        InsideOf index = c->inside_of[i - 1];
        if (priority < index.dielectric_priority && shift == 1) { insert_index = i; shift = 0; }
        c->inside_of[i] = c->inside_of[i - shift];
    }
    c->inside_of[insert_index].material_index = material_index;
    c->inside_of[insert_index].dielectric_priority = priority;
    c->inside_of_count++;
}
static void remove_material_from_inside_of(Ctx* c, int material_index) { /* :1384-1398 */
    int shift = 0;
    for (int i = 0; i < MAX_OVERLAP - 1; i++) {
        InsideOf index = c->inside_of[i];
        if (shift == 0 && material_index == index.material_index) shift = 1;
        c->inside_of[i] = c->inside_of[i + shift];
    }
    c->inside_of[MAX_OVERLAP - 1].material_index = 0;
    c->inside_of[MAX_OVERLAP - 1].dielectric_priority = INT_NEG_LIMIT;
    c->inside_of_count--;
}
static void material_chooser(Ctx* c, int is_inside, float dielectric_priority, int* object_material) { /* :1400-1419 */
    int mx = c->inside_of[0].material_index, my = c->inside_of[1].material_index;
    int same_first = mx == *object_material;
    int same_second = my == *object_material;
    int old_priority_is_higher = dielectric_priority < (float)c->inside_of[0].dielectric_priority;
    if (is_inside && same_first && !same_second) *object_material = my;
    else if (is_inside || same_first || old_priority_is_higher) *object_material = mx;
}
static Ray refract_ray(Ctx* c, const Ray* ray_in, RayHit* rh, int is_inside, v3 normal,
                       float eta_in, float eta_out) { /* :1422-1479 */
    if (is_inside) c->ignore_attenuation = 1;
    const Material* object_material = mat_at(c, rh->mtl_index);
    float eta = eta_in / eta_out;
    float cos_theta = ptm_min(v3_dot(v3_neg(ray_in->direction), normal), 1.0f);
    float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
    int cannot_refract = eta * sin_theta > 1.0f;
    /* || short-circuits: noise() is drawn only when refraction is possible */
    if (cannot_refract || reflectance(eta, cos_theta) * object_material->reflectivity > noise(c)) {
        return new_ray(offset_ray(rh->point, normal), v3_reflect(ray_in->direction, normal));
    }
    if (c->S->refraction_mode != 0) c->current_IOR = eta_out;
    if (c->S->refraction_mode == 2) {
        if (is_inside) remove_material_from_inside_of(c, rh->mtl_index);
        else add_material_to_inside_of(c, rh->mtl_index, object_material->dielectric_priority);
    }
    if (c->refraction_bounces < c->P->max_refraction_bounces - 1) c->refraction_bounces++;
    return new_ray(offset_ray(rh->point, v3_neg(normal)), v3_refract(ray_in->direction, normal, eta));
}
static Ray bounce_ray(Ctx* c, const Ray* ray_in, RayHit* rh) { /* :1481-1549 */
    int is_inside = v3_dot(rh->normal, ray_in->direction) > 0.0f;
    v3 normal = is_inside ? v3_neg(rh->normal) : rh->normal;
    int material = rh->mtl_index;
    const Material* object_material;
    float eta_in, eta_out;
    if (c->S->refraction_mode == 0) {
        object_material = mat_at(c, rh->mtl_index);
        float material_ior = object_material->ior;
        eta_in = is_inside ? material_ior : 1.0f;
        eta_out = is_inside ? 1.0f : material_ior;
    } else if (c->S->refraction_mode == 1) {
        object_material = mat_at(c, rh->mtl_index);
        eta_in = c->current_IOR;
        eta_out = is_inside ? 1.0f : object_material->ior;
    } else {
        eta_in = c->current_IOR;
        if (c->inside_of_count > 0) material_chooser(c, is_inside, (float)rh->dielectric_priority, &material);
        if (material == c->inside_of[0].material_index) {
            c->ignore_attenuation = 1;
            if (c->refraction_bounces < c->P->max_refraction_bounces - 1) c->refraction_bounces++;
            if (is_inside) remove_material_from_inside_of(c, rh->mtl_index);
            else add_material_to_inside_of(c, rh->mtl_index, rh->dielectric_priority);
            Ray r; /* new_ray(offset, ray_in.direction): the direction is normalised again */
            r = new_ray(offset_ray(rh->point, v3_neg(normal)), ray_in->direction);
            return r;
        }
        object_material = mat_at(c, material);
        eta_out = object_material->ior;
        rh->color = V4(object_material->albedo[0], object_material->albedo[1], object_material->albedo[2], 1.0f);
    }
    if (object_material->opacity < 1.0f) return refract_ray(c, ray_in, rh, is_inside, normal, eta_in, eta_out);
    if (object_material->metallic > 0.0f) return reflect_ray(c, ray_in, rh->point, normal, material);
    return scatter_ray(c, rh->point, normal);
}

static v4 cast_ray(Ctx* c, Ray ray, RayHit* last_hit, int32_t* primary_id) { /* :1553-1635 */
    Ray nr = ray;
    RayHit rh = empty_rayhit();
    c->refraction_bounces = 0;
    v4 new_color = V4(1, 1, 1, 1);
    v4 tex_col = V4(1, 1, 1, 1);
    int i = 0;
    int first = 1;
    if (c->render_mode_is_single_sample) {
        c->refraction_bounces = c->P->max_refraction_bounces;
        i = c->P->max_default_depth + c->P->max_refraction_bounces - 1;
    }
    for (; i < c->P->max_default_depth + c->refraction_bounces; i++) {
        c->st[PTO_ST_RAYS]++;
        int nodes_before = c->node_hit_count;
        if (c->use_bvh) rh = check_ray_hit_BVH(c, &nr);
        else rh = check_ray_hit(c, &nr);
        c->st[PTO_ST_BOX_TESTS] += c->node_hit_count - nodes_before;
        if (first) { *primary_id = rh.hit ? rh.object_id : -1; first = 0; }
        if (!rh.hit) {
            rh.color = c->S->use_emission ? V4(0, 0, 0, 0) : V4(0.7f, 0.7f, 0.9f, 1.0f); /* :179-186,665-668 */
            new_color = V4(new_color.x * rh.color.x, new_color.y * rh.color.y, new_color.z * rh.color.z, new_color.w * rh.color.w);
            break;
        }
        determine_rayhit(c, &nr, &rh);
        c->st[PTO_ST_MATERIAL_FETCH]++;
        if (c->S->use_emission && c->is_emissive) {
            new_color = V4(new_color.x * rh.color.x, new_color.y * rh.color.y, new_color.z * rh.color.z, new_color.w * rh.color.w);
            break;
        }
        tex_col = get_texture(c, &rh);
        nr = bounce_ray(c, &nr, &rh);
        if (!c->ignore_attenuation) {
            new_color.x *= rh.color.x * tex_col.x;
            new_color.y *= rh.color.y * tex_col.y;
            new_color.z *= rh.color.z * tex_col.z;
            new_color.w *= rh.color.w * tex_col.w;
        } else {
            c->ignore_attenuation = 0;
        }
    }
    if (c->S->use_emission && !c->is_emissive) new_color = V4(0, 0, 0, 0);
    if (c->show_bvh_view) { /* :1624-1628 */
        float a = c->show_node_count ? (float)c->node_hit_count / (float)c->P->node_count_threshold : 0.0f;
        float b = c->show_object_count ? (float)c->object_hit_count / (float)c->P->object_count_threshold : 0.0f;
        c->render_mode_color = (ptm_max(a, b) > 1.0f) ? V4(1, 1, 1, 1) : V4(b, 0.0f, a, 0.0f);
    }
    if (c->show_normal_view) { /* :1630-1632 */
        c->render_mode_color = V4((rh.normal.x + 1.0f) / 2.0f, (rh.normal.y + 1.0f) / 2.0f, (rh.normal.z + 1.0f) / 2.0f, 1.0f);
    }
    *last_hit = rh;
    return new_color;
}

static int v4_ne(v4 a, float s) { return a.x != s || a.y != s || a.z != s || a.w != s; }

/* One invocation of main(), ray_tracer.comp:1639-1705. gx,gy = gl_GlobalInvocationID.xy */
static void shade_pixel(const pto_scene* S, const Push* P, float* image, int img_w, int img_h,
                        int gx, int gy, pto_trace* trace, Ctx* c) {
    memset(c->st, 0, sizeof c->st);
    c->S = S; c->P = P; c->order = S->order;
    c->current_IOR = 1.0f; c->inside_of_count = 0;
    for (int i = 0; i < MAX_OVERLAP; i++) { c->inside_of[i].material_index = 0; c->inside_of[i].dielectric_priority = INT_NEG_LIMIT; }
    c->is_emissive = 0; c->refraction_bounces = 0; c->current_transform_index = -1; c->ignore_attenuation = 0;
    c->debug_color = V4(-1, -1, -1, -1); c->render_mode_color = V4(-1, -1, -1, -1);
    c->node_hit_count = 0; c->object_hit_count = 0;
    c->seed = 42.0f; c->value = 0.0f;

    v3 cam_pos = ld3(P->cam_pos), right = ld3(P->cam_right), up = ld3(P->cam_up), forward = ld3(P->cam_forward);
    v3 viewport_u = v3_scale(right, P->viewport_width);
    v3 viewport_v = v3_scale(v3_neg(up), P->viewport_height);
    v3 pixel_delta_u = v3_divs(viewport_u, (float)P->width);
    v3 pixel_delta_v = v3_divs(viewport_v, (float)P->height);
    v3 viewport_upper_left = v3_sub(v3_sub(cam_pos, v3_scale(forward, P->focal_length)),
                                    v3_divs(v3_add(viewport_u, viewport_v), 2.0f));
    v3 pixel00_loc = v3_add(viewport_upper_left, v3_scale(v3_add(pixel_delta_u, pixel_delta_v), 0.5f));
    int ux = gx + P->image_offset_x, uy = gy + P->image_offset_y;
    float UVx = (float)ux, UVy = (float)uy;
    v3 pixel_center = v3_add(v3_add(pixel00_loc, v3_scale(pixel_delta_u, UVx)), v3_scale(pixel_delta_v, UVy));
    c->seed = pto_hash(UVx, UVy, P->time);
    parse_flags(c);
    float jitter = pt_rand(P->time) - 1.0f;
    v3 offset = v3_add(v3_scale(v3_divs(pixel_delta_u, 2.0f), jitter), v3_scale(v3_divs(pixel_delta_v, 2.0f), jitter));
    v3 ray_direction = v3_normalize(v3_add(v3_sub(pixel_center, cam_pos), offset));
    Ray ray = new_ray(cam_pos, ray_direction);

    RayHit last; int32_t primary = -1;
    v4 col = cast_ray(c, ray, &last, &primary);
    col.x = ptm_pow(col.x, P->gamma); col.y = ptm_pow(col.y, P->gamma); col.z = ptm_pow(col.z, P->gamma);

    int in_image = ux >= 0 && uy >= 0 && ux < img_w && uy < img_h;
    if (c->multisample) {
        v4 prev = V4(0, 0, 0, 0); /* imageLoad outside the image returns 0 */
        if (in_image) { const float* p = image + ((size_t)uy * img_w + ux) * 4; prev = V4(p[0], p[1], p[2], p[3]); }
        float a = 1.0f / ptm_max(1.0f, P->frame);
        float ia = 1.0f - a;
        col = V4(prev.x * ia + col.x * a, prev.y * ia + col.y * a, prev.z * ia + col.z * a, prev.w * ia + col.w * a);
    }
    if (v4_ne(c->render_mode_color, -1.0f)) col = c->render_mode_color;
    if (v4_ne(c->debug_color, -1.0f)) col = c->debug_color;
    if (in_image) { float* p = image + ((size_t)uy * img_w + ux) * 4; p[0] = col.x; p[1] = col.y; p[2] = col.z; p[3] = col.w; }

    c->st[PTO_ST_OBJ_TESTS] = c->object_hit_count;
    if (trace && in_image) {
        size_t pi = (size_t)uy * img_w + ux;
        if (trace->primary_id) trace->primary_id[pi] = primary;
        if (trace->pixel_stats) {
            for (int k = 0; k < PTO_NSTAT; k++) trace->pixel_stats[pi * PTO_NSTAT + k] = (int32_t)c->st[k];
        }
    }
}

int pto_render(const pto_scene* S, const void* push128, float* image, int img_w, int img_h,
               int groups_x, int groups_y, pto_trace* trace, int nthreads) {
    if (!S || !push128 || !image) return -1;
    if (S->order < 2 || S->order > PTO_MAX_ORDER) return -2;
    Push P; memcpy(&P, push128, 128);
    int total = groups_x * groups_y;
    int64_t totals[PTO_NSTAT]; memset(totals, 0, sizeof totals);
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel num_threads(nthreads)
    {
        Ctx* c = (Ctx*)malloc(sizeof(Ctx));
        int64_t local[PTO_NSTAT]; memset(local, 0, sizeof local);
#pragma omp for schedule(dynamic, 8)
        for (int g = 0; g < total; g++) {
            int bx = g % groups_x, by = g / groups_x;
            for (int ly = 0; ly < 8; ly++)
                for (int lx = 0; lx < 8; lx++) {
                    shade_pixel(S, &P, image, img_w, img_h, bx * 8 + lx, by * 8 + ly, trace, c);
                    for (int k = 0; k < PTO_NSTAT; k++) local[k] += c->st[k];
                }
        }
#pragma omp critical
        for (int k = 0; k < PTO_NSTAT; k++) totals[k] += local[k];
        free(c);
    }
    if (trace && trace->totals) for (int k = 0; k < PTO_NSTAT; k++) trace->totals[k] = totals[k];
    return 0;
}

/* Small probes used by the known-answer tests. */
float pto_pcg_next(float* seed) {
    Ctx c; c.seed = *seed; float v = pcg(&c); *seed = c.seed; return v;
}
float pto_noise_next(float* seed) {
    Ctx c; c.seed = *seed; float v = noise(&c); *seed = c.seed; return v;
}
float pto_rand(float seed) { return pt_rand(seed); }
void pto_math_probe(int fn, int n, const float* a, const float* b, float* out) {
    for (int i = 0; i < n; i++) {
        switch (fn) {
        case 0: out[i] = ptm_sin(a[i]); break;
        case 1: out[i] = ptm_cos(a[i]); break;
        case 2: out[i] = ptm_acos(a[i]); break;
        case 3: out[i] = ptm_atan2(a[i], b[i]); break;
        case 4: out[i] = ptm_log2(a[i]); break;
        case 5: out[i] = ptm_exp2(a[i]); break;
        case 6: out[i] = ptm_pow(a[i], b[i]); break;
        case 7: out[i] = reflectance(a[i], b[i]); break;
        default: out[i] = 0.0f;
        }
    }
}
void pto_offset_ray(const float* p, const float* n, float* out) {
    v3 r = offset_ray(V3(p[0], p[1], p[2]), V3(n[0], n[1], n[2]));
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
