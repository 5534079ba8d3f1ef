/*
 * pt_oracle.h -- interface of the CPU oracle (TEST INFRASTRUCTURE ONLY; see
 * pt_oracle.c).  Plain C, loaded from Python tests through ctypes.
 */
#ifndef PT_ORACLE_H
#define PT_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PTO_MAX_ORDER 128

/* built-in procedural textures (shaders/procedural_textures/, two .comp files) */
#define PTO_PROC_CHECKER_BOARD 1
#define PTO_PROC_CHECKER_BOARD_SPATIAL 2

/* per-pixel / total statistics used for the algorithmic byte count */
enum {
    PTO_ST_RAYS = 0,       /* path segments = check_ray_hit_BVH calls */
    PTO_ST_BOX_TESTS,      /* hit_AABB calls (root + children) */
    PTO_ST_OBJ_TESTS,      /* hit_object calls */
    PTO_ST_SPHERE_TESTS,
    PTO_ST_TRI_TESTS,
    PTO_ST_SOCKETS,        /* mesh-socket transforms fetched */
    PTO_ST_HIT_SPHERE,     /* resolved hits by type */
    PTO_ST_HIT_PLANE,
    PTO_ST_HIT_TRIANGLE,
    PTO_ST_MATERIAL_FETCH,
    PTO_NSTAT
};

typedef struct {
    int width, height;
    const float* rgba; /* width*height*4 floats, level 0; NULL = filler texture */
} pto_texture;

typedef struct {
    const uint8_t* materials;  int n_materials;  /* 64-byte records */
    const uint8_t* spheres;    int n_spheres;    /* 32-byte records, buffer length() */
    const uint8_t* planes;     int n_planes;     /* 32-byte records, buffer length() */
    const uint8_t* triangles;  int n_triangles;  /* 80-byte records, floor(bytes/80) */
    const uint8_t* bvh;        int n_nodes;      /* 32-byte nodes */
    const int32_t* object_ids; int n_object_ids;
    const uint8_t* transforms; int n_transforms; /* 64-byte column-major mat4 */
    const pto_texture* textures; int n_textures;
    int order;            /* shader constant `order` */
    int refraction_mode;  /* use_refraction_tracking: 0, 1 or 2 */
    int use_emission;     /* USE_EMISSION */
    int early_t_break;    /* use_early_t_break */
    int n_proc;           /* number of spliced procedural functions */
    const int32_t* proc_builtin; /* function index i+1 -> PTO_PROC_* */
    /* Extension beyond the reference (its TODO at ray_tracer.comp:1071-1072): when non-NULL, six floats per
     * triangle record -- (u,v) of point1, point2, point3 -- and a triangle hit's uv becomes their barycentric
     * interpolation instead of the barycentric coordinates themselves.  NULL = the reference's behaviour. */
    const float* triangle_uvs; int n_triangle_uvs;
} pto_scene;

typedef struct {
    int32_t* primary_id;   /* per pixel: object_id of the first segment's hit, -1 on miss */
    int32_t* pixel_stats;  /* per pixel: PTO_NSTAT counters */
    int64_t* totals;       /* PTO_NSTAT totals over the dispatch */
} pto_trace;

/* One dispatch of (groups_x, groups_y, 1) work groups of 8x8 invocations.
 * image is the rgba32f storage image (read-modify-write). */
int pto_render(const pto_scene* scene, const void* push128, float* image, int image_w, int image_h,
               int groups_x, int groups_y, pto_trace* trace, int nthreads);

float pto_hash(float x, float y, float z);
float pto_pcg_next(float* seed);
float pto_noise_next(float* seed);
float pto_rand(float seed);
void pto_math_probe(int fn, int n, const float* a, const float* b, float* out);
void pto_offset_ray(const float* p, const float* n, float* out);

#ifdef __cplusplus
}
#endif
#endif
