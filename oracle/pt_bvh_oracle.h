/*
 * pt_bvh_oracle.h -- interface of the CPU BVH-builder oracle (TEST
 * INFRASTRUCTURE ONLY; see pt_bvh_oracle.c).
 */
#ifndef PT_BVH_ORACLE_H
#define PT_BVH_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* PTBVHTree.BVHType, pt_bvh_tree.gd:40-46 */
enum { PBO_EMPTY = 0, PBO_X_SORTED = 1, PBO_Y_SORTED = 2, PBO_Z_SORTED = 3, PBO_XYZ_SORTED = 4,
       PBO_SAH = 5 /* not a reference type: the binned-SAH rule libptk adds (pt_bvh_sweep_sah.gd is an empty stub) */ };

enum {
    PBO_OK = 0,
    PBO_ERR_NULL = -1,
    PBO_ERR_NODE_OVERFULL = -2, /* add_children assert: more children than order */
    PBO_ERR_TOO_DEEP = -3,      /* "too much depth" assert */
    PBO_ERR_BIN_RANGE = -4      /* bin index assert 0 <= index < segment_count */
};

typedef struct pbo_tree pbo_tree;

/* aabbs6: per object position.xyz,size.xyz (object.get_bvh_aabb()), in the
 * order create_disconnected_node_list visits objects (spheres, then triangles).
 * mesh_trees are adopted by the returned tree.  mesh_transforms12: Transform3D
 * of each PTMesh as basis rows (9) + origin (3). */
pbo_tree* pbo_build(int order, int type, int n_objects, const float* aabbs6, const int32_t* object_ids,
                    int n_meshes, pbo_tree** mesh_trees, const float* mesh_transforms12, const int32_t* mesh_indices);
pbo_tree* pbo_assemble(int order, int n_nodes, const int32_t* kinds, const int32_t* parents,
                       const int32_t* leaf_first, const int32_t* leaf_count,
                       int n_objects, const float* aabbs6, const int32_t* object_ids,
                       const int32_t* leaf_order, int n_leaf_order,
                       const float* socket_transforms12, const int32_t* socket_mesh_index);
int pbo_error(const pbo_tree* t);
int pbo_node_count(const pbo_tree* t);
int pbo_object_id_count(const pbo_tree* t);
void pbo_counts(const pbo_tree* t, int32_t* out4); /* inner, leaf, object, mesh_object */
int pbo_flatten(pbo_tree* t, uint8_t* node_bytes_out, int32_t* object_ids_out);
/* The reference's incremental refit (update_aabb chains) on a flattened tree; see pt_bvh_oracle.c. */
int pbo_refit(pbo_tree* t, int n_events, const int32_t* object_ids, const float* v12, uint8_t* node_bytes_io, int32_t* queued_out);
void pbo_free(pbo_tree* t);

int pbo_is_aabb_valid(const float* a6);
void pbo_merge_aabb(const float* a6, const float* b6, float* out6);
void pbo_xform_aabb(const float* t12, const float* a6, float* out6);
void pbo_godot_sort_desc(int n, const float* keys, int32_t* perm_out);

#ifdef __cplusplus
}
#endif
#endif
