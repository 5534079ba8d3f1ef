/*
 * pt_bvh_oracle.c -- CPU restatement of the reference BVH builder.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product;
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * build, load or call it.
 *
 * Restates, in plain C with Godot's float32 (real_t) semantics:
 *   proprietary_dataclasses/bvh_trees/pt_bvh_axis_sort.gd:62-84   (one node per object / mesh socket)
 *   proprietary_dataclasses/bvh_trees/pt_bvh_axis_sort.gd:156-229 (create_bvh_node_list)
 *   proprietary_dataclasses/bvh_trees/pt_bvh_axis_sort.gd:269-376 (_recursive_split_nodes)
 *   proprietary_dataclasses/bvh_trees/pt_bvh_tree.gd:457-524      (index_node / index_tree / index_subnodes)
 *   proprietary_dataclasses/bvh_trees/pt_bvh_tree.gd:780-791      (create_object_ids)
 *   proprietary_dataclasses/bvh_trees/pt_bvh_tree.gd:856-870,910-959 (set_aabb, node bytes)
 *   proprietary_dataclasses/pt_utils.gd:27-43,71-97               (is_aabb_valid, merge_aabb, aabb bytes)
 *
 * Third-party arithmetic that is NOT in /root/reference: Godot Engine 4.3
 * (project.godot config/features "4.3").  Restated here from its published
 * sources: core/math/aabb.h (merge_with, grow_by, abs, get_center,
 * get_longest_axis_index), core/math/transform_3d.cpp (xform(AABB)) and
 * core/templates/sort_array.h (SortArray: introsort with median-of-3 pivot,
 * threshold 16, heapsort fallback, final insertion sort) which is what
 * Array.sort_custom runs.
 *
 * PARITY PINNING: pinned by the reference's own vectors -- the 11 is_aabb_valid
 * and 6 merge_aabb cases of tests/unit_tests/test_utils.gd:53-106, and the three
 * hand-built trees of tests/test_helpers/ptbvhtree_setup_helper.gd:36-268 (node
 * order, counts, contiguity).  The output of PTBVHAxisSort itself (tree shape,
 * tie order of the unstable sort) is covered by no reference test: "parity
 * unpinned" for those.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "pt_bvh_oracle.h"

typedef struct { float x, y, z; } V3;
typedef struct { V3 position, size; } AABB;

#define MAX_DEPTH_LIMIT 32 /* pt_bvh_axis_sort.gd:10 */

static inline float ax(V3 v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); }
static inline V3 v3(float x, float y, float z) { V3 r = {x, y, z}; return r; }
static inline float fmin_godot(float a, float b) { return a < b ? a : b; } /* MIN(a,b) */

/* ---- Godot AABB (core/math/aabb.h), float32 ---------------------------- */
static AABB aabb_zero(void) { AABB a; memset(&a, 0, sizeof a); return a; }
static V3 aabb_end(AABB a) { return v3(a.position.x + a.size.x, a.position.y + a.size.y, a.position.z + a.size.z); }
static AABB aabb_abs(AABB a) {
    AABB r;
    r.position = v3(a.position.x + fmin_godot(a.size.x, 0.0f), a.position.y + fmin_godot(a.size.y, 0.0f),
                    a.position.z + fmin_godot(a.size.z, 0.0f));
    r.size = v3(fabsf(a.size.x), fabsf(a.size.y), fabsf(a.size.z));
    return r;
}
static int v3_eq(V3 a, V3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
static int aabb_eq(AABB a, AABB b) { return v3_eq(a.position, b.position) && v3_eq(a.size, b.size); }
static int aabb_is_finite(AABB a) {
    return isfinite(a.position.x) && isfinite(a.position.y) && isfinite(a.position.z) &&
           isfinite(a.size.x) && isfinite(a.size.y) && isfinite(a.size.z);
}
static AABB aabb_merge(AABB a, AABB b) { /* AABB::merge_with */
    V3 beg1 = a.position, beg2 = b.position;
    V3 end1 = aabb_end(a), end2 = aabb_end(b);
    V3 mn = v3(beg1.x < beg2.x ? beg1.x : beg2.x, beg1.y < beg2.y ? beg1.y : beg2.y, beg1.z < beg2.z ? beg1.z : beg2.z);
    V3 mx = v3(end1.x > end2.x ? end1.x : end2.x, end1.y > end2.y ? end1.y : end2.y, end1.z > end2.z ? end1.z : end2.z);
    AABB r; r.position = mn; r.size = v3(mx.x - mn.x, mx.y - mn.y, mx.z - mn.z);
    return r;
}
static AABB aabb_grow(AABB a, float by) { /* AABB::grow_by */
    a.position.x -= by; a.position.y -= by; a.position.z -= by;
    a.size.x += 2.0f * by; a.size.y += 2.0f * by; a.size.z += 2.0f * by;
    return a;
}
static V3 aabb_center(AABB a) { /* position + size * 0.5f */
    return v3(a.position.x + a.size.x * 0.5f, a.position.y + a.size.y * 0.5f, a.position.z + a.size.z * 0.5f);
}
static int aabb_longest_axis(AABB a) {
    int axis = 0; float m = a.size.x;
    if (a.size.y > m) { axis = 1; m = a.size.y; }
    if (a.size.z > m) { axis = 2; }
    return axis;
}
/* Transform3D::xform(AABB); t = basis rows (9 floats, row-major) + origin (3) */
static AABB xform_aabb(const float* t, AABB a) {
    float mn[3] = {a.position.x, a.position.y, a.position.z};
    V3 e = aabb_end(a);
    float mx[3] = {e.x, e.y, e.z};
    float tmin[3], tmax[3];
    for (int i = 0; i < 3; i++) {
        tmin[i] = tmax[i] = t[9 + i];
        for (int j = 0; j < 3; j++) {
            float ee = t[i * 3 + j] * mn[j];
            float ff = t[i * 3 + j] * mx[j];
            if (ee < ff) { tmin[i] += ee; tmax[i] += ff; }
            else { tmin[i] += ff; tmax[i] += ee; }
        }
    }
    AABB r; r.position = v3(tmin[0], tmin[1], tmin[2]);
    r.size = v3(tmax[0] - tmin[0], tmax[1] - tmin[1], tmax[2] - tmin[2]);
    return r;
}

/* pt_utils.gd:27-28 */
static int is_aabb_valid(AABB b) {
    int size_nonzero = !(b.size.x == 0.0f && b.size.y == 0.0f && b.size.z == 0.0f);
    return size_nonzero && aabb_eq(aabb_abs(b), b) && aabb_is_finite(b);
}
/* pt_utils.gd:34-43 */
static AABB merge_aabb(AABB a, AABB b) {
    int va = is_aabb_valid(a), vb = is_aabb_valid(b);
    if (!(va || vb)) return aabb_zero();
    if (!va) return b;
    if (!vb) return a;
    return aabb_merge(a, b);
}

int pbo_is_aabb_valid(const float* a6) {
    AABB a = {v3(a6[0], a6[1], a6[2]), v3(a6[3], a6[4], a6[5])};
    return is_aabb_valid(a);
}
void pbo_merge_aabb(const float* a6, const float* b6, float* out6) {
    AABB a = {v3(a6[0], a6[1], a6[2]), v3(a6[3], a6[4], a6[5])};
    AABB b = {v3(b6[0], b6[1], b6[2]), v3(b6[3], b6[4], b6[5])};
    AABB r = merge_aabb(a, b);
    out6[0] = r.position.x; out6[1] = r.position.y; out6[2] = r.position.z;
    out6[3] = r.size.x; out6[4] = r.size.y; out6[5] = r.size.z;
}
void pbo_xform_aabb(const float* t12, const float* a6, float* out6) {
    AABB a = {v3(a6[0], a6[1], a6[2]), v3(a6[3], a6[4], a6[5])};
    AABB r = xform_aabb(t12, a);
    out6[0] = r.position.x; out6[1] = r.position.y; out6[2] = r.position.z;
    out6[3] = r.size.x; out6[4] = r.size.y; out6[5] = r.size.z;
}

/* ---- tree ---------------------------------------------------------------- */
typedef struct Node Node;
struct pbo_tree;

struct Node {
    struct pbo_tree* tree;
    Node* parent;
    Node** children; int n_children, cap_children;
    int* objects; int n_objects, cap_objects; /* indices into tree->obj_* */
    AABB aabb;
    int is_mesh_socket;
    int is_inner;
    int index;     /* position in the flattened bvh_list */
    int oid_index; /* first slot in object_ids (leaves) */
};

struct pbo_tree {
    int order, type;
    Node* root;
    int n_objects;
    AABB* obj_aabb; int32_t* obj_id;
    Node** leaf_nodes; int n_leaf, cap_leaf;
    Node** all_nodes; int n_all, cap_all; /* ownership */
    int inner_count, leaf_count, object_count, mesh_object_count;
    /* sub-meshes of a scene tree */
    int n_meshes; struct pbo_tree** meshes;
    /* when this tree belongs to a mesh */
    float mesh_transform[12]; int mesh_index;
    struct pbo_tree* parent_tree;
    int axis;
    int error;
    int borrowed_objects;
};

static Node* node_new(struct pbo_tree* t, Node* parent) {
    Node* n = (Node*)calloc(1, sizeof(Node));
    n->tree = t; n->parent = parent; n->is_inner = 1; n->index = -1; n->oid_index = -1;
    if (t->n_all == t->cap_all) { t->cap_all = t->cap_all ? t->cap_all * 2 : 64; t->all_nodes = (Node**)realloc(t->all_nodes, sizeof(Node*) * t->cap_all); }
    t->all_nodes[t->n_all++] = n;
    return n;
}
static int node_size(const Node* n) { return n->n_children + n->n_objects; }

/* BVHNode.set_aabb, pt_bvh_tree.gd:856-870 */
static void node_set_aabb(Node* n) {
    n->aabb = aabb_zero();
    if (n->is_mesh_socket) {
        n->aabb = xform_aabb(n->children[0]->tree->mesh_transform, n->children[0]->aabb);
        return;
    }
    if (n->n_objects) {
        for (int i = 0; i < n->n_objects; i++) n->aabb = merge_aabb(n->aabb, n->tree->obj_aabb[n->objects[i]]);
        return;
    }
    if (n->n_children) {
        for (int i = 0; i < n->n_children; i++) n->aabb = merge_aabb(n->aabb, n->children[i]->aabb);
    }
}
static void node_add_child(Node* n, Node* c) { /* add_children([c]) + set_aabb, :885-895 */
    if (node_size(n) + 1 > n->tree->order) n->tree->error = PBO_ERR_NODE_OVERFULL;
    if (n->n_children == n->cap_children) { n->cap_children = n->cap_children ? n->cap_children * 2 : 8; n->children = (Node**)realloc(n->children, sizeof(Node*) * n->cap_children); }
    n->children[n->n_children++] = c;
    node_set_aabb(n);
}
static void node_add_children(Node* n, Node** cs, int k) { /* :885-891 */
    if (node_size(n) + k > n->tree->order) n->tree->error = PBO_ERR_NODE_OVERFULL;
    for (int i = 0; i < k; i++) {
        if (n->n_children == n->cap_children) { n->cap_children = n->cap_children ? n->cap_children * 2 : 8; n->children = (Node**)realloc(n->children, sizeof(Node*) * n->cap_children); }
        n->children[n->n_children++] = cs[i];
    }
    node_set_aabb(n);
}
static void node_add_object(Node* n, int obj) { /* :899-907 */
    if (n->n_objects == n->cap_objects) { n->cap_objects = n->cap_objects ? n->cap_objects * 2 : 8; n->objects = (int*)realloc(n->objects, sizeof(int) * n->cap_objects); }
    n->objects[n->n_objects++] = obj;
    node_set_aabb(n);
}
static void tree_push_leaf(struct pbo_tree* t, Node* n) {
    if (t->n_leaf == t->cap_leaf) { t->cap_leaf = t->cap_leaf ? t->cap_leaf * 2 : 64; t->leaf_nodes = (Node**)realloc(t->leaf_nodes, sizeof(Node*) * t->cap_leaf); }
    t->leaf_nodes[t->n_leaf++] = n;
}
/* index_node during creation (tree not yet scene owned, not yet a sub tree), pt_bvh_tree.gd:457-484 */
static void index_node(struct pbo_tree* t, Node* n) {
    if (n->is_inner) t->inner_count++;
    else { t->leaf_count++; t->object_count += node_size(n); tree_push_leaf(t, n); }
    node_set_aabb(n);
}

/* ---- Godot SortArray (core/templates/sort_array.h) on Node* ------------- */
typedef struct { int axis; } Cmp;
static inline int cmp_less(const Cmp* c, Node* a, Node* b) { /* pt_bvh_axis_sort.gd:180-181 */
    return ax(aabb_center(a->aabb), c->axis) > ax(aabb_center(b->aabb), c->axis);
}
static Node* median_of_3(const Cmp* c, Node* a, Node* b, Node* d) {
    if (cmp_less(c, a, b)) {
        if (cmp_less(c, b, d)) return b;
        else if (cmp_less(c, a, d)) return d;
        else return a;
    } else if (cmp_less(c, a, d)) return a;
    else if (cmp_less(c, b, d)) return d;
    else return b;
}
static int64_t bitlog(int64_t n) { int64_t k; for (k = 0; n != 1; n >>= 1) ++k; return k; }
static void push_heap(const Cmp* c, int64_t first, int64_t hole, int64_t top, Node* value, Node** a) {
    int64_t parent = (hole - 1) / 2;
    while (hole > top && cmp_less(c, a[first + parent], value)) {
        a[first + hole] = a[first + parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    a[first + hole] = value;
}
static void adjust_heap(const Cmp* c, int64_t first, int64_t hole, int64_t len, Node* value, Node** a) {
    int64_t top = hole;
    int64_t second = 2 * hole + 2;
    while (second < len) {
        if (cmp_less(c, a[first + second], a[first + (second - 1)])) second--;
        a[first + hole] = a[first + second];
        hole = second;
        second = 2 * (second + 1);
    }
    if (second == len) {
        a[first + hole] = a[first + (second - 1)];
        hole = second - 1;
    }
    push_heap(c, first, hole, top, value, a);
}
static void pop_heap4(const Cmp* c, int64_t first, int64_t last, int64_t result, Node* value, Node** a) {
    a[result] = a[first];
    adjust_heap(c, first, 0, last - first, value, a);
}
static void make_heap(const Cmp* c, int64_t first, int64_t last, Node** a) {
    if (last - first < 2) return;
    int64_t len = last - first;
    int64_t parent = (len - 2) / 2;
    for (;;) {
        adjust_heap(c, first, parent, len, a[first + parent], a);
        if (parent == 0) return;
        parent--;
    }
}
static void partial_sort(const Cmp* c, int64_t first, int64_t last, int64_t middle, Node** a) {
    make_heap(c, first, middle, a);
    for (int64_t i = middle; i < last; i++) {
        if (cmp_less(c, a[i], a[first])) pop_heap4(c, first, middle, i, a[i], a);
    }
    /* sort_heap */
    while (middle - first > 1) {
        pop_heap4(c, first, middle - 1, middle - 1, a[middle - 1], a);
        middle--;
    }
}
static int64_t partitioner(const Cmp* c, int64_t first, int64_t last, Node* pivot, Node** a) {
    for (;;) {
        while (cmp_less(c, a[first], pivot)) first++;
        last--;
        while (cmp_less(c, pivot, a[last])) last--;
        if (!(first < last)) return first;
        Node* tmp = a[first]; a[first] = a[last]; a[last] = tmp;
        first++;
    }
}
static void introsort(const Cmp* c, int64_t first, int64_t last, Node** a, int64_t max_depth) {
    while (last - first > 16) {
        if (max_depth == 0) { partial_sort(c, first, last, last, a); return; }
        max_depth--;
        int64_t cut = partitioner(c, first, last,
                                  median_of_3(c, a[first], a[first + (last - first) / 2], a[last - 1]), a);
        introsort(c, cut, last, a, max_depth);
        last = cut;
    }
}
static void unguarded_linear_insert(const Cmp* c, int64_t last, Node* value, Node** a) {
    int64_t next = last - 1;
    while (cmp_less(c, value, a[next])) { a[last] = a[next]; last = next; next--; }
    a[last] = value;
}
static void linear_insert(const Cmp* c, int64_t first, int64_t last, Node** a) {
    Node* val = a[last];
    if (cmp_less(c, val, a[first])) {
        for (int64_t i = last; i > first; i--) a[i] = a[i - 1];
        a[first] = val;
    } else unguarded_linear_insert(c, last, val, a);
}
static void insertion_sort(const Cmp* c, int64_t first, int64_t last, Node** a) {
    if (first == last) return;
    for (int64_t i = first + 1; i != last; i++) linear_insert(c, first, i, a);
}
static void godot_sort(const Cmp* c, Node** a, int64_t n) { /* SortArray::sort */
    if (n == 0) return;
    introsort(c, 0, n, a, bitlog(n) * 2);
    if (n > 16) {
        insertion_sort(c, 0, 16, a);
        for (int64_t i = 16; i != n; i++) unguarded_linear_insert(c, i, a[i], a);
    } else insertion_sort(c, 0, n, a);
}

/* Sort permutation of Godot's sort_custom for "descending key" (test probe). */
void pbo_godot_sort_desc(int n, const float* keys, int32_t* perm_out) {
    struct pbo_tree t; memset(&t, 0, sizeof t); t.order = 2;
    Node** a = (Node**)malloc(sizeof(Node*) * (n > 0 ? n : 1));
    Node* pool = (Node*)calloc(n > 0 ? n : 1, sizeof(Node));
    for (int i = 0; i < n; i++) { pool[i].aabb.position = v3(keys[i], 0, 0); pool[i].index = i; a[i] = &pool[i]; }
    Cmp c = {0};
    godot_sort(&c, a, n);
    for (int i = 0; i < n; i++) perm_out[i] = a[i]->index;
    free(a); free(pool);
}

/* ---- builder -------------------------------------------------------------- */
typedef struct { Node** v; int n, cap; } NodeVec;
static void nv_push(NodeVec* a, Node* n) {
    if (a->n == a->cap) { a->cap = a->cap ? a->cap * 2 : 8; a->v = (Node**)realloc(a->v, sizeof(Node*) * a->cap); }
    a->v[a->n++] = n;
}

/* get_array_aabb, pt_bvh_axis_sort.gd:512-516 */
static AABB get_array_aabb(Node** nodes, int n) {
    AABB a = aabb_zero();
    for (int i = 0; i < n; i++) a = merge_aabb(a, nodes[i]->aabb);
    return a;
}

/* _recursive_split_nodes, pt_bvh_axis_sort.gd:269-376.  Returns new nodes in out. */
static void recursive_split_nodes(struct pbo_tree* t, Node** node_list, int n, Node* parent, int depth, NodeVec* out) {
    const int order = t->order;
    int segment_count = order;
    AABB list_aabb = aabb_grow(aabb_abs(get_array_aabb(node_list, n)), 0.000001f);

    if (t->type == PBO_XYZ_SORTED) {
        t->axis = aabb_longest_axis(list_aabb);
        Cmp c = {t->axis};
        godot_sort(&c, node_list, n);
    }
    const int axis = t->axis;

    int mesh_count = 0;
    for (int i = 0; i < n; i++) if (node_list[i]->is_mesh_socket) mesh_count++;

    if (depth >= MAX_DEPTH_LIMIT) {
        if (mesh_count >= order) { /* warning only */ }
        else if (mesh_count > 0) segment_count = order - mesh_count;
    }
    if (depth >= MAX_DEPTH_LIMIT + 5) { t->error = PBO_ERR_TOO_DEEP; return; }

    NodeVec* bins = (NodeVec*)calloc(segment_count, sizeof(NodeVec));
    /* (size / segment_count)[axis]: Vector3 / real_t in float32, then widened to double */
    double segment = (double)(ax(list_aabb.size, axis) / (float)segment_count);
    for (int i = 0; i < n; i++) {
        V3 center = aabb_center(node_list[i]->aabb);
        float rel = ax(center, axis) - ax(list_aabb.position, axis);
        double q = floor((double)rel / segment);
        if (!(q >= 0.0 && q < (double)segment_count)) { t->error = PBO_ERR_BIN_RANGE; q = 0.0; }
        nv_push(&bins[(int)q], node_list[i]);
    }

    int n_bins = segment_count;
    int empty_sections = 0;
    NodeVec* first_non_empty = NULL;
    for (int b = 0; b < segment_count; b++) {
        if (bins[b].n == 0) empty_sections++;
        else first_non_empty = &bins[b];
    }
    NodeVec halves[2]; memset(halves, 0, sizeof halves);
    NodeVec* use = bins;
    if (empty_sections == segment_count - 1 && first_non_empty) {
        int half = first_non_empty->n / 2;
        for (int i = 0; i < half; i++) nv_push(&halves[0], first_non_empty->v[i]);
        for (int i = half; i < first_non_empty->n; i++) nv_push(&halves[1], first_non_empty->v[i]);
        use = halves; n_bins = 2;
    }

    NodeVec new_nodes; memset(&new_nodes, 0, sizeof new_nodes);
    for (int b = 0; b < n_bins && !t->error; b++) {
        NodeVec* nodes = &use[b];
        if (nodes->n == 0) continue;
        Node* new_node = node_new(t, parent);
        new_node->is_inner = 1;
        if (nodes->n > order && depth < MAX_DEPTH_LIMIT) {
            NodeVec sub; memset(&sub, 0, sizeof sub);
            recursive_split_nodes(t, nodes->v, nodes->n, new_node, depth + 1, &sub);
            node_add_children(new_node, sub.v, sub.n);
            free(sub.v);
            nv_push(&new_nodes, new_node);
            continue;
        }
        if (mesh_count == 0) {
            new_node->is_inner = 0;
            for (int i = 0; i < nodes->n; i++) node_add_object(new_node, nodes->v[i]->objects[0]);
        } else {
            Node* new_leaf = node_new(t, new_node);
            new_leaf->is_inner = 0;
            for (int i = 0; i < nodes->n; i++) {
                Node* nd = nodes->v[i];
                if (!nd->is_inner) node_add_object(new_leaf, nd->objects[0]);
                else {
                    nd->parent = new_node;
                    node_add_child(new_node, nd);
                    index_node(t, nd);
                }
            }
            if (node_size(new_leaf) > 0) {
                index_node(t, new_leaf);
                node_add_child(new_node, new_leaf);
            }
        }
        nv_push(&new_nodes, new_node);
    }
    for (int i = 0; i < new_nodes.n; i++) index_node(t, new_nodes.v[i]);
    for (int i = 0; i < new_nodes.n; i++) nv_push(out, new_nodes.v[i]);
    free(new_nodes.v);
    for (int b = 0; b < segment_count; b++) free(bins[b].v);
    free(bins); free(halves[0].v); free(halves[1].v);
}

/* ---- binned SAH (PBO_SAH) ------------------------------------------------------------------------
 * No counterpart in the reference (bvh_trees/pt_bvh_sweep_sah.gd is an empty class); this restates the rule
 * of libptk's PTK_BVH_SAH builder so that its output can be compared byte for byte.  Same node format, same
 * indexing, same leaf / socket conventions as the reference builder above; only the choice of which
 * elements share a node differs.
 *
 * One wide node (up to 8 children) is made from a list by up to three rounds of binary splits: every part
 * with more than 8 elements is split in two, parts with 8 or fewer are left alone.  A binary split looks at
 * the elements' box centres c = (lo + hi) * 0.5 (lo = position, hi = position + size): axis = longest side
 * of the centres' bounding box (x, then y, then z on ties), 16 equal bins along it, bin =
 * trunc((c - cmin) * (16 / extent)) clamped to 0..15; every bin keeps a count and the bounds of its
 * elements' boxes; the split plane after bin i (0..14) with the smallest area(L) * nL + area(R) * nR wins
 * (first on ties; area = 2 * ((dx * dy + dy * dz) + dz * dx)); the elements keep their order on both sides.
 * If the centres do not spread (extent <= 0, or 16 / extent not finite) or every element falls into one bin,
 * the list is cut in the middle.  Minima and maxima use the total order of the float bit patterns
 * (-0 < +0), so they do not depend on the order of evaluation. */
static uint32_t ord_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
static float ord_min(float a, float b) { return ord_bits(b) < ord_bits(a) ? b : a; }
static float ord_max(float a, float b) { return ord_bits(b) > ord_bits(a) ? b : a; }
typedef struct { float lo[3], hi[3]; int n; } SahBin;
static float sah_area(const float* lo, const float* hi) {
    float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    return 2.0f * ((dx * dy + dy * dz) + dz * dx);
}
static void elem_bounds(const Node* nd, float* lo, float* hi) {
    lo[0] = nd->aabb.position.x; lo[1] = nd->aabb.position.y; lo[2] = nd->aabb.position.z;
    hi[0] = nd->aabb.position.x + nd->aabb.size.x; hi[1] = nd->aabb.position.y + nd->aabb.size.y; hi[2] = nd->aabb.position.z + nd->aabb.size.z;
}
/* splits list[0..n) in place (stable); returns the size of the left part, or 0 when a list of 8 or fewer is
 * better left whole (no plane with area(L) * nL + area(R) * nR < area(all) * (n - 1), or no spread) */
static int sah_binary_split(Node** list, int n, int order) {
    float cmin[3], cmax[3];
    for (int i = 0; i < n; i++) {
        float lo[3], hi[3];
        elem_bounds(list[i], lo, hi);
        for (int a = 0; a < 3; a++) {
            float c = (lo[a] + hi[a]) * 0.5f;
            cmin[a] = i ? ord_min(cmin[a], c) : c;
            cmax[a] = i ? ord_max(cmax[a], c) : c;
        }
    }
    float e[3] = {cmax[0] - cmin[0], cmax[1] - cmin[1], cmax[2] - cmin[2]};
    int axis = 0; float m = e[0];
    if (e[1] > m) { axis = 1; m = e[1]; }
    if (e[2] > m) { axis = 2; m = e[2]; }
    int n_left = -1;
    float k = 16.0f / m;
    unsigned char* side = (unsigned char*)malloc(n > 0 ? n : 1);
    if (m > 0.0f && k < INFINITY) {
        SahBin bins[16]; memset(bins, 0, sizeof bins);
        unsigned char* bin_of = (unsigned char*)malloc(n);
        for (int i = 0; i < n; i++) {
            float lo[3], hi[3];
            elem_bounds(list[i], lo, hi);
            float q = ((lo[axis] + hi[axis]) * 0.5f - cmin[axis]) * k;
            int b = q >= 15.0f ? 15 : (q > 0.0f ? (int)q : 0);
            bin_of[i] = (unsigned char)b;
            SahBin* B = &bins[b];
            for (int a = 0; a < 3; a++) {
                B->lo[a] = B->n ? ord_min(B->lo[a], lo[a]) : lo[a];
                B->hi[a] = B->n ? ord_max(B->hi[a], hi[a]) : hi[a];
            }
            B->n++;
        }
        int best = -1; float best_cost = 0.0f;
        for (int p = 0; p < 15; p++) {
            SahBin L, R; memset(&L, 0, sizeof L); memset(&R, 0, sizeof R);
            for (int b = 0; b < 16; b++) {
                SahBin* D = b <= p ? &L : &R;
                if (!bins[b].n) continue;
                for (int a = 0; a < 3; a++) {
                    D->lo[a] = D->n ? ord_min(D->lo[a], bins[b].lo[a]) : bins[b].lo[a];
                    D->hi[a] = D->n ? ord_max(D->hi[a], bins[b].hi[a]) : bins[b].hi[a];
                }
                D->n += bins[b].n;
            }
            if (!L.n || !R.n) continue;
            float cost = sah_area(L.lo, L.hi) * (float)L.n + sah_area(R.lo, R.hi) * (float)R.n;
            if (best < 0 || cost < best_cost) { best = p; best_cost = cost; }
        }
        if (best >= 0 && n <= order) {  /* small enough for one leaf: split only if that pays for one more box test */
            SahBin P; memset(&P, 0, sizeof P);
            for (int b = 0; b < 16; b++) {
                if (!bins[b].n) continue;
                for (int a = 0; a < 3; a++) {
                    P.lo[a] = P.n ? ord_min(P.lo[a], bins[b].lo[a]) : bins[b].lo[a];
                    P.hi[a] = P.n ? ord_max(P.hi[a], bins[b].hi[a]) : bins[b].hi[a];
                }
                P.n += bins[b].n;
            }
            if (!(best_cost < sah_area(P.lo, P.hi) * (float)(n - 1))) best = -2;
        }
        if (best == -2) n_left = 0;
        if (best >= 0) {
            n_left = 0;
            for (int i = 0; i < n; i++) { side[i] = bin_of[i] > best; if (!side[i]) n_left++; }
        }
        free(bin_of);
    }
    if (n_left < 0) {  /* no spread: cut in the middle (a list that fits one leaf stays whole) */
        n_left = n <= order ? 0 : n / 2;
        for (int i = 0; i < n; i++) side[i] = i >= n_left;
    }
    if (n_left == 0) { free(side); return 0; }
    Node** tmp = (Node**)malloc(sizeof(Node*) * n);
    int l = 0, r = n_left;
    for (int i = 0; i < n; i++) { if (side[i]) tmp[r++] = list[i]; else tmp[l++] = list[i]; }
    memcpy(list, tmp, sizeof(Node*) * n);
    free(tmp); free(side);
    return n_left;
}
/* the children of one wide node over node_list; recursion, node kinds and index_node order as in
 * recursive_split_nodes above */
static void sah_split_nodes(struct pbo_tree* t, Node** node_list, int n, Node* parent, NodeVec* out) {
    const int order = t->order;
    int mesh_count = 0;
    for (int i = 0; i < n; i++) if (node_list[i]->is_mesh_socket) mesh_count++;
    int begin[9], parts = 1;
    begin[0] = 0; begin[1] = n;
    for (int round = 0; round < 3; round++) {
        int nb[9], np = 0;
        for (int s = 0; s < parts; s++) {
            int b = begin[s], e = begin[s + 1];
            nb[np++] = b;
            if (e - b >= 2) { int nl = sah_binary_split(node_list + b, e - b, order); if (nl > 0) nb[np++] = b + nl; }
        }
        nb[np] = n;
        memcpy(begin, nb, sizeof(int) * (np + 1));
        parts = np;
    }
    NodeVec new_nodes; memset(&new_nodes, 0, sizeof new_nodes);
    for (int s = 0; s < parts && !t->error; s++) {
        Node** nodes = node_list + begin[s];
        int cnt = begin[s + 1] - begin[s];
        if (cnt == 0) continue;
        Node* new_node = node_new(t, parent);
        new_node->is_inner = 1;
        if (cnt > order) {
            NodeVec sub; memset(&sub, 0, sizeof sub);
            sah_split_nodes(t, nodes, cnt, new_node, &sub);
            node_add_children(new_node, sub.v, sub.n);
            free(sub.v);
            nv_push(&new_nodes, new_node);
            continue;
        }
        if (mesh_count == 0) {
            new_node->is_inner = 0;
            for (int i = 0; i < cnt; i++) node_add_object(new_node, nodes[i]->objects[0]);
        } else {
            Node* new_leaf = node_new(t, new_node);
            new_leaf->is_inner = 0;
            for (int i = 0; i < cnt; i++) {
                Node* nd = nodes[i];
                if (!nd->is_inner) node_add_object(new_leaf, nd->objects[0]);
                else { nd->parent = new_node; node_add_child(new_node, nd); index_node(t, nd); }
            }
            if (node_size(new_leaf) > 0) { index_node(t, new_leaf); node_add_child(new_node, new_leaf); }
        }
        nv_push(&new_nodes, new_node);
    }
    for (int i = 0; i < new_nodes.n; i++) index_node(t, new_nodes.v[i]);
    for (int i = 0; i < new_nodes.n; i++) nv_push(out, new_nodes.v[i]);
    free(new_nodes.v);
}

pbo_tree* pbo_build(int order, int type, int n_objects, const float* aabbs6, const int32_t* object_ids,
                    int n_meshes, pbo_tree** mesh_trees, const float* mesh_transforms12, const int32_t* mesh_indices) {
    if (order < 2 || order > 128) return NULL;
    if (type < PBO_X_SORTED || type > PBO_SAH) return NULL;
    if (type == PBO_SAH && order != 8) return NULL; /* the SAH rule is defined for 8-wide nodes */
    if (n_objects + n_meshes <= 0) return NULL; /* "Cannot create bvh without objects or meshes." */
    struct pbo_tree* t = (struct pbo_tree*)calloc(1, sizeof *t);
    t->order = order; t->type = type; t->n_objects = n_objects;
    t->obj_aabb = (AABB*)malloc(sizeof(AABB) * (n_objects > 0 ? n_objects : 1));
    t->obj_id = (int32_t*)malloc(sizeof(int32_t) * (n_objects > 0 ? n_objects : 1));
    for (int i = 0; i < n_objects; i++) {
        t->obj_aabb[i].position = v3(aabbs6[i * 6 + 0], aabbs6[i * 6 + 1], aabbs6[i * 6 + 2]);
        t->obj_aabb[i].size = v3(aabbs6[i * 6 + 3], aabbs6[i * 6 + 4], aabbs6[i * 6 + 5]);
        t->obj_id[i] = object_ids[i];
    }
    t->root = node_new(t, NULL); /* PTBVHTree._init: root indexed, inner_count = 1 */
    t->inner_count = 1;
    t->n_meshes = n_meshes;
    t->meshes = (struct pbo_tree**)calloc(n_meshes > 0 ? n_meshes : 1, sizeof(*t->meshes));

    /* create_disconnected_node_list, pt_bvh_axis_sort.gd:62-84 */
    int n = n_objects + n_meshes;
    Node** list = (Node**)malloc(sizeof(Node*) * n);
    for (int i = 0; i < n_objects; i++) {
        Node* nd = node_new(t, NULL);
        nd->is_inner = 0;
        node_add_object(nd, i);
        list[i] = nd;
    }
    for (int m = 0; m < n_meshes; m++) {
        struct pbo_tree* mt = mesh_trees[m];
        t->meshes[m] = mt;
        memcpy(mt->mesh_transform, mesh_transforms12 + m * 12, sizeof(float) * 12);
        mt->mesh_index = mesh_indices[m];
        Node* nd = node_new(t, NULL);
        nd->is_mesh_socket = 1; nd->is_inner = 1;
        mt->root->parent = nd;
        node_add_child(nd, mt->root);
        list[n_objects + m] = nd;
    }
    /* create_bvh_node_list, pt_bvh_axis_sort.gd:183-202 */
    t->axis = 0;
    if (type == PBO_X_SORTED) t->axis = 0;
    if (type == PBO_Y_SORTED) t->axis = 1;
    if (type == PBO_Z_SORTED) t->axis = 2;
    if (type != PBO_XYZ_SORTED && type != PBO_SAH) { Cmp c = {t->axis}; godot_sort(&c, list, n); }
    NodeVec top; memset(&top, 0, sizeof top);
    if (type == PBO_SAH) sah_split_nodes(t, list, n, t->root, &top);
    else recursive_split_nodes(t, list, n, t->root, 0, &top);
    if (!t->error) node_add_children(t->root, top.v, top.n);
    free(top.v); free(list);
    /* :206-222 */
    for (int m = 0; m < n_meshes; m++) {
        struct pbo_tree* mt = t->meshes[m];
        for (int i = 0; i < mt->n_leaf; i++) tree_push_leaf(t, mt->leaf_nodes[i]);
        t->leaf_count += mt->leaf_count;
        t->inner_count += mt->inner_count;
        t->mesh_object_count += mt->object_count;
        mt->parent_tree = t;
    }
    /* create_object_ids (set_scene_as_owner), pt_bvh_tree.gd:780-791: slot order = leaf_nodes order */
    int idx = 0;
    for (int i = 0; i < t->n_leaf; i++) {
        t->leaf_nodes[i]->oid_index = idx;
        idx += t->leaf_nodes[i]->n_objects;
    }
    return t;
}

int pbo_error(const pbo_tree* t) { return t ? t->error : PBO_ERR_NULL; }

/* index_tree + index_subnodes, pt_bvh_tree.gd:488-524 */
static void count_subnodes(const Node* n, int* count) {
    *count += n->n_children;
    for (int i = 0; i < n->n_children; i++) if (n->children[i]->is_inner) count_subnodes(n->children[i], count);
}
int pbo_node_count(const pbo_tree* t) { int c = 1; count_subnodes(t->root, &c); return c; }
int pbo_object_id_count(const pbo_tree* t) { return t->object_count + t->mesh_object_count; }
void pbo_counts(const pbo_tree* t, int32_t* out4) {
    out4[0] = t->inner_count; out4[1] = t->leaf_count; out4[2] = t->object_count; out4[3] = t->mesh_object_count;
}

static void index_subnodes(Node* n, Node** list, int* k) {
    for (int i = 0; i < n->n_children; i++) { n->children[i]->index = *k; list[(*k)++] = n->children[i]; }
    for (int i = 0; i < n->n_children; i++) if (n->children[i]->is_inner) index_subnodes(n->children[i], list, k);
}

/* pt_utils.gd:71-97 + pt_bvh_tree.gd:910-959 */
static void node_bytes(const pbo_tree* scene, const Node* n, uint8_t* out32) {
    if (n == scene->root && node_size(n) == 0) { memset(out32, 0, 32); return; }
    int64_t node_index = 0, nsize = 0;
    if (n->is_inner) { node_index = n->children[0]->index; nsize = n->n_children; }
    else { node_index = -(int64_t)n->oid_index - 1; nsize = n->n_objects - 1; }
    node_index ^= nsize << 24;
    int32_t id32 = (int32_t)(uint32_t)(uint64_t)node_index;
    int32_t xf = -1;
    if (n->is_mesh_socket) xf = n->children[0]->tree->mesh_index;
    AABB a = aabb_abs(n->aabb);
    int zero = (n->aabb.size.x == 0.0f && n->aabb.size.y == 0.0f && n->aabb.size.z == 0.0f);
    float pad = zero ? 0.0f : 1e-6f;
    V3 e = aabb_end(a);
    float f[8];
    f[0] = a.position.x - pad; f[1] = a.position.y - pad; f[2] = a.position.z - pad;
    f[4] = e.x + pad; f[5] = e.y + pad; f[6] = e.z + pad;
    memcpy(out32, f, 32);
    memcpy(out32 + 12, &id32, 4);
    memcpy(out32 + 28, &xf, 4);
}

int pbo_flatten(pbo_tree* t, uint8_t* node_bytes_out, int32_t* object_ids_out) {
    if (!t || t->error) return t ? t->error : PBO_ERR_NULL;
    int n = pbo_node_count(t);
    Node** list = (Node**)malloc(sizeof(Node*) * n);
    int k = 0;
    t->root->index = 0; list[k++] = t->root;
    node_set_aabb(t->root); /* index_node(root_node) re-sets the root aabb */
    index_subnodes(t->root, list, &k);
    for (int i = 0; i < n; i++) node_bytes(t, list[i], node_bytes_out + (size_t)i * 32);
    for (int i = 0; i < t->n_leaf; i++) {
        const Node* lf = t->leaf_nodes[i];
        for (int j = 0; j < lf->n_objects; j++) object_ids_out[lf->oid_index + j] = lf->tree->obj_id[lf->objects[j]];
    }
    free(list);
    return 0;
}

/* ---- incremental refit ----------------------------------------------------
 * BVHNode.update_aabb (pt_bvh_tree.gd:873-882), reached from PTBVHTree.update_aabb(object) (:254-260) and
 * update_mesh_socket_aabb(mesh) (:263-270): recompute the node's box; if it is not is_equal_approx to the old
 * one, queue the node's index for upload (:295-302) and update the parent.  The upload later writes
 * to_byte_array() of every queued node (pt_renderer.gd:771-788), i.e. the box the node has by then.
 * Godot's Math::is_equal_approx(a, b) (core/math/math_funcs.h): equal, or |a - b| below max(1e-5 * |a|, 1e-5);
 * AABB::is_equal_approx compares position and size component-wise.  Not part of /root/reference: restated. */
static int approx1(float a, float b) {
    if (a == b) return 1;
    float tol = 0.00001f * fabsf(a);
    if (tol < 0.00001f) tol = 0.00001f;
    return fabsf(a - b) < tol;
}
static int aabb_equal_approx(AABB a, AABB b) {
    return approx1(a.position.x, b.position.x) && approx1(a.position.y, b.position.y) && approx1(a.position.z, b.position.z) &&
           approx1(a.size.x, b.size.x) && approx1(a.size.y, b.size.y) && approx1(a.size.z, b.size.z);
}
static void node_update_aabb(Node* n, uint8_t* queued_flag, int32_t* queued, int* n_queued) {
    AABB old = n->aabb;
    node_set_aabb(n);
    if (!aabb_equal_approx(n->aabb, old)) {
        if (!queued_flag[n->index]) { queued_flag[n->index] = 1; queued[(*n_queued)++] = n->index; }
        if (n->parent) node_update_aabb(n->parent, queued_flag, queued, n_queued);
    }
}
typedef struct { int32_t id; int tree; int idx; } ObjKey;
static int objkey_cmp(const void* a, const void* b) {
    int32_t x = ((const ObjKey*)a)->id, y = ((const ObjKey*)b)->id;
    return x < y ? -1 : (x > y ? 1 : 0);
}
/* events: object_ids[j] = an object's id with v12[j*12 .. +6) its new box, or mesh index | 4 << 24 with
 * v12[j*12 .. +12) the mesh's new transform.  node_bytes (the image pbo_flatten wrote, flattened before this
 * call) gets the records of the queued nodes rewritten; queued_out (room for pbo_node_count ints) their
 * indices in queue order.  Returns the number of queued nodes, or -1 for an unknown object / mesh. */
int pbo_refit(pbo_tree* t, int n_events, const int32_t* object_ids, const float* v12, uint8_t* node_bytes_io, int32_t* queued_out) {
    int n_trees = 1 + t->n_meshes;
    pbo_tree** trees = (pbo_tree**)malloc(sizeof(*trees) * n_trees);
    trees[0] = t;
    for (int m = 0; m < t->n_meshes; m++) trees[1 + m] = t->meshes[m];
    size_t total = 0;
    for (int k = 0; k < n_trees; k++) total += trees[k]->n_objects;
    ObjKey* keys = (ObjKey*)malloc(sizeof(ObjKey) * (total ? total : 1));
    Node*** leaf_of = (Node***)malloc(sizeof(Node**) * n_trees);
    size_t w = 0;
    for (int k = 0; k < n_trees; k++) {
        pbo_tree* tr = trees[k];
        leaf_of[k] = (Node**)calloc(tr->n_objects ? tr->n_objects : 1, sizeof(Node*));
        for (int i = 0; i < tr->n_objects; i++) { keys[w].id = tr->obj_id[i]; keys[w].tree = k; keys[w].idx = i; w++; }
        for (int i = 0; i < tr->n_all; i++) {
            Node* nd = tr->all_nodes[i];
            for (int j = 0; j < nd->n_objects; j++) leaf_of[k][nd->objects[j]] = nd;
        }
    }
    qsort(keys, total, sizeof(ObjKey), objkey_cmp);
    int n_nodes = pbo_node_count(t);
    uint8_t* flag = (uint8_t*)calloc(n_nodes, 1);
    Node** by_index = (Node**)calloc(n_nodes, sizeof(Node*));
    for (int k = 0; k < n_trees; k++)
        for (int i = 0; i < trees[k]->n_all; i++) { Node* nd = trees[k]->all_nodes[i]; if (nd->index >= 0 && nd->index < n_nodes) by_index[nd->index] = nd; }
    int n_queued = 0, rc = 0;
    for (int j = 0; j < n_events && rc == 0; j++) {
        int32_t id = object_ids[j];
        const float* v = v12 + (size_t)j * 12;
        if (((uint32_t)id >> 24) == 4u) {
            int mi = id & 0x00FFFFFF;
            pbo_tree* mt = NULL;
            for (int m = 0; m < t->n_meshes; m++) if (t->meshes[m]->mesh_index == mi) mt = t->meshes[m];
            if (!mt) { rc = -1; break; }
            memcpy(mt->mesh_transform, v, sizeof(float) * 12);
            node_update_aabb(mt->root->parent, flag, queued_out, &n_queued);
        } else {
            ObjKey probe; probe.id = id;
            ObjKey* hit = (ObjKey*)bsearch(&probe, keys, total, sizeof(ObjKey), objkey_cmp);
            if (!hit || !leaf_of[hit->tree][hit->idx]) { rc = -1; break; }
            AABB nb = {v3(v[0], v[1], v[2]), v3(v[3], v[4], v[5])};
            trees[hit->tree]->obj_aabb[hit->idx] = nb;
            node_update_aabb(leaf_of[hit->tree][hit->idx], flag, queued_out, &n_queued);
        }
    }
    if (rc == 0) for (int q = 0; q < n_queued; q++) node_bytes(t, by_index[queued_out[q]], node_bytes_io + (size_t)queued_out[q] * 32);
    for (int k = 0; k < n_trees; k++) free(leaf_of[k]);
    free(leaf_of); free(keys); free(flag); free(by_index); free(trees);
    return rc ? rc : n_queued;
}

void pbo_free(pbo_tree* t) {
    if (!t) return;
    for (int m = 0; m < t->n_meshes; m++) pbo_free(t->meshes[m]);
    for (int i = 0; i < t->n_all; i++) { free(t->all_nodes[i]->children); free(t->all_nodes[i]->objects); free(t->all_nodes[i]); }
    free(t->all_nodes); free(t->leaf_nodes);
    if (!t->borrowed_objects) { free(t->obj_aabb); free(t->obj_id); }
    free(t->meshes); free(t);
}

/* ---- hand-assembled trees (for the reference's fixture trees) -------------
 * A tree is described top-down: kinds[i] in {0 inner, 1 leaf, 2 mesh socket},
 * parents[i] (-1 for the root, which is node 0), and for leaves a list of
 * object ids.  Children attach to their parent in array order.  Flattening and
 * byte encoding then run through the same code as built trees. */
pbo_tree* pbo_assemble(int order, int n_nodes, const int32_t* kinds, const int32_t* parents,
                       const int32_t* leaf_first, const int32_t* leaf_count,
                       int n_objects, const float* aabbs6, const int32_t* object_ids,
                       const int32_t* leaf_order, int n_leaf_order,
                       const float* socket_transforms12, const int32_t* socket_mesh_index) {
    struct pbo_tree* t = (struct pbo_tree*)calloc(1, sizeof *t);
    t->order = order; t->type = PBO_X_SORTED; t->n_objects = n_objects;
    t->obj_aabb = (AABB*)malloc(sizeof(AABB) * (n_objects > 0 ? n_objects : 1));
    t->obj_id = (int32_t*)malloc(sizeof(int32_t) * (n_objects > 0 ? n_objects : 1));
    for (int i = 0; i < n_objects; i++) {
        t->obj_aabb[i].position = v3(aabbs6[i * 6 + 0], aabbs6[i * 6 + 1], aabbs6[i * 6 + 2]);
        t->obj_aabb[i].size = v3(aabbs6[i * 6 + 3], aabbs6[i * 6 + 4], aabbs6[i * 6 + 5]);
        t->obj_id[i] = object_ids[i];
    }
    t->meshes = (struct pbo_tree**)calloc(1, sizeof(*t->meshes));
    Node** nd = (Node**)calloc(n_nodes, sizeof(Node*));
    /* Sub-meshes are modelled as pseudo trees carrying only transform + index. */
    for (int i = 0; i < n_nodes; i++) {
        nd[i] = node_new(t, parents[i] >= 0 ? nd[parents[i]] : NULL);
        nd[i]->is_inner = kinds[i] != 1;
        nd[i]->is_mesh_socket = kinds[i] == 2;
        if (i == 0) t->root = nd[0];
    }
    /* leaves first (objects), then attach children bottom-up so set_aabb sees final boxes */
    for (int i = 0; i < n_nodes; i++)
        if (kinds[i] == 1) for (int j = 0; j < leaf_count[i]; j++) node_add_object(nd[i], leaf_first[i] + j);
    for (int i = n_nodes - 1; i >= 1; i--) {
        Node* p = nd[parents[i]];
        if (p->is_mesh_socket) {
            struct pbo_tree* pseudo = (struct pbo_tree*)calloc(1, sizeof *pseudo);
            pseudo->order = order;
            memcpy(pseudo->mesh_transform, socket_transforms12 + parents[i] * 12, sizeof(float) * 12);
            pseudo->mesh_index = socket_mesh_index[parents[i]];
            pseudo->obj_aabb = t->obj_aabb; pseudo->obj_id = t->obj_id; pseudo->borrowed_objects = 1;
            pseudo->parent_tree = t;
            /* the child's tree pointer selects transform/index; objects resolve through borrowed tables */
            nd[i]->tree = pseudo;
            t->meshes = (struct pbo_tree**)realloc(t->meshes, sizeof(*t->meshes) * (t->n_meshes + 1));
            t->meshes[t->n_meshes++] = pseudo;
        }
    }
    /* attach children in array order: iterate parents ascending, children ascending */
    for (int i = n_nodes - 1; i >= 0; i--) {
        /* collect children of i in ascending order */
        for (int j = 0; j < n_nodes; j++) if (parents[j] == i) {
            if (nd[i]->n_children == nd[i]->cap_children) { nd[i]->cap_children = nd[i]->cap_children ? nd[i]->cap_children * 2 : 8; nd[i]->children = (Node**)realloc(nd[i]->children, sizeof(Node*) * nd[i]->cap_children); }
            nd[i]->children[nd[i]->n_children++] = nd[j];
        }
        if (nd[i]->is_inner) node_set_aabb(nd[i]);
    }
    for (int i = 0; i < n_nodes; i++) {
        if (kinds[i] == 1) { t->leaf_count++; } else t->inner_count++;
    }
    int idx = 0;
    for (int i = 0; i < n_leaf_order; i++) {
        Node* lf = nd[leaf_order[i]];
        tree_push_leaf(t, lf);
        lf->oid_index = idx; idx += lf->n_objects;
        t->object_count += lf->n_objects;
    }
    free(nd);
    return t;
}
