// ptk_bvh.h -- device BVH builder interface (implementation in ptk_bvh.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <memory>
#include <string>
#include <vector>

// One edit of ptk_bvh_refit, as the device sees it: a start node (a leaf whose object moved, or the
// socket of a mesh whose transform changed) and its payload.
struct PtkRefitEvent {
    int32_t slot;    // object-id slot of the moved object, or -1 for a mesh event
    int32_t mesh;    // mesh slot of a mesh event, else -1
    float v[12];     // new box (position, size) / new Transform3D (basis rows, origin)
};

// A tree bound for rendering and kept refittable: the flattened topology (index_tree order) on host and
// device, the raw (unpadded) node boxes and object boxes the reference keeps in its BVHNodes, and the
// transforms of the merged meshes.  Produced by PtkBvh::bind.
struct PtkRefitState {
    int n_nodes = 0, n_ids = 0, n_meshes = 0;
    std::vector<int32_t> parent, depth, kind, slot_leaf, object_ids, mesh_socket_node, mesh_index;
    // device arrays, all in flat node order unless noted
    int32_t *d_first_child = nullptr, *d_n_children = nullptr, *d_kind = nullptr, *d_slot_first = nullptr, *d_slot_count = nullptr,
            *d_mesh_of_socket = nullptr, *d_node_id = nullptr, *d_xform = nullptr, *d_rec_off = nullptr, *d_rec_cnt = nullptr;
    float *d_raw_box = nullptr, *d_obj_box = nullptr /* per id slot */, *d_xf12 = nullptr /* per mesh slot */;
    // scratch of refit(): one device block and one pinned host block, kept between calls and grown when a call needs
    // more (a call then makes one upload, one download and one wait, and allocates nothing once they are large enough)
    void* d_ws = nullptr; void* h_ws = nullptr;
    size_t d_ws_bytes = 0, h_ws_bytes = 0;
    ~PtkRefitState();
    // Runs the edits in order with the reference's update_aabb semantics (pt_bvh_tree.gd:254-302,873-882).
    // Out: flat indices of the nodes the reference would queue for upload and their new 32-byte records.
    bool refit(hipStream_t stream, const std::vector<PtkRefitEvent>& events, std::vector<int32_t>& nodes_out,
               std::vector<uint8_t>& bytes_out, std::string& err);
};

// A built tree.  Holds the node topology produced by the builder kernels and
// produces the flattened byte image on request.
struct PtkBvh {
    virtual ~PtkBvh() {}
    bool adopted = false;  // merged into a scene tree as a mesh sub-tree
    virtual int flat_node_count() const = 0;
    virtual int flat_id_count() const = 0;
    virtual int mesh_count() const = 0;  // mesh sub-trees merged into this tree
    // Writes 32 bytes per node in index_tree order and the object id table (host pointers).
    virtual bool flatten(hipStream_t stream, uint8_t* node_bytes, int32_t* object_ids, std::string& err) = 0;
    // The same image written to device memory (d_node_bytes: flat_node_count() * 32 bytes, d_object_ids:
    // flat_id_count() ints) and, optionally, to host memory as well; state (may be null) receives what a later
    // refit needs.
    virtual bool bind(hipStream_t stream, uint8_t* d_node_bytes, int32_t* d_object_ids, uint8_t* h_node_bytes, int32_t* h_object_ids,
                      PtkRefitState* state, std::string& err) = 0;
};

// mesh_trees are shared with the new tree: it keeps them alive for as long as it can be flattened, whatever
// happens to the handles they were built under.  A mesh tree must not hold mesh sockets of its own.
std::shared_ptr<PtkBvh> ptk_bvh_build_device(hipStream_t stream, int order, int type, int n_objects,
                                             const float* aabbs6, const int32_t* object_ids,
                                             const std::vector<std::shared_ptr<PtkBvh>>& mesh_trees,
                                             const float* mesh_transforms12, const int32_t* mesh_indices,
                                             std::string& err);
