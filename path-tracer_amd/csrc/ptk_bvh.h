// ptk_bvh.h -- device BVH builder interface (implementation in ptk_bvh.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <memory>
#include <string>
#include <vector>

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
};

// mesh_trees are shared with the new tree: it keeps them alive for as long as it can be flattened, whatever
// happens to the handles they were built under.  A mesh tree must not hold mesh sockets of its own.
std::shared_ptr<PtkBvh> ptk_bvh_build_device(hipStream_t stream, int order, int type, int n_objects,
                                             const float* aabbs6, const int32_t* object_ids,
                                             const std::vector<std::shared_ptr<PtkBvh>>& mesh_trees,
                                             const float* mesh_transforms12, const int32_t* mesh_indices,
                                             std::string& err);
