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
    // Writes 32 bytes per node in index_tree order and the object id table (host pointers).
    virtual bool flatten(hipStream_t stream, uint8_t* node_bytes, int32_t* object_ids, std::string& err) = 0;
};

std::unique_ptr<PtkBvh> ptk_bvh_build_device(hipStream_t stream, int order, int type, int n_objects,
                                             const float* aabbs6, const int32_t* object_ids,
                                             const std::vector<PtkBvh*>& mesh_trees,
                                             const float* mesh_transforms12, const int32_t* mesh_indices,
                                             std::string& err);
