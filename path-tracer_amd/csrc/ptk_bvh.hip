// ptk_bvh.hip -- device BVH builder (placeholder until the builder kernels land).
#include "ptk_bvh.h"

std::unique_ptr<PtkBvh> ptk_bvh_build_device(hipStream_t, int, int, int, const float*, const int32_t*,
                                             const std::vector<PtkBvh*>&, const float*, const int32_t*,
                                             std::string& err) {
    err = "the device BVH builder is not part of this build";
    return nullptr;
}
