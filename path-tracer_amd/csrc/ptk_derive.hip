// ptk_derive.hip -- the traversal structures of a bound scene, derived and checked on the device.
//
// Before the first dispatch over a new scene the library checks every index the kernels can reach through the bound
// node buffer (pt_bvh_tree.gd:910-959 encodes it, ray_tracer.comp:1115-1181, 1267-1317 walk it), finds the deepest the
// traversal stack can get, and compiles the inner nodes into the 256-byte blocks the block traversal reads
// (PtkWavefront::blocks).  For a large tree that walk is milliseconds of host time; here it runs level by level on the
// GPU over the buffers where they lie:
//
//   k_dv_level    one launch per tree level: every node of the level is checked, its children appended to the next
//                 level (the levels, one after the other, are the tree in breadth-first order)
//   k_dv_count / k_dv_tops / k_dv_number   a scan over that order numbers the inner nodes (their blocks) and the
//                 world-level mesh sockets: the top of the tree gets the first numbers, which is what the traversal
//                 kernel keeps in LDS
//   k_dv_fill     writes the blocks, the socket table and the node -> child-slot map
//   k_dv_records  material indices of all primitive records; k_dv_slots: which id slot names a primitive
//
// A failed check only raises PtkDeriveInfo::error: the host then walks its own copy of the same bytes to say what is
// wrong where.  Nothing here reads through an index before it has been compared with its array's length.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ptk_device.h"

namespace {

constexpr int DV_BLOCK = 1024;
constexpr uint32_t DV_SMALL = 4096;  // a level of at most this many nodes is run by one workgroup, in order, so that the
                                     // numbering of the top of the tree does not depend on timing

__device__ inline int dv_sext(int v, int off, int bits) { return ((int)((uint32_t)v << (32 - off - bits))) >> (32 - bits); }

// exclusive prefix sum over the workgroup's DV_BLOCK threads; *total = the sum.  s_wave: 16 words of LDS.
__device__ uint32_t dv_block_scan(uint32_t v, uint32_t* total, uint32_t* s_wave) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t y = __shfl_up(x, d, 64);
        if (lane >= d) x += y;
    }
    __syncthreads();  // (s_wave may still be read from the call before)
    if (lane == 63) s_wave[wave] = x;
    __syncthreads();
    if (wave == 0) {
        uint32_t t = lane < DV_BLOCK / 64 ? s_wave[lane] : 0u;
#pragma unroll
        for (int d = 1; d < DV_BLOCK / 64; d <<= 1) {
            const uint32_t y = __shfl_up(t, d, 64);
            if (lane >= d) t += y;
        }
        if (lane < DV_BLOCK / 64) s_wave[lane] = t;
    }
    __syncthreads();
    const uint32_t base = wave ? s_wave[wave - 1] : 0u;
    *total = s_wave[DV_BLOCK / 64 - 1];
    return base + x - v;
}

struct DvNode {
    int id, idx, size, x;
};
__device__ inline DvNode dv_node(const PtkDerive& D, uint32_t i) {
    DvNode n;
    n.id = D.nodes[(size_t)i * 8 + 3];
    n.x = D.nodes[(size_t)i * 8 + 7];
    n.idx = dv_sext(n.id, 0, 24);
    const int s = dv_sext(n.id, 24, 8);
    n.size = s < 0 ? -s : s;
    return n;
}

}  // namespace

__global__ __launch_bounds__(64) void k_dv_init(PtkDerive D) {
    if (threadIdx.x == 0) {
        PtkDeriveInfo z = {};
        z.total = 1; z.begin = 0; z.end = 1;
        *D.info = z;
        D.ent_node[0] = 0u;
        D.ent_aux[0] = 0u;
    }
}

// One level of the tree: entries [begin, end) of the level list.  Checks of the host's walker, in its words:
// a node is reachable once; its id is in the encoding PTBVHTree writes; its transform index is in range and, on a leaf,
// absent; a leaf's id slots lie inside the id table and name primitives inside their buffers; an inner node has at
// most PTK_MAX_WIDTH children, inside the node buffer.
__global__ __launch_bounds__(DV_BLOCK) void k_dv_level(PtkDerive D) {
    __shared__ uint32_t s_wave[DV_BLOCK / 64];
    __shared__ uint32_t s_base, s_max, s_blocks;
    __shared__ bool s_last;
    PtkDeriveInfo* const I = D.info;
    if (I->finished) return;  // (set by the last workgroup of an earlier launch only: every workgroup sees the same)
    const uint32_t begin = I->begin, end = I->end, depth = I->level;
    const uint32_t n = end - begin;
    const bool small = n <= DV_SMALL;
    if (threadIdx.x == 0) { s_max = 0u; s_blocks = 0u; }
    __syncthreads();
    if (!small || blockIdx.x == 0) {
        const uint32_t first = small ? 0u : blockIdx.x * (uint32_t)DV_BLOCK;
        const uint32_t stride = small ? (uint32_t)DV_BLOCK : gridDim.x * (uint32_t)DV_BLOCK;
        for (uint32_t k0 = first; k0 < n; k0 += stride) {
            const uint32_t k = k0 + threadIdx.x;
            uint32_t n_children = 0, child_first = 0, child_mesh = 0, child_below = 0;
            bool is_block = false;
            if (k < n) {
                const uint32_t e = D.ent_node[begin + k];
                const uint32_t i = e & 0x7FFFFFFFu;
                const bool in_mesh = (e >> 31) != 0u;
                const uint32_t below = D.ent_aux[begin + k] & 0xFFFFu;
                bool bad = false;
                const uint32_t bit = 1u << (i & 31u);
                if (atomicOr(D.visited + (i >> 5), bit) & bit) bad = true;  // reachable twice: not a tree
                const DvNode nd = dv_node(D, i);
                if ((nd.idx > 0 && nd.id <= 0) || (nd.idx < 0 && nd.id >= 0)) bad = true;
                if (nd.x >= D.n_xf) bad = true;
                if (nd.x >= 0 && nd.idx < 0) bad = true;
                if (nd.x >= 0 && in_mesh) I->nested_socket = 1u;
                if (!bad) {
                    if (nd.idx <= 0) {
                        const int base = (nd.idx < 0 ? -nd.idx : nd.idx) - 1;
                        if (nd.size > 0 && (base < 0 || base + nd.size > D.n_ids)) bad = true;
                        else {
                            for (int j = 0; j < nd.size; j++) {
                                const int oid = D.ids[base + j];
                                const int type = dv_sext(oid, 24, 8), oi = dv_sext(oid, 0, 24);
                                if (type == 1 && (oi < 0 || oi >= D.n_sph)) bad = true;
                                if (type == 3 && (oi < 0 || oi >= D.n_tri)) bad = true;
                            }
                        }
                    } else if (nd.size > PTK_MAX_WIDTH || nd.idx + nd.size > D.n_nodes || below + (uint32_t)nd.size > 4096u) {
                        bad = true;  // (a stack of thousands of entries: the host reports how many the tree can need)
                    } else {
                        n_children = (uint32_t)nd.size;
                        child_first = (uint32_t)nd.idx;
                        child_mesh = (in_mesh || nd.x >= 0) ? 0x80000000u : 0u;
                        child_below = below + (uint32_t)nd.size - 1u;
                        atomicMax(&s_max, below + (uint32_t)nd.size);
                        if (nd.x >= 0 && !in_mesh) atomicAdd(&I->n_sockets, 1u);  // (a handful per scene)
                        else is_block = true;
                    }
                }
                if (bad) I->error = 1u;
            }
            const uint32_t nb = (uint32_t)__syncthreads_count(is_block ? 1 : 0);
            uint32_t tot;
            const uint32_t off = dv_block_scan(n_children, &tot, s_wave);
            if (threadIdx.x == 0) {
                s_base = tot ? atomicAdd(&I->total, tot) : 0u;
                s_blocks += nb;
            }
            __syncthreads();
            if (n_children) {
                const uint32_t pos = s_base + off;
                if (pos + n_children > (uint32_t)D.n_nodes) {
                    I->error = 1u;  // more entries than nodes: some node is listed by two parents
                } else {
                    for (uint32_t l = 0; l < n_children; l++) {
                        D.ent_node[pos + l] = (child_first + l) | child_mesh;
                        D.ent_aux[pos + l] = child_below | ((depth + 1u) << 16);
                    }
                }
            }
            __syncthreads();
        }
    }
    if (threadIdx.x == 0) {
        if (s_max) atomicMax(&I->stack_max, s_max);
        if (s_blocks) atomicAdd(&I->n_blocks, s_blocks);
    }
    // the workgroup that is through last closes the level
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) s_last = atomicAdd(&I->done_blocks, 1u) == gridDim.x - 1u;
    __syncthreads();
    if (s_last && threadIdx.x == 0) {
        __threadfence();
        uint32_t tot = atomicAdd(&I->total, 0u);
        if (tot > (uint32_t)D.n_nodes) tot = (uint32_t)D.n_nodes;  // (an error is on record; entries past the list were not written)
        I->begin = end;
        I->end = tot;
        I->level = depth + 1u;
        I->done_blocks = 0u;
        if (tot == end || atomicAdd(&I->error, 0u) != 0u) I->finished = 1u;
        else if (depth + 1u > 512u) { I->error = 1u; I->finished = 1u; }  // deeper than 512 levels
    }
}

namespace {
// what entry p of the level list contributes to the numbering: .x a block (inner node that is not a world-level mesh
// socket), .y a world-level socket
__device__ inline uint2 dv_flags(const PtkDerive& D, uint32_t p) {
    const uint32_t e = D.ent_node[p];
    const DvNode nd = dv_node(D, e & 0x7FFFFFFFu);
    if (nd.idx <= 0) return make_uint2(0u, 0u);
    const bool sock = nd.x >= 0 && (e >> 31) == 0u;
    return sock ? make_uint2(0u, 1u) : make_uint2(1u, 0u);
}
}  // namespace

__global__ __launch_bounds__(DV_BLOCK) void k_dv_count(PtkDerive D) {
    __shared__ uint32_t s_wave[DV_BLOCK / 64];
    const uint32_t total = D.info->end;  // (the list's length once the levels are through)
    const uint32_t p = blockIdx.x * (uint32_t)DV_BLOCK + threadIdx.x;
    uint2 f = make_uint2(0u, 0u);
    if (p < total) f = dv_flags(D, p);
    uint32_t tb, ts;
    dv_block_scan(f.x, &tb, s_wave);
    dv_block_scan(f.y, &ts, s_wave);
    if (threadIdx.x == 0) D.partial[blockIdx.x] = make_uint2(tb, ts);
}

__global__ __launch_bounds__(DV_BLOCK) void k_dv_tops(PtkDerive D, uint32_t n_parts) {  // one workgroup
    __shared__ uint32_t s_wave[DV_BLOCK / 64];
    uint32_t carry_b = 0, carry_s = 0;
    for (uint32_t k0 = 0; k0 < n_parts; k0 += DV_BLOCK) {
        const uint32_t k = k0 + threadIdx.x;
        const uint2 v = k < n_parts ? D.partial[k] : make_uint2(0u, 0u);
        uint32_t tb, ts;
        const uint32_t ob = dv_block_scan(v.x, &tb, s_wave);
        const uint32_t os = dv_block_scan(v.y, &ts, s_wave);
        if (k < n_parts) D.partial[k] = make_uint2(carry_b + ob, carry_s + os);
        carry_b += tb;
        carry_s += ts;
    }
}

__global__ __launch_bounds__(DV_BLOCK) void k_dv_number(PtkDerive D) {
    __shared__ uint32_t s_wave[DV_BLOCK / 64];
    const uint32_t total = D.info->end;
    const uint32_t p = blockIdx.x * (uint32_t)DV_BLOCK + threadIdx.x;
    uint2 f = make_uint2(0u, 0u);
    if (p < total) f = dv_flags(D, p);
    uint32_t tb, ts;
    const uint32_t ob = dv_block_scan(f.x, &tb, s_wave);
    const uint32_t os = dv_block_scan(f.y, &ts, s_wave);
    const uint2 base = D.partial[blockIdx.x];
    if (p < total) {
        const uint32_t i = D.ent_node[p] & 0x7FFFFFFFu;
        if (f.x) D.node_blk[i] = (int32_t)(base.x + ob);
        if (f.y) D.node_sock[i] = (int32_t)(base.y + os);
    }
}

namespace {
// what the traversal visits for child node c (PtkWavefront::blocks): a leaf's id word, a world-level socket's mark,
// or the child's block and child count
__device__ inline uint32_t dv_entry(const PtkDerive& D, uint32_t c, bool c_in_mesh) {
    const DvNode nd = dv_node(D, c);
    if (nd.idx <= 0) return (uint32_t)nd.id;
    if (nd.x >= 0 && !c_in_mesh) return 0x7F000000u | (uint32_t)D.node_sock[c];
    return (uint32_t)D.node_blk[c] | ((uint32_t)nd.size << 24);
}
}  // namespace

// 64 threads per entry of the level list, one per word of its block (if it has one); sockets fill their table row.
__global__ __launch_bounds__(256) void k_dv_fill(PtkDerive D, float* __restrict__ blocks, int32_t* __restrict__ sockets) {
    const uint32_t total = D.info->end;
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    const uint32_t p = t >> 6, w = t & 63u;
    if (p >= total) return;
    const uint32_t e = D.ent_node[p];
    const uint32_t i = e & 0x7FFFFFFFu;
    const bool in_mesh = (e >> 31) != 0u;
    const DvNode nd = dv_node(D, i);
    if (nd.idx <= 0) return;
    if (nd.x >= 0 && !in_mesh) {
        if (w == 0) {
            const int32_t s = D.node_sock[i];
            sockets[(size_t)s * 2] = (int32_t)dv_entry(D, (uint32_t)nd.idx, true);
            sockets[(size_t)s * 2 + 1] = nd.x;
        }
        return;
    }
    const int32_t b = D.node_blk[i];
    const uint32_t plane = w >> 3, l = w & 7u;
    uint32_t v = 0u;
    if (plane < 7u && (int)l < nd.size) {
        const uint32_t c = (uint32_t)nd.idx + l;
        if (plane < 6u) {
            const int word_of[6] = {0, 1, 2, 4, 5, 6};
            v = (uint32_t)D.nodes[(size_t)c * 8 + word_of[plane]];
        } else {
            v = dv_entry(D, c, in_mesh || nd.x >= 0);
            D.node_slot[c] = b * 8 + (int32_t)l;
        }
    } else if (w == 56u) {
        v = D.ent_aux[p] >> 16;  // first spare word: the node's level (measurement builds read it)
    }
    ((uint32_t*)blocks)[(size_t)b * 64 + w] = v;
}

// Material indices of every primitive record (validation refuses a record the kernels could resolve with an index
// outside the material buffer), and the materials of the planes that can be hit.
__global__ __launch_bounds__(256) void k_dv_records(PtkDerive D) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    int m = 0;
    bool mark = false;
    if (t < D.n_sph) {
        m = __float_as_int(D.spheres[(size_t)t * 2 + 1].x);
    } else if (t < D.n_sph + D.n_pln) {
        const int i = t - D.n_sph;
        const float4 pl = D.planes[(size_t)i * 2];
        m = __float_as_int(D.planes[(size_t)i * 2 + 1].x);
        mark = !(pl.x == 0.0f && pl.y == 0.0f && pl.z == 0.0f);
    } else if (t < D.n_sph + D.n_pln + D.n_tri) {
        const int i = t - D.n_sph - D.n_pln;
        m = __float_as_int(D.triangles[(size_t)i * 5 + 4].x);
    } else {
        return;
    }
    if (m < 0 || m >= D.n_mats) D.info->error = 1u;
    else if (mark) D.mat_used[m] = 1;
}

// Which id slot names a primitive (-1 none, -2 several), which materials the listed primitives use.
__global__ __launch_bounds__(256) void k_dv_slots(PtkDerive D) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= D.n_ids) return;
    const int oid = D.ids[i];
    const int type = dv_sext(oid, 24, 8), oi = dv_sext(oid, 0, 24);
    int m = -1;
    if (type == 3 && oi >= 0 && oi < D.n_tri) {
        D.info->has_tri = 1u;
        if (atomicCAS(D.tri_slot + oi, -1, i) != -1) D.tri_slot[oi] = -2;
        m = __float_as_int(D.triangles[(size_t)oi * 5 + 4].x);
    } else if (type == 1 && oi >= 0 && oi < D.n_sph) {
        if (atomicCAS(D.sph_slot + oi, -1, i) != -1) D.sph_slot[oi] = -2;
        m = __float_as_int(D.spheres[(size_t)oi * 2 + 1].x);
    }
    if (m >= 0 && m < D.n_mats) D.mat_used[m] = 1;
}

extern "C" hipError_t ptk_launch_derive_begin(const PtkDerive* d, hipStream_t stream) {
    hipLaunchKernelGGL(k_dv_init, dim3(1), dim3(64), 0, stream, *d);
    if (d->n_sph + d->n_pln + d->n_tri > 0)
        hipLaunchKernelGGL(k_dv_records, dim3((unsigned)((d->n_sph + d->n_pln + d->n_tri + 255) / 256)), dim3(256), 0, stream, *d);
    if (d->n_ids > 0) hipLaunchKernelGGL(k_dv_slots, dim3((unsigned)((d->n_ids + 255) / 256)), dim3(256), 0, stream, *d);
    return hipGetLastError();
}

extern "C" hipError_t ptk_launch_derive_levels(const PtkDerive* d, int n_levels, hipStream_t stream) {
    unsigned grid = (unsigned)((d->n_nodes + DV_BLOCK - 1) / DV_BLOCK);
    grid = grid < 1u ? 1u : (grid > 256u ? 256u : grid);
    for (int k = 0; k < n_levels; k++) hipLaunchKernelGGL(k_dv_level, dim3(grid), dim3(DV_BLOCK), 0, stream, *d);
    return hipGetLastError();
}

extern "C" hipError_t ptk_launch_derive_number(const PtkDerive* d, uint32_t n_reached, hipStream_t stream) {
    const uint32_t n_parts = (n_reached + DV_BLOCK - 1) / DV_BLOCK;
    if (n_parts == 0) return hipSuccess;
    hipLaunchKernelGGL(k_dv_count, dim3(n_parts), dim3(DV_BLOCK), 0, stream, *d);
    hipLaunchKernelGGL(k_dv_tops, dim3(1), dim3(DV_BLOCK), 0, stream, *d, n_parts);
    hipLaunchKernelGGL(k_dv_number, dim3(n_parts), dim3(DV_BLOCK), 0, stream, *d);
    return hipGetLastError();
}

extern "C" hipError_t ptk_launch_derive_fill(const PtkDerive* d, uint32_t n_reached, float* blocks, int32_t* sockets, hipStream_t stream) {
    if (n_reached == 0) return hipSuccess;
    const size_t threads = (size_t)n_reached * 64;
    hipLaunchKernelGGL(k_dv_fill, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, *d, blocks, sockets);
    return hipGetLastError();
}
