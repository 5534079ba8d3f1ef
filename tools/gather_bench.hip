// gather_bench.hip -- how fast does one CU gather 16-byte pieces?  Every lane reads 14 pieces per
// round, as the traversal kernel's inner step does, in four address patterns:
//   own   : a lane reads the 14 pieces of its own 256-byte block (what the kernel does today)
//   quad  : the four lanes of a quad read 64 contiguous bytes per instruction, block by block
//   own8  : as `own` with 8-byte loads;  own4: with 4-byte loads
// Prints lane-accesses per clock per CU (at the measured kernel time and 2.4 GHz).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ inline uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int MODE>
__global__ __launch_bounds__(64) void gather(const float4* __restrict__ table, uint32_t n_blocks, int rounds, float* out) {
    const uint32_t tid = blockIdx.x * 64 + threadIdx.x;
    uint32_t s = mix(tid + 1);
    float acc = 0.0f;
    for (int r = 0; r < rounds; r++) {
        s = mix(s + r);
        const uint32_t blk = s % n_blocks;
        if (MODE == 0) {
            const float4* p = table + (size_t)blk * 16;
            float4 q[14];
#pragma unroll
            for (int k = 0; k < 14; k++) q[k] = p[k];
#pragma unroll
            for (int k = 0; k < 14; k++) acc += q[k].x + q[k].w;
        } else if (MODE == 1) {
            float4 q[14];
            const int lane = threadIdx.x, qd = lane & 3;
            const float4* base[4];
#pragma unroll
            for (int o = 0; o < 4; o++) base[o] = table + (size_t)__shfl(blk, (lane & ~3) | o) * 16 + qd;
#pragma unroll
            for (int k = 0; k < 14; k++) q[k] = base[k & 3][(k >> 2) * 4];
#pragma unroll
            for (int k = 0; k < 14; k++) acc += q[k].x + q[k].w;
        } else if (MODE == 2) {
            const float2* p = (const float2*)(table + (size_t)blk * 16);
            float2 q[14];
#pragma unroll
            for (int k = 0; k < 14; k++) q[k] = p[k * 2];
#pragma unroll
            for (int k = 0; k < 14; k++) acc += q[k].x + q[k].y;
        } else {
            const float* p = (const float*)(table + (size_t)blk * 16);
            float q[14];
#pragma unroll
            for (int k = 0; k < 14; k++) q[k] = p[k * 4];
#pragma unroll
            for (int k = 0; k < 14; k++) acc += q[k];
        }
    }
    out[tid] = acc;
}

int main(int argc, char** argv) {
    const size_t table_mb = argc > 1 ? atoi(argv[1]) : 4;
    const uint32_t n_blocks = (uint32_t)(table_mb * 1024 * 1024 / 256);
    const int rounds = 2000, waves = 256 * 16;
    float4* table; float* out;
    CHECK(hipMalloc(&table, (size_t)n_blocks * 256));
    CHECK(hipMemset(table, 0, (size_t)n_blocks * 256));
    CHECK(hipMalloc(&out, waves * 64 * 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const char* names[4] = {"own 16B", "quad 16B", "own 8B", "own 4B"};
    for (int mode = 0; mode < 4; mode++) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; rep++) {
            CHECK(hipEventRecord(e0));
            if (mode == 0) hipLaunchKernelGGL(gather<0>, dim3(waves), dim3(64), 0, 0, table, n_blocks, rounds, out);
            if (mode == 1) hipLaunchKernelGGL(gather<1>, dim3(waves), dim3(64), 0, 0, table, n_blocks, rounds, out);
            if (mode == 2) hipLaunchKernelGGL(gather<2>, dim3(waves), dim3(64), 0, 0, table, n_blocks, rounds, out);
            if (mode == 3) hipLaunchKernelGGL(gather<3>, dim3(waves), dim3(64), 0, 0, table, n_blocks, rounds, out);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        const double accesses = (double)waves * 64 * rounds * 14;
        printf("table %zu MB  %-9s %.3f ms  %.3f lane-accesses/clk/CU  %.2f TB/s useful\n", table_mb, names[mode], best,
               accesses / (best * 1e-3 * 2.4e9) / 256, accesses * (mode == 2 ? 8 : mode == 3 ? 4 : 16) / (best * 1e-3) / 1e12);
    }
    return 0;
}
