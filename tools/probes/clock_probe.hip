// Shader clock seen by a single wave on an otherwise idle GPU: cycles (clock64) per 100 MHz wall tick
// (wall_clock64) and cycles per dependent v_add.  hipcc --offload-arch=gfx950 clock_probe.hip -o clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(float* out, long long* t, int n) {
    long long c0 = clock64(), w0 = wall_clock64();
    float x = out[0];
    for (int i = 0; i < n; i += 32) {
#pragma unroll
        for (int k = 0; k < 32; k++) x = x + 1.0f;
    }
    long long c1 = clock64(), w1 = wall_clock64();
    out[0] = x;
    if (threadIdx.x == 0) { t[0] = c1 - c0; t[1] = w1 - w0; }
}
int main() {
    float* d; long long* t; long long h[2];
    hipMalloc(&d, 4); hipMalloc(&t, 16); hipMemset(d, 0, 4);
    for (int n : {1000, 100000, 1000000, 10000000}) {
        probe<<<1, 64>>>(d, t, n);
        hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
        printf("n=%d clock64 %lld wall(100MHz) %lld -> %.0f MHz, %.2f cycles per add, %.3f ns per add\n", n, h[0], h[1],
               100.0 * h[0] / h[1], (double)h[0] / n, 10.0 * h[1] / n);
    }
    return 0;
}
