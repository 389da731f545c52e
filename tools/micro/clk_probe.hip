// micro-probe: effective shader clock seen by (a) one busy wave on an otherwise idle GPU, (b) the same
// with all CUs kept busy by other workgroups.  clock64() counts shader-engine cycles, wall_clock64()
// a constant 100 MHz.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_probe(long long* out, int heavy_blocks, int iters) {
    double a = threadIdx.x * 1e-9 + 1.0;
    if (blockIdx.x == 0) {
        long long c0 = clock64(), w0 = wall_clock64();
        for (int i = 0; i < iters; i++) a = a * 1.0000001 + 1e-12;   // dependent fp64 chain
        long long c1 = clock64(), w1 = wall_clock64();
        if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; }
    } else {
        for (int i = 0; i < iters * 4; i++) a = a * 1.0000001 + 1e-12;
    }
    if (a == 123.0) out[2] = 1;
}
int main() {
    long long* d; hipMalloc(&d, 64);
    for (int blocks : {1, 257, 2049}) for (int rep = 0; rep < 2; rep++) {
        hipMemset(d, 0, 64);
        hipLaunchKernelGGL(k_probe, dim3(blocks), dim3(64), 0, 0, d, blocks, 200000);
        hipDeviceSynchronize();
        long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("blocks=%d: %lld core cycles in %.1f us -> %.0f MHz; %.2f cycles per dependent fma\n", blocks, h[0], h[1] / 100.0,
               h[0] / (h[1] / 100.0), (double)h[0] / 200000);
    }
    return 0;
}
