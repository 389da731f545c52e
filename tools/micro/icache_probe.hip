// Cost of cold instruction fetch for short kernels: a kernel of N KB of straight-line code (one wave per CU), launched
// (a) back to back with itself (its code stays in the 64 KB instruction cache) and (b) alternating with a second large
// kernel that evicts it, as the event chain's k_track / k_update alternate.  hipcc --offload-arch=gfx950 -O2 icache_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define R256(x) R16(R16(x))
template <int REP, int SALT>
__global__ void k_code(double* p, double a, double b) {
    double x = p[threadIdx.x], y = a, z = b;
#pragma unroll
    for (int r = 0; r < REP; r++) {
        // 256 x 3 dependent-free-ish fp64 ops = 768 instructions ~ 6 KB per repetition
        R256(x = x * y + z; y = y + (double)SALT; z = z - x;)
    }
    p[threadIdx.x] = x + y + z;
}
template <class F, class G>
static float run(F f, G g, int iters, bool alternate) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; i++) { f(); if (alternate) g(); }
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int i = 0; i < iters; i++) { f(); if (alternate) g(); }
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / iters;
}
int main() {
    double* p; hipMalloc(&p, 8 * 256 * 512); hipMemset(p, 0, 8 * 256 * 512);
    const dim3 g(130), b(256);
    auto small1 = [&]() { hipLaunchKernelGGL((k_code<1, 1>), g, b, 0, 0, p, 1.0, 2.0); };
    auto mid4 = [&]() { hipLaunchKernelGGL((k_code<4, 2>), g, b, 0, 0, p, 1.0, 2.0); };
    auto big12 = [&]() { hipLaunchKernelGGL((k_code<12, 3>), g, b, 0, 0, p, 1.0, 2.0); };
    auto evict = [&]() { hipLaunchKernelGGL((k_code<24, 4>), g, b, 0, 0, p, 1.0, 2.0); };
    auto empty = [&]() {};
    float e = run(evict, empty, 200, false);
    printf("evictor alone (24 reps, ~150 KB of code): %.2f us per launch\n", e);
    struct { const char* name; float alone, alt; } rows[3];
    rows[0] = {"1 rep  (~6 KB)", run(small1, empty, 500, false), run(small1, evict, 200, true) - e};
    rows[1] = {"4 reps (~25 KB)", run(mid4, empty, 500, false), run(mid4, evict, 200, true) - e};
    rows[2] = {"12 reps (~75 KB)", run(big12, empty, 500, false), run(big12, evict, 200, true) - e};
    for (auto& r : rows) printf("%-18s warm %.2f us   after the evictor %.2f us\n", r.name, r.alone, r.alt);
    return 0;
}
