// micro-benchmark: latency of a grid-wide barrier for G workgroups, and of the record exchange
// pattern of a persistent event kernel (every WG writes a record, barrier, every WG reads all)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

// variant 0: one counter, everybody spins on it
__device__ __forceinline__ void gb0(unsigned* ctr, unsigned& epoch, unsigned G) {
    __syncthreads();
    if (threadIdx.x == 0) {
        epoch += G;
        __threadfence();
        unsigned v = atomicAdd(ctr, 1u) + 1u;
        long spins = 0;
        while (v < epoch) {
            v = __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (++spins > 20000000) break;
        }
        __threadfence();
    }
    __syncthreads();
}
// variant 1: arrival counter + separate release flag (128 B apart); the last arriver releases
__device__ __forceinline__ void gb1(unsigned* ctr, unsigned& epoch, unsigned G) {
    __syncthreads();
    if (threadIdx.x == 0) {
        epoch += 1;
        __threadfence();
        unsigned v = atomicAdd(ctr, 1u) + 1u;
        if (v == epoch * G) __hip_atomic_store(ctr + 32, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        else {
            long spins = 0;
            while (__hip_atomic_load(ctr + 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch)
                if (++spins > 20000000) break;
        }
        __threadfence();
    }
    __syncthreads();
}

template <int V>
__global__ void k_bar(unsigned* ctr, int iters, double* recs, double* sink, int stride) {
    if (blockIdx.x % stride != 0) return;      // stride 8: only the workgroups of one XCD work
    const unsigned G = gridDim.x / stride, wg = blockIdx.x / stride;
    unsigned epoch = 0;
    double acc = 0;
    for (int i = 0; i < iters; i++) {
        if (threadIdx.x == 0) { recs[2 * wg] = (double)(i + wg); recs[2 * wg + 1] = 1.0; }
        if (V == 0) gb0(ctr, epoch, G); else gb1(ctr, epoch, G);
        double s = 0;
        for (unsigned k = threadIdx.x; k < G; k += blockDim.x) s += recs[2 * k] - (double)(i + k);  // must be 0
        acc += s;
        if (V == 0) gb0(ctr, epoch, G); else gb1(ctr, epoch, G);
    }
    if (acc != 0.0) atomicAdd(sink, 1.0);  // counts stale reads
}

template <int V>
void run(int G, int T, int stride, unsigned* ctr, double* recs, double* sink) {
    const int iters = 2000;
    hipMemset(ctr, 0, 512); hipMemset(sink, 0, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_bar<V>, dim3(G * stride), dim3(T), 0, 0, ctr, 10, recs, sink, stride);
    hipDeviceSynchronize();
    hipMemset(ctr, 0, 512);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_bar<V>, dim3(G * stride), dim3(T), 0, 0, ctr, iters, recs, sink, stride);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double stale = 0; hipMemcpy(&stale, sink, 8, hipMemcpyDeviceToHost);
    printf("variant %d G=%d T=%d stride=%d: %.3f us per (write, barrier, read all, barrier); stale=%g\n", V, G, T, stride, ms * 1e3 / iters, stale);
}

int main() {
    unsigned* ctr; double* sink; double* recs;
    hipMalloc(&ctr, 512); hipMalloc(&sink, 8); hipMalloc(&recs, 16 * 4096);
    for (int G : {8, 16, 32, 33, 64}) for (int stride : {1, 8}) {
        run<0>(G, 1024, stride, ctr, recs, sink);
        run<1>(G, 1024, stride, ctr, recs, sink);
    }
    run<1>(128, 256, 1, ctr, recs, sink);
    run<1>(256, 256, 1, ctr, recs, sink);
    return 0;
}
