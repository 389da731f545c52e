// Development aid: rocBLAS dgemm rates at the shapes of the split-weight solver's appends (MI355X).
// build: hipcc --offload-arch=gfx950 -O2 -o gemm_shapes gemm_shapes.hip -lrocblas
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <cstdio>
#include <cstdlib>
__global__ void k_fillv(double* v, int64_t cnt) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < cnt) v[i] = 1.0 + (double)(i % 17) * 0.01;
}
#define CK(x) do { auto e_ = (x); if ((int)e_ != 0) { printf("fail %s = %d line %d\n", #x, (int)e_, __LINE__); exit(1); } } while (0)
int main() {
    rocblas_handle h; CK(rocblas_create_handle(&h)); hipStream_t s; CK(hipStreamCreate(&s)); CK(rocblas_set_stream(h, s));
    const double one = 1.0, zero = 0.0;
    const int64_t K = 60000, ldw = 65536;
    double *W, *T, *X;
    CK(hipMalloc(&W, sizeof(double) * ldw * 8192)); CK(hipMalloc(&T, sizeof(double) * ldw * 4096)); CK(hipMalloc(&X, sizeof(double) * ldw * 4096));
    hipLaunchKernelGGL(k_fillv, dim3((ldw * 8192 + 255) / 256), dim3(256), 0, s, W, ldw * 8192);
    hipLaunchKernelGGL(k_fillv, dim3((ldw * 4096 + 255) / 256), dim3(256), 0, s, T, ldw * 4096);
    CK(hipStreamSynchronize(s));
    auto run = [&](const char* name, rocblas_operation ta, rocblas_operation tb, int64_t m, int64_t n, int64_t k, const double* A, int64_t lda, const double* B, int64_t ldb,
                   double* C, int64_t ldc) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        CK(rocblas_dgemm_64(h, ta, tb, m, n, k, &one, A, lda, B, ldb, &zero, C, ldc));
        hipEventRecord(a, s);
        for (int i = 0; i < 3; i++) CK(rocblas_dgemm_64(h, ta, tb, m, n, k, &one, A, lda, B, ldb, &zero, C, ldc));
        hipEventRecord(b, s); hipEventSynchronize(b);
        float ms = 0; hipEventElapsedTime(&ms, a, b);
        printf("%-58s m=%ld n=%ld k=%ld: %.4f s  %.1f TF/s\n", name, (long)m, (long)n, (long)k, ms * 1e-3 / 3, 2.0 * m * n * k / (ms * 1e-3 / 3) * 1e-12);
        fflush(stdout);
    };
    for (int64_t k : {64, 256, 1024, 4096}) {
        run("T = Wpanel * B (NN, rows of a panel)", rocblas_operation_none, rocblas_operation_none, 8192, k, K, W, ldw, T, ldw, X, ldw);
        run("X = T^T * Wpanel (TN, M = k)", rocblas_operation_transpose, rocblas_operation_none, k, 8192, K, T, ldw, W, ldw, X, 4096);
        run("X^T = Wpanel^T * T (TN, M = panel)", rocblas_operation_transpose, rocblas_operation_none, 8192, k, K, W, ldw, T, ldw, X, ldw);
        run("rows = Li * X (NN, k x f x k)", rocblas_operation_none, rocblas_operation_none, k, K, k, T, ldw, X, 4096, W, ldw);
        run("rows = Li * (X^T)^T (NT, k x f x k)", rocblas_operation_none, rocblas_operation_transpose, k, K, k, T, ldw, X, ldw, W, ldw);
    }
    return 0;
}
