// Development aid: rates of the rocBLAS / rocSOLVER fp64 routines the split-weight solver leans on (MI355X).
// build: hipcc --offload-arch=gfx950 -O2 -o blas_bench blas_bench.hip -lrocblas -lrocsolver ; run: ./blas_bench [f ...]
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <rocsolver/rocsolver.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void k_fill(double* A, int64_t f, int64_t ld) {  // SPD: diagonally dominant, smooth off-diagonal
    const int64_t c = blockIdx.y, r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= f) return;
    A[c * ld + r] = (r == c) ? (double)f : 1.0 / (1.0 + (double)(r > c ? r - c : c - r));
}
__global__ void k_fillv(double* v, int64_t cnt) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < cnt) v[i] = 1.0 + (double)(i % 17) * 0.01;
}
#define CK(x) do { auto e_ = (x); if ((int)e_ != 0) { printf("fail %s = %d line %d\n", #x, (int)e_, __LINE__); exit(1); } } while (0)
template <class F> double timeit(hipStream_t s, F f, int reps = 1) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a, s); for (int i = 0; i < reps; i++) f(); hipEventRecord(b, s); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b); hipEventDestroy(a); hipEventDestroy(b); return ms * 1e-3 / reps;
}
int main(int argc, char** argv) {
    std::vector<int64_t> fs; for (int i = 1; i < argc; i++) fs.push_back(atoll(argv[i]));
    if (fs.empty()) fs = {8192, 32768};
    rocblas_handle h; CK(rocblas_create_handle(&h)); hipStream_t s; CK(hipStreamCreate(&s)); CK(rocblas_set_stream(h, s));
    const double one = 1.0, mone = -1.0, zero = 0.0;
    for (int64_t f : fs) {
        const int64_t ld = f, k = 512;
        double *A, *B, *C, *v, *u; int64_t* info;
        CK(hipMalloc(&A, sizeof(double) * ld * f)); CK(hipMalloc(&B, sizeof(double) * f * k)); CK(hipMalloc(&C, sizeof(double) * f * k));
        CK(hipMalloc(&v, sizeof(double) * f)); CK(hipMalloc(&u, sizeof(double) * f)); CK(hipMalloc(&info, 8));
        hipLaunchKernelGGL(k_fill, dim3((f + 255) / 256, f), dim3(256), 0, s, A, f, ld);
        hipLaunchKernelGGL(k_fillv, dim3((f * k + 255) / 256), dim3(256), 0, s, B, f * k);
        hipLaunchKernelGGL(k_fillv, dim3((f + 255) / 256), dim3(256), 0, s, v, f);
        CK(hipStreamSynchronize(s));
        double t;
        t = timeit(s, [&] { CK(rocblas_dgemm_64(h, rocblas_operation_none, rocblas_operation_none, f, k, f, &one, A, ld, B, f, &zero, C, f)); }, 2);
        printf("f=%ld dgemm  f x f x %ld: %.4f s  %.1f TF/s\n", (long)f, (long)k, t, 2.0 * f * f * k / t * 1e-12);
        t = timeit(s, [&] { CK(rocblas_dgemm_64(h, rocblas_operation_transpose, rocblas_operation_none, k, k, f, &one, B, f, C, f, &zero, A, ld)); }, 2);
        printf("f=%ld dgemm  k x k x f (TN): %.4f s  %.1f TF/s\n", (long)f, t, 2.0 * f * k * k / t * 1e-12);
        hipLaunchKernelGGL(k_fill, dim3((f + 255) / 256, f), dim3(256), 0, s, A, f, ld);
        t = timeit(s, [&] { CK(rocblas_dgemv_64(h, rocblas_operation_none, f, f, &one, A, ld, v, 1, &zero, u, 1)); }, 3);
        printf("f=%ld dgemv N: %.5f s  %.0f GB/s\n", (long)f, t, 8.0 * f * f / t * 1e-9);
        t = timeit(s, [&] { CK(rocblas_dgemv_64(h, rocblas_operation_transpose, f, f, &one, A, ld, v, 1, &zero, u, 1)); }, 3);
        printf("f=%ld dgemv T: %.5f s  %.0f GB/s\n", (long)f, t, 8.0 * f * f / t * 1e-9);
        t = timeit(s, [&] { CK(rocsolver_dpotrf_64(h, rocblas_fill_lower, f, A, ld, info)); });
        int64_t hinfo = -1; CK(hipMemcpy(&hinfo, info, 8, hipMemcpyDeviceToHost));
        printf("f=%ld dpotrf: %.4f s  %.1f TF/s (info %ld)\n", (long)f, t, f * (double)f * f / 3.0 / t * 1e-12, (long)hinfo);
        t = timeit(s, [&] { CK(rocblas_dtrsm_64(h, rocblas_side_left, rocblas_fill_lower, rocblas_operation_none, rocblas_diagonal_non_unit, f, k, &one, A, ld, B, f)); });
        printf("f=%ld dtrsm L\\B (k=%ld): %.4f s  %.1f TF/s\n", (long)f, (long)k, t, (double)f * f * k / t * 1e-12);
        CK(hipMemcpyAsync(u, v, sizeof(double) * f, hipMemcpyDeviceToDevice, s));
        t = timeit(s, [&] { CK(rocblas_dtrsv_64(h, rocblas_fill_lower, rocblas_operation_none, rocblas_diagonal_non_unit, f, A, ld, u, 1)); });
        printf("f=%ld dtrsv N: %.5f s  %.0f GB/s\n", (long)f, t, 4.0 * f * f / t * 1e-9);
        t = timeit(s, [&] { CK(rocblas_dtrsv_64(h, rocblas_fill_lower, rocblas_operation_transpose, rocblas_diagonal_non_unit, f, A, ld, u, 1)); });
        printf("f=%ld dtrsv T: %.5f s  %.0f GB/s\n", (long)f, t, 4.0 * f * f / t * 1e-9);
        t = timeit(s, [&] { CK(rocblas_dsyrk_64(h, rocblas_fill_lower, rocblas_operation_none, f, k, &mone, B, f, &one, A, ld)); });
        printf("f=%ld dsyrk f x f x %ld: %.4f s  %.1f TF/s\n", (long)f, (long)k, t, (double)f * f * k / t * 1e-12);
        fflush(stdout);
        hipFree(A); hipFree(B); hipFree(C); hipFree(v); hipFree(u); hipFree(info);
    }
    return 0;
}
