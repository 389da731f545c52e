// fnn_splits.hip -- circular split weights (non-negative least squares) on gfx950.
//
// The "next" row N1 of SURVEY.md 8(f): given the circular ordering, the weights of the n(n-1)/2 circular splits are
// the solution of the non-negative least-squares problem min |A x - d|, x >= 0 - what the reference's live path computes
// with a dense design matrix and a third-party Lawson-Hanson solver (FastNN.java:401-454), here on the implicit
// operators of CircularSplitWeights.java (:603-731 calculateAtx / calculateAb) with the re-ordering of the distances by
// the circular ordering restored (setupD :202-211) and the result in the index space of the live path
// (FastNN.java:405-419).  Method: the closed-form unconstrained optimum if it is feasible; else Lawson-Hanson "from
// below" on closed-form entries of A^T A with an explicitly maintained inverse of the free block (rank-one updates by
// rocBLAS, rebuilds by rocSOLVER), Kuhn-Tucker test on a fresh inverse; the reference's own active-set / conjugate-
// gradient method (:366-557 runActiveConjugate, :769-831 circularConjugateGrads, :283-337 worstIndices) remains as the
// fallback and behind FNN_SW_REFERENCE_METHOD=1 (it stops short of the optimum by ~1e-5: DESIGN.md section 7).
//
// GPU formulation.  All vectors of the method (split weights x, distances d, residuals ...) are
// kept as the strict upper triangle of dense n x n fp64 arrays: entry [i][j], i < j, is the split
// (i, j) = cycle positions {i+1 .. j}, or the pair of positions (i, j).  The reference evaluates
// A b and A^T y with anti-diagonal recurrences (n - 1 dependent sweeps); here both are O(1)
// gathers from a 2-D inclusive prefix sum P of the argument (row scan, LDS-tiled transpose, row
// scan; P is held transposed):
//   (A b)[a][b]   = sum of x over the splits that separate positions a < b
//                 = rect(i in [0,a-1], j in [a,b-1]) + rect(i in [a,b-1], j in [b,n-1])
//   (A^T y)[i][j] = sum of y over the pairs separated by split (i,j), S = {i+1..j}
//                 = (RS[j] - RS[i]) - 2 (P[j][j] - P[i][j]),  RS = prefix of the row sums of the
//                   symmetric completion of y.
// Everything is bandwidth-bound elementwise / scan work on n^2 doubles; the control flow of the
// solver (a handful of scalars per step) runs on the host.  The optimum is unique (A is square and non-singular); the
// tests hold the weights to 1e-6 of a dense NNLS solve of the live path's problem where that is computable (n <= 64),
// to the known weights of generated circular metrics up to 4096 taxa, and to the Kuhn-Tucker conditions beyond.
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "fnn_engine.h"  // fnn::fail / the thread-local error message shared with fnn_hip.hip

namespace fnnsw {

constexpr double CG_EPSILON = 1e-8;  // CircularSplitWeights.java:54
constexpr int T = 256;

#define SWOK(x) ((x) == hipSuccess)

// ---------------------------------------------------------------- 2-D prefix sum
// row scan in place: one workgroup per row
// Scalars of the conjugate-gradient loop live on the device (one host read per CG_CHECK iterations);
// every kernel of an iteration returns at once when the loop has ended (sc[SC_DONE]).
enum { SC_RHO = 0, SC_RHO_OLD, SC_PW, SC_TOL2, SC_DONE, SC_K, SC_ALPHA, SC_BETA, SC_KMAX, SC_WORDS = 16 };
#define CG_SKIP(sc) do { if ((sc) != nullptr && (sc)[SC_DONE] != 0.0) return; } while (0)

// row scan: dst row = inclusive prefix of src row; one workgroup per row
__global__ __launch_bounds__(T) void k_rowscan(const double* src, double* a, int n, int64_t ld, const double* sc) {
    __shared__ double wsum[T / 64];
    __shared__ double carry_s;
    CG_SKIP(sc);
    const double* srow = src + (int64_t)blockIdx.x * ld;
    double* row = a + (int64_t)blockIdx.x * ld;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0.0;
    __syncthreads();
    for (int base = 0; base < n; base += T) {
        const int i = base + (int)threadIdx.x;
        double v = i < n ? srow[i] : 0.0;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const double t = __shfl_up(v, d, 64);
            if (lane >= d) v += t;
        }
        if (lane == 63) wsum[w] = v;
        __syncthreads();
        double pre = carry_s;
        for (int k = 0; k < w; k++) pre += wsum[k];
        v += pre;
        if (i < n) row[i] = v;
        __syncthreads();
        if (threadIdx.x == T - 1) carry_s = v;
        __syncthreads();
    }
}

// out = in^T (n x n), 32 x 32 tiles through LDS
__global__ __launch_bounds__(T) void k_transpose(const double* in, double* out, int n, int64_t ld, const double* sc) {
    __shared__ double tile[32][33];
    CG_SKIP(sc);
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    for (int k = ty; k < 32; k += 8) {
        const int r = by + k, c = bx + tx;
        tile[k][tx] = (r < n && c < n) ? in[(int64_t)r * ld + c] : 0.0;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int r = bx + k, c = by + tx;
        if (r < n && c < n) out[(int64_t)r * ld + c] = tile[tx][k];
    }
}

// ---------------------------------------------------------------- operators (gathers from the transposed prefix Pt[j][i] = P[i][j])
__device__ __forceinline__ double PT(const double* Pt, int64_t ld, int i, int j) {  // P[i][j], -1 -> 0
    return (i < 0 || j < 0) ? 0.0 : Pt[(int64_t)j * ld + i];
}

// d[a][b] = (A x)[a][b] for a < b, from the prefix of x
__global__ __launch_bounds__(T) void k_ab(const double* Pt, double* d, int n, int64_t ld, const double* sc) {
    CG_SKIP(sc);
    const int b = blockIdx.x * T + threadIdx.x, a = blockIdx.y;
    if (b >= n || a >= b) return;
    const double first = PT(Pt, ld, a - 1, b - 1) - PT(Pt, ld, a - 1, a - 1);
    const double second = (PT(Pt, ld, b - 1, n - 1) - PT(Pt, ld, a - 1, n - 1)) - (PT(Pt, ld, b - 1, b - 1) - PT(Pt, ld, a - 1, b - 1));
    d[(int64_t)a * ld + b] = first + second;
}

// RS = inclusive prefix of the row sums of the symmetric completion (one workgroup)
__global__ __launch_bounds__(1024) void k_scan1(const double* Qt, double* v, int n, int64_t ld, const double* sc) {
    __shared__ double wsum[16];
    __shared__ double carry_s;
    CG_SKIP(sc);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0.0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int i = base + (int)threadIdx.x;
        double x = 0.0;
        if (i < n) x = (PT(Qt, ld, i, n - 1) - PT(Qt, ld, i - 1, n - 1)) + (PT(Qt, ld, n - 1, i) - PT(Qt, ld, n - 1, i - 1));
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const double t = __shfl_up(x, d, 64);
            if (lane >= d) x += t;
        }
        if (lane == 63) wsum[w] = x;
        __syncthreads();
        double pre = carry_s;
        for (int k = 0; k < w; k++) pre += wsum[k];
        x += pre;
        if (i < n) v[i] = x;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = x;
        __syncthreads();
    }
}
// p[i][j] = (A^T y)[i][j] for i < j
// (mask: entries of the active set are written as 0, circularConjugateGrads :817-819)
__global__ __launch_bounds__(T) void k_atx(const double* Qt, const double* RS, double* p, int n, int64_t ld, const uint8_t* mask,
                                           const double* sc) {
    CG_SKIP(sc);
    const int j = blockIdx.x * T + threadIdx.x, i = blockIdx.y;
    if (j >= n || i >= j) return;
    const double inside = PT(Qt, ld, j, j) - PT(Qt, ld, i, j);
    const int64_t k = (int64_t)i * ld + j;
    p[k] = (mask != nullptr && mask[k]) ? 0.0 : (RS[j] - RS[i]) - 2.0 * inside;
}

// ---------------------------------------------------------------- setup kernels
// d'[a][b] = D[ord[a+1]-1][ord[b+1]-1] for a < b, 0 elsewhere (restored setupD)
__global__ __launch_bounds__(T) void k_reorder(const double* D, int64_t ldD, const int32_t* ord, double* d, int n, int64_t ld) {
    const int b = blockIdx.x * T + threadIdx.x, a = blockIdx.y;
    if (b >= n) return;
    double v = 0.0;
    if (a < b) v = D[(int64_t)(ord[a + 1] - 1) * ldD + (ord[b + 1] - 1)];
    d[(int64_t)a * ld + b] = v;
}
// Chepoi & Fichet closed form of the unconstrained optimum (runUnconstrainedLS :247-271), with dd(a,b)
// the re-ordered distance of positions a, b (either order, 0 on the diagonal):
//   x[i][j] = ( dd(i,j) + dd(i+1,j+1) - dd(i,j+1) - dd(i+1,j) ) / 2, positions taken modulo n
__device__ __forceinline__ double dd(const double* d, int64_t ld, int a, int b) {
    if (a == b) return 0.0;
    return a < b ? d[(int64_t)a * ld + b] : d[(int64_t)b * ld + a];
}
__global__ __launch_bounds__(T) void k_unconstrained(const double* d, double* x, int n, int64_t ld) {
    const int j = blockIdx.x * T + threadIdx.x, i = blockIdx.y;
    if (j >= n) return;
    double v = 0.0;
    if (i < j) {
        const int i1 = i + 1, j1 = (j + 1) % n;
        v = (dd(d, ld, i, j) + dd(d, ld, i1, j1) - dd(d, ld, i, j1) - dd(d, ld, i1, j)) / 2.0;
    }
    x[(int64_t)i * ld + j] = v;
}

// ---------------------------------------------------------------- vector kernels over the strict upper triangle
enum { OP_COPY = 0, OP_R_INIT, OP_P_UPDATE, OP_W_MASK, OP_XR_UPDATE, OP_CONTRACT, OP_MOVE_OLD, OP_GRAD };
struct VecArgs {
    double* a; double* b; const double* c; const double* e; uint8_t* act;
    double s0, s1;
    int n; int64_t ld;
    const double* sc;  // device scalars of the CG loop (nullptr: s0 / s1 are the arguments)
};
template <int OP>
__global__ __launch_bounds__(T) void k_vec(VecArgs g) {
    CG_SKIP(g.sc);
    const int j = blockIdx.x * T + threadIdx.x, i = blockIdx.y;
    if (j >= g.n || i >= j) return;
    const int64_t k = (int64_t)i * g.ld + j;
    if (OP == OP_COPY) g.a[k] = g.c[k];
    else if (OP == OP_R_INIT) g.a[k] = g.act[k] ? 0.0 : g.c[k] - g.a[k];            // r = active ? 0 : b - r   (:785-789)
    else if (OP == OP_P_UPDATE) g.a[k] = (g.sc[SC_K] == 0.0) ? g.c[k] : g.c[k] + g.sc[SC_BETA] * g.a[k];  // p = r (k = 1) | r + beta p (:801-808)
    else if (OP == OP_W_MASK) { if (g.act[k]) g.a[k] = 0.0; }                         // w = 0 on the active set  (:817-819)
    else if (OP == OP_XR_UPDATE) { const double al = g.sc[SC_ALPHA]; g.a[k] += al * g.c[k]; g.b[k] -= al * g.e[k]; }  // x += alpha p; r -= alpha w (:826-829)
    else if (OP == OP_CONTRACT) {                                                     // worstIndices + contraction (:411-430)
        const double v = g.a[k];
        if (v < g.s0 || (g.s1 != 0.0 && v == g.s0)) { g.a[k] = 0.0; g.act[k] = 1; }
    } else if (OP == OP_MOVE_OLD) { if (!g.act[k]) g.b[k] += g.s0 * (g.a[k] - g.b[k]); }  // old_x += min_xi (x - old_x) (:452-454)
    else if (OP == OP_GRAD) g.a[k] = (g.a[k] - g.c[k]) * 2.0;                          // r = 2 (AtWAx - AtWd)     (:478-479)
}

// reductions: per-workgroup partials, folded by a second launch (fixed order: reproducible)
struct Best { double v; int64_t k; };
enum { RD_DOT = 0, RD_COUNT_NEG, RD_COUNT_LT, RD_COUNT_EQ, RD_MIN_RATIO, RD_MIN_ACTIVE_GRAD, RD_ANY_NEG, RD_MAX_UNMASKED };
template <int RD>
__global__ __launch_bounds__(T) void k_reduce(const double* a, const double* b, const uint8_t* act, double s0, int n, int64_t ld,
                                              double* partial, Best* bpartial, const double* sc) {
    __shared__ double sh[T / 64];
    __shared__ Best shb[T / 64];
    CG_SKIP(sc);
    double acc = 0.0;
    Best best{INFINITY, INT64_MAX};
    for (int i = blockIdx.y; i < n; i += gridDim.y)
        for (int j = blockIdx.x * T + threadIdx.x; j < n; j += gridDim.x * T) {
            if (i >= j) continue;
            const int64_t k = (int64_t)i * ld + j;
            if (RD == RD_DOT) acc += a[k] * b[k];
            else if (RD == RD_COUNT_NEG) acc += a[k] < 0.0 ? 1.0 : 0.0;
            else if (RD == RD_COUNT_LT) acc += (a[k] < 0.0 && a[k] < s0) ? 1.0 : 0.0;
            else if (RD == RD_COUNT_EQ) acc += (a[k] < 0.0 && a[k] == s0) ? 1.0 : 0.0;
            else if (RD == RD_MIN_RATIO) {  // first minimum of old_x / (old_x - x) over x < 0 (:434-445)
                if (a[k] < 0.0) {
                    const double xi = b[k] / (b[k] - a[k]);
                    if (xi < best.v || (xi == best.v && k < best.k)) { best.v = xi; best.k = k; }
                }
            } else if (RD == RD_MAX_UNMASKED) {  // first maximum of a over the entries outside the mask (as the minimum of -a)
                if (!act[k]) {
                    const double gv = -a[k];
                    if (gv < best.v || (gv == best.v && k < best.k)) { best.v = gv; best.k = k; }
                }
            } else if (RD == RD_MIN_ACTIVE_GRAD) {  // first minimum of the gradient over the active set (:480-487)
                if (act[k]) {
                    const double gv = a[k];
                    if (gv < best.v || (gv == best.v && k < best.k)) { best.v = gv; best.k = k; }
                }
            }
        }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (RD == RD_MIN_RATIO || RD == RD_MIN_ACTIVE_GRAD || RD == RD_MAX_UNMASKED) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            Best o;
            o.v = __shfl_down(best.v, off, 64);
            o.k = __shfl_down((long long)best.k, off, 64);
            if (o.v < best.v || (o.v == best.v && o.k < best.k)) best = o;
        }
        if (lane == 0) shb[w] = best;
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int q = 1; q < T / 64; q++)
                if (shb[q].v < best.v || (shb[q].v == best.v && shb[q].k < best.k)) best = shb[q];
            bpartial[blockIdx.y * gridDim.x + blockIdx.x] = best;
        }
    } else {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) acc += __shfl_down(acc, off, 64);
        if (lane == 0) sh[w] = acc;
        __syncthreads();
        if (threadIdx.x == 0) partial[blockIdx.y * gridDim.x + blockIdx.x] = ((sh[0] + sh[1]) + sh[2]) + sh[3];
    }
}
// the scalars of one CG step from the partial sums of the preceding k_reduce<RD_DOT> (one workgroup;
// fixed summation order).  stage 0: rho of the start residual; 1: alpha = rho / p.w; 2: the new rho,
// beta, the iteration count and the loop condition  rho > e_0^2 && k < kmax  (:795)
__global__ __launch_bounds__(T) void k_cg_scalar(const double* partial, int np, double* sc, int stage) {
    __shared__ double sh[T / 64];
    if (stage != 0 && sc[SC_DONE] != 0.0) return;
    double acc = 0.0;
    for (int i = threadIdx.x; i < np; i += T) acc += partial[i];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x != 0) return;
    const double sum = ((sh[0] + sh[1]) + sh[2]) + sh[3];
    if (stage == 0) {
        sc[SC_RHO] = sum; sc[SC_RHO_OLD] = 0.0; sc[SC_K] = 0.0;
        sc[SC_DONE] = (sum > sc[SC_TOL2] && 0.0 < sc[SC_KMAX]) ? 0.0 : 1.0;
    } else if (stage == 1) {
        sc[SC_PW] = sum;
        sc[SC_ALPHA] = sc[SC_RHO] / sum;
    } else {
        const double rho_old = sc[SC_RHO];
        sc[SC_RHO_OLD] = rho_old;
        sc[SC_RHO] = sum;
        sc[SC_BETA] = sum / rho_old;
        const double k = sc[SC_K] + 1.0;
        sc[SC_K] = k;
        sc[SC_DONE] = (sum > sc[SC_TOL2] && k < sc[SC_KMAX]) ? 0.0 : 1.0;
    }
}
// extreme negative values for the bisection of the cutoff
__global__ __launch_bounds__(T) void k_minneg(const double* a, int n, int64_t ld, double* partial) {
    __shared__ double sh[T / 64];
    double mn = 0.0;
    for (int i = blockIdx.y; i < n; i += gridDim.y)
        for (int j = blockIdx.x * T + threadIdx.x; j < n; j += gridDim.x * T)
            if (i < j) { const double v = a[(int64_t)i * ld + j]; if (v < mn) mn = v; }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { const double o = __shfl_down(mn, off, 64); if (o < mn) mn = o; }
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = mn;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int q = 1; q < T / 64; q++) if (sh[q] < mn) mn = sh[q];
        partial[blockIdx.y * gridDim.x + blockIdx.x] = mn;
    }
}
__global__ __launch_bounds__(T) void k_fill(double* a, int64_t count, double v) {
    const int64_t i = (int64_t)blockIdx.x * T + threadIdx.x;
    if (i < count) a[i] = v;
}
// live[k] over (i, j), 0 <= i < j <= n-1, from the fast algorithm's x
__global__ __launch_bounds__(T) void k_to_live(const double* x, double* live, int n, int64_t ld) {
    const int j = blockIdx.x * T + threadIdx.x, i = blockIdx.y;
    if (j >= n || i >= j) return;
    const int64_t k = ((int64_t)(2 * n - i - 1) * i) / 2 + (j - i - 1);  // row-major index of (i, j), i < j
    const double v = (i >= 1) ? x[(int64_t)(i - 1) * ld + (j - 1)] : x[(int64_t)(j - 1) * ld + (n - 1)];
    live[k] = v;
}

// the weights above `thr` (live index order, FastNN.java:455), appended in any order: {index, weight}; *count receives their number
__global__ __launch_bounds__(T) void k_pick_positive(const double* live, int64_t N, double thr, int64_t* idx, double* w, unsigned long long* count, unsigned long long cap) {
    const int64_t k = (int64_t)blockIdx.x * T + threadIdx.x;
    const bool take = k < N && live[k] > thr;
    const unsigned long long ball = __ballot(take);
    if (ball == 0) return;
    const int lane = threadIdx.x & 63, first = __ffsll((long long)ball) - 1;
    unsigned long long base = 0;
    if (lane == first) base = atomicAdd(count, (unsigned long long)__popcll(ball));
    base = __shfl(base, first, 64);
    if (take) {
        const unsigned long long at = base + __popcll(ball & ((1ULL << lane) - 1ULL));
        if (at < cap) { idx[at] = k; w[at] = live[k]; }
    }
}
// The solver's own Kuhn-Tucker check of the returned weights (g = A^T (A x - d) from the implicit operators): per workgroup
// {max -x, max |g| over x > 0, max -g over x <= 0, max |c|} over the strict upper triangle
__global__ __launch_bounds__(T) void k_kkt(const double* x, const double* g, const double* c, int n, int64_t ld, double* partial) {
    __shared__ double sh[T / 64][4];
    double v0 = 0.0, v1 = 0.0, v2 = 0.0, v3 = 0.0;
    for (int i = blockIdx.y; i < n; i += gridDim.y)
        for (int j = blockIdx.x * T + threadIdx.x; j < n; j += gridDim.x * T) {
            if (i >= j) continue;
            const int64_t k = (int64_t)i * ld + j;
            const double xv = x[k], gv = g[k], cv = fabs(c[k]);
            if (-xv > v0 || xv != xv) v0 = xv != xv ? INFINITY : -xv;
            if (xv > 0.0) { if (fabs(gv) > v1 || gv != gv) v1 = gv != gv ? INFINITY : fabs(gv); }
            else if (-gv > v2 || gv != gv) v2 = gv != gv ? INFINITY : -gv;
            if (cv > v3) v3 = cv;
        }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        v0 = fmax(v0, __shfl_down(v0, off, 64)); v1 = fmax(v1, __shfl_down(v1, off, 64));
        v2 = fmax(v2, __shfl_down(v2, off, 64)); v3 = fmax(v3, __shfl_down(v3, off, 64));
    }
    if ((threadIdx.x & 63) == 0) { double* q = sh[threadIdx.x >> 6]; q[0] = v0; q[1] = v1; q[2] = v2; q[3] = v3; }
    __syncthreads();
    if (threadIdx.x < 4) {
        double m = sh[0][threadIdx.x];
        for (int q = 1; q < T / 64; q++) m = fmax(m, sh[q][threadIdx.x]);
        partial[4 * (blockIdx.y * gridDim.x + blockIdx.x) + threadIdx.x] = m;
    }
}
// test hook (FNN_SW_FAULT_PERTURB=1): the first positive weight of the grid is doubled, so that the call's own check has something to find
__global__ void k_fault_perturb(double* x, int n, int64_t ld) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    for (int i = 0; i < n; i++)
        for (int j = i + 1; j < n; j++)
            if (x[(int64_t)i * ld + j] > 0.0) { x[(int64_t)i * ld + j] *= 2.0; return; }
}
// a = a - c over the strict upper triangle
__global__ __launch_bounds__(T) void k_sub(double* a, const double* c, int n, int64_t ld) {
    const int j = blockIdx.x * T + threadIdx.x, i = blockIdx.y;
    if (j >= n || i >= j) return;
    const int64_t k = (int64_t)i * ld + j;
    a[k] -= c[k];
}

// ---------------------------------------------------------------- "from below": block active-set method on the normal equations
// The reference's method starts from the unconstrained optimum (every split free) and contracts; on distances that are
// far from circular (random matrices: only ~2.4 n of the n(n-1)/2 splits end up with a positive weight) almost all of its
// work is conjugate-gradient iterations on huge, ill-conditioned free sets.  Lawson & Hanson's active-set method - what the
// reference's LIVE path runs (edu.rit.numeric.NonNegativeLeastSquares, FastNN.java:401-454) - grows the free set F from
// nothing and solves each sub-problem exactly.  Round 2 let ONE split enter per step (4.2 n steps, level-2 work and six
// host round trips each: 2-3.5 h at 32768 taxa).  Here whole BLOCKS enter and leave:
//   * candidates are the local maxima of the multiplier w = A^T (d - A x) on the (i, j) grid (w is smooth there: the
//     largest values sit in clusters of nearly identical splits, of which at most one survives), the largest ~10 % |F| of them;
//   * the free set's normal equations H_FF = L L^T are held as W = L^-1 (lower triangular): a block that enters appends
//     rows to W and never touches existing entries (two triangular GEMMs, one small Cholesky) - numerically benign, unlike
//     the explicitly updated inverse of round 2, whose errors compound per update;
//   * a split that leaves is not eliminated but CONSTRAINED to zero: with Y = W[:, R] the columns of the splits R that
//     left, the sub-problem's solution is x = W^T (z - Y (Y^T Y)^-1 Y^T z), z = W c - a projection, exact and stable;
//     when R exceeds ~15 % of the factor the factor is rebuilt from H's closed form (Cholesky + triangular inverse);
//   * all splits with a negative weight leave at once; the step is kept only if the objective -c_F.x_F/2 fell, otherwise
//     the block is taken back (the appended rows are simply dropped), shrunk, and at a single split the Lawson-Hanson
//     ratio step - whose descent is guaranteed - takes over.
// H has a closed form: the number of position pairs that two circular splits S, T both separate is
// |S n T| |S^c n T^c| + |S n T^c| |S^c n T|.
__device__ __forceinline__ double h_entry(int n, int i, int j, int k, int l) {  // splits (i,j), (k,l): positions {i+1..j}, {k+1..l}
    const int lo = i > k ? i : k, hi = j < l ? j : l;
    const double st = hi > lo ? (double)(hi - lo) : 0.0, s = (double)(j - i), t = (double)(l - k);
    return st * ((double)n - s - t + st) + (s - st) * (t - st);
}
// out[r + c ldo] = H[Fa[r], Fb[c]] (column-major); lower_only: entries above the diagonal are written as zeros
__global__ __launch_bounds__(T) void k_hblock(const int2* Fa, int64_t fa, const int2* Fb, int64_t fb, int n, double* out, int64_t ldo,
                                              int lower_only) {
    const int64_t r = (int64_t)blockIdx.x * T + threadIdx.x;
    if (r >= fa) return;
    const int2 a = Fa[r];
    for (int64_t c = blockIdx.y; c < fb; c += gridDim.y) {
        const int2 b = Fb[c];
        out[c * ldo + r] = (lower_only && r < c) ? 0.0 : h_entry(n, a.x, a.y, b.x, b.y);
    }
}
__global__ __launch_bounds__(T) void k_gather(const int2* F, int64_t f, const double* grid, int64_t ld, double* out) {
    const int64_t p = (int64_t)blockIdx.x * T + threadIdx.x;
    if (p < f) out[p] = grid[(int64_t)F[p].x * ld + F[p].y];
}
__global__ __launch_bounds__(T) void k_scatter(const int2* F, int64_t f, const double* v, double* grid, int64_t ld) {
    const int64_t p = (int64_t)blockIdx.x * T + threadIdx.x;
    if (p < f) grid[(int64_t)F[p].x * ld + F[p].y] = v[p];
}
__global__ __launch_bounds__(T) void k_mask(const int2* F, int64_t f, uint8_t* mask, int64_t ld, uint8_t v) {
    const int64_t p = (int64_t)blockIdx.x * T + threadIdx.x;
    if (p < f) mask[(int64_t)F[p].x * ld + F[p].y] = v;
}
// A lower-triangular inverse factor as COLUMN PANELS of width pw: panel q holds the rows [q pw, cap) of its pw columns,
// column-major with leading dimension cap - q pw - the zero half above the diagonal blocks is not stored (half the memory
// of a square - 9/16 with the eight panels used - which is what bounds the factor's capacity at 32768 taxa).  Every product with
// the factor runs panel by panel.  pw >= cap: one panel = plain square storage (the Gram factor of the departed columns).
struct TriStore {
    double* base;
    int64_t cap, pw;
    __host__ __device__ int64_t off(int64_t q) const { return pw * (q * cap - pw * (q * (q - 1) / 2)); }
    __host__ __device__ int64_t ld(int64_t q) const { return cap - q * pw; }
    __host__ __device__ double* panel(int64_t q) const { return base + off(q); }
    __host__ __device__ double* at(int64_t row, int64_t col) const {  // row >= the first row of the column's panel
        const int64_t q = col / pw, q0 = q * pw;
        return base + off(q) + (col - q0) * ld(q) + (row - q0);
    }
    static int64_t elems(int64_t cap, int64_t pw) {
        int64_t tot = 0;
        for (int64_t q0 = 0; q0 < cap; q0 += pw) tot += std::min(pw, cap - q0) * (cap - q0);
        return tot;
    }
};
// Y[:, q] = W[0:f, list[q]] for a panel-stored factor (rows above the column's panel are zeros)
__global__ __launch_bounds__(T) void k_gather_cols_tri(TriStore W, int64_t f, const int32_t* list, int64_t cnt, double* Y, int64_t ldy) {
    const int64_t r = (int64_t)blockIdx.x * T + threadIdx.x;
    if (r >= f) return;
    for (int64_t q = blockIdx.y; q < cnt; q += gridDim.y) {
        const int64_t c = list[q];
        Y[q * ldy + r] = r >= (c / W.pw) * W.pw ? *W.at(r, c) : 0.0;
    }
}
// Y[f0 + i, q] = W[f0 + i, list[q]], i < k (list[q] < f0: the rows lie inside the columns' panels)
__global__ __launch_bounds__(T) void k_gather_rows_tri(TriStore W, int64_t f0, int64_t k, const int32_t* list, int64_t cnt, double* Y, int64_t ldy) {
    const int64_t i = (int64_t)blockIdx.x * T + threadIdx.x;
    if (i >= k) return;
    for (int64_t q = blockIdx.y; q < cnt; q += gridDim.y) Y[q * ldy + f0 + i] = *W.at(f0 + i, list[q]);
}
// Y[:, q] = W[0:f, list[q]] (columns of the splits that left)
__global__ __launch_bounds__(T) void k_gather_cols(const double* W, int64_t ldw, int64_t f, const int32_t* list, int64_t cnt, double* Y, int64_t ldy) {
    const int64_t r = (int64_t)blockIdx.x * T + threadIdx.x;
    if (r >= f) return;
    for (int64_t q = blockIdx.y; q < cnt; q += gridDim.y) Y[q * ldy + r] = W[(int64_t)list[q] * ldw + r];
}
// out (m x m, ldo) = a[idx, idx] (a: lda), the sub-block of the kept rows / columns
__global__ __launch_bounds__(T) void k_gather_sym(const double* a, int64_t lda, const int32_t* idx, int64_t m, double* out, int64_t ldo) {
    const int64_t r = (int64_t)blockIdx.x * T + threadIdx.x;
    if (r >= m) return;
    const int64_t ri = idx[r];
    for (int64_t c = blockIdx.y; c < m; c += gridDim.y) out[c * ldo + r] = a[(int64_t)idx[c] * lda + ri];
}
__global__ __launch_bounds__(T) void k_gather_vec(const double* a, const int32_t* idx, int64_t m, double* out) {
    const int64_t r = (int64_t)blockIdx.x * T + threadIdx.x;
    if (r < m) out[r] = a[idx[r]];
}
__global__ __launch_bounds__(T) void k_gather_int2(const int2* a, const int32_t* idx, int64_t m, int2* out) {
    const int64_t r = (int64_t)blockIdx.x * T + threadIdx.x;
    if (r < m) out[r] = a[idx[r]];
}
// out (nn x m, ldo) = a^T (a: m x nn, lda)
__global__ __launch_bounds__(T) void k_transpose_small(const double* a, int64_t lda, int64_t m, int64_t nn, double* out, int64_t ldo) {
    const int64_t r = (int64_t)blockIdx.x * T + threadIdx.x;  // row of a
    if (r >= m) return;
    for (int64_t c = blockIdx.y; c < nn; c += gridDim.y) out[r * ldo + c] = a[c * lda + r];
}
// copy the lower triangle of a (m x m, lda) into b (ldb), zeros above the diagonal
__global__ __launch_bounds__(T) void k_copy_lower(const double* a, int64_t lda, double* b, int64_t ldb, int64_t m) {
    const int64_t r = (int64_t)blockIdx.x * T + threadIdx.x;
    if (r >= m) return;
    for (int64_t c = blockIdx.y; c < m; c += gridDim.y) b[c * ldb + r] = r >= c ? a[c * lda + r] : 0.0;
}
// rows x cols block (column-major: the rows of a column are adjacent) filled / copied by kernels: the runtime's 2-D fill
// (hipMemset2DAsync -> fillBufferAligned2D) ran at 6 GB/s on the pitched blocks of the factor - 209 ms for the 1.2 GB of zeros
// above a block's new columns, 6.8 s of a 32768-taxon solve (profiles/r03/r03_splits_kernel_stats_n32768.csv) - and the 2-D
// device-to-device copies go through the copy engines.
__global__ __launch_bounds__(T) void k_fill2d(double* dst, int64_t ldd, int64_t rows, int64_t cols, double v) {
    const int64_t r = (int64_t)blockIdx.x * T + threadIdx.x;
    if (r >= rows) return;
    for (int64_t c = blockIdx.y; c < cols; c += gridDim.y) dst[c * ldd + r] = v;
}
__global__ __launch_bounds__(T) void k_copy2d(const double* src, int64_t lds_, double* dst, int64_t ldd, int64_t rows, int64_t cols) {
    const int64_t r = (int64_t)blockIdx.x * T + threadIdx.x;
    if (r >= rows) return;
    for (int64_t c = blockIdx.y; c < cols; c += gridDim.y) dst[c * ldd + r] = src[c * lds_ + r];
}
constexpr int TRI_NB = 64;
// inverse of the Cholesky factor of a symmetric positive definite m x m block (m <= 64; lower triangle of S read):
// Li (lower triangular, zeros above) with Li^T Li = S^-1.  ONE wave; lane t owns ROW t of the matrix in registers (every index
// below is a compile-time constant: the loops are unrolled, 256 VGPR, no scratch); the entry of another row that a step needs
// is the same for all lanes and comes straight out of the owning lane's register (two v_readlane_b32 into a scalar pair
// that feeds the FMA): no LDS round trip on the 4032 multiply-adds.  Left-looking Cholesky, then lane t computes COLUMN t of
// the inverse by forward substitution.  Measured (profiles/r03/r03_splits_kernel_stats_n4096.csv): 64 us per block - a first
// version that walked an LDS copy with dependent reads took 170 us (45 % of a 4096-taxon solve).  What bounds it now is the
// fetch of its own ~100 KB of straight-line code, executed once per launch (the same 64 us with broadcast LDS reads in place of
// the lane reads); a rolled loop would index the register rows at run time, at about the same cost per term.
// Rows beyond m are padded with the identity.  A pivot that is not positive reports its (1-based, offset by `off`)
// position through atomicMin on *info.  (Not on the bit-exact path: FMAs are welcome here.)
__device__ __forceinline__ double lane_value(double v, int src) {  // the value lane `src` holds (src a compile-time constant)
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, src), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), src);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__global__ __launch_bounds__(TRI_NB) void k_invchol_small(const double* S, int64_t lds, double* Li, int64_t ldl, int m, long long off, long long* info) {
    __shared__ double Ls[TRI_NB][TRI_NB + 1];
    __shared__ double rd[TRI_NB];
    const int t = threadIdx.x;
    double a[TRI_NB];
#pragma unroll
    for (int c = 0; c < TRI_NB; c++) a[c] = (t < m && c <= t) ? S[(int64_t)c * lds + t] : (c == t ? 1.0 : 0.0);
#pragma unroll
    for (int j = 0; j < TRI_NB; j++) {
        double sj = a[j];
#pragma unroll
        for (int l = 0; l < j; l++) sj = __builtin_fma(-a[l], lane_value(a[l], j), sj);  // (row t) . (row j) over the finished columns
        double piv = lane_value(sj, j);
        if (!(piv > 0.0)) {
            if (t == j && j < m) atomicMin(info, off + j + 1);
            piv = 1.0;
        }
        const double dj = __builtin_sqrt(piv);
        a[j] = t == j ? dj : (t > j ? sj / dj : 0.0);
        if (t == j) rd[j] = 1.0 / dj;
    }
    __syncthreads();
    // column t of L^-1: x_i = (delta_it - sum_{l < i} L_il x_l) / L_ii (x_l = 0 for l < t falls out of the recurrence)
    double x[TRI_NB];
#pragma unroll
    for (int i = 0; i < TRI_NB; i++) {
        double acc = i == t ? 1.0 : 0.0;
#pragma unroll
        for (int l = 0; l < i; l++) acc = __builtin_fma(-lane_value(a[l], i), x[l], acc);
        x[i] = acc * rd[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < TRI_NB; i++) Ls[i][t] = x[i];  // X[i][t]
    __syncthreads();
    for (int c = 0; c < m; c++)
        if (t < m) Li[(int64_t)c * ldl + t] = Ls[t][c];
}
// local maxima of the multiplier on the (i, j) grid: entry (i, j), i < j, not masked, w > tol, and no unmasked neighbour
// within `rad` (Chebyshev) is larger in the order (w, index).  Winners are appended to the candidate list.
__global__ __launch_bounds__(T) void k_candidates(const double* w, const uint8_t* mask, int n, int64_t ld, double tol, int rad,
                                                  double* key, int64_t* idx, unsigned long long* count, unsigned long long capacity) {
    const int j = blockIdx.x * T + threadIdx.x, i = blockIdx.y;
    bool win = false;
    int64_t k = 0;
    double v = 0.0;
    if (j < n && i < j) {
        k = (int64_t)i * ld + j;
        v = w[k];
        if (!mask[k] && v > tol) {
            win = true;
            for (int di = -rad; di <= rad && win; di++) {
                const int ii = i + di;
                if (ii < 0 || ii >= n) continue;
                for (int dj = -rad; dj <= rad; dj++) {
                    const int jj = j + dj;
                    if ((di == 0 && dj == 0) || jj <= ii || jj >= n) continue;
                    const int64_t kk = (int64_t)ii * ld + jj;
                    const double u = w[kk];
                    if (!mask[kk] && (u > v || (u == v && kk < k))) { win = false; break; }
                }
            }
        }
    }
    const unsigned long long ball = __ballot(win);
    if (ball == 0) return;
    const int lane = threadIdx.x & 63;
    unsigned long long base = 0;
    if (lane == __ffsll((long long)ball) - 1) base = atomicAdd(count, (unsigned long long)__popcll(ball));
    base = __shfl(base, __ffsll((long long)ball) - 1, 64);
    if (win) {
        const unsigned long long at = base + __popcll(ball & ((1ULL << lane) - 1ULL));
        if (at < capacity) { key[at] = v; idx[at] = k; }
    }
}
// C (m x nn, ldc) = beta C + alpha * sum over the `parts` partial products at P + q * stride (each m x nn, ld m)
__global__ __launch_bounds__(T) void k_sum_parts(const double* P, int64_t stride, int parts, int64_t m, int64_t nn, double alpha, double beta, double* C,
                                                 int64_t ldc) {
    const int64_t r = (int64_t)blockIdx.x * T + threadIdx.x;
    if (r >= m) return;
    for (int64_t c = blockIdx.y; c < nn; c += gridDim.y) {
        double acc = 0.0;
        for (int q = 0; q < parts; q++) acc += P[(int64_t)q * stride + c * m + r];
        C[c * ldc + r] = (beta == 0.0 ? 0.0 : beta * C[c * ldc + r]) + alpha * acc;
    }
}
// A block of at most SMALLK columns: out (f x k) = Wl (lower triangular) * Bm, Wl read once.  One thread per row, a
// workgroup per 256 rows and column chunk; partial sums per chunk, added up in chunk order by k_sum_parts.
constexpr int SMALLK = 8;
constexpr int SK_CHUNK = 2048;
__global__ __launch_bounds__(T) void k_tri_times_small(TriStore Wl, int64_t f, const double* Bm, int64_t ldb, int k, double* part) {
    const int64_t row = (int64_t)blockIdx.x * T + threadIdx.x;
    const int64_t c0 = (int64_t)blockIdx.y * SK_CHUNK;
    if ((int64_t)blockIdx.x * T + T - 1 < c0) return;  // the whole row block lies above the diagonal of this chunk: zeros (the buffer is cleared)
    double acc[SMALLK];
#pragma unroll
    for (int a = 0; a < SMALLK; a++) acc[a] = 0.0;
    if (row < f) {
        int64_t c1 = c0 + SK_CHUNK < f ? c0 + SK_CHUNK : f;
        if (row + 1 < c1) c1 = row + 1;
        for (int64_t c = c0; c < c1; c++) {
            const double w = *Wl.at(row, c);
#pragma unroll
            for (int a = 0; a < SMALLK; a++)
                if (a < k) acc[a] += w * Bm[(int64_t)a * ldb + c];
        }
#pragma unroll
        for (int a = 0; a < SMALLK; a++)
            if (a < k) part[((int64_t)blockIdx.y * k + a) * f + row] = acc[a];
    }
}
// out (k x f, ldo) = Tm^T (f x k) * Wl (lower triangular): one workgroup per column of Wl, Wl read once
__global__ __launch_bounds__(T) void k_t_times_tri_small(const double* Tm, int64_t ldt, int k, TriStore Wl, int64_t f, double* out, int64_t ldo) {
    __shared__ double sh[T / 64][SMALLK];
    const int64_t j = blockIdx.x;
    double acc[SMALLK];
#pragma unroll
    for (int a = 0; a < SMALLK; a++) acc[a] = 0.0;
    const double* colj = Wl.at(j, j) - 0;  // column j from its diagonal entry down (contiguous inside its panel)
    for (int64_t l = j + threadIdx.x; l < f; l += T) {
        const double w = colj[l - j];
#pragma unroll
        for (int a = 0; a < SMALLK; a++)
            if (a < k) acc[a] += w * Tm[(int64_t)a * ldt + l];
    }
#pragma unroll
    for (int a = 0; a < SMALLK; a++) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) acc[a] += __shfl_down(acc[a], off, 64);
    }
    if ((threadIdx.x & 63) == 0)
        for (int a = 0; a < SMALLK; a++) sh[threadIdx.x >> 6][a] = acc[a];
    __syncthreads();
    if (threadIdx.x < k) out[j * ldo + threadIdx.x] = ((sh[0][threadIdx.x] + sh[1][threadIdx.x]) + sh[2][threadIdx.x]) + sh[3][threadIdx.x];
}
__global__ __launch_bounds__(T) void k_idx_to_split(const int64_t* idx, int64_t cnt, int64_t ld, int2* out) {
    const int64_t p = (int64_t)blockIdx.x * T + threadIdx.x;
    if (p < cnt) out[p] = make_int2((int)(idx[p] / ld), (int)(idx[p] % ld));
}

// ---------------------------------------------------------------- device buffers kept between calls
// hipMalloc of the solver's ~220 GB costs 0.5-4.8 s from box to box (page tables: `fnn_sw_stats.t_alloc_s`).  Buffers of >= 64 MiB
// go back to a per-process pool instead of the driver and are handed out again to a request of exactly that size on that
// device (a second solve of the same n: every one of them); a failed hipMalloc - or a caller who wants the memory,
// fnn_split_weights_release_cache - empties the pool.  The pool's bytes count as free when the factor is sized.
struct BufferPool {
    struct Item { void* p; size_t bytes; int dev; };
    std::vector<Item> items;
    std::mutex mu;
    static constexpr size_t kMin = (size_t)64 << 20;
    void* take(size_t bytes, int dev) {
        std::lock_guard<std::mutex> g(mu);
        for (size_t i = 0; i < items.size(); i++)
            if (items[i].bytes == bytes && items[i].dev == dev) { void* p = items[i].p; items.erase(items.begin() + (long)i); return p; }
        return nullptr;
    }
    void give(void* p, size_t bytes, int dev) {
        if (bytes < kMin) { (void)hipFree(p); return; }
        std::lock_guard<std::mutex> g(mu);
        items.push_back(Item{p, bytes, dev});
    }
    size_t pooled(int dev) {
        std::lock_guard<std::mutex> g(mu);
        size_t t = 0;
        for (const Item& it : items) if (it.dev == dev) t += it.bytes;
        return t;
    }
    void flush() {
        std::lock_guard<std::mutex> g(mu);
        for (const Item& it : items) (void)hipFree(it.p);
        items.clear();
    }
};
inline BufferPool& pool() { static BufferPool* p = new BufferPool; return *p; }  // (never destroyed: no hipFree after the runtime has gone)

// ---------------------------------------------------------------- host driver
struct Solver {
    int n = 0;
    int64_t ld = 0;
    hipStream_t s = nullptr;
    std::vector<void*> allocs;
    std::vector<size_t> alloc_bytes;  // (parallel to allocs)
    int dev = 0;
    double *d = nullptr, *x = nullptr, *r = nullptr, *w = nullptr, *p = nullptr, *y = nullptr, *old_x = nullptr, *atwd = nullptr;
    double *P = nullptr, *Pt = nullptr, *rs = nullptr, *partial = nullptr, *live = nullptr, *Dm = nullptr, *sc = nullptr;
    int32_t* ord = nullptr;
    uint8_t* act = nullptr;
    Best* bpartial = nullptr;
    dim3 grid2, gred;
    int64_t st_outer = 0, st_cg = 0, st_it = 0;
    bool ok = true;

    template <class Tp>
    Tp* alloc(size_t count) {
        const size_t bytes = sizeof(Tp) * (count ? count : 1);
        void* p_ = pool().take(bytes, dev);
        if (!p_ && !SWOK(hipMalloc(&p_, bytes))) {
            (void)hipGetLastError();
            pool().flush();  // (what the pool holds may be what is missing)
            if (!SWOK(hipMalloc(&p_, bytes))) { ok = false; return nullptr; }
        }
        allocs.push_back(p_);
        alloc_bytes.push_back(bytes);
        return (Tp*)p_;
    }
    void release(void* p_) {  // one allocation back to the device (after the work queued on it)
        (void)hipStreamSynchronize(s);
        for (size_t i = 0; i < allocs.size(); i++)
            if (allocs[i] == p_) { pool().give(p_, alloc_bytes[i], dev); allocs.erase(allocs.begin() + (long)i); alloc_bytes.erase(alloc_bytes.begin() + (long)i); return; }
    }
    size_t block_mark = SIZE_MAX;  // allocs[block_mark ..) belong to the block active-set method
    void release_block_buffers() {
        (void)hipStreamSynchronize(s);
        if (block_mark == SIZE_MAX) return;
        while (allocs.size() > block_mark) { pool().give(allocs.back(), alloc_bytes.back(), dev); allocs.pop_back(); alloc_bytes.pop_back(); }
        ok = true;
    }
    ~Solver() {
        if (s) (void)hipStreamSynchronize(s);
        for (size_t i = 0; i < allocs.size(); i++) pool().give(allocs[i], alloc_bytes[i], dev);
        if (s) (void)hipStreamDestroy(s);
    }
    // 2-D inclusive prefix of src (upper triangle, zeros elsewhere) -> Pt (transposed)
    void prefix(const double* src, const double* skip) {
        hipLaunchKernelGGL(k_rowscan, dim3(n), dim3(T), 0, s, src, P, n, ld, skip);
        hipLaunchKernelGGL(k_transpose, dim3((n + 31) / 32, (n + 31) / 32), dim3(T), 0, s, P, Pt, n, ld, skip);
        hipLaunchKernelGGL(k_rowscan, dim3(n), dim3(T), 0, s, Pt, Pt, n, ld, skip);
    }
    void Ab(const double* b_, double* out, const double* skip = nullptr) {  // out = A b
        prefix(b_, skip);
        hipLaunchKernelGGL(k_ab, grid2, dim3(T), 0, s, Pt, out, n, ld, skip);
    }
    void Atx(const double* y_, double* out, const uint8_t* mask = nullptr, const double* skip = nullptr) {  // out = A^T y
        prefix(y_, skip);
        hipLaunchKernelGGL(k_scan1, dim3(1), dim3(1024), 0, s, Pt, rs, n, ld, skip);
        hipLaunchKernelGGL(k_atx, grid2, dim3(T), 0, s, Pt, rs, out, n, ld, mask, skip);
    }
    template <int OP>
    void vec(double* a, double* b, const double* c, const double* e, double s0 = 0.0, double s1 = 0.0, const double* scp = nullptr) {
        VecArgs g{a, b, c, e, act, s0, s1, n, ld, scp};
        hipLaunchKernelGGL(k_vec<OP>, grid2, dim3(T), 0, s, g);
    }
    void dot_partials(const double* a, const double* b, const double* skip) {
        hipLaunchKernelGGL(k_reduce<RD_DOT>, gred, dim3(T), 0, s, a, b, act, 0.0, n, ld, partial, bpartial, skip);
    }
    template <int RD>
    double reduce_sum(const double* a, const double* b, double s0 = 0.0) {
        hipLaunchKernelGGL(k_reduce<RD>, gred, dim3(T), 0, s, a, b, act, s0, n, ld, partial, bpartial, (const double*)nullptr);
        std::vector<double> h((size_t)gred.x * gred.y);
        (void)hipMemcpyAsync(h.data(), partial, sizeof(double) * h.size(), hipMemcpyDeviceToHost, s);
        (void)hipStreamSynchronize(s);
        double acc = 0.0;
        for (double v : h) acc += v;
        return acc;
    }
    template <int RD>
    Best reduce_best(const double* a, const double* b) {
        hipLaunchKernelGGL(k_reduce<RD>, gred, dim3(T), 0, s, a, b, act, 0.0, n, ld, partial, bpartial, (const double*)nullptr);
        std::vector<Best> h((size_t)gred.x * gred.y);
        (void)hipMemcpyAsync(h.data(), bpartial, sizeof(Best) * h.size(), hipMemcpyDeviceToHost, s);
        (void)hipStreamSynchronize(s);
        Best best{INFINITY, INT64_MAX};
        for (const Best& o : h)
            if (o.v < best.v || (o.v == best.v && o.k < best.k)) best = o;
        return best;
    }
    double min_negative(const double* a) {
        hipLaunchKernelGGL(k_minneg, gred, dim3(T), 0, s, a, n, ld, partial);
        std::vector<double> h((size_t)gred.x * gred.y);
        (void)hipMemcpyAsync(h.data(), partial, sizeof(double) * h.size(), hipMemcpyDeviceToHost, s);
        (void)hipStreamSynchronize(s);
        double mn = 0.0;
        for (double v : h) if (v < mn) mn = v;
        return mn;
    }

    // circularConjugateGrads (:769-831), W = 1.  The loop's scalars stay on the device; the host enqueues
    // CG_CHECK iterations at a time and reads the loop condition once per bunch (the kernels of an
    // iteration that comes after the end of the loop return at once, so the iteration count is exactly
    // the reference's).
    static constexpr int CG_CHECK = 16;
    double tol2 = -1.0;  // (CG_EPSILON * ||AtWd||)^2, constant over the solve
    void cg() {
        const int np = (int)(gred.x * gred.y);
        if (tol2 < 0.0) { const double e_0 = CG_EPSILON * std::sqrt(reduce_sum<RD_DOT>(atwd, atwd)); tol2 = e_0 * e_0; }
        double hsc[SC_WORDS] = {0};
        hsc[SC_TOL2] = tol2;
        hsc[SC_KMAX] = (double)((int64_t)n * (n - 1) / 2);
        (void)hipMemcpyAsync(sc, hsc, sizeof(hsc), hipMemcpyHostToDevice, s);
        (void)hipStreamSynchronize(s);  // (hsc is a stack buffer)
        Ab(x, y);
        Atx(y, r);
        vec<OP_R_INIT>(r, nullptr, atwd, nullptr);
        dot_partials(r, r, nullptr);
        hipLaunchKernelGGL(k_cg_scalar, dim3(1), dim3(T), 0, s, partial, np, sc, 0);
        for (;;) {
            for (int it = 0; it < CG_CHECK; it++) {
                vec<OP_P_UPDATE>(p, nullptr, r, nullptr, 0.0, 0.0, sc);
                Ab(p, y, sc);
                Atx(y, w, act, sc);
                dot_partials(p, w, sc);
                hipLaunchKernelGGL(k_cg_scalar, dim3(1), dim3(T), 0, s, partial, np, sc, 1);
                vec<OP_XR_UPDATE>(x, r, p, w, 0.0, 0.0, sc);
                dot_partials(r, r, sc);
                hipLaunchKernelGGL(k_cg_scalar, dim3(1), dim3(T), 0, s, partial, np, sc, 2);
            }
            (void)hipMemcpyAsync(hsc, sc, sizeof(hsc), hipMemcpyDeviceToHost, s);
            (void)hipStreamSynchronize(s);
            if (hsc[SC_DONE] != 0.0) break;
        }
        st_cg++;
        st_it += (int64_t)hsc[SC_K];
    }
    // worstIndices(x, 0.6) + contraction (:411-430).  Returns false if nothing is negative.
    bool contract_worst() {
        const int64_t num_neg = (int64_t)reduce_sum<RD_COUNT_NEG>(x, nullptr);
        if (num_neg == 0) return false;
        const int64_t nkept = (int64_t)std::ceil(0.6 * (double)num_neg);
        // cutoff = nkept-th smallest negative value.  count(x < t) is monotone in t; bisect over the
        // bit patterns of the negative doubles (larger pattern = more negative), at most 64 steps:
        // invariant count(x < val(lo)) < nkept <= count(x < val(hi)), lo more negative than hi
        auto val = [](uint64_t bits) { double v; std::memcpy(&v, &bits, 8); return v; };
        const double mn = min_negative(x);
        uint64_t lo, hi = 0x8000000000000000ULL;  // hi = -0.0: count(x < -0.0) = num_neg >= nkept
        std::memcpy(&lo, &mn, 8);                  // count(x < min) = 0 < nkept
        while (lo - hi > 1) {
            const uint64_t mid = hi + (lo - hi) / 2;
            const int64_t c = (int64_t)reduce_sum<RD_COUNT_LT>(x, nullptr, val(mid));
            if (c >= nkept) hi = mid; else lo = mid;
        }
        // no double lies strictly between val(lo) and val(hi): the nkept-th smallest value is val(lo).
        // Values below it all go; of the values equal to it the reference takes the first
        // nkept - count(<) by index - here they are taken together (the same unless equal negative
        // weights straddle the 60 % mark; the optimum reached does not depend on it)
        vec<OP_CONTRACT>(x, nullptr, nullptr, nullptr, val(lo), 1.0);
        return true;
    }

    // ------------------------------------------------------------ block active-set method (see the comment above h_entry)
    int64_t st_lh_steps = 0, st_lh_refactor = 0, st_solves = 0, st_adds = 0, st_dels = 0, st_rejects = 0, st_ratio_steps = 0, st_screened = 0, st_dead_wanting = 0, st_revived = 0, st_dels_new = 0;
    double t_ops = 0, t_sel = 0, t_append = 0, t_solve = 0, t_dead = 0, t_refactor = 0;  // host wall clock per phase (FNN_SW_LOG)
    struct Blk {
        rocblas_handle bh = nullptr;
        int64_t cap = 0, kmax = 0, rcap = 0, f = 0, r = 0, pw = 0;
        TriStore Ws{nullptr, 0, 0}, LCs{nullptr, 0, 0};  // the factor (column panels) and the Gram factor (one panel = square)
        double *W = nullptr, *B = nullptr, *Tb = nullptr, *X = nullptr, *S = nullptr, *S0 = nullptr, *Li = nullptr, *tmp2 = nullptr;
        double *Y = nullptr, *CR = nullptr, *LC = nullptr, *CRb = nullptr, *CRw = nullptr;
        double *z = nullptr, *v = nullptr, *xs = nullptr, *cK = nullptr, *lam = nullptr, *gF = nullptr;
        int2 *dF = nullptr, *dscr = nullptr;
        int32_t *dlist = nullptr, *dlist2 = nullptr;
        double *wk = nullptr, *sk1 = nullptr, *sk2 = nullptr;
        int64_t* dinfo = nullptr;
        double* ckey = nullptr; double* ckey2 = nullptr; int64_t* cidx = nullptr; int64_t* cidx2 = nullptr;
        unsigned long long* ccount = nullptr; void* sort_tmp = nullptr; size_t sort_bytes = 0; int64_t ccap = 0;
    } bk;
    bool blas_ok = true;
    double gemm_flops = 0.0;
    int64_t cap_want = 0;        // > 0: the factor's capacity for this attempt (a retry after FNN_SW_GIVEUP_CAPACITY), else the default for n
    int giveup = 0;              // FNN_SW_GIVEUP_* of the block method
    const char* giveup_text = "";
    double final_tol_rel = 0.0, t_alloc_s = 0.0;
    int64_t n_set_aside = 0, f_peak = 0, capacity = 0;
    bool have_atwd = false;
    static int64_t up64(int64_t v) { return (v + 63) / 64 * 64; }
    static dim3 g1(int64_t c) { return dim3((unsigned)((c + T - 1) / T)); }
    static dim3 g2(int64_t rows, int64_t cols) { return dim3((unsigned)((rows + T - 1) / T), (unsigned)std::max<int64_t>(1, std::min<int64_t>(cols, 16384))); }
    void gemm(rocblas_operation ta, rocblas_operation tb, int64_t m, int64_t nn, int64_t k, double alpha, const double* A, int64_t lda, const double* Bm,
              int64_t ldb, double beta, double* Cm, int64_t ldc) {
        if (m <= 0 || nn <= 0) return;
        gemm_flops += 2.0 * (double)m * (double)nn * (double)k;
        if (rocblas_dgemm_64(bk.bh, ta, tb, m, nn, k, &alpha, A, lda, Bm, ldb, &beta, Cm, ldc) != rocblas_status_success) blas_ok = false;
    }
    void gemv(rocblas_operation ta, int64_t m, int64_t nn, double alpha, const double* A, int64_t lda, const double* xv, double beta, double* yv) {
        if (m <= 0 || nn <= 0) return;
        if (rocblas_dgemv_64(bk.bh, ta, m, nn, &alpha, A, lda, xv, 1, &beta, yv, 1) != rocblas_status_success) blas_ok = false;
    }
    // panels that skip the zero half of a lower-triangular operand
    // out (f x k, ldo) = Wl (f x f lower triangular, panel-stored) * Bm (f x k, ldb): panel by panel, each adding into the rows it reaches
    void tri_times(const TriStore& Wl, int64_t f, const double* Bm, int64_t ldb, int64_t k, double* out, int64_t ldo) {
        const int64_t chunks = (f + SK_CHUNK - 1) / SK_CHUNK;
        if (k <= SMALLK && chunks * k * f <= bk.cap * bk.kmax) {  // a few columns: one pass over Wl (partial sums per column chunk in bk.X)
            (void)hipMemsetAsync(bk.X, 0, sizeof(double) * (size_t)(chunks * k * f), s);
            hipLaunchKernelGGL(k_tri_times_small, dim3((unsigned)((f + T - 1) / T), (unsigned)chunks), dim3(T), 0, s, Wl, f, Bm, ldb, (int)k, bk.X);
            hipLaunchKernelGGL(k_sum_parts, g2(f, k), dim3(T), 0, s, bk.X, k * f, (int)chunks, f, k, 1.0, 0.0, out, ldo);
            return;
        }
        // column chunks of about f / 8 (never across a storage panel): the zero triangle above a chunk's diagonal is the only waste
        const int64_t gw = std::min(Wl.pw, std::max<int64_t>(1024, up64((f + 7) / 8)));
        for (int64_t c0 = 0; c0 < f;) {
            const int64_t q = c0 / Wl.pw, c1 = std::min({c0 + gw, (q + 1) * Wl.pw, f});
            gemm(rocblas_operation_none, rocblas_operation_none, f - c0, k, c1 - c0, 1.0, Wl.at(c0, c0), Wl.ld(q), Bm + c0, ldb, c0 == 0 ? 0.0 : 1.0, out + c0, ldo);
            c0 = c1;
        }
    }
    // out (k x f, ldo) = Tm^T (Tm: f x k, ldt) * Wl (f x f lower triangular, panel-stored)
    void t_times_tri(const double* Tm, int64_t ldt, int64_t k, const TriStore& Wl, int64_t f, double* out, int64_t ldo) {
        if (k <= SMALLK) {
            hipLaunchKernelGGL(k_t_times_tri_small, dim3((unsigned)f), dim3(T), 0, s, Tm, ldt, (int)k, Wl, f, out, ldo);
            return;
        }
        const int64_t gw = std::min(Wl.pw, std::max<int64_t>(1024, up64((f + 7) / 8)));
        for (int64_t c0 = 0; c0 < f;) {
            const int64_t q = c0 / Wl.pw, c1 = std::min({c0 + gw, (q + 1) * Wl.pw, f});
            gemm(rocblas_operation_transpose, rocblas_operation_none, k, c1 - c0, f - c0, 1.0, Tm + c0, ldt, Wl.at(c0, c0), Wl.ld(q), 0.0, out + c0 * ldo, ldo);
            c0 = c1;
        }
    }
    // C (m x nn) = beta C + alpha A^T B with a long inner dimension kk (A: kk x m, B: kk x nn): rocBLAS has no split-K form of
    // this shape (3 TF/s); here the inner dimension is cut into slices that run as one strided-batched GEMM into partial
    // products (workspace bk.X), added up in slice order
    void tn_splitk(int64_t m, int64_t nn, int64_t kk, double alpha, const double* A, int64_t lda, const double* Bm, int64_t ldb, double beta, double* Cm,
                   int64_t ldc) {
        if (m <= 0 || nn <= 0) return;
        const int64_t room = (bk.cap * bk.kmax) / std::max<int64_t>(m * nn, 1);
        int64_t parts = std::min<int64_t>({room, (int64_t)32, (kk + 1023) / 1024});
        if (parts < 2 || m * nn > (int64_t)1 << 26) {
            gemm(rocblas_operation_transpose, rocblas_operation_none, m, nn, kk, alpha, A, lda, Bm, ldb, beta, Cm, ldc);
            return;
        }
        const int64_t slice = kk / parts, rest = kk - slice * parts;  // `parts` equal slices, the remainder as one more
        const double one = 1.0, zero = 0.0;
        if (rocblas_dgemm_strided_batched_64(bk.bh, rocblas_operation_transpose, rocblas_operation_none, m, nn, slice, &one, A, lda, slice, Bm, ldb, slice, &zero,
                                             bk.X, m, m * nn, parts) != rocblas_status_success) blas_ok = false;
        int np = (int)parts;
        if (rest > 0 && parts < room) {
            gemm(rocblas_operation_transpose, rocblas_operation_none, m, nn, rest, 1.0, A + slice * parts, lda, Bm + slice * parts, ldb, 0.0, bk.X + parts * m * nn, m);
            np++;
        } else if (rest > 0) {
            gemm(rocblas_operation_transpose, rocblas_operation_none, m, nn, rest, 1.0, A + slice * parts, lda, Bm + slice * parts, ldb, 1.0, bk.X + (parts - 1) * m * nn, m);
        }
        hipLaunchKernelGGL(k_sum_parts, g2(m, nn), dim3(T), 0, s, bk.X, m * nn, np, m, nn, alpha, beta, Cm, ldc);
    }
    // Li = inverse of the Cholesky factor of S (m x m, lower triangle read, destroyed), zeros above the diagonal: recursive halves,
    // GEMMs and the in-LDS kernel for blocks of <= 64.  Everything is enqueued; a block that is not positive definite leaves the
    // 1-based position of its first bad pivot in *bk.dinfo (read it with invchol_info() after the call).
    void invchol_rec(double* S, int64_t lds, double* Li, int64_t ldl, int64_t m, int64_t off) {
        if (m <= TRI_NB) {
            hipLaunchKernelGGL(k_invchol_small, dim3(1), dim3(TRI_NB), 0, s, S, lds, Li, ldl, (int)m, (long long)off, (long long*)bk.dinfo);
            return;
        }
        const int64_t m1 = std::min(m - 1, up64((m + 1) / 2)), m2 = m - m1;
        double *S21 = S + m1, *S22 = S + m1 + m1 * lds, *Li21 = Li + m1, *Li22 = Li + m1 + m1 * ldl;
        double *t1 = bk.tmp2, *t2 = bk.tmp2 + m2 * m1;
        invchol_rec(S, lds, Li, ldl, m1, off);
        gemm(rocblas_operation_none, rocblas_operation_transpose, m2, m1, m1, 1.0, S21, lds, Li, ldl, 0.0, t1, m2);           // L21 = S21 Li11^T
        const double mone = -1.0, one = 1.0;
        if (rocblas_dsyrk_64(bk.bh, rocblas_fill_lower, rocblas_operation_none, m2, m1, &mone, t1, m2, &one, S22, lds) != rocblas_status_success) blas_ok = false;
        gemm(rocblas_operation_none, rocblas_operation_none, m2, m1, m1, 1.0, t1, m2, Li, ldl, 0.0, t2, m2);                    // L21 Li11
        invchol_rec(S22, lds, Li22, ldl, m2, off + m1);
        gemm(rocblas_operation_none, rocblas_operation_none, m2, m1, m2, -1.0, Li22, ldl, t2, m2, 0.0, Li21, ldl);              // Li21 = -Li22 L21 Li11
        hipLaunchKernelGGL(k_fill2d, g2(m1, m2), dim3(T), 0, s, Li + m1 * ldl, ldl, m1, m2, 0.0);
    }
    int64_t invchol(double* S, int64_t lds, double* Li, int64_t ldl, int64_t m) {  // 0, or the 1-based position of the first pivot that is not positive
        if (m <= 0) return 0;
        const int64_t big = INT64_MAX;
        (void)hipMemcpyAsync(bk.dinfo, &big, sizeof(big), hipMemcpyHostToDevice, s);
        (void)hipStreamSynchronize(s);
        invchol_rec(S, lds, Li, ldl, m, 0);
        int64_t info = 0;
        (void)hipMemcpyAsync(&info, bk.dinfo, sizeof(info), hipMemcpyDeviceToHost, s);
        (void)hipStreamSynchronize(s);
        return info == INT64_MAX ? 0 : info;
    }
    double wall() { (void)hipStreamSynchronize(s); return fnn::now_s(); }
    std::map<std::string, double> tsub;  // finer split of the phases (FNN_SW_LOG)
    double t_lap = 0.0;
    bool lap_on = false;
    void lap(const char* name) {
        if (!lap_on) return;
        const double t = wall();
        if (name) tsub[name] += t - t_lap;
        t_lap = t;
    }

    bool block_active_set() {
        const bool log = std::getenv("FNN_SW_LOG") != nullptr;
        lap_on = log;
        auto envd = [](const char* k, double dflt) { const char* e = std::getenv(k); return e ? std::atof(e) : dflt; };
        const double kfrac = envd("FNN_SW_KFRAC", 0.20);
        double rfrac = envd("FNN_SW_RFRAC", 0.15);
        const int rad = (int)envd("FNN_SW_NMS", 3);
        const int64_t N = (int64_t)n * (n - 1) / 2;
        Blk& b = bk;
        b = Blk{};  // (a retry with a larger capacity starts from nothing)
        giveup = 0; giveup_text = ""; blas_ok = true; ok = true;
        block_mark = allocs.size();
        // Capacity of the factor (splits that are in + the departed ones still inside + the entering block): random distances end
        // with ~2.4 n positive splits, tree-like ones with more (tree + 5 % noise: 3.8 n) - as many as device memory allows, up
        // to 6 n: the factor takes cap^2 doubles, the block buffers, the departed columns and their Gram matrices 0.61 cap^2 more.
        auto sized = [&](double factor, int kdiv, int rdiv) {
            b.cap = up64(std::min<int64_t>(N, std::max<int64_t>({(int64_t)(factor * n) + 1024, std::min<int64_t>(8 * (int64_t)n + 64, 20000), 512})));
            if (cap_want > 0) b.cap = up64(std::min<int64_t>(N, cap_want));
            b.kmax = std::min<int64_t>({b.cap, std::max<int64_t>(64, up64(b.cap / kdiv)), (int64_t)8192});  // (a block costs 4 k^2 f beside its 4 k f^2: keep k / f small)
            b.rcap = std::min<int64_t>(b.cap, up64(b.cap / rdiv) + 64);
            b.pw = std::max<int64_t>(1024, up64(b.cap / 8));
            if (const char* e = std::getenv("FNN_SW_PANEL")) { const int64_t v = std::atoll(e); if (v >= 64) b.pw = up64(v); }
            return 8.0 * ((double)TriStore::elems(b.cap, b.pw) + 3.0 * (double)b.cap * b.kmax + (double)b.cap * b.rcap + 4.0 * (double)b.rcap * b.rcap + 3.0 * (double)b.kmax * b.kmax +
                          0.5 * (double)std::max(b.rcap, b.kmax) * std::max(b.rcap, b.kmax)) + 64.0 * (double)b.cap + 48.0 * (double)std::max<int64_t>(1 << 16, N / 16 + 1024);
        };
        size_t free_b = 0, total_b = 0;
        if (!SWOK(hipMemGetInfo(&free_b, &total_b))) free_b = 0;
        free_b += pool().pooled(dev);  // (handed out again to this very call, or flushed if the sizes differ)
        const double budget = 0.92 * (double)free_b;
        const double want = envd("FNN_SW_CAP", 0.0);
        // {capacity / n, block = capacity / kdiv, departed columns = capacity / rdiv}: the roomy shapes first; where memory is
        // short (32768 taxa) a shape with smaller side buffers buys capacity (4 n instead of 3.5 n)
        struct Shape { double f; int kdiv, rdiv; };
        const Shape shapes[] = {{6.0, 12, 5}, {5.0, 12, 5}, {4.5, 12, 5}, {4.0, 12, 5}, {4.0, 16, 8}, {3.5, 12, 5}, {3.25, 12, 5}};
        Shape pick = shapes[6];
        if (cap_want > 0) {
            // a retry with a larger factor: the wanted capacity with the roomiest side buffers that fit, else as much as fits
            const Shape tries[] = {{0.0, 12, 5}, {0.0, 16, 8}, {0.0, 24, 12}};
            bool fits = false;
            for (int round = 0; round < 40 && !fits; round++) {
                for (const Shape& sh : tries) { pick = sh; if (sized(0.0, sh.kdiv, sh.rdiv) <= budget) { fits = true; break; } }
                if (!fits) cap_want = (int64_t)(0.9 * (double)cap_want);
            }
            if (!fits) { giveup = FNN_SW_GIVEUP_SETUP; giveup_text = "no factor fits the device memory"; return false; }
        } else {
        for (const Shape& sh : shapes) {
            pick = sh;
            if (want > 0.0) { pick.f = want; break; }
            if (sized(sh.f, sh.kdiv, sh.rdiv) <= budget) break;
        }
        }
        (void)sized(pick.f, pick.kdiv, pick.rdiv);
        // (the departed columns' buffers bound how many may stay in the factor: up to this share of it whatever its size, and beyond
        //  that share as long as the buffers keep room for one more block's departures - the rebuild rule in the loop below)
        const double rfrac_buf = 0.75 * (double)b.rcap / (double)b.cap;
        if (log) std::fprintf(stderr, "  [sw] capacity %lld splits (%.2f n), block <= %lld, departed <= %lld; device memory free %.1f GB\n", (long long)b.cap,
                              (double)b.cap / n, (long long)b.kmax, (long long)b.rcap, (double)free_b * 1e-9);
        capacity = b.cap;
        if (rocblas_create_handle(&b.bh) != rocblas_status_success) { giveup = FNN_SW_GIVEUP_SETUP; giveup_text = "rocblas_create_handle failed"; return false; }
        struct HandleGuard { rocblas_handle h; ~HandleGuard() { rocblas_destroy_handle(h); } } guard{b.bh};
        rocblas_set_stream(b.bh, s);
        rocblas_set_pointer_mode(b.bh, rocblas_pointer_mode_host);
        const double t_alloc0 = wall();
        b.W = alloc<double>((size_t)TriStore::elems(b.cap, b.pw));
        b.Ws = TriStore{b.W, b.cap, b.pw};
        b.B = alloc<double>((size_t)b.cap * b.kmax); b.Tb = alloc<double>((size_t)b.cap * b.kmax); b.X = alloc<double>((size_t)b.cap * b.kmax);
        b.S = alloc<double>((size_t)b.kmax * b.kmax); b.S0 = alloc<double>((size_t)b.kmax * b.kmax); b.Li = alloc<double>((size_t)b.kmax * b.kmax);
        b.tmp2 = alloc<double>((size_t)(std::max(b.rcap, b.kmax) * std::max(b.rcap, b.kmax) / 2 + 4096));
        b.Y = alloc<double>((size_t)b.cap * b.rcap);
        b.CR = alloc<double>((size_t)b.rcap * b.rcap); b.LC = alloc<double>((size_t)b.rcap * b.rcap);
        b.LCs = TriStore{b.LC, b.rcap, b.rcap}; b.CRb = alloc<double>((size_t)b.rcap * b.rcap); b.CRw = alloc<double>((size_t)b.rcap * b.rcap);
        b.z = alloc<double>((size_t)b.cap); b.v = alloc<double>((size_t)b.cap); b.xs = alloc<double>((size_t)b.cap); b.cK = alloc<double>((size_t)b.cap);
        b.lam = alloc<double>((size_t)b.rcap); b.gF = alloc<double>((size_t)b.cap);
        b.dF = alloc<int2>((size_t)b.cap); b.dscr = alloc<int2>((size_t)b.cap); b.dlist = alloc<int32_t>((size_t)b.cap); b.dlist2 = alloc<int32_t>((size_t)b.cap);
        b.wk = alloc<double>((size_t)b.cap); b.sk1 = alloc<double>((size_t)b.cap); b.sk2 = alloc<double>((size_t)b.cap); b.dinfo = alloc<int64_t>(1);
        b.ccap = std::min<int64_t>(N, std::max<int64_t>(1 << 16, N / 16 + 1024));
        b.ckey = alloc<double>((size_t)b.ccap); b.ckey2 = alloc<double>((size_t)b.ccap);
        b.cidx = alloc<int64_t>((size_t)b.ccap); b.cidx2 = alloc<int64_t>((size_t)b.ccap);
        b.ccount = alloc<unsigned long long>(1);
        if (hipcub::DeviceRadixSort::SortPairsDescending(nullptr, b.sort_bytes, b.ckey, b.ckey2, b.cidx, b.cidx2, (int)std::min<int64_t>(b.ccap, INT32_MAX), 0,
                                                         64, s) != hipSuccess) { giveup = FNN_SW_GIVEUP_SETUP; giveup_text = "hipcub size query failed"; return false; }
        b.sort_tmp = alloc<uint8_t>(b.sort_bytes + 16);
        if (!ok) { giveup = FNN_SW_GIVEUP_SETUP; giveup_text = "device allocation of the factor's buffers failed"; return false; }
        t_alloc_s = wall() - t_alloc0;
        if (log) std::fprintf(stderr, "  [sw] buffers allocated in %.2f s\n", t_alloc_s);

        // host state: the factor's splits in factor order, their weights, who left
        std::vector<int2> F;
        std::vector<double> xw;         // weight per factor position (0 for the splits that left)
        std::vector<double> cF;         // c = A^T d at the factor positions
        std::vector<uint8_t> dead;
        std::vector<int32_t> deadlist;  // factor positions of the splits that left, in the order of Y's columns
        std::vector<int2> banned;
        double phi = 0.0;               // objective - c_F . x_F / 2 of the current point (a sub-problem's exact minimiser)

        Atx(d, atwd);  // c = A^T d
        (void)hipMemsetAsync(act, 0, (size_t)n * (size_t)ld, s);
        double cmax = 0.0;
        { const Best bb = reduce_best<RD_MAX_UNMASKED>(atwd, nullptr); cmax = bb.k == INT64_MAX ? 0.0 : -bb.v; }
        double tol = 1e-12 * (cmax > 0.0 ? cmax : 1.0);  // multipliers above it are candidates (raised to the measured noise floor, below)
        const double tol_cap = 1e-10 * (cmax > 0.0 ? cmax : 1.0);

        auto upload_F = [&](int64_t from, int64_t to) {
            if (to > from) (void)hipMemcpyAsync(b.dF + from, F.data() + from, sizeof(int2) * (size_t)(to - from), hipMemcpyHostToDevice, s);
        };
        auto set_mask = [&](const std::vector<int2>& list, uint8_t v) {  // (through the scratch list: the factor's own list stays as it is)
            if (list.empty()) return;
            (void)hipMemcpyAsync(b.dscr, list.data(), sizeof(int2) * list.size(), hipMemcpyHostToDevice, s);
            hipLaunchKernelGGL(k_mask, g1((int64_t)list.size()), dim3(T), 0, s, b.dscr, (int64_t)list.size(), act, ld, v);
            (void)hipStreamSynchronize(s);
        };
        // LC = inverse of the Cholesky factor of the Gram matrix CR of the departed columns (lower triangular), from scratch
        auto factor_gram = [&]() -> bool {
            if (b.r == 0) return true;
            hipLaunchKernelGGL(k_copy_lower, g2(b.r, b.r), dim3(T), 0, s, b.CR, b.rcap, b.CRw, b.rcap, b.r);
            return invchol(b.CRw, b.rcap, b.LC, b.rcap, b.r) == 0;
        };
        // The core of an append, shared by the factor W (rows for splits that enter) and by LC (rows for columns that depart).
        // phase 1: T = Wi Bc, S = S - T^T T for a lower-triangular inverse factor Wi (f x f); phase 2: rows f .. f + k of Wi
        // from T, the Cholesky factor's inverse Li of S:  [-Li T^T Wi | Li], zeros above.
        auto rows_phase1 = [&](const TriStore& Wi, int64_t f, const double* Bc, int64_t ldb, int64_t k, double* Tm, int64_t ldt, double* Sm, int64_t lds) {
            if (f == 0) return;
            tri_times(Wi, f, Bc, ldb, k, Tm, ldt);
            tn_splitk(k, k, f, -1.0, Tm, ldt, Tm, ldt, 1.0, Sm, lds);
        };
        auto rows_phase2 = [&](const TriStore& Wi, int64_t f, const double* Tm, int64_t ldt, const double* Lim, int64_t ldl, int64_t k) {
            if (f > 0) {
                t_times_tri(Tm, ldt, k, Wi, f, b.X, b.kmax);  // X = T^T Wi
                for (int64_t q = 0, q0 = 0; q0 < f; q++, q0 += Wi.pw) {  // new rows -Li X, panel by panel
                    const int64_t w = std::min(Wi.pw, f - q0);
                    gemm(rocblas_operation_none, rocblas_operation_none, k, w, k, -1.0, Lim, ldl, b.X + q0 * b.kmax, b.kmax, 0.0, Wi.panel(q) + (f - q0), Wi.ld(q));
                }
            }
            for (int64_t c0 = f; c0 < f + k;) {  // the new columns: zeros above (inside their panel), then Li
                const int64_t q = c0 / Wi.pw, q0 = q * Wi.pw, c1 = std::min(f + k, q0 + Wi.pw);
                if (f > q0)
                    hipLaunchKernelGGL(k_fill2d, g2(f - q0, c1 - c0), dim3(T), 0, s, Wi.at(q0, c0), Wi.ld(q), f - q0, c1 - c0, 0.0);
                // rows [max(f, q0), f + k) of the columns c0 .. c1 (the rows of Li above a panel that starts inside the block are its zeros)
                const int64_t r0 = std::max(f, q0);
                hipLaunchKernelGGL(k_copy2d, g2(f + k - r0, c1 - c0), dim3(T), 0, s, Lim + (c0 - f) * ldl + (r0 - f), ldl, Wi.at(r0, c0), Wi.ld(q), f + k - r0, c1 - c0);
                c0 = c1;
            }
        };
        // append the k splits at b.dF[f .. f + k) with multipliers wK (device, may be null: no screening).  Screening: the block's
        // weights in the joint sub-problem are S^-1 w_K with S the block's Schur complement - known before the rows are formed;
        // splits whose weight would not be positive are dropped here, at half the price and without ever entering the factor.
        // `kept` receives the indices (into the block) of the splits that entered; returns their number, -1 on failure.
        auto append = [&](int64_t k, const double* wK, std::vector<int32_t>& kept) -> int64_t {
            const double t0 = log ? wall() : 0.0;
            const int64_t f = b.f;
            int2* dK = b.dF + f;
            kept.resize((size_t)k);
            for (int64_t q = 0; q < k; q++) kept[(size_t)q] = (int32_t)q;
            lap(nullptr);
            hipLaunchKernelGGL(k_hblock, g2(k, k), dim3(T), 0, s, dK, k, dK, k, n, b.S, b.kmax, 0);
            if (f > 0) hipLaunchKernelGGL(k_hblock, g2(f, k), dim3(T), 0, s, b.dF, f, dK, k, n, b.B, b.cap, 0);
            lap("append.hblock");
            rows_phase1(b.Ws, f, b.B, b.cap, k, b.Tb, b.cap, b.S, b.kmax);  // T = L^-1 B = L21^T, S = H_KK - L21 L21^T
            lap("append.W*B+syrk");
            hipLaunchKernelGGL(k_copy2d, g2(k, k), dim3(T), 0, s, b.S, b.kmax, b.S0, b.kmax, k, k);
            const double* wcur = wK;
            std::vector<double> sK;
            bool changed = false;
            for (int round = 0;; round++) {
                const int64_t kk = (int64_t)kept.size();
                const int64_t info = invchol(b.S, b.kmax, b.Li, b.kmax, kk);  // Li = L22^-1
                if (info > 0) {  // not positive definite at split `info`: the block ends in front of it
                    kept.resize((size_t)(info - 1)); changed = true;
                    if (kept.empty()) return 0;
                    (void)hipMemcpyAsync(b.dlist2, kept.data(), sizeof(int32_t) * kept.size(), hipMemcpyHostToDevice, s);
                    hipLaunchKernelGGL(k_gather_sym, g2((int64_t)kept.size(), (int64_t)kept.size()), dim3(T), 0, s, b.S0, b.kmax, b.dlist2, (int64_t)kept.size(), b.S, b.kmax);
                    if (wK) { hipLaunchKernelGGL(k_gather_vec, g1((int64_t)kept.size()), dim3(T), 0, s, wK, b.dlist2, (int64_t)kept.size(), b.wk); wcur = b.wk; }
                    (void)hipStreamSynchronize(s);
                    continue;
                }
                if (!wK || round >= 12) break;
                // the block's weights in the joint sub-problem: S^-1 w_K = Li^T (Li w_K)
                gemv(rocblas_operation_none, kk, kk, 1.0, b.Li, b.kmax, wcur, 0.0, b.sk1);
                gemv(rocblas_operation_transpose, kk, kk, 1.0, b.Li, b.kmax, b.sk1, 0.0, b.sk2);
                sK.resize((size_t)kk);
                (void)hipMemcpyAsync(sK.data(), b.sk2, sizeof(double) * (size_t)kk, hipMemcpyDeviceToHost, s);
                (void)hipStreamSynchronize(s);
                std::vector<int32_t> nk;
                for (int64_t q = 0; q < kk; q++) if (sK[(size_t)q] > 0.0) nk.push_back(kept[(size_t)q]);
                if ((int64_t)nk.size() == kk) break;
                st_screened += kk - (int64_t)nk.size();
                if (nk.empty()) nk.push_back(kept[0]);  // (a single split with a positive multiplier always gets a positive weight)
                kept.swap(nk); changed = true;
                (void)hipMemcpyAsync(b.dlist2, kept.data(), sizeof(int32_t) * kept.size(), hipMemcpyHostToDevice, s);
                hipLaunchKernelGGL(k_gather_sym, g2((int64_t)kept.size(), (int64_t)kept.size()), dim3(T), 0, s, b.S0, b.kmax, b.dlist2, (int64_t)kept.size(), b.S, b.kmax);
                hipLaunchKernelGGL(k_gather_vec, g1((int64_t)kept.size()), dim3(T), 0, s, wK, b.dlist2, (int64_t)kept.size(), b.wk);
                wcur = b.wk;
                (void)hipStreamSynchronize(s);
            }
            lap("append.potrf+trtri+screen");
            k = (int64_t)kept.size();
            const double* Tuse = b.Tb;
            if (changed) {  // T's columns and the split list follow the kept ones (b.B is free now)
                (void)hipMemcpyAsync(b.dlist2, kept.data(), sizeof(int32_t) * kept.size(), hipMemcpyHostToDevice, s);
                if (f > 0) { hipLaunchKernelGGL(k_gather_cols, g2(f, k), dim3(T), 0, s, b.Tb, b.cap, f, b.dlist2, k, b.B, b.cap); Tuse = b.B; }
                hipLaunchKernelGGL(k_gather_int2, g1(k), dim3(T), 0, s, dK, b.dlist2, k, b.dscr);
                (void)hipMemcpyAsync(dK, b.dscr, sizeof(int2) * (size_t)k, hipMemcpyDeviceToDevice, s);
                (void)hipStreamSynchronize(s);
            }
            hipLaunchKernelGGL(k_gather, g1(k), dim3(T), 0, s, dK, k, atwd, ld, b.cK);
            rows_phase2(b.Ws, f, Tuse, b.cap, b.Li, b.kmax, k);
            if (f > 0) gemv(rocblas_operation_transpose, f, k, -1.0, Tuse, b.cap, b.z, 1.0, b.cK);  // c_K - L21 z
            gemv(rocblas_operation_none, k, k, 1.0, b.Li, b.kmax, b.cK, 0.0, b.z + f);                // z_K = L22^-1 (c_K - L21 z)
            lap("append.rows");
            if (b.r > 0) {  // the departed columns grow by the new rows; so does their Gram matrix
                hipLaunchKernelGGL(k_gather_rows_tri, g2(k, b.r), dim3(T), 0, s, b.Ws, f, k, b.dlist, b.r, b.Y, b.cap);
                const double one = 1.0;
                if (rocblas_dsyrk_64(b.bh, rocblas_fill_lower, rocblas_operation_transpose, b.r, k, &one, b.Y + f, b.cap, &one, b.CR, b.rcap) != rocblas_status_success)
                    blas_ok = false;
                if (!factor_gram()) return -1;
                lap("append.gram");
            }
            b.f = f + k;
            if (log) t_append += wall() - t0;
            return blas_ok ? k : -1;
        };
        // rebuild the factor from H's closed form for the splits that are still in: the compacted list appended in blocks
        auto refactor = [&]() -> bool {
            const double t0 = log ? wall() : 0.0;
            st_lh_refactor++;
            if (log) {  // where in the factor did the departed splits sit?
                std::vector<double> pos;
                for (size_t p = 0; p < F.size(); p++) if (dead[p]) pos.push_back((double)p / (double)F.size());
                if (!pos.empty())
                    std::fprintf(stderr, "  [sw] rebuild at %zu with %zu departed; their positions: 10%% %.2f 25%% %.2f 50%% %.2f 75%% %.2f\n", F.size(), pos.size(),
                                 pos[pos.size() / 10], pos[pos.size() / 4], pos[pos.size() / 2], pos[pos.size() * 3 / 4]);
            }
            std::vector<int2> gone;
            size_t q = 0;
            for (size_t p = 0; p < F.size(); p++) {
                if (dead[p]) { gone.push_back(F[p]); continue; }
                F[q] = F[p]; xw[q] = xw[p]; cF[q] = cF[p]; q++;
            }
            F.resize(q); xw.resize(q); cF.resize(q); dead.assign(q, 0); deadlist.clear();
            set_mask(gone, 0);  // they may enter again
            b.f = 0; b.r = 0;
            if (q == 0) return true;
            upload_F(0, (int64_t)q);
            std::vector<int32_t> kept;
            while (b.f < (int64_t)q) {
                const int64_t k = std::min<int64_t>({b.kmax, (int64_t)4096, (int64_t)q - b.f});
                if (append(k, nullptr, kept) != k) return false;
            }
            (void)hipStreamSynchronize(s);
            if (log) { const double dt = wall() - t0; t_refactor += dt; t_append -= dt; }
            return blas_ok;
        };
        // the splits at the factor positions `idxs` leave: their columns join Y, the Gram matrix and its factor grow
        auto depart = [&](const std::vector<int32_t>& idxs) -> bool {
            const double t0 = log ? wall() : 0.0;
            const int64_t nn = (int64_t)idxs.size(), r = b.r, f = b.f;
            lap(nullptr);
            (void)hipMemcpyAsync(b.dlist + r, idxs.data(), sizeof(int32_t) * (size_t)nn, hipMemcpyHostToDevice, s);
            hipLaunchKernelGGL(k_gather_cols_tri, g2(f, nn), dim3(T), 0, s, b.Ws, f, b.dlist + r, nn, b.Y + r * b.cap, b.cap);
            // CR[r : r + nn, 0 : r + nn] = YN^T [Y, YN]
            tn_splitk(nn, r + nn, f, 1.0, b.Y + r * b.cap, b.cap, b.Y, b.cap, 0.0, b.CR + r, b.rcap);
            (void)hipStreamSynchronize(s);  // (idxs may be a temporary)
            lap("depart.gather+gram");
            for (int32_t p : idxs) { dead[(size_t)p] = 1; xw[(size_t)p] = 0.0; deadlist.push_back(p); }
            bool fine = true;
            if (r == 0 || nn > b.kmax) { b.r = r + nn; fine = factor_gram(); }
            else {  // border the inverse factor: cross block CR[r:, 0:r]^T, diagonal block CR[r:, r:]
                hipLaunchKernelGGL(k_transpose_small, g2(nn, r), dim3(T), 0, s, b.CR + r, b.rcap, nn, r, b.B, b.cap);  // (r x nn)
                hipLaunchKernelGGL(k_copy2d, g2(nn, nn), dim3(T), 0, s, b.CR + r + r * b.rcap, b.rcap, b.S, b.kmax, nn, nn);
                rows_phase1(b.LCs, r, b.B, b.cap, nn, b.Tb, b.cap, b.S, b.kmax);
                if (invchol(b.S, b.kmax, b.Li, b.kmax, nn) != 0) { b.r = r + nn; fine = factor_gram(); }
                else {
                    rows_phase2(b.LCs, r, b.Tb, b.cap, b.Li, b.kmax, nn);
                    b.r = r + nn;
                }
            }
            lap("depart.gram_factor");
            if (log) t_dead += wall() - t0;
            return fine && blas_ok;
        };
        // Departed splits (factor positions `rev`) come BACK: a split that left is only constrained to zero - its row is still in the
        // factor - so un-constraining it costs the compaction of Y's columns and of the Gram matrix and a fresh Gram factor
        // (r^3, milliseconds), where the route through a rebuild and the candidates re-appends it at f^2 per split.
        auto revive = [&](const std::vector<int32_t>& rev) -> bool {
            const double t0 = log ? wall() : 0.0;
            const int64_t r = b.r, f = b.f;
            std::vector<uint8_t> back((size_t)f, 0);
            for (int32_t p : rev) { back[(size_t)p] = 1; dead[(size_t)p] = 0; }
            std::vector<int32_t> keepcols, newdead;
            for (int64_t q = 0; q < r; q++)
                if (!back[(size_t)deadlist[(size_t)q]]) { keepcols.push_back((int32_t)q); newdead.push_back(deadlist[(size_t)q]); }
            const int64_t rn = (int64_t)keepcols.size();
            st_revived += r - rn;
            if (rn > 0) {
                (void)hipMemcpyAsync(b.dlist2, keepcols.data(), sizeof(int32_t) * (size_t)rn, hipMemcpyHostToDevice, s);
                hipLaunchKernelGGL(k_gather_sym, g2(rn, rn), dim3(T), 0, s, b.CR, b.rcap, b.dlist2, rn, b.CRw, b.rcap);
                hipLaunchKernelGGL(k_copy2d, g2(rn, rn), dim3(T), 0, s, b.CRw, b.rcap, b.CR, b.rcap, rn, rn);
                // (a kept column's new index is never larger than its old one: chunks of new columns can be written in place)
                for (int64_t c0 = 0; c0 < rn; c0 += b.kmax) {
                    const int64_t cnt = std::min<int64_t>(b.kmax, rn - c0);
                    hipLaunchKernelGGL(k_gather_cols, g2(f, cnt), dim3(T), 0, s, b.Y, b.cap, f, b.dlist2 + c0, cnt, b.B, b.cap);
                    hipLaunchKernelGGL(k_copy2d, g2(f, cnt), dim3(T), 0, s, b.B, b.cap, b.Y + c0 * b.cap, b.cap, f, cnt);
                }
                (void)hipMemcpyAsync(b.dlist, newdead.data(), sizeof(int32_t) * (size_t)rn, hipMemcpyHostToDevice, s);
                (void)hipStreamSynchronize(s);  // (the host vectors are temporaries)
            }
            deadlist.swap(newdead);
            b.r = rn;
            const bool fine = factor_gram();
            if (log) t_dead += wall() - t0;
            return fine && blas_ok;
        };
        // minimiser of the sub-problem on the splits that are in: xs = W^T (z - Y (Y^T Y)^-1 Y^T z), copied to `out`
        auto solve = [&](std::vector<double>& out) -> bool {
            const double t0 = log ? wall() : 0.0;
            st_solves++;
            const int64_t f = b.f, r = b.r;
            out.assign((size_t)f, 0.0);
            if (f == 0) return true;
            const double* vv = b.z;
            if (r > 0) {
                gemv(rocblas_operation_transpose, f, r, 1.0, b.Y, b.cap, b.z, 0.0, b.lam);
                gemv(rocblas_operation_none, r, r, 1.0, b.LC, b.rcap, b.lam, 0.0, b.sk1);       // (Y^T Y)^-1 = LC^T LC
                gemv(rocblas_operation_transpose, r, r, 1.0, b.LC, b.rcap, b.sk1, 0.0, b.lam);
                (void)hipMemcpyAsync(b.v, b.z, sizeof(double) * (size_t)f, hipMemcpyDeviceToDevice, s);
                gemv(rocblas_operation_none, f, r, -1.0, b.Y, b.cap, b.lam, 1.0, b.v);
                vv = b.v;
            }
            for (int64_t q = 0, q0 = 0; q0 < f; q++, q0 += b.Ws.pw) {
                const int64_t w = std::min(b.Ws.pw, f - q0);
                gemv(rocblas_operation_transpose, f - q0, w, 1.0, b.Ws.panel(q), b.Ws.ld(q), vv + q0, 0.0, b.xs + q0);
            }
            (void)hipMemcpyAsync(out.data(), b.xs, sizeof(double) * (size_t)f, hipMemcpyDeviceToHost, s);
            (void)hipStreamSynchronize(s);
            for (size_t p = 0; p < (size_t)f; p++) if (dead[p]) out[p] = 0.0;
            if (log) t_solve += wall() - t0;
            return blas_ok;
        };
        auto objective = [&](const std::vector<double>& xs) {
            double acc = 0.0;
            for (size_t p = 0; p < xs.size(); p++) if (!dead[p]) acc += cF[p] * xs[p];
            return -0.5 * acc;
        };
        std::vector<double> sbuf;
        // every weight that is not positive leaves, until the sub-problem's minimiser is feasible (false: out of room / failure)
        bool over_rcap = false;  // settle_all failed for lack of room for the departed columns (not a numerical failure)
        int64_t f_step0 = 0;     // (statistics: factor size before this step's block entered)
        auto settle_all = [&]() -> bool {
            over_rcap = false;
            for (;;) {
                if (!solve(sbuf)) return false;
                std::vector<int32_t> out;
                for (size_t p = 0; p < sbuf.size(); p++) if (!dead[p] && !(sbuf[p] > 0.0)) out.push_back((int32_t)p);
                if (out.empty()) return true;
                if (b.r + (int64_t)out.size() > b.rcap) { over_rcap = true; return false; }
                st_dels += (int64_t)out.size();
                for (int32_t p : out) st_dels_new += p >= f_step0 ? 1 : 0;
                if (!depart(out)) return false;
            }
        };

        const int64_t kmin = std::max<int64_t>(8, n / 32);
        int64_t k_limit = b.kmax;
        bool ratio_mode = false, done = false, good = ok, fresh = false, banned_rechecked = false;
        // FNN_SW_REVIVE: 0 = departed splits only come back through a rebuild (round 3), 1 = at the start of every step, 2 = only when no
        // candidate is left (the case that used to force a rebuild "to look again")
        const int revive_mode = (int)envd("FNN_SW_REVIVE", 2.0);
        const bool revive_on = revive_mode != 0;
        const int64_t revive_min_f = (int64_t)envd("FNN_SW_REVIVE_MINF", 0.0);
        bool revive_blocked = false;   // a step that only brought departed splits back did not descend: not again before a step moves
        bool force_rebuild = false;    // a step ran out of room for its departed columns (taken back): rebuild before the next one

        auto give = [&](int why, const char* text) { giveup = why; giveup_text = text; good = false; };
        int stall = 0;  // steps in a row whose descent is not measurable (below 1e-13 |objective|)
        const int64_t max_outer = 40 * (int64_t)n + 1000;
        while (good && !done && st_lh_steps < max_outer) {
            st_lh_steps++;
            f_step0 = INT64_MAX;
            // ---- multipliers: r = c - A^T A x on the grid; the entries at the factor's splits go to the host
            double t0 = log ? wall() : 0.0;
            (void)hipMemsetAsync(this->x, 0, sizeof(double) * (size_t)n * (size_t)ld, s);
            if (b.f) {
                (void)hipMemcpyAsync(b.xs, xw.data(), sizeof(double) * (size_t)b.f, hipMemcpyHostToDevice, s);
                hipLaunchKernelGGL(k_scatter, g1(b.f), dim3(T), 0, s, b.dF, b.f, b.xs, this->x, ld);
            }
            Ab(this->x, y);
            Atx(y, r);
            std::vector<double> gF((size_t)b.f);
            if (b.f) {
                hipLaunchKernelGGL(k_gather, g1(b.f), dim3(T), 0, s, b.dF, b.f, r, ld, b.gF);
                (void)hipMemcpyAsync(gF.data(), b.gF, sizeof(double) * (size_t)b.f, hipMemcpyDeviceToHost, s);
            }
            vec<OP_R_INIT>(r, nullptr, atwd, nullptr);  // r = masked ? 0 : c - r
            if (log) t_ops += wall() - t0;
            t0 = log ? wall() : 0.0;
            (void)hipMemsetAsync(b.ccount, 0, sizeof(unsigned long long), s);
            hipLaunchKernelGGL(k_candidates, grid2, dim3(T), 0, s, r, act, n, ld, tol, rad, b.ckey, b.cidx, b.ccount, (unsigned long long)b.ccap);
            unsigned long long ncand64 = 0;
            (void)hipMemcpyAsync(&ncand64, b.ccount, sizeof(ncand64), hipMemcpyDeviceToHost, s);
            (void)hipStreamSynchronize(s);
            const int64_t ncand = (int64_t)std::min<unsigned long long>(ncand64, (unsigned long long)b.ccap);
            // (gF holds A^T A x at the factor's splits: c - gF is the multiplier of a split that left, the drift at one that is in)
            double drift = 0.0, wdead = 0.0;
            int64_t dead_wanting = 0;
            std::vector<int32_t> rev;  // departed splits whose multiplier is positive again: they come back with this step (revive)
            for (size_t p = 0; p < (size_t)b.f; p++) {
                const double g = cF[p] - gF[p];
                if (dead[p]) {
                    wdead = std::max(wdead, g); dead_wanting += g > tol ? 1 : 0;
                    // (FNN_SW_REVIVE_MINF: only from that many splits on - 8192 was tried: 82.2 s instead of 73.0 s at 32768 taxa)
                    if (g > tol && revive_on && !revive_blocked && b.f >= revive_min_f) rev.push_back((int32_t)p);
                } else drift = std::max(drift, std::fabs(g));
            }
            st_dead_wanting += dead_wanting;
            if (revive_mode == 2 && ncand != 0) rev.clear();
            const int64_t nlive = b.f - b.r;
            int64_t k = (int64_t)std::min<double>((double)b.kmax, std::max<double>((double)kmin, kfrac * (double)std::max<int64_t>(nlive, 1)));
            k = std::max<int64_t>(1, std::min<int64_t>({k, ncand, k_limit}));
            const bool no_cand = ncand == 0;
            if (!no_cand && b.r == 0 && b.f + k > b.cap && b.f < b.cap) k = b.cap - b.f;  // (nothing to rebuild away: fill the factor to the brim first)
            // (the drift of the gradient on the splits that are in: half of what the solver's own Kuhn-Tucker check allows at the end)
            // (departed splits in the factor: each costs every append 4 k f flops and every departure its share of the Gram matrix; a
            //  rebuild costs 2/3 f^3 - FNN_SW_RFRAC is the share of the factor at which the rebuild is taken)
            const double r_limit = std::min(rfrac * (double)b.f, std::max(rfrac_buf * (double)b.f, (double)b.rcap - 1.5 * (double)k));
            if ((no_cand && rev.empty() && (wdead > tol || (drift > 5e-10 * cmax && !fresh))) ||
                (!no_cand && (b.f + k > b.cap || (double)(b.r - (int64_t)rev.size()) > r_limit || (force_rebuild && b.r > 0)))) {
                force_rebuild = false;
                // a split that left wants back in, the factor has drifted or is full of departed splits: rebuild it, solve, look again
                if (!no_cand && b.r == 0) { give(FNN_SW_GIVEUP_CAPACITY, "the free set outgrew the dense factor"); break; }  // the caller decides about the reference's route
                if (!refactor()) { give(FNN_SW_GIVEUP_NUMERIC, "rebuild of the factor failed"); break; }
                if (!settle_all()) { give(over_rcap ? FNN_SW_GIVEUP_DEPARTED : FNN_SW_GIVEUP_NUMERIC, over_rcap ? "departed splits outgrew their buffers after a rebuild" : "sub-problem solve failed"); break; }
                xw = sbuf; phi = objective(xw); fresh = true;
                continue;
            }
            if (no_cand && rev.empty()) {
                // Splits that were set aside (numerically dependent on the factor at the time, or no measurable descent) are still
                // masked: before the method may call this the optimum they get one more look against the current factor.
                if (!banned.empty() && !banned_rechecked) { set_mask(banned, 0); banned.clear(); banned_rechecked = true; continue; }
                n_set_aside = (int64_t)banned.size();
                done = true; break;
            }
            // ---- the block: the largest local maxima of the multiplier (none: a step that only brings departed splits back)
            const int64_t f0 = b.f, r0 = b.r;
            f_step0 = f0;
            const std::vector<uint8_t> dead0 = dead;
            const std::vector<double> x0 = xw;
            const std::vector<int32_t> deadlist0 = deadlist;
            if (r0 > 0) hipLaunchKernelGGL(k_copy2d, g2(r0, r0), dim3(T), 0, s, b.CR, b.rcap, b.CRb, b.rcap, r0, r0);
            int64_t kin = 0;
            int2 entered = make_int2(-1, -1);
            if (!no_cand) {
                if (hipcub::DeviceRadixSort::SortPairsDescending(b.sort_tmp, b.sort_bytes, b.ckey, b.ckey2, b.cidx, b.cidx2, (int)ncand, 0, 64, s) != hipSuccess) {
                    give(FNN_SW_GIVEUP_NUMERIC, "candidate sort failed"); break;
                }
                hipLaunchKernelGGL(k_idx_to_split, g1(k), dim3(T), 0, s, b.cidx2, k, ld, b.dF + b.f);
                F.resize((size_t)(b.f + k));
                (void)hipMemcpyAsync(F.data() + b.f, b.dF + b.f, sizeof(int2) * (size_t)k, hipMemcpyDeviceToHost, s);
                (void)hipStreamSynchronize(s);
                // (tried: returns only for multipliers that would have made the block - fewer returns, but more one-split solves: 78.0 s
                //  instead of 74.4 s at 32768 taxa)
                if (log) t_sel += wall() - t0;
                // ---- it enters, the weights that are not positive leave, the objective decides
                const int2 first = F[(size_t)f0];
                std::vector<int32_t> kept;
                kin = append(k, b.ckey2, kept);
                if (kin < 0) { give(FNN_SW_GIVEUP_NUMERIC, "append failed (BLAS / Cholesky)"); break; }
                if (kin == 0) {  // the first split is numerically dependent on the factor: set it aside until progress is made
                    F.resize((size_t)f0);
                    banned.push_back(first); set_mask({first}, 1);
                    if (rev.empty()) continue;
                } else {
                    for (int64_t q = 0; q < kin; q++) F[(size_t)(f0 + q)] = F[(size_t)(f0 + kept[(size_t)q])];
                    F.resize((size_t)(f0 + kin));
                    entered = F[(size_t)f0];
                    hipLaunchKernelGGL(k_mask, g1(kin), dim3(T), 0, s, b.dF + f0, kin, act, ld, (uint8_t)1);
                    st_adds += kin;
                    xw.resize((size_t)b.f, 0.0); dead.resize((size_t)b.f, 0); cF.resize((size_t)b.f);
                    hipLaunchKernelGGL(k_gather, g1(kin), dim3(T), 0, s, b.dF + f0, kin, atwd, ld, b.xs);
                    (void)hipMemcpyAsync(cF.data() + f0, b.xs, sizeof(double) * (size_t)kin, hipMemcpyDeviceToHost, s);
                    (void)hipStreamSynchronize(s);
                }
            }
            const bool revived = !rev.empty();
            if (revived && !revive(rev)) { give(FNN_SW_GIVEUP_NUMERIC, "Gram factor of the departed columns failed"); break; }
            bool feasible = true;
            if (!ratio_mode) { feasible = settle_all(); if (!feasible && over_rcap) force_rebuild = true; }
            else {  // Lawson & Hanson: as far towards the sub-problem's minimiser as feasibility allows; what reaches zero leaves
                std::vector<double> xcur = xw;
                for (;;) {
                    if (!solve(sbuf)) { feasible = false; break; }
                    double alpha = 2.0;
                    for (size_t p = 0; p < sbuf.size(); p++)
                        if (!dead[p] && !(sbuf[p] > 0.0)) alpha = std::min(alpha, xcur[p] / (xcur[p] - sbuf[p]));
                    if (alpha > 1.0) break;
                    st_ratio_steps++;
                    std::vector<int32_t> out;
                    for (size_t p = 0; p < sbuf.size(); p++) {
                        if (dead[p]) continue;
                        const bool neg = !(sbuf[p] > 0.0);
                        const bool hit = neg && !(xcur[p] / (xcur[p] - sbuf[p]) > alpha);  // (0 / 0 counts as a hit)
                        xcur[p] = hit ? 0.0 : xcur[p] + alpha * (sbuf[p] - xcur[p]);
                        if (hit || (neg && !(xcur[p] > 0.0))) out.push_back((int32_t)p);
                    }
                    if (b.r + (int64_t)out.size() > b.rcap) { feasible = false; force_rebuild = true; break; }
                    st_dels += (int64_t)out.size();
                    for (int32_t p : out) st_dels_new += p >= f_step0 ? 1 : 0;
                    if (!depart(out)) { feasible = false; break; }
                }
            }
            if (!blas_ok) { give(FNN_SW_GIVEUP_NUMERIC, "a BLAS call failed"); break; }
            f_peak = std::max(f_peak, b.f);
            const double phi_new = feasible ? objective(sbuf) : INFINITY;
            // Lawson & Hanson's step moves along the segment to the minimiser of a convex quadratic: it cannot ascend, so an
            // "ascent" in the objective's last digits is rounding and the step is taken; the other step has to descend.
            const double phi_eps = 1e-13 * std::fabs(phi);
            if (!(ratio_mode ? phi_new < phi + phi_eps : phi_new < phi)) {  // no descent: the factor as it was before this block
                st_rejects++;
                if (log) std::fprintf(stderr, "  [sw] step %lld: block of %lld (+%lld departed splits back) at |F| = %lld taken back (%.17g vs %.17g)%s\n", (long long)st_lh_steps,
                                      (long long)kin, (long long)rev.size(), (long long)(f0 - r0), phi_new, phi, ratio_mode ? " [ratio step]" : "");
                if (b.f > f0) hipLaunchKernelGGL(k_mask, g1(b.f - f0), dim3(T), 0, s, b.dF + f0, b.f - f0, act, ld, (uint8_t)0);
                b.f = f0; b.r = r0;
                F.resize((size_t)f0); dead = dead0; xw = x0; deadlist = deadlist0; cF.resize((size_t)f0);
                if (r0 > 0) {
                    hipLaunchKernelGGL(k_copy2d, g2(r0, r0), dim3(T), 0, s, b.CRb, b.rcap, b.CR, b.rcap, r0, r0);
                    (void)hipMemcpyAsync(b.dlist, deadlist.data(), sizeof(int32_t) * (size_t)r0, hipMemcpyHostToDevice, s);
                    // (departed splits had come back with this step: Y's columns were compacted - fetch them from the factor again)
                    if (revived) hipLaunchKernelGGL(k_gather_cols_tri, g2(f0, r0), dim3(T), 0, s, b.Ws, f0, b.dlist, r0, b.Y, b.cap);
                    (void)hipStreamSynchronize(s);
                    if (!factor_gram()) { give(FNN_SW_GIVEUP_NUMERIC, "Gram factor of the departed columns failed"); break; }
                }
                if (revived) revive_blocked = true;    // (no returns again before a step has moved)
                if (revived && kin == 0) continue;     // (nothing had entered: only the returns are taken back)
                // the same block once more with Lawson & Hanson's step, which cannot ascend; if that made no progress either (rounding
                // noise at this level): a quarter of the block, and a single split that does not move is set aside
                if (ratio_mode) {
                    if (kin == 1) {
                        // One split, the step that cannot ascend, and still no descent: the objective no longer resolves what this
                        // multiplier is worth.  If the block's largest multiplier is below 1e-10 max|A^T d| (a tenth of what the
                        // Kuhn-Tucker certificate of the tests allows), that level IS the noise floor of this problem - tree-like
                        // distances at 32768 taxa: objective 9e10, changes in its 14th digit, thousands of such candidates, each
                        // worth a step - and becomes the candidates' threshold; a larger one is set aside until a step moves.
                        double wtop = 0.0;
                        (void)hipMemcpyAsync(&wtop, b.ckey2, sizeof(double), hipMemcpyDeviceToHost, s);
                        (void)hipStreamSynchronize(s);
                        if (wtop <= tol_cap) {
                            if (log) std::fprintf(stderr, "  [sw] noise floor: candidates' threshold %.3g -> %.3g (%.3g max|A^T d|)\n", tol, wtop, wtop / (cmax > 0.0 ? cmax : 1.0));
                            tol = std::max(tol, wtop);
                        } else {
                            banned.push_back(entered); set_mask({entered}, 1);
                        }
                    }
                    k_limit = std::max<int64_t>(1, kin / 4);
                } else {
                    ratio_mode = true; k_limit = kin;
                }
                continue;
            }
            const bool moved = phi - phi_new > phi_eps;  // (a descent in the last digits is not progress)
            phi = phi_new; xw = sbuf; fresh = false;
            stall = moved ? 0 : stall + 1;
            if (stall >= 8) {
                // Eight steps that the objective cannot tell apart: the candidates' multipliers are at the noise floor of this
                // problem (tree-like distances at 32768 taxa: objective 9e10, thousands of candidates around 1e-11 max|A^T d|,
                // each good for a step in the 14th digit).  Their level becomes the candidates' threshold, as long as it is
                // below 1e-10 max|A^T d| (a tenth of what the tests' Kuhn-Tucker certificate allows); above that only after
                // 200 such steps, with a line in the log, so that the method ends.
                double wtop = 0.0;
                (void)hipMemcpyAsync(&wtop, b.ckey2, sizeof(double), hipMemcpyDeviceToHost, s);
                (void)hipStreamSynchronize(s);
                if (wtop <= tol_cap || stall >= 200) {
                    if (log || wtop > tol_cap) std::fprintf(stderr, "  [sw] noise floor after %d steps without measurable descent: candidates' threshold %.3g -> %.3g (%.3g max|A^T d|)\n",
                                                            stall, tol, wtop, wtop / (cmax > 0.0 ? cmax : 1.0));
                    tol = std::max(tol, wtop);
                    stall = 0;
                }
            }
            k_limit = k_limit > b.kmax / 2 ? b.kmax : 2 * k_limit;
            // near the end (few candidates, each displacing one split) the guaranteed step stays on.  (Tried, round 4: the guaranteed step
            // only for the retry of a block that did not descend, and for eight steps after three such blocks in a row - 75.6 s instead
            // of 73.0 s at 32768 taxa, 13.9 s instead of 13.4 s at 16384: the method's path is sensitive to such rules, +-10 %.)
            ratio_mode = ratio_mode && ncand <= 2 * kmin;
            if (moved) revive_blocked = false;
            if (moved) banned_rechecked = false;
            if (moved && !banned.empty()) { set_mask(banned, 0); banned.clear(); }  // progress: the splits set aside may be looked at again
            if (log && (st_lh_steps % 10 == 0 || st_lh_steps < 5))
                std::fprintf(stderr, "  [sw] step %lld: |F| = %lld (+%lld departed in the factor) candidates %lld block %lld solves %lld objective %.12g | ops %.2f sel %.2f "
                             "append %.2f solve %.2f depart %.2f refactor %.2f s\n", (long long)st_lh_steps, (long long)(b.f - b.r), (long long)b.r, (long long)ncand,
                             (long long)kin, (long long)st_solves, phi, t_ops, t_sel, t_append, t_solve, t_dead, t_refactor);
        }
        if (!done && good) give(FNN_SW_GIVEUP_STEPS, "step limit reached");
        final_tol_rel = tol / (cmax > 0.0 ? cmax : 1.0);
        have_atwd = true;
        if (good) {  // the optimum on the grid
            (void)hipMemsetAsync(this->x, 0, sizeof(double) * (size_t)n * (size_t)ld, s);
            if (b.f) {
                (void)hipMemcpyAsync(b.xs, xw.data(), sizeof(double) * (size_t)b.f, hipMemcpyHostToDevice, s);
                hipLaunchKernelGGL(k_scatter, g1(b.f), dim3(T), 0, s, b.dF, b.f, b.xs, this->x, ld);
            }
            (void)hipStreamSynchronize(s);
        }
        if (log) std::fprintf(stderr, "  [sw] %s: %lld steps, %lld solves, %lld entered (%lld screened out before), %lld left, %lld taken back, %lld ratio steps, %lld factorisations; |F| = %lld | ops %.2f "
                              "sel %.2f append %.2f solve %.2f depart %.2f refactor %.2f s\n", good ? "done" : "gave up", (long long)st_lh_steps, (long long)st_solves,
                              (long long)st_adds, (long long)st_screened, (long long)st_dels, (long long)st_rejects, (long long)st_ratio_steps, (long long)st_lh_refactor, (long long)(b.f - b.r),
                              t_ops, t_sel, t_append, t_solve, t_dead, t_refactor);
        if (log) std::fprintf(stderr, "  [sw]   GEMM work through gemm(): %.3e flop; departed splits with a positive multiplier, summed over the steps: %lld; brought back: %lld; "
                              "departures of splits in the step they entered: %lld\n", gemm_flops, (long long)st_dead_wanting, (long long)st_revived, (long long)st_dels_new);
        if (log) for (const auto& kv : tsub) std::fprintf(stderr, "  [sw]   %-28s %8.3f s\n", kv.first.c_str(), kv.second);
        return good;
    }

    // runActiveConjugate (:366-557)
    void active_conjugate() {
        hipLaunchKernelGGL(k_unconstrained, dim3((n + T - 1) / T, n), dim3(T), 0, s, d, x, n, ld);
        if (reduce_sum<RD_COUNT_NEG>(x, nullptr) == 0.0) return;
        (void)hipMemsetAsync(act, 0, (size_t)n * (size_t)ld, s);
        hipLaunchKernelGGL(k_fill, dim3((unsigned)((n * ld + T - 1) / T)), dim3(T), 0, s, old_x, (int64_t)n * ld, 1.0);  // Arrays.fill(old_x, 1.0) (:383)
        Atx(d, atwd);
        bool first_pass = true;
        for (;;) {
            st_outer++;
            for (;;) {
                if (!first_pass) cg();
                first_pass = false;
                if (contract_worst()) cg();
                const Best mr = reduce_best<RD_MIN_RATIO>(x, old_x);
                if (mr.k == INT64_MAX) break;  // feasible
                vec<OP_MOVE_OLD>(x, old_x, nullptr, nullptr, mr.v);
                const uint8_t one = 1;
                const double zero = 0.0;
                (void)hipMemcpyAsync(act + mr.k, &one, 1, hipMemcpyHostToDevice, s);
                (void)hipMemcpyAsync(x + mr.k, &zero, sizeof(double), hipMemcpyHostToDevice, s);
                (void)hipStreamSynchronize(s);
            }
            Ab(x, y);
            Atx(y, r);
            vec<OP_GRAD>(r, nullptr, atwd, nullptr);
            const Best mg = reduce_best<RD_MIN_ACTIVE_GRAD>(r, nullptr);
            if (mg.k == INT64_MAX || mg.v > -0.0000001) break;
            const uint8_t zero8 = 0;
            (void)hipMemcpyAsync(act + mg.k, &zero8, 1, hipMemcpyHostToDevice, s);
            (void)hipStreamSynchronize(s);
        }
    }
};

}  // namespace fnnsw

// weights_out != nullptr: all n (n - 1) / 2 weights; else the weights above `thr` as (index, weight) pairs in index order
static int32_t split_weights_impl(const double* D, int32_t n, int64_t ldD, const int32_t* ordering, int32_t device, double* weights_out,
                                  double thr, int64_t* idx_out, double* w_out, int64_t cap_out, int64_t* count_out, fnn_sw_stats* stats) {
    using namespace fnnsw;
    if (!D || !ordering || n < 2 || ldD < n || (!weights_out && (!idx_out || !w_out || !count_out || cap_out < 0)))
        return fnn::fail(FNN_EINVAL, "fnn_split_weights_f64: bad arguments");
    {
        std::vector<char> seen((size_t)n + 1, 0);
        for (int i = 1; i <= n; i++) {
            if (ordering[i] < 1 || ordering[i] > n || seen[(size_t)ordering[i]]) return fnn::fail(FNN_EINVAL, "fnn_split_weights_f64: ordering is not a permutation of 1..n");
            seen[(size_t)ordering[i]] = 1;
        }
    }
    int cnt = 0;
    if (!SWOK(hipGetDeviceCount(&cnt)) || cnt <= 0) return fnn::fail(FNN_EHIP, "no HIP device available");
    if (device < 0 || device >= cnt || !SWOK(hipSetDevice(device))) return fnn::fail(FNN_EINVAL, "device ordinal out of range");
    struct Events {  // (destroyed on every exit path)
        hipEvent_t e0 = nullptr, e1 = nullptr;
        ~Events() { if (e0) (void)hipEventDestroy(e0); if (e1) (void)hipEventDestroy(e1); }
    } ev;
    Solver S;
    S.dev = device;
    S.n = n;
    S.ld = ((int64_t)n + 31) / 32 * 32 + 32;
    if (!SWOK(hipStreamCreateWithFlags(&S.s, hipStreamNonBlocking))) return fnn::fail(FNN_EHIP, "hipStreamCreate failed");
    const size_t NN = (size_t)n * (size_t)S.ld;
    // (w, p, old_x belong to the reference's method only and are allocated when it runs)
    S.d = S.alloc<double>(NN); S.x = S.alloc<double>(NN); S.r = S.alloc<double>(NN);
    S.y = S.alloc<double>(NN); S.atwd = S.alloc<double>(NN);
    S.P = S.alloc<double>(NN); S.Pt = S.alloc<double>(NN); S.rs = S.alloc<double>((size_t)n + 8);
    S.act = S.alloc<uint8_t>(NN);
    S.Dm = S.alloc<double>((size_t)n * (size_t)n);
    S.ord = S.alloc<int32_t>((size_t)n + 1);
    S.live = S.alloc<double>((size_t)n * (n - 1) / 2);
    S.grid2 = dim3((unsigned)((n + T - 1) / T), (unsigned)n);
    S.gred = dim3((unsigned)((n + T - 1) / T), (unsigned)(n < 256 ? n : 256));
    S.partial = S.alloc<double>(4 * (size_t)S.gred.x * S.gred.y);
    S.sc = S.alloc<double>(SC_WORDS);
    S.bpartial = S.alloc<Best>((size_t)S.gred.x * S.gred.y);
    if (!S.ok) return fnn::fail(FNN_ENOMEM, "fnn_split_weights_f64: device allocation failed");
    for (double* v : {S.d, S.x, S.r, S.y, S.atwd, S.P, S.Pt}) (void)hipMemsetAsync(v, 0, sizeof(double) * NN, S.s);
    if (!SWOK(hipMemcpy2DAsync(S.Dm, sizeof(double) * (size_t)n, D, sizeof(double) * (size_t)ldD, sizeof(double) * (size_t)n, (size_t)n,
                               hipMemcpyHostToDevice, S.s)) ||
        !SWOK(hipMemcpyAsync(S.ord, ordering, sizeof(int32_t) * ((size_t)n + 1), hipMemcpyHostToDevice, S.s)) ||
        !SWOK(hipStreamSynchronize(S.s)))
        return fnn::fail(FNN_EHIP, "fnn_split_weights_f64: upload failed");
    if (!SWOK(hipEventCreate(&ev.e0)) || !SWOK(hipEventCreate(&ev.e1))) return fnn::fail(FNN_EHIP, "fnn_split_weights_f64: hipEventCreate failed");
    (void)hipEventRecord(ev.e0, S.s);
    hipLaunchKernelGGL(k_reorder, dim3((unsigned)((n + T - 1) / T), (unsigned)n), dim3(T), 0, S.s, S.Dm, (int64_t)n, S.ord, S.d, n, S.ld);
    S.release(S.Dm);  // (the raw copy has served: its n^2 doubles go back to the pool before the factor is sized)
    S.Dm = nullptr;
    // the closed form if it is feasible; else from below (block active-set method, exact sub-problems); the reference's own
    // method (from above, conjugate gradients) where the free set is too large for a dense factor AND that route is affordable
    int route = FNN_SW_ROUTE_CLOSED_FORM;
    hipLaunchKernelGGL(k_unconstrained, dim3((unsigned)((n + T - 1) / T), (unsigned)n), dim3(T), 0, S.s, S.d, S.x, n, S.ld);
    if (S.reduce_sum<RD_COUNT_NEG>(S.x, nullptr) != 0.0) {
        const bool want_reference = std::getenv("FNN_SW_REFERENCE_METHOD") != nullptr;
        // The block method's factor is sized for the common case (random or tree-like distances: 2.4 n ... 3.8 n positive splits) so
        // that small problems do not pay for a 100-GB allocation; distances whose optimum has more positive splits (nearly
        // circular metrics: measured 19.5 n at 1024 taxa with 1 % noise) outgrow it: try again with four times the capacity, up to
        // what device memory holds (~150 000 splits, whatever n) or all n (n - 1) / 2 splits.
        bool from_below = false;
        if (!want_reference) {
            const int64_t Nall = (int64_t)n * (n - 1) / 2;
            for (int attempt = 0; attempt < 6; attempt++) {
                from_below = S.block_active_set();
                if (from_below || S.giveup != FNN_SW_GIVEUP_CAPACITY || S.capacity >= Nall || std::getenv("FNN_SW_CAP")) break;
                const int64_t prev = S.capacity;
                S.release_block_buffers();
                S.cap_want = std::min<int64_t>(Nall, 4 * prev);
                size_t fb = 0, tb = 0;
                if (hipMemGetInfo(&fb, &tb) != hipSuccess) break;
                fb += pool().pooled(device);
                // (the largest factor that can fit at all: cap^2 * 8 B * 9/16 for the factor alone)
                const int64_t hard = (int64_t)std::sqrt(0.9 * (double)fb / (8.0 * 0.62 * 1.7));
                if (S.cap_want > hard) S.cap_want = hard;
                if (S.cap_want < prev + prev / 4) { S.cap_want = 0; break; }  // no meaningful growth left: the give-up stands
                std::fprintf(stderr, "fnn_split_weights_f64: the free set outgrew the factor's capacity of %lld splits at n = %d; once more with %lld\n",
                             (long long)prev, n, (long long)S.cap_want);
            }
        }
        route = from_below ? FNN_SW_ROUTE_FROM_BELOW : FNN_SW_ROUTE_REFERENCE;
        if (!from_below) {
            if (!want_reference) {
                // The block method gave up.  Say so, always: the route that follows is the reference's own - correct, but
                // O(n^2) work per conjugate-gradient iteration and thousands of iterations on large problems.
                int64_t max_n = 1024;  // (measured: a 1024-taxon nearly circular input takes 80 s on that route, 2048 taxa more than 5 min)
                if (const char* e = std::getenv("FNN_SW_REFERENCE_MAX_N")) max_n = std::atoll(e);
                const bool allow = std::getenv("FNN_SW_ALLOW_REFERENCE_ROUTE") != nullptr || n <= max_n;
                std::fprintf(stderr, "fnn_split_weights_f64: the block active-set method gave up at n = %d (%s; capacity %lld splits, free set peaked at %lld): %s\n",
                             n, S.giveup_text, (long long)S.capacity, (long long)S.f_peak,
                             std::getenv("FNN_SW_NO_REFERENCE_ROUTE") ? "FNN_SW_NO_REFERENCE_ROUTE is set"
                             : allow ? "taking the reference's active-set / conjugate-gradient route"
                                     : "the reference's conjugate-gradient route is not taken automatically above FNN_SW_REFERENCE_MAX_N taxa");
                if (stats) {
                    std::memset(stats, 0, sizeof(*stats));
                    stats->giveup_reason = S.giveup; stats->capacity = S.capacity; stats->free_set_peak = S.f_peak;
                    stats->outer_iterations = S.st_lh_steps; stats->entered = S.st_adds; stats->screened_out = S.st_screened; stats->departed = S.st_dels;
                }
                if (std::getenv("FNN_SW_NO_REFERENCE_ROUTE") || !allow) {
                    const bool cap = S.giveup == FNN_SW_GIVEUP_CAPACITY || S.giveup == FNN_SW_GIVEUP_DEPARTED || S.giveup == FNN_SW_GIVEUP_SETUP;
                    return fnn::fail(cap ? FNN_ECAPACITY : FNN_EHIP,
                                     std::string("fnn_split_weights_f64: the block active-set method gave up (") + S.giveup_text + "); the optimum of these distances has more "
                                     "positive splits than the dense factor holds (" + std::to_string((long long)S.capacity) + "), and the reference's conjugate-gradient route "
                                     "is not taken automatically at this size (FNN_SW_ALLOW_REFERENCE_ROUTE=1 takes it)");
                }
            }
            S.release_block_buffers();
            S.w = S.alloc<double>(NN); S.p = S.alloc<double>(NN); S.old_x = S.alloc<double>(NN);
            if (!S.ok) return fnn::fail(FNN_ENOMEM, "fnn_split_weights_f64: device allocation failed");
            for (double* v : {S.w, S.p, S.old_x}) (void)hipMemsetAsync(v, 0, sizeof(double) * NN, S.s);
            S.st_lh_steps = 0; S.st_solves = 0;
            S.active_conjugate();
            S.have_atwd = true;
        }
    }
    if (std::getenv("FNN_SW_FAULT_PERTURB")) hipLaunchKernelGGL(k_fault_perturb, dim3(1), dim3(1), 0, S.s, S.x, n, S.ld);
    hipLaunchKernelGGL(k_to_live, S.grid2, dim3(T), 0, S.s, S.x, S.live, n, S.ld);
    (void)hipEventRecord(ev.e1, S.s);
    // The solver's own Kuhn-Tucker check of what it returns: g = A^T (A x - d) from the implicit operators, which share
    // nothing with the factor the block method solved with (outside the timed span: ~0.15 s at 32768 taxa)
    if (!S.have_atwd) S.Atx(S.d, S.atwd);
    S.Ab(S.x, S.y);
    hipLaunchKernelGGL(k_sub, S.grid2, dim3(T), 0, S.s, S.y, S.d, n, S.ld);
    S.Atx(S.y, S.r);
    hipLaunchKernelGGL(k_kkt, S.gred, dim3(T), 0, S.s, S.x, S.r, S.atwd, n, S.ld, S.partial);
    std::vector<double> kp(4 * (size_t)S.gred.x * S.gred.y);
    (void)hipMemcpyAsync(kp.data(), S.partial, sizeof(double) * kp.size(), hipMemcpyDeviceToHost, S.s);
    hipError_t e = hipStreamSynchronize(S.s);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, ev.e0, ev.e1);
    if (e != hipSuccess || hipGetLastError() != hipSuccess) return fnn::fail(FNN_EHIP, std::string("fnn_split_weights_f64: ") + hipGetErrorString(e));
    double kk[4] = {0.0, 0.0, 0.0, 0.0};
    for (size_t q = 0; q < kp.size(); q += 4)
        for (int a = 0; a < 4; a++) kk[a] = (kp[q + a] > kk[a] || kp[q + a] != kp[q + a]) ? kp[q + a] : kk[a];
    const double scale = kk[3] > 0.0 ? kk[3] : 1.0;
    const double viol = std::max({kk[0], kk[1] / scale, kk[2] / scale});
    const bool certified = viol <= 1e-9;  // (NaN compares false)
    // what goes back to the host: every weight, or only those above the threshold (the reference keeps x[k] > 1e-6, FastNN.java:455:
    // 77 000 of 5.4e8 weights at 32768 taxa - the dense array is 4.3 GB over the bus and a host pass over it)
    const int64_t Nw = (int64_t)n * (n - 1) / 2;
    int64_t npos = 0;
    {
        unsigned long long* dcount = S.alloc<unsigned long long>(1);
        const int64_t pcap = weights_out ? 0 : cap_out;
        int64_t* didx = S.alloc<int64_t>((size_t)std::max<int64_t>(pcap, 1));
        double* dw = S.alloc<double>((size_t)std::max<int64_t>(pcap, 1));
        if (!S.ok) return fnn::fail(FNN_ENOMEM, "fnn_split_weights_f64: device allocation failed");
        (void)hipMemsetAsync(dcount, 0, sizeof(unsigned long long), S.s);
        hipLaunchKernelGGL(k_pick_positive, dim3((unsigned)((Nw + T - 1) / T)), dim3(T), 0, S.s, S.live, Nw, weights_out ? 0.000001 : thr, didx, dw, dcount,
                           (unsigned long long)pcap);
        unsigned long long c = 0;
        if (!SWOK(hipMemcpyAsync(&c, dcount, sizeof(c), hipMemcpyDeviceToHost, S.s)) || !SWOK(hipStreamSynchronize(S.s)))
            return fnn::fail(FNN_EHIP, "fnn_split_weights_f64: download failed");
        npos = (int64_t)c;
        if (weights_out) {
            if (!SWOK(hipMemcpy(weights_out, S.live, sizeof(double) * (size_t)Nw, hipMemcpyDeviceToHost)))
                return fnn::fail(FNN_EHIP, "fnn_split_weights_f64: download failed");
        } else {
            *count_out = npos;
            const int64_t take = std::min(npos, cap_out);
            std::vector<int64_t> hi((size_t)take);
            std::vector<double> hw((size_t)take);
            if (take > 0 && (!SWOK(hipMemcpy(hi.data(), didx, sizeof(int64_t) * (size_t)take, hipMemcpyDeviceToHost)) ||
                             !SWOK(hipMemcpy(hw.data(), dw, sizeof(double) * (size_t)take, hipMemcpyDeviceToHost))))
                return fnn::fail(FNN_EHIP, "fnn_split_weights_f64: download failed");
            std::vector<int64_t> perm((size_t)take);
            for (int64_t q = 0; q < take; q++) perm[(size_t)q] = q;
            std::sort(perm.begin(), perm.end(), [&](int64_t a, int64_t b) { return hi[(size_t)a] < hi[(size_t)b]; });  // the reference's list order
            for (int64_t q = 0; q < take; q++) { idx_out[q] = hi[(size_t)perm[(size_t)q]]; w_out[q] = hw[(size_t)perm[(size_t)q]]; }
        }
    }
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        const bool from_below = route == FNN_SW_ROUTE_FROM_BELOW;
        stats->outer_iterations = from_below ? S.st_lh_steps : S.st_outer;
        stats->cg_calls = S.st_cg;
        stats->cg_iterations = S.st_it;
        stats->reserved[0] = from_below ? 1 : 0;        // method: 1 = from below (block active-set, dense factor), 0 = the reference's (or the closed form)
        stats->reserved[1] = S.st_lh_refactor;
        stats->reserved[2] = S.st_solves;               // sub-problems solved (from below)
        stats->t_solve_s = ms * 1e-3;
        stats->nsplits = npos;                          // weights above the threshold (FastNN.java:455: 1e-6; the sparse call: its own)
        stats->route = route;
        stats->certified = certified ? 1 : 0;
        stats->kkt_violation = viol;
        stats->final_threshold_rel = from_below ? S.final_tol_rel : 0.0;
        stats->n_set_aside = from_below ? S.n_set_aside : 0;
        stats->capacity = S.capacity;
        stats->free_set_peak = S.f_peak;
        stats->giveup_reason = S.giveup;
        stats->entered = S.st_adds; stats->screened_out = S.st_screened; stats->departed = S.st_dels;
        stats->t_alloc_s = S.t_alloc_s;
    }
    if (route == FNN_SW_ROUTE_FROM_BELOW && !certified) {
        char buf[320];
        std::snprintf(buf, sizeof(buf), "fnn_split_weights_f64: the weights fail the Kuhn-Tucker check: violation %.3g of max|A^T d| (min x %.3g, gradient on positive weights %.3g, "
                      "on zero weights %.3g; candidates' final threshold %.3g, %lld splits set aside): not the certified optimum",
                      viol, -kk[0], kk[1] / scale, kk[2] / scale, S.final_tol_rel, (long long)S.n_set_aside);
        std::fprintf(stderr, "%s\n", buf);
        return fnn::fail(FNN_EINEXACT, buf);
    }
    return FNN_OK;
}

extern "C" int32_t fnn_split_weights_f64(const double* D, int32_t n, int64_t ldD, const int32_t* ordering, int32_t device,
                                         double* weights_out, fnn_sw_stats* stats) {
    if (!weights_out) return fnn::fail(FNN_EINVAL, "fnn_split_weights_f64: bad arguments");
    return split_weights_impl(D, n, ldD, ordering, device, weights_out, 0.0, nullptr, nullptr, 0, nullptr, stats);
}
extern "C" int32_t fnn_split_weights_release_cache(void) {
    fnnsw::pool().flush();
    return FNN_OK;
}
extern "C" int32_t fnn_split_weights_sparse_f64(const double* D, int32_t n, int64_t ldD, const int32_t* ordering, int32_t device, double threshold,
                                                int64_t* index_out, double* weight_out, int64_t capacity, int64_t* count_out, fnn_sw_stats* stats) {
    return split_weights_impl(D, n, ldD, ordering, device, nullptr, threshold, index_out, weight_out, capacity, count_out, stats);
}
