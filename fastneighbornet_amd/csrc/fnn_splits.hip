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
#include <rocsolver/rocsolver.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
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

// ---------------------------------------------------------------- "from below": Lawson-Hanson on the normal equations
// The reference's method starts from the unconstrained optimum (every split free) and contracts; on distances that are
// far from circular (random matrices: only ~2 n of the n(n-1)/2 splits end up with a positive weight) almost all of its
// work is conjugate-gradient iterations on huge, ill-conditioned free sets.  Lawson & Hanson's active-set method - what the
// reference's LIVE path runs (edu.rit.numeric.NonNegativeLeastSquares, FastNN.java:401-454) - grows the free set F from
// nothing, one split at a time, and solves each sub-problem exactly; here F's normal equations H_FF z = c_F are dense
// (|F| ~ 2 n) with a Cholesky factor that is extended in place, and the only O(n^2) work per step is the gradient
// A^T (d - A x) by the same prefix-sum operators.  H has a closed form: the number of position pairs that two circular
// splits S, T both separate is |S n T| |S^c n T^c| + |S n T^c| |S^c n T|.
__device__ __forceinline__ double h_entry(int n, int i, int j, int k, int l) {  // splits (i,j), (k,l): positions {i+1..j}, {k+1..l}
    const int lo = i > k ? i : k, hi = j < l ? j : l;
    const double st = hi > lo ? (double)(hi - lo) : 0.0, s = (double)(j - i), t = (double)(l - k);
    return st * ((double)n - s - t + st) + (s - st) * (t - st);
}
// the whole symmetric H_FF (F = list of splits), column-major, for a fresh inverse
__global__ __launch_bounds__(T) void k_hfill(const int2* F, int f, int n, double* G, int64_t ldg) {
    const int r = blockIdx.x * T + threadIdx.x, c = blockIdx.y;
    if (r >= f) return;
    G[(int64_t)c * ldg + r] = h_entry(n, F[r].x, F[r].y, F[c].x, F[c].y);
}
// h[p] = H[F[p], F[f]] for p <= f (the new split is the last of the list)
__global__ __launch_bounds__(T) void k_hcol(const int2* F, int f, int n, double* h) {
    const int p = blockIdx.x * T + threadIdx.x;
    if (p > f) return;
    h[p] = h_entry(n, F[p].x, F[p].y, F[f].x, F[f].y);
}
// the inverse of the bordered matrix: new row / column f = -s u, corner s (the leading block got + s u u^T by dger)
__global__ __launch_bounds__(T) void k_border(double* G, int64_t ldg, int f, const double* u, double s_) {
    const int p = blockIdx.x * T + threadIdx.x;
    if (p < f) { const double v = -s_ * u[p]; G[(int64_t)f * ldg + p] = v; G[(int64_t)p * ldg + f] = v; }
    else if (p == f) G[(int64_t)f * ldg + f] = s_;
}
// exchange rows / columns p and q (p < q) of the symmetric G (leading f x f block)
__global__ __launch_bounds__(T) void k_swap_rc(double* G, int64_t ldg, int f, int p, int q) {
    const int i = blockIdx.x * T + threadIdx.x;
    if (i >= f) return;
    if (i != p && i != q) {
        const double a = G[(int64_t)p * ldg + i], b = G[(int64_t)q * ldg + i];
        G[(int64_t)p * ldg + i] = b; G[(int64_t)q * ldg + i] = a;
        G[(int64_t)i * ldg + p] = b; G[(int64_t)i * ldg + q] = a;
    } else if (i == p) {
        const double a = G[(int64_t)p * ldg + p], b = G[(int64_t)q * ldg + q];
        G[(int64_t)p * ldg + p] = b; G[(int64_t)q * ldg + q] = a;  // (G[p][q] = G[q][p] stays)
    }
}
// copy the lower triangle of the leading f x f block onto the upper one (after potri)
__global__ __launch_bounds__(T) void k_symmetrize(double* G, int64_t ldg, int f) {
    const int r = blockIdx.x * T + threadIdx.x, c = blockIdx.y;
    if (r >= f || c >= r) return;
    G[(int64_t)r * ldg + c] = G[(int64_t)c * ldg + r];
}
__global__ __launch_bounds__(T) void k_gather(const int2* F, int f, const double* grid, int64_t ld, double* out) {
    const int p = blockIdx.x * T + threadIdx.x;
    if (p < f) out[p] = grid[(int64_t)F[p].x * ld + F[p].y];
}
__global__ __launch_bounds__(T) void k_scatter(const int2* F, int f, const double* v, double* grid, uint8_t* mask, int64_t ld) {
    const int p = blockIdx.x * T + threadIdx.x;
    if (p < f) { const int64_t k = (int64_t)F[p].x * ld + F[p].y; grid[k] = v[p]; mask[k] = 1; }
}

// ---------------------------------------------------------------- host driver
struct Solver {
    int n = 0;
    int64_t ld = 0;
    hipStream_t s = nullptr;
    std::vector<void*> allocs;
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
        void* p_ = nullptr;
        if (!SWOK(hipMalloc(&p_, sizeof(Tp) * (count ? count : 1)))) { ok = false; return nullptr; }
        allocs.push_back(p_);
        return (Tp*)p_;
    }
    ~Solver() {
        for (void* p_ : allocs) (void)hipFree(p_);
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

    // Lawson-Hanson with the explicit inverse G = H_FF^-1 of the free set's normal equations, kept current by rank-one
    // updates: a split that enters borders G (one matrix-vector product, one rank-one update), a split that leaves is
    // swapped to the end and eliminated by the Schur complement of its diagonal entry (one rank-one update), the
    // sub-problem's solution is one more matrix-vector product - every step is O(|F|^2) of fully parallel, bandwidth-bound
    // work (a Cholesky factor would need sequential triangular solves and a fresh factorisation after every removal:
    // measured 10 x slower at 2048 taxa).  Rounding drift of the updates is watched through the gradient on F, which
    // the next step computes anyway: beyond 1e-9 (relative) G is rebuilt from H_FF's closed form (potrf + potri); the last
    // step always ends on a freshly built inverse.  Returns false if the free set outgrows its capacity (distances close
    // to a circular metric with many positive splits): the caller then runs the reference's method, from the other end.
    int64_t st_lh_steps = 0, st_lh_refactor = 0;
    bool lawson_hanson() {
        const int64_t N = (int64_t)n * (n - 1) / 2;
        const int64_t cap = std::min<int64_t>(N, std::max<int64_t>(8 * (int64_t)n + 64, 256));
        rocblas_handle bh = nullptr;
        if (rocblas_create_handle(&bh) != rocblas_status_success) return false;
        rocblas_set_stream(bh, s);
        rocblas_set_pointer_mode(bh, rocblas_pointer_mode_host);
        double* G = alloc<double>((size_t)cap * (size_t)cap);
        int2* dF = alloc<int2>((size_t)cap);
        double* dv = alloc<double>((size_t)cap);
        double* dh = alloc<double>((size_t)cap);
        double* du = alloc<double>((size_t)cap);
        rocblas_int* dinfo = alloc<rocblas_int>(1);
        bool good = ok;
        std::vector<int2> F;
        std::vector<double> xF, z, cF;
        const double one = 1.0, zero = 0.0;
        auto blocks = [](int64_t c) { return dim3((unsigned)((c + T - 1) / T)); };
        auto upload_F = [&]() { if (!F.empty()) (void)hipMemcpyAsync(dF, F.data(), sizeof(int2) * F.size(), hipMemcpyHostToDevice, s); };
        auto rebuild = [&]() {  // G = H_FF^-1 from the closed form
            const int f = (int)F.size();
            st_lh_refactor++;
            if (f == 0) return true;
            upload_F();
            hipLaunchKernelGGL(k_hfill, dim3((unsigned)((f + T - 1) / T), (unsigned)f), dim3(T), 0, s, dF, f, n, G, cap);
            if (rocsolver_dpotrf(bh, rocblas_fill_lower, f, G, (rocblas_int)cap, dinfo) != rocblas_status_success) return false;
            rocblas_int info = 0;
            (void)hipMemcpyAsync(&info, dinfo, sizeof(info), hipMemcpyDeviceToHost, s);
            (void)hipStreamSynchronize(s);
            if (info != 0) return false;
            if (rocsolver_dpotri(bh, rocblas_fill_lower, f, G, (rocblas_int)cap, dinfo) != rocblas_status_success) return false;
            hipLaunchKernelGGL(k_symmetrize, dim3((unsigned)((f + T - 1) / T), (unsigned)f), dim3(T), 0, s, G, cap, f);
            return true;
        };
        auto solve = [&]() {  // z = G c_F
            const int f = (int)F.size();
            z.assign((size_t)f, 0.0);
            if (f == 0) return true;
            (void)hipMemcpyAsync(dv, cF.data(), sizeof(double) * (size_t)f, hipMemcpyHostToDevice, s);
            if (rocblas_dgemv(bh, rocblas_operation_none, f, f, &one, G, (rocblas_int)cap, dv, 1, &zero, du, 1) != rocblas_status_success) return false;
            (void)hipMemcpyAsync(z.data(), du, sizeof(double) * (size_t)f, hipMemcpyDeviceToHost, s);
            (void)hipStreamSynchronize(s);
            return true;
        };
        auto remove_at = [&](int p) {  // split F[p] leaves: swap it to the end, eliminate
            const int f = (int)F.size(), q = f - 1;
            if (p != q) {
                hipLaunchKernelGGL(k_swap_rc, blocks(f), dim3(T), 0, s, G, cap, f, p, q);
                std::swap(F[(size_t)p], F[(size_t)q]); std::swap(xF[(size_t)p], xF[(size_t)q]); std::swap(cF[(size_t)p], cF[(size_t)q]);
            }
            double gamma = 0.0;
            (void)hipMemcpyAsync(&gamma, G + (int64_t)q * cap + q, sizeof(double), hipMemcpyDeviceToHost, s);
            (void)hipStreamSynchronize(s);
            F.pop_back(); xF.pop_back(); cF.pop_back();
            if (!(gamma > 0.0)) return false;
            const double a = -1.0 / gamma;
            // (the column is copied first: dger must not read what it writes)
            if (q > 0) {
                (void)hipMemcpyAsync(dh, G + (int64_t)q * cap, sizeof(double) * (size_t)q, hipMemcpyDeviceToDevice, s);
                if (rocblas_dger(bh, q, q, &a, dh, 1, dh, 1, G, (rocblas_int)cap) != rocblas_status_success) return false;
            }
            return true;
        };
        Atx(d, atwd);  // c = A^T d
        double cmax = 0.0;
        {
            (void)hipMemsetAsync(act, 0, (size_t)n * (size_t)ld, s);
            const Best b = reduce_best<RD_MAX_UNMASKED>(atwd, nullptr);
            cmax = b.k == INT64_MAX ? 0.0 : -b.v;
        }
        const double tol = 1e-12 * (cmax > 0.0 ? cmax : 1.0);
        std::vector<int64_t> banned;
        const int64_t max_steps = 8 * cap + 1000;
        bool fresh = true, done = false;  // fresh: G was just rebuilt (the Kuhn-Tucker test only counts on a fresh inverse)
        // move from xF towards z as far as feasibility allows; what reaches zero leaves F; repeat until z > 0
        auto settle = [&]() {
            for (;;) {
                double alpha = 2.0;
                for (size_t p = 0; p < F.size(); p++)
                    if (z[p] <= 0.0) { const double al = xF[p] / (xF[p] - z[p]); if (al < alpha) alpha = al; }
                if (alpha > 1.0) { xF = z; return true; }
                std::vector<int2> out;
                for (size_t p = 0; p < F.size(); p++) {
                    const double v = xF[p] + alpha * (z[p] - xF[p]);
                    const bool keep = v > 0.0 && !(z[p] <= 0.0 && xF[p] / (xF[p] - z[p]) <= alpha);
                    if (keep) xF[p] = v; else out.push_back(F[p]);
                }
                for (const int2& t : out) {
                    int p = -1;
                    for (size_t q = 0; q < F.size(); q++) if (F[q].x == t.x && F[q].y == t.y) { p = (int)q; break; }
                    if (p < 0 || !remove_at(p)) return false;
                }
                fresh = false;
                if (!solve()) return false;
            }
        };
        while (good && !done && st_lh_steps < max_steps) {
            st_lh_steps++;
            // x on the grid (zero outside F), mask = F (+ the splits that were rejected since the last successful step)
            const int f = (int)F.size();
            (void)hipMemsetAsync(x, 0, sizeof(double) * (size_t)n * (size_t)ld, s);
            (void)hipMemsetAsync(act, 0, (size_t)n * (size_t)ld, s);
            if (f) {
                upload_F();
                (void)hipMemcpyAsync(dv, xF.data(), sizeof(double) * (size_t)f, hipMemcpyHostToDevice, s);
                hipLaunchKernelGGL(k_scatter, blocks(f), dim3(T), 0, s, dF, f, dv, x, act, ld);
            }
            for (int64_t k : banned) { const uint8_t one8 = 1; (void)hipMemcpyAsync(act + k, &one8, 1, hipMemcpyHostToDevice, s); }
            // r = A^T A x: on F, c - r shows the drift of the updated inverse; outside, c - r is the multiplier w
            Ab(x, y);
            Atx(y, r);
            double drift = 0.0;
            if (f) {
                hipLaunchKernelGGL(k_gather, blocks(f), dim3(T), 0, s, dF, f, r, ld, dh);
                std::vector<double> rf((size_t)f);
                (void)hipMemcpyAsync(rf.data(), dh, sizeof(double) * (size_t)f, hipMemcpyDeviceToHost, s);
                (void)hipStreamSynchronize(s);
                for (int p = 0; p < f; p++) drift = std::max(drift, std::fabs(cF[(size_t)p] - rf[(size_t)p]));
            }
            vec<OP_R_INIT>(r, nullptr, atwd, nullptr);  // r = masked ? 0 : c - r
            const Best b = reduce_best<RD_MAX_UNMASKED>(r, nullptr);
            const bool kkt = b.k == INT64_MAX || -b.v <= tol;
            if ((drift > 1e-9 * cmax || kkt) && !fresh) {
                // the updated inverse has drifted, or this looks like the end: the same free set once more on a freshly
                // built inverse, then look again
                if (!rebuild() || !solve()) { good = false; break; }
                fresh = true;
                if (!settle()) { good = false; break; }
                continue;
            }
            if (kkt) { done = true; break; }  // Kuhn-Tucker on a fresh inverse: no split outside F wants in
            if ((int64_t)F.size() + 1 > cap) { good = false; break; }
            // the split with the largest multiplier enters: border the inverse
            const int2 t = make_int2((int)(b.k / ld), (int)(b.k % ld));
            double ct = 0.0;
            (void)hipMemcpyAsync(&ct, atwd + b.k, sizeof(double), hipMemcpyDeviceToHost, s);
            F.push_back(t); xF.push_back(0.0);
            upload_F();
            hipLaunchKernelGGL(k_hcol, blocks(f + 1), dim3(T), 0, s, dF, f, n, dh);
            double eta = 0.0, hu = 0.0;
            (void)hipMemcpyAsync(&eta, dh + f, sizeof(double), hipMemcpyDeviceToHost, s);
            if (f > 0) {
                if (rocblas_dgemv(bh, rocblas_operation_none, f, f, &one, G, (rocblas_int)cap, dh, 1, &zero, du, 1) != rocblas_status_success ||
                    rocblas_ddot(bh, f, dh, 1, du, 1, &hu) != rocblas_status_success) { good = false; break; }
            }
            (void)hipStreamSynchronize(s);
            cF.push_back(ct);
            const double schur = eta - hu;
            bool accepted = schur > 1e-10 * eta;
            if (accepted) {
                const double sinv = 1.0 / schur;
                if (f > 0 && rocblas_dger(bh, f, f, &sinv, du, 1, du, 1, G, (rocblas_int)cap) != rocblas_status_success) { good = false; break; }
                hipLaunchKernelGGL(k_border, blocks(f + 1), dim3(T), 0, s, G, cap, f, du, sinv);
                fresh = false;
                if (!solve()) { good = false; break; }
                accepted = z.back() > 0.0;  // (Lawson & Hanson's check on the entering variable)
                if (!accepted && !remove_at(f)) { good = false; break; }
            } else { F.pop_back(); xF.pop_back(); cF.pop_back(); }
            if (!accepted) {  // numerically dependent on F, or not a descent direction after all: leave it out for now
                banned.push_back(b.k);
                if (!solve()) { good = false; break; }
                continue;
            }
            banned.clear();
            if (!settle()) { good = false; break; }
        }
        if (!done) good = false;
        if (good) {  // the optimum on the grid
            const int f = (int)F.size();
            (void)hipMemsetAsync(x, 0, sizeof(double) * (size_t)n * (size_t)ld, s);
            if (f) {
                upload_F();
                (void)hipMemcpyAsync(dv, xF.data(), sizeof(double) * (size_t)f, hipMemcpyHostToDevice, s);
                hipLaunchKernelGGL(k_scatter, blocks(f), dim3(T), 0, s, dF, f, dv, x, act, ld);
            }
            (void)hipStreamSynchronize(s);
        }
        rocblas_destroy_handle(bh);
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

extern "C" int32_t fnn_split_weights_f64(const double* D, int32_t n, int64_t ldD, const int32_t* ordering, int32_t device,
                                         double* weights_out, fnn_sw_stats* stats) {
    using namespace fnnsw;
    if (!D || !ordering || !weights_out || n < 2 || ldD < n) return fnn::fail(FNN_EINVAL, "fnn_split_weights_f64: bad arguments");
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
    hipEvent_t e0, e1;
    Solver S;
    S.n = n;
    S.ld = ((int64_t)n + 31) / 32 * 32 + 32;
    if (!SWOK(hipStreamCreateWithFlags(&S.s, hipStreamNonBlocking))) return fnn::fail(FNN_EHIP, "hipStreamCreate failed");
    const size_t NN = (size_t)n * (size_t)S.ld;
    S.d = S.alloc<double>(NN); S.x = S.alloc<double>(NN); S.r = S.alloc<double>(NN); S.w = S.alloc<double>(NN);
    S.p = S.alloc<double>(NN); S.y = S.alloc<double>(NN); S.old_x = S.alloc<double>(NN); S.atwd = S.alloc<double>(NN);
    S.P = S.alloc<double>(NN); S.Pt = S.alloc<double>(NN); S.rs = S.alloc<double>((size_t)n + 8);
    S.act = S.alloc<uint8_t>(NN);
    S.Dm = S.alloc<double>((size_t)n * (size_t)n);
    S.ord = S.alloc<int32_t>((size_t)n + 1);
    S.live = S.alloc<double>((size_t)n * (n - 1) / 2);
    S.grid2 = dim3((unsigned)((n + T - 1) / T), (unsigned)n);
    S.gred = dim3((unsigned)((n + T - 1) / T), (unsigned)(n < 256 ? n : 256));
    S.partial = S.alloc<double>((size_t)S.gred.x * S.gred.y);
    S.sc = S.alloc<double>(SC_WORDS);
    S.bpartial = S.alloc<Best>((size_t)S.gred.x * S.gred.y);
    if (!S.ok) return fnn::fail(FNN_ENOMEM, "fnn_split_weights_f64: device allocation failed");
    for (double* v : {S.d, S.x, S.r, S.w, S.p, S.y, S.old_x, S.atwd, S.P, S.Pt}) (void)hipMemsetAsync(v, 0, sizeof(double) * NN, S.s);
    if (!SWOK(hipMemcpy2DAsync(S.Dm, sizeof(double) * (size_t)n, D, sizeof(double) * (size_t)ldD, sizeof(double) * (size_t)n, (size_t)n,
                               hipMemcpyHostToDevice, S.s)) ||
        !SWOK(hipMemcpyAsync(S.ord, ordering, sizeof(int32_t) * ((size_t)n + 1), hipMemcpyHostToDevice, S.s)) ||
        !SWOK(hipStreamSynchronize(S.s)))
        return fnn::fail(FNN_EHIP, "fnn_split_weights_f64: upload failed");
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, S.s);
    hipLaunchKernelGGL(k_reorder, dim3((unsigned)((n + T - 1) / T), (unsigned)n), dim3(T), 0, S.s, S.Dm, (int64_t)n, S.ord, S.d, n, S.ld);
    // the closed form if it is feasible; else from below (Lawson-Hanson, exact sub-problems); the reference's own method
    // (from above, conjugate gradients) where the free set is too large for a dense factor
    bool from_below = false;
    hipLaunchKernelGGL(k_unconstrained, dim3((unsigned)((n + T - 1) / T), (unsigned)n), dim3(T), 0, S.s, S.d, S.x, n, S.ld);
    if (S.reduce_sum<RD_COUNT_NEG>(S.x, nullptr) != 0.0) {
        from_below = !std::getenv("FNN_SW_REFERENCE_METHOD") && S.lawson_hanson();
        if (!from_below) { S.st_lh_steps = 0; S.active_conjugate(); }
    }
    hipLaunchKernelGGL(k_to_live, S.grid2, dim3(T), 0, S.s, S.x, S.live, n, S.ld);
    (void)hipEventRecord(e1, S.s);
    hipError_t e = hipStreamSynchronize(S.s);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (e != hipSuccess || hipGetLastError() != hipSuccess) return fnn::fail(FNN_EHIP, std::string("fnn_split_weights_f64: ") + hipGetErrorString(e));
    if (!SWOK(hipMemcpy(weights_out, S.live, sizeof(double) * (size_t)n * (n - 1) / 2, hipMemcpyDeviceToHost)))
        return fnn::fail(FNN_EHIP, "fnn_split_weights_f64: download failed");
    if (stats) {
        stats->outer_iterations = from_below ? S.st_lh_steps : S.st_outer;
        stats->cg_calls = S.st_cg;
        stats->cg_iterations = S.st_it;
        stats->reserved[0] = from_below ? 1 : 0;        // method: 1 = from below (Lawson-Hanson, dense factor), 0 = the reference's (or the closed form)
        stats->reserved[1] = S.st_lh_refactor;
        stats->t_solve_s = ms * 1e-3;
        int64_t pos = 0;
        for (int64_t k = 0; k < (int64_t)n * (n - 1) / 2; k++) pos += weights_out[k] > 0.000001 ? 1 : 0;  // FastNN.java:455 threshold
        stats->nsplits = pos;
    }
    return FNN_OK;
}
