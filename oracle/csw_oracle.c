/*
 * csw_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, fp64) of the circular split-weight estimation of
 * JacobPorter/FastNeighborNet, CircularSplitWeights.java: Chepoi-Fichet closed form
 * (runUnconstrainedLS :247-271), the implicit operators A^T y and A b (calculateAtx :603-636,
 * calculateAb :647-731, rowsum :571-590), the conjugate-gradient solve on the free variables
 * (circularConjugateGrads :769-831), worstIndices (:283-337) and the active-set loop
 * (runActiveConjugate :366-557) for ordinary least squares (W = 1, setupV :217-219).
 *
 * Two deliberate differences from the file as checked in:
 *   * setupD (:202-211) no longer re-orders the distances by the circular ordering (the loop is
 *     commented out), which makes the class correct for the identity ordering only.  The
 *     restatement restores the commented-out loop: d'[a][b] = dist(ord[a+1], ord[b+1]).
 *   * the class is dead code on the reference's live path (FastNN.java:509-511 is commented
 *     out); the live path solves the same non-negative least-squares problem with a dense
 *     design matrix (FastNN.java:401-454).  The optimum is unique, so both agree to solver
 *     tolerance.  csw_live_index() maps a split of the fast algorithm to the live path's index.
 *
 * PARITY UNPINNED against the reference itself (no JVM here, no fixtures in the reference); the
 * restatement is pinned mathematically: tests compare it with a dense NNLS solve (scipy) of the
 * live path's design matrix and with the known weights of synthetic circular metrics.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define CG_EPSILON 1e-8 /* CircularSplitWeights.java:54 */

/* :571-590 */
static double rowsum(int n, const double* d, int k) {
    double r = 0;
    long index = 0;
    if (k > 0) {
        index = k - 1;
        for (int i = 0; i < k; i++) {
            r += d[index];
            index += (n - i - 2);
        }
        index++;
    }
    for (int j = k + 1; j < n; j++) r += d[index++];
    return r;
}

/* :603-636  p = A^T d */
void csw_calculate_atx(int n, const double* d, double* p) {
    long index = 0;
    for (int i = 0; i < n - 1; i++) {
        p[index] = rowsum(n, d, i + 1);
        index += (n - i - 1);
    }
    index = 1;
    for (int i = 0; i < n - 2; i++) {
        p[index] = p[index - 1] + p[index + (n - i - 2)] - 2 * d[index + (n - i - 2)];
        index += (n - i - 2) + 1;
    }
    for (int k = 3; k <= n - 1; k++) {
        index = k - 1;
        for (int i = 0; i <= n - k - 1; i++) {
            p[index] = p[index - 1] + p[index + n - i - 2] - p[index + n - i - 3] - 2.0 * d[index + n - i - 2];
            index += (n - i - 2) + 1;
        }
    }
}

/* :647-731  d = A b */
void csw_calculate_ab(int n, const double* b, double* d) {
    double d_ij;
    long index;
    long dindex = 0;
    for (int i = 0; i <= n - 2; i++) {
        d_ij = 0.0;
        index = i - 1;
        for (int k = 0; k <= i - 1; k++) {
            d_ij += b[index];
            index += (n - k - 2);
        }
        index++;
        for (int k = i + 1; k <= n - 1; k++) d_ij += b[index++];
        d[dindex] = d_ij;
        dindex += (n - i - 2) + 1;
    }
    index = 1;
    for (int i = 0; i <= n - 3; i++) {
        d[index] = d[index - 1] + d[index + (n - i - 2)] - 2 * b[index - 1];
        index += 1 + (n - i - 2);
    }
    for (int k = 3; k <= n - 1; k++) {
        index = k - 1;
        for (int i = 0; i <= n - k - 1; i++) {
            d[index] = d[index - 1] + d[index + (n - i - 2)] - d[index + (n - i - 2) - 1] - 2.0 * b[index - 1];
            index += 1 + (n - i - 2);
        }
    }
}

/* :247-271  Chepoi & Fichet */
void csw_unconstrained_ls(int n, const double* d, double* x) {
    long index = 0;
    for (int i = 0; i <= n - 3; i++) {
        x[index] = (d[index] + d[index + (n - i - 2) + 1] - d[index + 1]) / 2.0;
        index++;
        for (int j = i + 2; j <= n - 2; j++) {
            x[index] = (d[index] + d[index + (n - i - 2) + 1] - d[index + 1] - d[index + (n - i - 2)]) / 2.0;
            index++;
        }
        if (i == 0) x[index] = (d[0] + d[n - 2] - d[2 * n - 4]) / 2.0;
        else x[index] = (d[index] + d[i] - d[i - 1] - d[index + (n - i - 2)]) / 2.0;
        index++;
    }
    x[index] = (d[index] + d[n - 2] - d[n - 3]) / 2.0;
}

static double norm(const double* x, long n) { /* :743-752 */
    double ss = 0.0;
    for (long k = 0; k < n; k++) ss += x[k] * x[k];
    return ss;
}

static int cmp_double(const void* a, const void* b) {
    double x = *(const double*)a, y = *(const double*)b;
    return (x > y) - (x < y);
}

/* :283-337; returns the number of indices written to result (0: none) */
static long worst_indices(const double* x, long n, double propKept, long** result_out) {
    *result_out = NULL;
    if (propKept == 0) return 0;
    long numNeg = 0;
    for (long i = 0; i < n; i++)
        if (x[i] < 0.0) numNeg++;
    if (numNeg == 0) return 0;
    double* xcopy = (double*)malloc(sizeof(double) * (size_t)numNeg);
    long j = 0;
    for (long i = 0; i < n; i++)
        if (x[i] < 0.0) xcopy[j++] = x[i];
    qsort(xcopy, (size_t)numNeg, sizeof(double), cmp_double);
    long nkept = (long)ceil(propKept * (double)numNeg);
    double cutoff = xcopy[nkept - 1];
    free(xcopy);
    long* result = (long*)malloc(sizeof(long) * (size_t)nkept);
    long front = 0, back = nkept - 1;
    for (long i = 0; i < n; i++) {
        if (x[i] < cutoff) result[front++] = i;
        else if (x[i] == cutoff) {
            if (back >= front) result[back--] = i;
        }
    }
    *result_out = result;
    return nkept;
}

/* :769-831 (W = 1) */
static long cg(int ntax, long npairs, double* r, double* w, double* p, double* y, const double* b,
               const unsigned char* active, double* x) {
    long kmax = (long)ntax * (ntax - 1) / 2;
    csw_calculate_ab(ntax, x, y);
    csw_calculate_atx(ntax, y, r);
    for (long k = 0; k < npairs; k++) r[k] = active[k] ? 0.0 : b[k] - r[k];
    double rho = norm(r, npairs), rho_old = 0;
    double e_0 = CG_EPSILON * sqrt(norm(b, npairs));
    long k = 0;
    while ((rho > e_0 * e_0) && (k < kmax)) {
        k = k + 1;
        if (k == 1) memcpy(p, r, sizeof(double) * (size_t)npairs);
        else {
            double beta = rho / rho_old;
            for (long i = 0; i < npairs; i++) p[i] = r[i] + beta * p[i];
        }
        csw_calculate_ab(ntax, p, y);
        csw_calculate_atx(ntax, y, w);
        for (long i = 0; i < npairs; i++)
            if (active[i]) w[i] = 0.0;
        double alpha = 0.0;
        for (long i = 0; i < npairs; i++) alpha += p[i] * w[i];
        alpha = rho / alpha;
        for (long i = 0; i < npairs; i++) {
            x[i] += alpha * p[i];
            r[i] -= alpha * w[i];
        }
        rho_old = rho;
        rho = norm(r, npairs);
    }
    return k;
}

/* :366-557 (OLS, collapse_many_negs = true, useMax = false).  stats3 (may be NULL):
 * {outer iterations, CG calls, CG iterations}. */
void csw_active_conjugate(int ntax, const double* d, double* x, long* stats3) {
    long npairs = (long)ntax * (ntax - 1) / 2;
    long st_outer = 0, st_cg = 0, st_it = 0;
    csw_unconstrained_ls(ntax, d, x);
    int all_positive = 1;
    for (long k = 0; k < npairs && all_positive; k++)
        if (x[k] < 0.0) all_positive = 0;
    if (all_positive) {
        if (stats3) { stats3[0] = 0; stats3[1] = 0; stats3[2] = 0; }
        return;
    }
    size_t B = sizeof(double) * (size_t)npairs;
    double* r = (double*)calloc(1, B);
    double* w = (double*)calloc(1, B);
    double* p = (double*)calloc(1, B);
    double* y = (double*)calloc(1, B);
    double* old_x = (double*)malloc(B);
    double* AtWd = (double*)calloc(1, B);
    unsigned char* active = (unsigned char*)calloc(1, (size_t)npairs);
    for (long k = 0; k < npairs; k++) old_x[k] = 1.0;
    for (long k = 0; k < npairs; k++) y[k] = d[k];
    csw_calculate_atx(ntax, y, AtWd);
    int first_pass = 1;
    for (;;) {
        st_outer++;
        for (;;) {
            if (!first_pass) { st_it += cg(ntax, npairs, r, w, p, y, AtWd, active, x); st_cg++; }
            first_pass = 0;
            {
                long* contract = NULL;
                long num = worst_indices(x, npairs, 0.6, &contract);
                if (contract != NULL) {
                    for (long k = 0; k < num; k++) {
                        x[contract[k]] = 0.0;
                        active[contract[k]] = 1;
                    }
                    free(contract);
                    st_it += cg(ntax, npairs, r, w, p, y, AtWd, active, x);
                    st_cg++;
                }
            }
            long min_i = -1;
            double min_xi = -1.0;
            for (long i = 0; i < npairs; i++) {
                if (x[i] < 0.0) {
                    double xi = (old_x[i]) / (old_x[i] - x[i]);
                    if ((min_i == -1) || (xi < min_xi)) {
                        min_i = i;
                        min_xi = xi;
                    }
                }
            }
            if (min_i == -1) break;
            for (long i = 0; i < npairs; i++)
                if (!active[i]) old_x[i] += min_xi * (x[i] - old_x[i]);
            active[min_i] = 1;
            x[min_i] = 0.0;
        }
        csw_calculate_ab(ntax, x, y);
        csw_calculate_atx(ntax, y, r);
        long min_i = -1;
        double min_grad = 1.0;
        for (long i = 0; i < npairs; i++) {
            r[i] -= AtWd[i];
            r[i] *= 2.0;
            if (active[i]) {
                double grad_ij = r[i];
                if ((min_i == -1) || (grad_ij < min_grad)) {
                    min_i = i;
                    min_grad = grad_ij;
                }
            }
        }
        if ((min_i == -1) || (min_grad > -0.0000001)) break;
        active[min_i] = 0;
    }
    free(r); free(w); free(p); free(y); free(old_x); free(AtWd); free(active);
    if (stats3) { stats3[0] = st_outer; stats3[1] = st_cg; stats3[2] = st_it; }
}

/* restored setupD (:202-211, commented-out loop): packed upper triangle of the distances
 * re-ordered by the circular ordering (ordering[1..n], 1-based taxon ids; D is n x n row-major) */
void csw_setup_d(int n, const double* D, const int32_t* ordering, double* d) {
    long index = 0;
    for (int i = 0; i < n; i++)
        for (int j = i + 1; j < n; j++)
            d[index++] = D[(size_t)(ordering[i + 1] - 1) * (size_t)n + (size_t)(ordering[j + 1] - 1)];
}

/* index of the fast algorithm's split (i, j), i < j (positions i+1 .. j of the cycle, 0-based):
 * (2n - i - 3) i / 2 + j - 1  (:27-35) */
long csw_pair_index(int n, int i, int j) { return ((long)(2 * n - i - 3) * i) / 2 + j - 1; }

/* live path (FastNN.java:405-419): split k = (i, j), 0 <= i < j <= n-1, taxa ordering[i+1 .. j]; in the
 * fast algorithm's numbering that is split (i-1, j-1) for i >= 1 and, for i = 0, the complement
 * {j .. n-1} = split (j-1, n-1).  Weights of the live index space from x. */
void csw_to_live(int n, const double* x, double* live) {
    long k = 0;
    for (int i = 0; i < n; i++)
        for (int j = i + 1; j < n; j++, k++)
            live[k] = (i >= 1) ? x[csw_pair_index(n, i - 1, j - 1)] : x[csw_pair_index(n, j - 1, n - 1)];
}

/* whole path: distances + circular ordering -> non-negative least-squares split weights in the live
 * path's index order */
void csw_split_weights(int n, const double* D, const int32_t* ordering, double* live, long* stats3) {
    long npairs = (long)n * (n - 1) / 2;
    double* d = (double*)malloc(sizeof(double) * (size_t)npairs);
    double* x = (double*)calloc((size_t)npairs, sizeof(double));
    csw_setup_d(n, D, ordering, d);
    csw_active_conjugate(n, d, x, stats3);
    csw_to_live(n, x, live);
    free(d);
    free(x);
}
