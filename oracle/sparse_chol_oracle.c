/*
 * oracle/sparse_chol_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Sparse restatement of the third-party factorisation behind the reference's
 *   R/vecchia_prediction.R:74-81   V.ord = t(Matrix::chol(revMat(tcrossprod(U.y))))
 *   R/vecchia_prediction.R:124-125 Matrix::solve(V.ord, .), Matrix::solve(t(V.ord), .)
 *   R/vecchia_likelihood.R:88      Matrix::solve(V.ord, rev(z2), system='L')
 * Matrix::chol on a dsCMatrix with its default pivot = FALSE is CHOLMOD's Cholesky
 * in the NATURAL order (no fill-reducing permutation): A = L L^T, L lower triangular.
 * CHOLMOD is not in this image; this file restates the published up-looking algorithm
 * (elimination tree by Liu's ancestor compression; row k of L = the reach of the
 * entries of A(0:k-1, k) in the tree; one sparse triangular solve per row), which
 * yields the same L in exact arithmetic (the Cholesky factor is unique), summation
 * order aside.  Checked against numpy's dense Cholesky in tests/test_oracle.py.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's parity legs may load it.
 *
 * Storage: compressed sparse column, 0-based, long indices.  The input holds the UPPER
 * triangle of A by columns (entries with row > column are ignored), i.e. the lower
 * triangle by rows; the output L is CSC, the diagonal entry FIRST in every column,
 * rows ascending.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* elimination tree of A (upper triangle by columns): parent[k] or -1 */
static void spo_etree(long n, const long *Ap, const long *Ai, long *parent, long *ancestor)
{
    for (long k = 0; k < n; ++k) {
        parent[k] = -1;
        ancestor[k] = -1;
        for (long p = Ap[k]; p < Ap[k + 1]; ++p) {
            long i = Ai[p];
            while (i != -1 && i < k) {             /* walk from i towards the root, compressing the path onto k */
                long next = ancestor[i];
                ancestor[i] = k;
                if (next == -1) parent[i] = k;
                i = next;
            }
        }
    }
}

/* pattern of row k of L (strictly below the diagonal in column terms: columns < k), returned in stack[top..n-1]
 * in an order in which every column comes after all columns it depends on; flag[] marks visited nodes with k */
static long spo_reach(long n, const long *Ap, const long *Ai, long k, const long *parent, long *stack, long *path,
                      long *flag)
{
    long top = n;
    flag[k] = k;
    for (long p = Ap[k]; p < Ap[k + 1]; ++p) {
        long i = Ai[p];
        if (i >= k) continue;
        long len = 0;
        while (flag[i] != k) {                     /* climb until a node already in the pattern */
            path[len++] = i;
            flag[i] = k;
            i = parent[i];
        }
        while (len > 0) stack[--top] = path[--len];
    }
    return top;
}

/* Symbolic pass: number of entries of every column of L (diagonal included) into colcount[n]; returns nnz(L). */
long oracle_sparse_chol_symbolic(long n, const long *Ap, const long *Ai, long *parent, long *colcount)
{
    long *ancestor = (long *)malloc(sizeof(long) * (size_t)(n > 0 ? n : 1));
    long *stack = (long *)malloc(sizeof(long) * (size_t)(n > 0 ? n : 1));
    long *path = (long *)malloc(sizeof(long) * (size_t)(n > 0 ? n : 1));
    long *flag = (long *)malloc(sizeof(long) * (size_t)(n > 0 ? n : 1));
    if (!ancestor || !stack || !path || !flag) { free(ancestor); free(stack); free(path); free(flag); return -1; }
    spo_etree(n, Ap, Ai, parent, ancestor);
    for (long k = 0; k < n; ++k) { colcount[k] = 1; flag[k] = -1; }
    long nnz = n;
    for (long k = 0; k < n; ++k) {
        long top = spo_reach(n, Ap, Ai, k, parent, stack, path, flag);
        for (long t = top; t < n; ++t) colcount[stack[t]]++;
        nnz += n - top;
    }
    free(ancestor); free(stack); free(path); free(flag);
    return nnz;
}

/* Numeric up-looking Cholesky.  Lp[n+1] must hold the column pointers (prefix sums of colcount), Li / Lx sized nnz(L).
 * Returns 0, or k+1 when the k-th pivot is not positive (the factorisation stops there, like CHOLMOD's "not positive
 * definite" error which Matrix::chol turns into an R error).
 * Two instances of the numeric routines: double (the oracle proper) and x87 long double (the adjudicator of
 * tests/: the same factorisation to ~cond * 1e-19, against which both the oracle's and the HIP path's posterior means
 * are measured where they differ by more than the flat tolerance). */
#define SPO_DEFINE(REAL, SUF, SQRT)                                                                                    \
long oracle_sparse_chol_numeric##SUF(long n, const long *Ap, const long *Ai, const REAL *Ax, const long *parent,       \
                                     const long *Lp, long *Li, REAL *Lx)                                               \
{                                                                                                                      \
    long *stack = (long *)malloc(sizeof(long) * (size_t)(n > 0 ? n : 1));                                              \
    long *path = (long *)malloc(sizeof(long) * (size_t)(n > 0 ? n : 1));                                               \
    long *flag = (long *)malloc(sizeof(long) * (size_t)(n > 0 ? n : 1));                                               \
    long *fill = (long *)malloc(sizeof(long) * (size_t)(n > 0 ? n : 1));     /* next free slot of every column */       \
    REAL *x = (REAL *)calloc((size_t)(n > 0 ? n : 1), sizeof(REAL));                                                   \
    long status = 0;                                                                                                   \
    if (!stack || !path || !flag || !fill || !x) { status = -1; goto done; }                                           \
    for (long k = 0; k < n; ++k) { flag[k] = -1; fill[k] = Lp[k]; }                                                    \
    for (long k = 0; k < n; ++k) {                                                                                     \
        long top = spo_reach(n, Ap, Ai, k, parent, stack, path, flag);                                                 \
        REAL d = 0;                                                                                                    \
        for (long p = Ap[k]; p < Ap[k + 1]; ++p) {         /* scatter column k of the upper triangle */                \
            long i = Ai[p];                                                                                            \
            if (i < k) x[i] = Ax[p];                                                                                   \
            else if (i == k) d = Ax[p];                                                                                \
        }                                                                                                              \
        for (long t = top; t < n; ++t) {                   /* L(0:k-1,0:k-1) y = A(0:k-1,k), y = row k of L */          \
            long i = stack[t];                                                                                         \
            REAL lki = x[i] / Lx[Lp[i]];                    /* diagonal is the first entry of column i */               \
            x[i] = 0;                                                                                                  \
            for (long p = Lp[i] + 1; p < fill[i]; ++p) x[Li[p]] -= Lx[p] * lki;                                        \
            d -= lki * lki;                                                                                            \
            long q = fill[i]++;                                                                                        \
            Li[q] = k;                                                                                                 \
            Lx[q] = lki;                                                                                               \
        }                                                                                                              \
        if (!(d > 0)) { status = k + 1; goto done; }                                                                   \
        long q = fill[k]++;                                                                                            \
        Li[q] = k;                                                                                                     \
        Lx[q] = SQRT(d);                                                                                               \
    }                                                                                                                  \
done:                                                                                                                  \
    free(stack); free(path); free(flag); free(fill); free(x);                                                          \
    return status;                                                                                                     \
}                                                                                                                      \
/* x <- L^{-1} x (CSC lower triangular, diagonal first in every column) */                                             \
void oracle_sparse_lsolve##SUF(long n, const long *Lp, const long *Li, const REAL *Lx, REAL *x)                        \
{                                                                                                                      \
    for (long j = 0; j < n; ++j) {                                                                                     \
        x[j] /= Lx[Lp[j]];                                                                                             \
        const REAL xj = x[j];                                                                                          \
        for (long p = Lp[j] + 1; p < Lp[j + 1]; ++p) x[Li[p]] -= Lx[p] * xj;                                           \
    }                                                                                                                  \
}                                                                                                                      \
/* x <- L^{-T} x */                                                                                                    \
void oracle_sparse_ltsolve##SUF(long n, const long *Lp, const long *Li, const REAL *Lx, REAL *x)                       \
{                                                                                                                      \
    for (long j = n - 1; j >= 0; --j) {                                                                                \
        REAL s = x[j];                                                                                                 \
        for (long p = Lp[j] + 1; p < Lp[j + 1]; ++p) s -= Lx[p] * x[Li[p]];                                            \
        x[j] = s / Lx[Lp[j]];                                                                                          \
    }                                                                                                                  \
}

SPO_DEFINE(double, , sqrt)
SPO_DEFINE(long double, _ld, sqrtl)
