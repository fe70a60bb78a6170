/* oracle/o_linalg.c -- TEST INFRASTRUCTURE (see oracle.h).
 * CPU restatement of eul/LinAlg.cpp dense helpers on flat row-major double arrays, plus the
 * small direct solver that stands in for PETSc's PCLU on per-column systems. */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

/* C = A(ni x nk) B(nk x nj), naive ijk with the running sum kept in C -- Mult_IP :87-97 */
void orc_mult(int ni, int nj, int nk, double* A, double* B, double* C) {
    int i, j, k;
    for (i = 0; i < ni; i++)
        for (j = 0; j < nj; j++) {
            C[i*nj+j] = 0.0;
            for (k = 0; k < nk; k++) C[i*nj+j] += A[i*nk+k]*B[k*nj+j];
        }
}

/* C[i][j] = A[i][j] d[j]  (full x diagonal) -- Mult_FD_IP :115-132 */
void orc_mult_fd(int ni, int nj, int nk, double* A, double* d, double* C) {
    int i, j;
    (void)nk;
    for (i = 0; i < ni; i++)
        for (j = 0; j < nj; j++) C[i*nj+j] = A[i*nj+j]*d[j];
}

/* C[i][j] = d[i] B[i][j]  (diagonal x full) -- Mult_DF_IP :104-112 */
void orc_mult_df(int ni, int nj, int nk, double* d, double* B, double* C) {
    int i, j;
    (void)nk;
    for (i = 0; i < ni; i++)
        for (j = 0; j < nj; j++) C[i*nj+j] = d[i]*B[i*nj+j];
}

/* B = A^T -- Tran_IP :151-159 */
void orc_tran(int ni, int nj, double* A, double* B) {
    int i, j;
    for (i = 0; i < ni; i++)
        for (j = 0; j < nj; j++) B[j*ni+i] = A[i*nj+j];
}

/* b = A x -- Ax_b :162-170 */
void orc_axb(int ni, int nj, double* A, double* x, double* b) {
    int i, j;
    for (i = 0; i < ni; i++) {
        b[i] = 0.0;
        for (j = 0; j < nj; j++) b[i] += A[i*nj+j]*x[j];
    }
}

/* Gauss-Jordan inverse with full pivoting; error 1 = pivot reused, 2 = |pivot| < 1e-12.
 * Inv :186-269: identical search order (>= keeps the LAST maximal entry), row swap,
 * normalise, eliminate, and the final column un-permutation. */
int orc_inv(double* A, double* Ainv, int n) {
    int err = 0, i, j, k, l, ll, irow = 0, icol = 0;
    int* indxc = (int*)malloc(sizeof(int)*n);
    int* indxr = (int*)malloc(sizeof(int)*n);
    int* ipiv  = (int*)malloc(sizeof(int)*n);
    double big, dum, pivinv, t;

    for (i = 0; i < n*n; i++) Ainv[i] = A[i];
    for (j = 0; j < n; j++) ipiv[j] = 0;
    for (i = 0; i < n; i++) {
        big = 0.0;
        for (j = 0; j < n; j++) {
            if (ipiv[j] == 1) continue;
            for (k = 0; k < n; k++) {
                if (ipiv[k] == 0) {
                    if (fabs(Ainv[j*n+k]) >= big) { big = fabs(Ainv[j*n+k]); irow = j; icol = k; }
                } else if (ipiv[k] > 1) err = 1;
            }
        }
        ++ipiv[icol];
        if (irow != icol)
            for (l = 0; l < n; l++) { t = Ainv[irow*n+l]; Ainv[irow*n+l] = Ainv[icol*n+l]; Ainv[icol*n+l] = t; }
        indxr[i] = irow;
        indxc[i] = icol;
        if (fabs(Ainv[icol*n+icol]) < 1.0e-12) err = 2;
        pivinv = 1.0/Ainv[icol*n+icol];
        Ainv[icol*n+icol] = 1.0;
        for (l = 0; l < n; l++) Ainv[icol*n+l] *= pivinv;
        for (ll = 0; ll < n; ll++) {
            if (ll == icol) continue;
            dum = Ainv[ll*n+icol];
            Ainv[ll*n+icol] = 0.0;
            for (l = 0; l < n; l++) Ainv[ll*n+l] -= Ainv[icol*n+l]*dum;
        }
    }
    for (l = n-1; l >= 0; l--) {
        if (indxr[l] == indxc[l]) continue;
        for (k = 0; k < n; k++) { t = Ainv[k*n+indxr[l]]; Ainv[k*n+indxr[l]] = Ainv[k*n+indxc[l]]; Ainv[k*n+indxc[l]] = t; }
    }
    free(indxc); free(indxr); free(ipiv);
    return err;
}

/* ---- swappable table used by every assembly routine ---------------------------------- */
static const orc_linalg orc_builtin = { orc_mult, orc_mult_fd, orc_mult_df, orc_tran, orc_axb, orc_inv };
const orc_linalg* orc_la = &orc_builtin;
static orc_linalg orc_user;

void orc_set_linalg(const orc_linalg* la) {
    if (!la) { orc_la = &orc_builtin; return; }
    orc_user = *la;
    if (!orc_user.mult)    orc_user.mult    = orc_mult;
    if (!orc_user.mult_fd) orc_user.mult_fd = orc_mult_fd;
    if (!orc_user.mult_df) orc_user.mult_df = orc_mult_df;
    if (!orc_user.tran)    orc_user.tran    = orc_tran;
    if (!orc_user.axb)     orc_user.axb     = orc_axb;
    if (!orc_user.inv)     orc_user.inv     = orc_inv;
    orc_la = &orc_user;
}

/* Dense LU with partial pivoting: stand-in for PETSc KSPPREONLY/PCLU on MATSEQAIJ column
 * systems (third-party arithmetic outside /root/reference; any correct direct solver agrees
 * to round-off).  A is overwritten. */
int orc_dense_solve(int n, double* A, double* b, double* x) {
    int i, j, k, piv;
    double big, t, f;
    for (i = 0; i < n; i++) x[i] = b[i];
    for (k = 0; k < n; k++) {
        piv = k; big = fabs(A[k*n+k]);
        for (i = k+1; i < n; i++) if (fabs(A[i*n+k]) > big) { big = fabs(A[i*n+k]); piv = i; }
        if (big == 0.0) return 1;
        if (piv != k) {
            for (j = 0; j < n; j++) { t = A[k*n+j]; A[k*n+j] = A[piv*n+j]; A[piv*n+j] = t; }
            t = x[k]; x[k] = x[piv]; x[piv] = t;
        }
        for (i = k+1; i < n; i++) {
            f = A[i*n+k]/A[k*n+k];
            if (f == 0.0) continue;
            for (j = k; j < n; j++) A[i*n+j] -= f*A[k*n+j];
            x[i] -= f*x[k];
        }
    }
    for (i = n-1; i >= 0; i--) {
        t = x[i];
        for (j = i+1; j < n; j++) t -= A[i*n+j]*x[j];
        x[i] = t/A[i*n+i];
    }
    return 0;
}
