/* oracle/o_vertops.c -- TEST INFRASTRUCTURE (see oracle.h).
 * CPU restatement of the per-column operators of eul/VertOps.cpp and of the column Schur
 * solve eul/VertSolve.cpp:677-823.  Column vectors are indexed k*n2e+i (VertOps.cpp:215-217).
 * Where the reference fills tiny MATSEQAIJ matrices and chains MatMatMult/PCLU (PETSc, not in
 * /root/reference) this file fills DENSE row-major matrices in the same MatSetValues order and
 * uses plain dense products / LU: same algebra, obviously-correct, slow -- a checker. */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

extern const orc_linalg* orc_la;

#define RD 287.0
#define CP 1004.5
#define CV 717.5
#define P0 100000.0
#define SCALE 1.0e+8

/* WtQW = Wt diag(c) W -- the Mult_FD_IP + Mult_IP pair every Assemble* ends with */
static void wqw(const orc_patch* p, double* c, double* tmp, double* out) {
    orc_la->mult_fd(p->n2e, p->mp12, p->mp12, p->Wt, c, tmp);
    orc_la->mult(p->n2e, p->n2e, p->mp12, tmp, p->W, out);
}
/* interpolate a column 2-form at level-slot k to quad point ii: sum_j f[k*n2+j] W[ii][j] */
static double w_interp(const orc_patch* p, const double* f, int k, int ii) {
    double r = 0.0; int j;
    for (j = 0; j < p->n2e; j++) r += f[k*p->n2e + j]*p->W[ii*p->n2e + j];
    return r;
}
/* MatSetValues of an n2e x n2e block at block position (bi,bj) of a dense (.. x ldc) matrix */
static void put(const orc_patch* p, double* M, int ldc, int bi, int bj, const double* blk, int add) {
    int i, j, n2 = p->n2e;
    for (i = 0; i < n2; i++) for (j = 0; j < n2; j++) {
        double* d = &M[(size_t)(bi*n2 + i)*ldc + bj*n2 + j];
        if (add) *d += blk[i*n2+j]; else *d = blk[i*n2+j];
    }
}

int orc_colop_dims(const orc_patch* p, int colop, int* rows, int* cols) {
    int nk = p->nk, n2 = p->n2e;
    switch (colop) {
    case ORC_V_CONST: case ORC_V_CONST_INV: case ORC_V_CONST_RHO: case ORC_V_CONST_RHO_INV:
    case ORC_V_CONST_THETA: case ORC_V_EOS_BLOCK:
        *rows = nk*n2; *cols = nk*n2; return 0;
    case ORC_V_LINEAR: case ORC_V_LINEAR_INV: case ORC_V_LINEAR_RT: case ORC_V_LINEAR_THETA: case ORC_V_RAYLEIGH:
        *rows = (nk-1)*n2; *cols = (nk-1)*n2; return 0;
    case ORC_V_LINEAR_RHO2: *rows = (nk+1)*n2; *cols = (nk+1)*n2; return 0;
    case ORC_V_LINCON:  *rows = (nk-1)*n2; *cols = nk*n2; return 0;
    case ORC_V_LINCON2: *rows = (nk+1)*n2; *cols = nk*n2; return 0;
    case ORC_V_CONLIN: case ORC_V_CONLIN_W: case ORC_V_CONLIN_RHODPI:
        *rows = nk*n2; *cols = (nk-1)*n2; return 0;
    }
    return 1;
}

int orc_colop_dense(const orc_patch* p, int colop, int ex, int ey, int flag,
                    const double* f1, const double* f2, double* out) {
    int rows, cols, kk, ii, nk = p->nk, mp12 = p->mp12, n2 = p->n2e, ei = ey*p->nElsX + ex;
    int iq[128];
    double Q0[128], QT[128], QB[128];
    double *tmp = (double*)malloc(sizeof(double)*n2*mp12), *blk = (double*)malloc(sizeof(double)*n2*n2),
           *binv = (double*)malloc(sizeof(double)*n2*n2), *t1 = (double*)malloc(sizeof(double)*n2*n2),
           *t2 = (double*)malloc(sizeof(double)*n2*n2);
    const double* det = p->det + (size_t)ei*mp12;
    if (orc_colop_dims(p, colop, &rows, &cols)) return 1;
    memset(out, 0, sizeof(double)*(size_t)rows*cols);
    orc_elindsq_l(p, ex, ey, iq);
#define TH(k, ii)  p->thick[(size_t)(k)*p->n0q + iq[ii]]
#define THI(k, ii) p->thickInv[(size_t)(k)*p->n0q + iq[ii]]

    if (colop == ORC_V_RAYLEIGH) {   /* AssembleRayleigh :826-888: top three interfaces */
        static const double wgt[3] = {0.5, 0.25, 0.125};
        int s;
        for (s = 0; s < 3; s++) {
            for (ii = 0; ii < mp12; ii++) {
                Q0[ii] = p->Q[ii]*(SCALE/det[ii]);
                Q0[ii] *= wgt[s]*(TH(nk-1-s, ii) + TH(nk-2-s, ii));
            }
            wqw(p, Q0, tmp, blk);
            put(p, out, cols, nk-2-s, nk-2-s, blk, 1);
        }
        goto done;
    }
    if (colop == ORC_V_LINEAR_INV) { /* AssembleLinearInv :411-443 */
        for (kk = 0; kk < nk-1; kk++) {
            for (ii = 0; ii < mp12; ii++) {
                Q0[ii] = p->Q[ii]*(SCALE/det[ii]);
                Q0[ii] *= 0.5*(TH(kk, ii) + TH(kk+1, ii));
            }
            wqw(p, Q0, tmp, blk);
            orc_la->inv(blk, binv, n2);
            put(p, out, cols, kk, kk, binv, 1);
        }
        goto done;
    }

    for (kk = 0; kk < nk; kk++) {
        double rk, tb, tt, wb, wt;
        switch (colop) {
        case ORC_V_CONST:       /* AssembleConst :201-220 */
        case ORC_V_CONST_INV:   /* AssembleConstInv :803-818 */
            for (ii = 0; ii < mp12; ii++) { Q0[ii] = p->Q[ii]*(SCALE/det[ii]); Q0[ii] *= THI(kk, ii); }
            wqw(p, Q0, tmp, blk);
            if (colop == ORC_V_CONST) put(p, out, cols, kk, kk, blk, 0);
            else { orc_la->inv(blk, binv, n2); put(p, out, cols, kk, kk, binv, 1); }
            break;
        case ORC_V_CONST_RHO:     /* AssembleConstWithRho :508-532 */
        case ORC_V_CONST_RHO_INV: /* AssembleConstWithRhoInv :461-486 */
            for (ii = 0; ii < mp12; ii++) {
                Q0[ii] = p->Q[ii]*(SCALE/det[ii]);
                Q0[ii] *= THI(kk, ii);
                rk = w_interp(p, f1, kk, ii);
                Q0[ii] *= rk/(TH(kk, ii)*det[ii]);
            }
            wqw(p, Q0, tmp, blk);
            if (colop == ORC_V_CONST_RHO) put(p, out, cols, kk, kk, blk, 1);
            else { orc_la->inv(blk, binv, n2); put(p, out, cols, kk, kk, binv, 1); }
            break;
        case ORC_V_CONST_THETA:   /* AssembleConstWithTheta :946-971 */
            for (ii = 0; ii < mp12; ii++) {
                Q0[ii] = p->Q[ii]*(SCALE/det[ii]);
                Q0[ii] *= THI(kk, ii);
                tb = w_interp(p, f1, kk, ii); tt = w_interp(p, f1, kk+1, ii);
                Q0[ii] *= 0.5*(tb + tt)/det[ii];
            }
            wqw(p, Q0, tmp, blk);
            put(p, out, cols, kk, kk, blk, 1);
            break;
        case ORC_V_EOS_BLOCK:     /* Assemble_EOS_Block :1162-1196  B . B(rt)^-1 . B */
            for (ii = 0; ii < mp12; ii++) {
                Q0[ii] = p->Q[ii]*(SCALE/det[ii]);
                Q0[ii] *= THI(kk, ii);
                rk = w_interp(p, f1, kk, ii);
                Q0[ii] *= rk/(TH(kk, ii)*det[ii]);
            }
            wqw(p, Q0, tmp, blk);
            orc_la->inv(blk, binv, n2);
            for (ii = 0; ii < mp12; ii++) { Q0[ii] = p->Q[ii]*(SCALE/det[ii]); Q0[ii] *= THI(kk, ii); }
            wqw(p, Q0, tmp, blk);
            orc_la->mult(n2, n2, n2, binv, blk, t1);
            orc_la->mult(n2, n2, n2, blk, t1, t2);
            put(p, out, cols, kk, kk, t2, 1);
            break;
        case ORC_V_LINEAR:        /* AssembleLinear :242-267 */
            for (ii = 0; ii < mp12; ii++) { Q0[ii] = p->Q[ii]*(SCALE/det[ii]); Q0[ii] *= 0.5*TH(kk, ii); }
            wqw(p, Q0, tmp, blk);
            if (kk > 0) put(p, out, cols, kk-1, kk-1, blk, 1);
            if (kk < nk-1) put(p, out, cols, kk, kk, blk, 1);
            break;
        case ORC_V_LINEAR_RT:     /* AssembleLinearWithRT :621-662 ; flag = do_internal */
            if (kk > 0 && kk < nk-1 && !flag) break;
            for (ii = 0; ii < mp12; ii++) {
                Q0[ii] = p->Q[ii]*(SCALE/det[ii]);
                rk = w_interp(p, f1, kk, ii);
                if (!flag) rk *= THI(kk, ii);
                Q0[ii] *= 0.5*rk/det[ii];
            }
            wqw(p, Q0, tmp, blk);
            if (kk > 0) put(p, out, cols, kk-1, kk-1, blk, 1);
            if (kk < nk-1) put(p, out, cols, kk, kk, blk, 1);
            break;
        case ORC_V_LINEAR_THETA:  /* AssembleLinearWithTheta :685-725 */
            for (ii = 0; ii < mp12; ii++) {
                QB[ii] = p->Q[ii]*(SCALE/det[ii]);
                QB[ii] *= 0.5*TH(kk, ii);
                QT[ii] = QB[ii];
                tb = w_interp(p, f1, kk, ii); tt = w_interp(p, f1, kk+1, ii);
                QB[ii] *= tb/det[ii];
                QT[ii] *= tt/det[ii];
            }
            if (kk > 0) { wqw(p, QB, tmp, blk); put(p, out, cols, kk-1, kk-1, blk, 1); }
            if (kk < nk-1) { wqw(p, QT, tmp, blk); put(p, out, cols, kk, kk, blk, 1); }
            break;
        case ORC_V_LINEAR_RHO2:   /* AssembleLinearWithRho2 :375-403 (nk+1 interfaces, no bcs) */
            for (ii = 0; ii < mp12; ii++) {
                Q0[ii] = p->Q[ii]*(SCALE/det[ii]);
                rk = w_interp(p, f1, kk, ii);
                Q0[ii] *= 0.5*rk/det[ii];
            }
            wqw(p, Q0, tmp, blk);
            put(p, out, cols, kk, kk, blk, 1);
            put(p, out, cols, kk+1, kk+1, blk, 1);
            break;
        case ORC_V_LINCON:        /* AssembleLinCon :285-313 */
        case ORC_V_LINCON2:       /* AssembleLinCon2 :331-355 */
        case ORC_V_CONLIN:        /* AssembleConLin :901-924 */
            for (ii = 0; ii < mp12; ii++) { Q0[ii] = p->Q[ii]*(SCALE/det[ii]); Q0[ii] *= 0.5; }
            wqw(p, Q0, tmp, blk);
            if (colop == ORC_V_LINCON) {
                if (kk > 0) put(p, out, cols, kk-1, kk, blk, 1);
                if (kk < nk-1) put(p, out, cols, kk, kk, blk, 1);
            } else if (colop == ORC_V_LINCON2) {
                put(p, out, cols, kk, kk, blk, 1);
                put(p, out, cols, kk+1, kk, blk, 1);
            } else {
                if (kk > 0) put(p, out, cols, kk, kk-1, blk, 1);
                if (kk < nk-1) put(p, out, cols, kk, kk, blk, 1);
            }
            break;
        case ORC_V_CONLIN_W:      /* AssembleConLinWithW :551-600 ; f1 = velz on interfaces */
            if (kk > 0) {
                for (ii = 0; ii < mp12; ii++) {
                    Q0[ii] = p->Q[ii]*(SCALE/det[ii]);
                    wb = w_interp(p, f1, kk-1, ii);
                    Q0[ii] *= 0.5*wb/det[ii];
                }
                wqw(p, Q0, tmp, blk);
                put(p, out, cols, kk, kk-1, blk, 0);
            }
            if (kk < nk-1) {
                for (ii = 0; ii < mp12; ii++) {
                    Q0[ii] = p->Q[ii]*(SCALE/det[ii]);
                    wt = w_interp(p, f1, kk, ii);
                    Q0[ii] *= 0.5*wt/det[ii];
                }
                wqw(p, Q0, tmp, blk);
                put(p, out, cols, kk, kk, blk, 0);
            }
            break;
        case ORC_V_CONLIN_RHODPI: /* AssembleConLinWithRhodPi :1323-1373 ; f1 = theta(levels) f2 = dpi(interfaces) */
            if (kk > 0) {
                for (ii = 0; ii < mp12; ii++) {
                    Q0[ii] = p->Q[ii]*(SCALE/det[ii]);
                    wb = w_interp(p, f2, kk-1, ii); tb = w_interp(p, f1, kk, ii);
                    tb *= THI(kk, ii);
                    Q0[ii] *= 0.5*wb*tb/(det[ii]*det[ii]);
                }
                wqw(p, Q0, tmp, blk);
                put(p, out, cols, kk, kk-1, blk, 0);
            }
            if (kk < nk-1) {
                for (ii = 0; ii < mp12; ii++) {
                    Q0[ii] = p->Q[ii]*(SCALE/det[ii]);
                    wt = w_interp(p, f2, kk, ii); tt = w_interp(p, f1, kk, ii);
                    tt *= THI(kk, ii);
                    Q0[ii] *= 0.5*wt*tt/(det[ii]*det[ii]);
                }
                wqw(p, Q0, tmp, blk);
                put(p, out, cols, kk, kk, blk, 0);
            }
            break;
        default:
            free(tmp); free(blk); free(binv); free(t1); free(t2);
            return 1;
        }
    }
done:
    free(tmp); free(blk); free(binv); free(t1); free(t2);
    return 0;
}

/* shared tail of the EOS vectors: out[k*n2+j] = 2 * sum_q WtQ[j][q] rtq[q], WtQ = Wt diag(0.5 w SCALE) */
static void eos_project(const orc_patch* p, const double* rtq, double* out_k) {
    int ii, jj, mp12 = p->mp12, n2 = p->n2e;
    double Q0[128];
    double* WtQ = (double*)malloc(sizeof(double)*n2*mp12);
    for (ii = 0; ii < mp12; ii++) Q0[ii] = 0.5*p->Q[ii]*SCALE;
    orc_la->mult_fd(n2, mp12, mp12, p->Wt, Q0, WtQ);
    for (jj = 0; jj < n2; jj++) {
        double r = 0.0;
        for (ii = 0; ii < mp12; ii++) r += WtQ[jj*mp12+ii]*rtq[ii];
        r *= 2.0;
        out_k[jj] = r;
    }
    free(WtQ);
}

/* Assemble_EOS_Residual :987-1047 */
void orc_eos_residual(const orc_patch* p, int ex, int ey, const double* rt, const double* exner, double* out) {
    int kk, ii, ei = ey*p->nElsX + ex, iq[128];
    double rtq[128];
    const double* det = p->det + (size_t)ei*p->mp12;
    orc_elindsq_l(p, ex, ey, iq);
    for (kk = 0; kk < p->nk; kk++) {
        for (ii = 0; ii < p->mp12; ii++) {
            double rk = w_interp(p, rt, kk, ii), ek = w_interp(p, exner, kk, ii);
            rk *= 1.0/(det[ii]*TH(kk, ii));
            ek *= 1.0/(det[ii]*TH(kk, ii));
            rtq[ii] = log(ek) - (RD/CV)*log(rk) - log(CP) - (RD/CV)*log(RD/P0);
        }
        eos_project(p, rtq, out + kk*p->n2e);
    }
}

/* Assemble_EOS_RHS :732-787 */
void orc_eos_rhs(const orc_patch* p, int ex, int ey, const double* rt, double factor, double exponent, double* out) {
    int kk, ii, ei = ey*p->nElsX + ex, iq[128];
    double rtq[128];
    const double* det = p->det + (size_t)ei*p->mp12;
    orc_elindsq_l(p, ex, ey, iq);
    for (kk = 0; kk < p->nk; kk++) {
        for (ii = 0; ii < p->mp12; ii++) {
            double rk = w_interp(p, rt, kk, ii);
            rk *= 1.0/(det[ii]*TH(kk, ii));
            rtq[ii] = factor*pow(rk, exponent);
        }
        eos_project(p, rtq, out + kk*p->n2e);
    }
}

/* AssembleConstWithLogThetaPlusEta :1204-1255 (eta may be NULL) */
void orc_const_log_theta_plus_eta(const orc_patch* p, int ex, int ey, const double* theta, const double* eta, double* out) {
    int kk, ii, jj, ei = ey*p->nElsX + ex, iq[128], mp12 = p->mp12;
    double rtq[128];
    const double* det = p->det + (size_t)ei*mp12;
    orc_elindsq_l(p, ex, ey, iq);
    for (kk = 0; kk < p->nk; kk++) {
        for (ii = 0; ii < mp12; ii++) {
            double tb, ek, fac;
            rtq[ii] = p->Q[ii];
            tb = w_interp(p, theta, kk, ii);
            fac = log(tb/(TH(kk, ii)*det[ii]));
            if (eta) { ek = w_interp(p, eta, kk, ii); fac += ek/(TH(kk, ii)*det[ii]); }
            rtq[ii] *= (SCALE*fac);
        }
        for (jj = 0; jj < p->n2e; jj++) {
            double r = 0.0;
            for (ii = 0; ii < mp12; ii++) r += p->Wt[jj*mp12+ii]*rtq[ii];
            out[kk*p->n2e + jj] = r;
        }
    }
}

/* AssembleConstWithRhoExpEta :1257-1305 */
void orc_const_rho_exp_eta(const orc_patch* p, int ex, int ey, const double* rho, const double* eta, double* out) {
    int kk, ii, jj, ei = ey*p->nElsX + ex, iq[128], mp12 = p->mp12;
    double rtq[128];
    const double* det = p->det + (size_t)ei*mp12;
    orc_elindsq_l(p, ex, ey, iq);
    for (kk = 0; kk < p->nk; kk++) {
        for (ii = 0; ii < mp12; ii++) {
            double rk = w_interp(p, rho, kk, ii), ek = w_interp(p, eta, kk, ii);
            rtq[ii] = p->Q[ii];
            rk *= 1.0/(TH(kk, ii)*det[ii]);
            ek *= 1.0/(TH(kk, ii)*det[ii]);
            rtq[ii] *= (SCALE*rk*exp(ek));
        }
        for (jj = 0; jj < p->n2e; jj++) {
            double r = 0.0;
            for (ii = 0; ii < mp12; ii++) r += p->Wt[jj*mp12+ii]*rtq[ii];
            out[kk*p->n2e + jj] = r;
        }
    }
}

/* ---- dense helpers for the Mat chains (PETSc MatMatMult / MatMult stand-ins) ------------ */
static double* dmat(int r, int c) { return (double*)calloc(((size_t)r*c) > 0 ? (size_t)r*c : 1, sizeof(double)); }
static void mm(int m, int k, int n, const double* A, const double* B, double* C) {
    int i, j, l;
    memset(C, 0, sizeof(double)*(size_t)m*n);
    for (i = 0; i < m; i++) for (l = 0; l < k; l++) {
        double a = A[(size_t)i*k+l];
        if (a == 0.0) continue;
        for (j = 0; j < n; j++) C[(size_t)i*n+j] += a*B[(size_t)l*n+j];
    }
}
static void mv(int m, int n, const double* A, const double* x, double* y) {
    int i, j;
    for (i = 0; i < m; i++) { double s = 0.0; for (j = 0; j < n; j++) s += A[(size_t)i*n+j]*x[j]; y[i] = s; }
}

/* diagTheta_L2 eul/VertSolve.cpp:339-349 for one column */
int orc_diag_theta_L2(const orc_patch* p, int ex, int ey, const double* rho, const double* rt, double* theta) {
    int N = p->nk*p->n2e, err;
    double *VB = dmat(N, N), *frt = dmat(N, 1);
    orc_colop_dense(p, ORC_V_CONST, ex, ey, 0, NULL, NULL, VB);
    mv(N, N, VB, rt, frt);
    orc_colop_dense(p, ORC_V_CONST_RHO, ex, ey, 0, rho, NULL, VB);
    err = orc_dense_solve(N, VB, frt, theta);
    free(VB); free(frt);
    return err;
}

/* diagTheta2 eul/VertSolve.cpp:306-315 for one column: theta on nk+1 interfaces */
int orc_diag_theta2(const orc_patch* p, int ex, int ey, const double* rho, const double* rt, double* theta) {
    int N = p->nk*p->n2e, Np = (p->nk+1)*p->n2e, err;
    double *VAB2 = dmat(Np, N), *VA2 = dmat(Np, Np), *frt = dmat(Np, 1);
    orc_colop_dense(p, ORC_V_LINCON2, ex, ey, 0, NULL, NULL, VAB2);
    mv(Np, N, VAB2, rt, frt);
    orc_colop_dense(p, ORC_V_LINEAR_RHO2, ex, ey, 0, rho, NULL, VA2);
    err = orc_dense_solve(Np, VA2, frt, theta);
    free(VAB2); free(VA2); free(frt);
    return err;
}

/* VertSolve::solve_schur_column_eta eul/VertSolve.cpp:677-823, dense restatement.
 * (velz is accepted and unused, as in the reference.) */
int orc_solve_schur_column_eta(const orc_patch* p, int ex, int ey, double dt,
        const double* theta, const double* velz, const double* rho, const double* eta, const double* pi,
        double* F_u, double* F_rho, double* F_eta, double* F_pi,
        double* d_u, double* d_rho, double* d_eta, double* d_pi, double* Lpi_out) {
    int nk = p->nk, n2 = p->n2e, N = nk*n2, Nm = (nk-1)*n2, i, j, k, err;
    double *VB = dmat(N, N), *VA_inv = dmat(Nm, Nm), *VB_inv = dmat(N, N), *VA = dmat(Nm, Nm);
    double *V10 = dmat(N, Nm), *V01 = dmat(Nm, N);
    double *DTV1 = dmat(Nm, N), *GRAD = dmat(Nm, N), *VBA = dmat(N, Nm), *G_rt = dmat(Nm, N), *G_pi = dmat(Nm, N);
    double *X = dmat(Nm, Nm), *DX = dmat(N, Nm), *D_rho = dmat(N, Nm);
    double *N_pi = dmat(N, N), *N_rho = dmat(N, N), *GV = dmat(Nm, N), *L_eta = dmat(Nm, Nm);
    double *CM = dmat(N, N), *DIV = dmat(N, Nm), *L_pi = dmat(N, N);
    double *tA1 = dmat(Nm, 1), *tA2 = dmat(Nm, 1), *tB1 = dmat(N, 1);
    (void)velz;

    /* vertOps() eul/VertOps.cpp:134-163 */
    for (k = 0; k < nk; k++) for (i = 0; i < n2; i++) {
        if (k > 0)    V10[(size_t)(k*n2+i)*Nm + (k-1)*n2 + i] = -1.0;
        if (k < nk-1) V10[(size_t)(k*n2+i)*Nm + k*n2 + i] = +1.0;
    }
    for (i = 0; i < N; i++) for (j = 0; j < Nm; j++) V01[(size_t)j*N + i] = -V10[(size_t)i*Nm + j];

    orc_colop_dense(p, ORC_V_CONST, ex, ey, 0, NULL, NULL, VB);          /* :690 */
    orc_colop_dense(p, ORC_V_LINEAR_INV, ex, ey, 0, NULL, NULL, VA_inv);  /* :691 */
    orc_colop_dense(p, ORC_V_CONST_INV, ex, ey, 0, NULL, NULL, VB_inv);   /* :692 */
    mm(Nm, N, N, V01, VB, DTV1);                                          /* :694 */
    mm(Nm, Nm, N, VA_inv, DTV1, GRAD);                                    /* :695 grad operator */

    mv(Nm, N, GRAD, pi, tA2);                                             /* :700 pressure gradient */
    orc_colop_dense(p, ORC_V_CONLIN_RHODPI, ex, ey, 0, theta, tA2, VBA);  /* :701 */
    for (i = 0; i < N; i++) for (j = 0; j < Nm; j++) G_rt[(size_t)j*N + i] = VBA[(size_t)i*Nm + j]*(0.5*dt); /* :702-703 */

    orc_colop_dense(p, ORC_V_LINEAR_RT, ex, ey, 1, theta, NULL, VA);      /* :709 */
    mm(Nm, Nm, N, VA, GRAD, G_pi);                                        /* :710 */
    for (i = 0; i < Nm*N; i++) G_pi[i] *= 0.5*dt;                         /* :711 */

    orc_colop_dense(p, ORC_V_LINEAR_RT, ex, ey, 1, rho, NULL, VA);        /* :716 */
    mm(Nm, Nm, Nm, VA_inv, VA, X);                                        /* :717 */
    mm(N, Nm, Nm, V10, X, DX);                                            /* :720 */
    mm(N, N, Nm, VB, DX, D_rho);                                          /* :723 */
    for (i = 0; i < N*Nm; i++) D_rho[i] *= 0.5*dt;                        /* :724 */

    mv(Nm, N, GRAD, eta, tA2);                                            /* :729 entropy gradient */
    orc_colop_dense(p, ORC_V_CONLIN_W, ex, ey, 0, tA2, NULL, VBA);        /* :730 */
    for (i = 0; i < N*Nm; i++) VBA[i] *= 0.5*dt;                          /* :731 */

    orc_colop_dense(p, ORC_V_EOS_BLOCK, ex, ey, 0, pi, NULL, N_pi);       /* :736 */
    orc_colop_dense(p, ORC_V_EOS_BLOCK, ex, ey, 0, rho, NULL, N_rho);     /* :739 */

    mm(Nm, N, N, G_rt, VB_inv, GV);                                       /* :742 */
    mm(Nm, N, Nm, GV, VBA, L_eta);                                        /* :743 */
    orc_colop_dense(p, ORC_V_LINEAR, ex, ey, 0, NULL, NULL, VA);          /* :744 */
    for (i = 0; i < Nm*Nm; i++) L_eta[i] = -1.0*L_eta[i] + VA[i];         /* :745 MatAYPX */
    for (i = 0; i < Nm; i++) tA1[i] = 1.0/L_eta[(size_t)i*Nm + i];        /* :749-751 lumped inverse */

    mm(N, N, N, N_rho, VB_inv, CM);                                       /* :754 */
    mm(N, N, Nm, CM, D_rho, DIV);                                         /* :759 */
    for (i = 0; i < N*Nm; i++) DIV[i] += VBA[i];                          /* :760 */
    for (i = 0; i < N; i++) for (j = 0; j < Nm; j++) DIV[(size_t)i*Nm + j] *= tA1[j]; /* :761 column scale */

    mm(N, Nm, N, DIV, G_pi, L_pi);                                        /* :766 */
    for (i = 0; i < N*N; i++) L_pi[i] = (-1.0*RD/CV)*L_pi[i] + N_pi[i];   /* :767 */
    if (Lpi_out) memcpy(Lpi_out, L_pi, sizeof(double)*(size_t)N*N);

    mv(Nm, N, GV, F_eta, tA2);                                            /* :772 */
    for (i = 0; i < Nm; i++) F_u[i] += -1.0*tA2[i];                       /* :773 */
    for (i = 0; i < N; i++) F_pi[i] *= -1.0;                              /* :775 */
    mv(N, Nm, DIV, F_u, tB1);                                             /* :776 */
    for (i = 0; i < N; i++) F_pi[i] += (+1.0*RD/CV)*tB1[i];               /* :777 */
    mv(N, N, CM, F_rho, tB1);                                             /* :778 */
    for (i = 0; i < N; i++) F_pi[i] += (-1.0*RD/CV)*tB1[i];               /* :779 */
    for (i = 0; i < N; i++) F_pi[i] += (-1.0*RD/CV)*F_eta[i];             /* :780 */

    err = orc_dense_solve(N, L_pi, F_pi, d_pi);                           /* :783-789 PCLU */

    mv(Nm, N, G_pi, d_pi, tA2);                                           /* :792 */
    for (i = 0; i < Nm; i++) { F_u[i] += tA2[i]; F_u[i] *= -1.0; d_u[i] = tA1[i]*F_u[i]; } /* :793-795 */

    mv(N, Nm, VBA, d_u, tB1);                                             /* :807 */
    for (i = 0; i < N; i++) { F_eta[i] += tB1[i]; F_eta[i] *= -1.0; }     /* :808-809 */
    mv(N, N, VB_inv, F_eta, d_eta);                                       /* :810 */

    mv(N, Nm, D_rho, d_u, tB1);                                           /* :812 */
    for (i = 0; i < N; i++) { F_rho[i] += tB1[i]; F_rho[i] *= -1.0; }     /* :813-814 */
    mv(N, N, VB_inv, F_rho, d_rho);                                       /* :815 */

    free(VB); free(VA_inv); free(VB_inv); free(VA); free(V10); free(V01); free(DTV1); free(GRAD);
    free(VBA); free(G_rt); free(G_pi); free(X); free(DX); free(D_rho); free(N_pi); free(N_rho);
    free(GV); free(L_eta); free(CM); free(DIV); free(L_pi); free(tA1); free(tA2); free(tB1);
    return err;
}

/* ==== Held-Suarez / Strang-splitting column rows (C2-C6 remainder) ========================= */

/* AssembleLinearWithRayleighInv :1380-1413 (which ORC_V_LINEAR_RAYLEIGH_INV, param = dt_fric),
 * Assemble_EOS_BlockInv :1049-1142 (ORC_V_EOS_BLOCK_INV, f1 = rt, f2 = theta or NULL),
 * AssembleLinearWithRho2_up :1415-1490 (ORC_V_LINEAR_RHO2_UP, f1 = rho, param = dt, uh = [nk][n1] local 1-forms),
 * AssembleLinCon2_up :1492-1561 (ORC_V_LINCON2_UP, param = dt, uh). */
int orc_colop_dims_ex(const orc_patch* p, int colop, int* rows, int* cols) {
    int nk = p->nk, n2 = p->n2e;
    switch (colop) {
    case ORC_V_LINEAR_RAYLEIGH_INV: *rows = (nk-1)*n2; *cols = (nk-1)*n2; return 0;
    case ORC_V_EOS_BLOCK_INV: *rows = nk*n2; *cols = nk*n2; return 0;
    case ORC_V_LINEAR_RHO2_UP: *rows = (nk+1)*n2; *cols = (nk+1)*n2; return 0;
    case ORC_V_LINCON2_UP: *rows = (nk+1)*n2; *cols = nk*n2; return 0;
    }
    return orc_colop_dims(p, colop, rows, cols);
}

int orc_colop_dense_ex(const orc_patch* p, int colop, int ex, int ey, int flag, double param,
                       const double* f1, const double* f2, const double* uh, double* out) {
    int rows, cols, kk, ii, jj, nk = p->nk, mp1 = p->mp1, mp12 = p->mp12, n2 = p->n2e, nn = p->n, ei = ey*p->nElsX + ex;
    int iq[128];
    double Q0[128];
    double *tmp, *blk, *binv, *BinvB, *B_BinvB, *Wt;
    const double* det = p->det + (size_t)ei*mp12;
    if (colop < ORC_V_LINEAR_RAYLEIGH_INV) return orc_colop_dense(p, colop, ex, ey, flag, f1, f2, out);
    if (orc_colop_dims_ex(p, colop, &rows, &cols)) return 1;
    tmp = (double*)malloc(sizeof(double)*n2*mp12); blk = (double*)malloc(sizeof(double)*n2*n2);
    binv = (double*)malloc(sizeof(double)*n2*n2); BinvB = (double*)malloc(sizeof(double)*n2*n2);
    B_BinvB = (double*)malloc(sizeof(double)*n2*n2); Wt = (double*)malloc(sizeof(double)*n2*mp12);
    memset(out, 0, sizeof(double)*(size_t)rows*cols);
    orc_elindsq_l(p, ex, ey, iq);

    if (colop == ORC_V_LINEAR_RAYLEIGH_INV) {
        for (kk = 0; kk < nk-1; kk++) {
            for (ii = 0; ii < mp12; ii++) {
                Q0[ii]  = p->Q[ii]*(SCALE/det[ii]);
                Q0[ii] *= 0.5*(TH(kk+0, ii) + TH(kk+1, ii));
                if (kk == nk-1)      Q0[ii] *= (1.0 + 1.00*param);
                else if (kk == nk-2) Q0[ii] *= (1.0 + 0.50*param);
                else if (kk == nk-3) Q0[ii] *= (1.0 + 0.25*param);
            }
            wqw(p, Q0, tmp, blk);
            orc_la->inv(blk, binv, n2);
            put(p, out, cols, kk, kk, binv, 1);
        }
    } else if (colop == ORC_V_EOS_BLOCK_INV) {
        for (kk = 0; kk < nk; kk++) {
            double tk, tkp1;
            for (ii = 0; ii < mp12; ii++) {                       /* the layer-wise inverse matrix :1071-1085 */
                Q0[ii] = p->Q[ii]*(SCALE/det[ii]);
                Q0[ii] *= THI(kk, ii);
                tk = w_interp(p, f1, kk, ii);
                Q0[ii] *= tk/(TH(kk, ii)*det[ii]);
            }
            wqw(p, Q0, tmp, blk);
            orc_la->inv(blk, binv, n2);
            for (ii = 0; ii < mp12; ii++) { Q0[ii] = p->Q[ii]*(SCALE/det[ii]); Q0[ii] *= THI(kk, ii); }   /* :1088-1094 */
            wqw(p, Q0, tmp, blk);
            orc_la->mult(n2, n2, n2, binv, blk, BinvB);           /* :1097-1098 */
            orc_la->mult(n2, n2, n2, blk, BinvB, B_BinvB);
            if (f2) {                                             /* rho correction :1101-1126 */
                orc_la->inv(blk, binv, n2);
                for (ii = 0; ii < mp12; ii++) {
                    Q0[ii]  = p->Q[ii]*(SCALE/det[ii]);
                    Q0[ii] *= THI(kk, ii);
                    tk = w_interp(p, f2, kk, ii); tkp1 = w_interp(p, f2, kk+1, ii);
                    Q0[ii] *= 0.5*(tk + tkp1)/det[ii];
                }
                wqw(p, Q0, tmp, blk);
                orc_la->mult(n2, n2, n2, blk, binv, BinvB);
                for (ii = 0; ii < n2; ii++) BinvB[ii*n2+ii] += 1.0;
                orc_la->mult(n2, n2, n2, BinvB, B_BinvB, blk);
                for (ii = 0; ii < n2*n2; ii++) B_BinvB[ii] = blk[ii];
            }
            orc_la->inv(B_BinvB, binv, n2);                       /* :1128 */
            put(p, out, cols, kk, kk, binv, 1);
        }
    } else {   /* the two *_up assemblies: test functions at x_q + dt * (local velocity) */
        for (kk = 0; kk < nk; kk++) {
            const double* uArray = uh + (size_t)kk*p->n1;
            for (ii = 0; ii < mp12; ii++) {
                const double* jac = &p->J[((size_t)ei*mp12 + ii)*4];
                double ug[2], ul[2], _ex[16], _ey[16];
                Q0[ii] = p->Q[ii]*(SCALE/det[ii]);
                if (colop == ORC_V_LINEAR_RHO2_UP) {
                    double rk = w_interp(p, f1, kk, ii);
                    Q0[ii] *= 0.5*rk/det[ii];
                } else Q0[ii] *= 0.5;
                orc_interp1_g(p, ex, ey, ii%mp1, ii/mp1, uArray, ug);
                ul[0] = (+jac[3]*ug[0] - jac[1]*ug[1])/det[ii];
                ul[1] = (-jac[2]*ug[0] + jac[0]*ug[1])/det[ii];
                ul[0] *= THI(kk, ii);
                ul[1] *= THI(kk, ii);
                for (jj = 0; jj < nn; jj++) {
                    _ex[jj] = orc_edge_eval(nn, p->nx, p->qx[ii%mp1] + param*ul[0], jj);
                    _ey[jj] = orc_edge_eval(nn, p->nx, p->qx[ii/mp1] + param*ul[1], jj);
                }
                for (jj = 0; jj < n2; jj++) Wt[jj*mp12+ii] = _ex[jj%nn]*_ey[jj/nn];
            }
            orc_la->mult_fd(n2, mp12, mp12, Wt, Q0, tmp);
            orc_la->mult(n2, n2, mp12, tmp, p->W, blk);
            if (colop == ORC_V_LINEAR_RHO2_UP) {
                put(p, out, cols, kk, kk, blk, 1);
                put(p, out, cols, kk+1, kk+1, blk, 1);
            } else {
                put(p, out, cols, kk, kk, blk, 1);
                put(p, out, cols, kk+1, kk, blk, 1);
            }
        }
    }
    free(tmp); free(blk); free(binv); free(BinvB); free(B_BinvB); free(Wt);
    return 0;
}

/* diagTheta_up eul/VertSolve.cpp:354-384 for one column */
int orc_diag_theta_up(const orc_patch* p, int ex, int ey, double dt, const double* rho, const double* rt,
                      const double* uh, double* theta) {
    int N = p->nk*p->n2e, Np = (p->nk+1)*p->n2e, err;
    double *VAB2 = dmat(Np, N), *VA2 = dmat(Np, Np), *frt = dmat(Np, 1);
    orc_colop_dense_ex(p, ORC_V_LINCON2_UP, ex, ey, 0, dt, NULL, NULL, uh, VAB2);
    mv(Np, N, VAB2, rt, frt);
    orc_colop_dense_ex(p, ORC_V_LINEAR_RHO2_UP, ex, ey, 0, dt, rho, NULL, uh, VA2);
    err = orc_dense_solve(Np, VA2, frt, theta);
    free(VAB2); free(VA2); free(frt);
    return err;
}

/* compute_k_T eul/VertOps.cpp:1563-1587 */
static double hs_k_T(double phi, double exner, double exner_s, double theta) {
    double pr       = pow(exner/CP, CP/RD);
    double ps       = pow(exner_s/CP, CP/RD);
    double sigma    = pr/ps;
    double sigma_b  = 0.7;
    double theta_eq;
    double k_a      = 2.8935185185185185e-07;
    double k_s      = 2.8935185185185184e-06;
    double k_t      = 0.0;
    double t_eq     = 315.0 - 60.0*sin(phi)*sin(phi) - 10.0*log(pr)*cos(phi)*cos(phi);
    t_eq *= pow(pr, RD/CP);
    if (t_eq < 200.0) t_eq = 200.0;
    theta_eq = t_eq*pow(1.0/pr, RD/CP);
    if (sigma > sigma_b) {
        k_t  = (k_s - k_a)*(sigma - sigma_b)/(1.0 - sigma_b);
        k_t *= pow(cos(phi), 4.0);
    }
    k_t += k_a;
    return k_t*(theta - theta_eq);
}

/* AssembleTempForcing_HS :1589-1633 ; theta on nk+1 interfaces */
void orc_temp_forcing_hs(const orc_patch* p, int ex, int ey, const double* exner, const double* theta,
                         const double* rho, double* vec) {
    int kk, ii, jj, ll, ei = ey*p->nElsX + ex, mp12 = p->mp12, n2 = p->n2e, iq[128];
    double _e[128], _r[128], _tb[128], _tt[128], _es[128], k_t[128];
    const double* det = p->det + (size_t)ei*mp12;
    orc_elindsq_l(p, ex, ey, iq);
    memset(vec, 0, sizeof(double)*p->nk*n2);
    for (kk = 0; kk < p->nk; kk++) {
        for (ii = 0; ii < mp12; ii++) {
            _e[ii] = _tb[ii] = _tt[ii] = _r[ii] = _es[ii] = 0.0;
            for (ll = 0; ll < n2; ll++) {
                _e[ii]  += exner[kk*n2+ll]*p->W[ii*n2+ll];
                _tb[ii] += theta[(kk+0)*n2+ll]*p->W[ii*n2+ll];
                _tt[ii] += theta[(kk+1)*n2+ll]*p->W[ii*n2+ll];
                _r[ii]  += rho[kk*n2+ll]*p->W[ii*n2+ll];
                _es[ii] += exner[ll]*p->W[ii*n2+ll];
            }
            _e[ii]  /= (det[ii]*TH(kk, ii));
            _tb[ii] /= (det[ii]);
            _tt[ii] /= (det[ii]);
            _r[ii]  /= (det[ii]*TH(kk, ii));
            _es[ii] /= (det[ii]*TH(0, ii));
            k_t[ii] = hs_k_T(p->sq[2*iq[ii]+1], _e[ii], _es[ii], 0.5*(_tb[ii] + _tt[ii]));
        }
        for (jj = 0; jj < n2; jj++)
            for (ii = 0; ii < mp12; ii++)
                vec[kk*n2+jj] += p->Wt[jj*mp12+ii]*p->Q[ii]*SCALE*_r[ii]*k_t[ii];
    }
}

/* VertSolve::solve_schur_column_3 eul/VertSolve.cpp:504-675 (RAYLEIGH defined, :32), dense restatement.
 * flags: 1 = RAYLEIGH undefined, 2 = VBA(velz) not re-assembled -- together the box twin box/VertSolve.cpp:879-1058.
 * F_* modified in place as the reference does; d_* outputs; Lrt_out (N x N) optional. */
#define RAYLEIGH (4.0/120.0)
int orc_solve_schur_column_3(const orc_patch* p, int ex, int ey, double dt, int flags,
        const double* theta, const double* velz, const double* rho, const double* rt, const double* pi,
        double* F_u, double* F_rho, double* F_rt, double* F_pi,
        double* d_u, double* d_rho, double* d_rt, double* d_pi, double* Lrt_out) {
    int nk = p->nk, n2 = p->n2e, N = nk*n2, Nm = (nk-1)*n2, i, j, k, err;
    double *M_u_inv = dmat(Nm, Nm), *M_rho_inv = dmat(N, N), *M_rt = dmat(N, N), *N_pi_inv = dmat(N, N);
    double *VB = dmat(N, N), *VA_inv = dmat(Nm, Nm), *VB_inv = dmat(N, N), *VA = dmat(Nm, Nm), *VBA = dmat(N, Nm), *VAB = dmat(Nm, N);
    double *V10 = dmat(N, Nm), *V01 = dmat(Nm, N);
    double *t_mn = dmat(Nm, N), *t_mn2 = dmat(Nm, N), *t_mm = dmat(Nm, Nm), *t_nm = dmat(N, Nm), *t_nn = dmat(N, N), *t_nn2 = dmat(N, N);
    double *G_rt = dmat(Nm, N), *G_pi = dmat(Nm, N), *D_rho = dmat(N, Nm), *D_rt = dmat(N, Nm), *N_rt = dmat(N, N);
    double *Q_rt_rho = dmat(N, N), *QM = dmat(N, N), *GN = dmat(Nm, N), *GG = dmat(Nm, N), *DD = dmat(N, Nm), *DDM = dmat(N, Nm);
    double *L = dmat(N, N);
    double *tA1 = dmat(Nm, 1), *tA2 = dmat(Nm, 1), *tB1 = dmat(N, 1);

    for (k = 0; k < nk; k++) for (i = 0; i < n2; i++) {
        if (k > 0)    V10[(size_t)(k*n2+i)*Nm + (k-1)*n2 + i] = -1.0;
        if (k < nk-1) V10[(size_t)(k*n2+i)*Nm + k*n2 + i] = +1.0;
    }
    for (i = 0; i < N; i++) for (j = 0; j < Nm; j++) V01[(size_t)j*N + i] = -V10[(size_t)i*Nm + j];

    if (flags & 1) orc_colop_dense(p, ORC_V_LINEAR_INV, ex, ey, 0, NULL, NULL, M_u_inv);                      /* :522 (box: RAYLEIGH undefined) */
    else orc_colop_dense_ex(p, ORC_V_LINEAR_RAYLEIGH_INV, ex, ey, 0, 0.5*dt*RAYLEIGH, NULL, NULL, NULL, M_u_inv);  /* :520 */
    orc_colop_dense(p, ORC_V_CONST, ex, ey, 0, NULL, NULL, M_rt);                 /* :524 */
    orc_colop_dense(p, ORC_V_CONST_INV, ex, ey, 0, NULL, NULL, M_rho_inv);        /* :525 */
    orc_colop_dense_ex(p, ORC_V_EOS_BLOCK_INV, ex, ey, 0, 0.0, pi, NULL, NULL, N_pi_inv);   /* :526 */
    orc_colop_dense(p, ORC_V_CONST, ex, ey, 0, NULL, NULL, VB);                   /* :527 */
    mv(N, N, VB, pi, tB1);                                                        /* :528 */
    mv(Nm, N, V01, tB1, tA1);                                                     /* :529 */
    orc_colop_dense(p, ORC_V_LINEAR_INV, ex, ey, 0, NULL, NULL, VA_inv);          /* :530 */
    mv(Nm, Nm, VA_inv, tA1, tA2);                                                 /* :531 pressure gradient */
    orc_colop_dense(p, ORC_V_CONLIN_W, ex, ey, 0, tA2, NULL, VBA);                /* :532 */
    for (i = 0; i < N; i++) for (j = 0; j < Nm; j++) VAB[(size_t)j*N + i] = VBA[(size_t)i*Nm + j];   /* :533 */
    orc_colop_dense(p, ORC_V_CONST_RHO_INV, ex, ey, 0, rho, NULL, VB_inv);        /* :536 */
    mm(Nm, N, N, VAB, VB_inv, t_mn);                                              /* :537 */
    mm(Nm, N, N, t_mn, VB, G_rt);                                                 /* :540 */
    for (i = 0; i < Nm*N; i++) G_rt[i] *= 0.5*dt;                                 /* :541 */

    mm(Nm, N, N, V01, VB, t_mn);                                                  /* :546 pc_DTV1 */
    mm(Nm, Nm, N, VA_inv, t_mn, t_mn2);                                           /* :550 */
    orc_colop_dense(p, ORC_V_LINEAR_THETA, ex, ey, 0, theta, NULL, VA);           /* :553 */
    mm(Nm, Nm, N, VA, t_mn2, G_pi);                                               /* :554 */
    for (i = 0; i < Nm*N; i++) G_pi[i] *= 0.5*dt;                                 /* :555 */

    orc_colop_dense(p, ORC_V_LINEAR_RT, ex, ey, 1, rho, NULL, VA);                /* :559 */
    mm(Nm, Nm, Nm, VA_inv, VA, t_mm);                                             /* :561 */
    mm(N, Nm, Nm, V10, t_mm, t_nm);                                               /* :564 */
    mm(N, N, Nm, VB, t_nm, D_rho);                                                /* :568 */
    for (i = 0; i < N*Nm; i++) D_rho[i] *= 0.5*dt;                                /* :569 */

    orc_colop_dense(p, ORC_V_CONST_RHO, ex, ey, 0, rt, NULL, t_nn);               /* :573 */
    mm(N, N, Nm, t_nn, V10, D_rt);                                                /* :574 */
    for (i = 0; i < N*Nm; i++) D_rt[i] *= 0.5*dt;                                 /* :575 */

    orc_colop_dense(p, ORC_V_CONST_RHO_INV, ex, ey, 0, rt, NULL, VB_inv);         /* :580 */
    mm(N, N, N, VB_inv, VB, t_nn);                                                /* :581 */
    mm(N, N, N, VB, t_nn, N_rt);                                                  /* :584 */
    for (i = 0; i < N*N; i++) N_rt[i] *= -1.0*RD/CV;                              /* :585 */

    orc_colop_dense(p, ORC_V_CONST_THETA, ex, ey, 0, theta, NULL, t_nn);          /* :589 */
    mm(Nm, N, N, V01, t_nn, t_mn);                                                /* :590 */
    mm(Nm, Nm, N, VA_inv, t_mn, t_mn2);                                           /* :594 */
    if (!(flags & 2)) orc_colop_dense(p, ORC_V_CONLIN_W, ex, ey, 0, velz, NULL, VBA);   /* :597 ; absent in box/VertSolve.cpp:978-981 */
    mm(N, Nm, N, VBA, t_mn2, Q_rt_rho);                                           /* :598 */
    for (i = 0; i < N*N; i++) Q_rt_rho[i] *= 0.5*dt;                              /* :599 */

    mm(N, N, N, Q_rt_rho, M_rho_inv, QM);                                         /* :603 */
    mm(Nm, N, N, G_pi, N_pi_inv, GN);                                             /* :607 */
    mm(Nm, N, N, GN, N_rt, GG);                                                   /* :611 */
    for (i = 0; i < Nm*N; i++) GG[i] = -1.0*GG[i] + G_rt[i];                      /* :612 MatAYPX */

    mm(N, N, Nm, QM, D_rho, DD);                                                  /* :618 */
    for (i = 0; i < N*Nm; i++) DD[i] = -1.0*DD[i] + D_rt[i];                      /* :619 */
    mm(N, Nm, Nm, DD, M_u_inv, DDM);                                              /* :622 */
    mm(N, Nm, N, DDM, GG, L);                                                     /* :626 */
    for (i = 0; i < N*N; i++) L[i] = -1.0*L[i] + M_rt[i];                         /* :627 */
    if (Lrt_out) memcpy(Lrt_out, L, sizeof(double)*(size_t)N*N);

    mv(N, N, QM, F_rho, tB1);  for (i = 0; i < N; i++) F_rt[i] += -1.0*tB1[i];    /* :632-633 */
    mv(Nm, N, GN, F_pi, tA1);  for (i = 0; i < Nm; i++) F_u[i] += -1.0*tA1[i];    /* :635-636 */
    mv(N, Nm, DDM, F_u, tB1);  for (i = 0; i < N; i++) F_rt[i] += -1.0*tB1[i];    /* :638-639 */

    for (i = 0; i < N; i++) F_rt[i] *= -1.0;                                      /* :652 */
    err = orc_dense_solve(N, L, F_rt, d_rt);                                      /* :653 PCLU */

    mv(Nm, N, GG, d_rt, tA1);                                                     /* :656 */
    for (i = 0; i < Nm; i++) { F_u[i] += tA1[i]; F_u[i] *= -1.0; }                /* :657-658 */
    mv(Nm, Nm, M_u_inv, F_u, d_u);                                                /* :659 */
    mv(N, N, N_rt, d_rt, tB1);                                                    /* :661 */
    for (i = 0; i < N; i++) { F_pi[i] += tB1[i]; F_pi[i] *= -1.0; }               /* :662-663 */
    mv(N, N, N_pi_inv, F_pi, d_pi);                                               /* :664 */
    mv(N, Nm, D_rho, d_u, tB1);                                                   /* :666 */
    for (i = 0; i < N; i++) { F_rho[i] += tB1[i]; F_rho[i] *= -1.0; }             /* :667-668 */
    mv(N, N, M_rho_inv, F_rho, d_rho);                                            /* :669 */

    free(M_u_inv); free(M_rho_inv); free(M_rt); free(N_pi_inv); free(VB); free(VA_inv); free(VB_inv); free(VA);
    free(VBA); free(VAB); free(V10); free(V01); free(t_mn); free(t_mn2); free(t_mm); free(t_nm); free(t_nn); free(t_nn2);
    free(G_rt); free(G_pi); free(D_rho); free(D_rt); free(N_rt); free(Q_rt_rho); free(QM); free(GN); free(GG);
    free(DD); free(DDM); free(L); free(tA1); free(tA2); free(tB1);
    return err;
}
