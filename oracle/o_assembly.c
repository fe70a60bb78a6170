/* oracle/o_assembly.c -- TEST INFRASTRUCTURE (see oracle.h).
 * CPU restatement of the horizontal operator classes of eul/Assembly.cpp: per element the
 * quadrature-point coefficients (Jacobian metric x weights x interpolated field x 1/thickness),
 * then  B_row^T . diag(c) . B_col  by Mult_FD_IP + Mult_IP exactly as the reference composes
 * them.  Element matrices are returned to the caller (what the reference hands to
 * MatSetValues); orc_op_apply is the single-rank MatMult content on local ghosted vectors. */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

extern const orc_linalg* orc_la;

#define J00 jac[0]
#define J01 jac[1]
#define J10 jac[2]
#define J11 jac[3]

int orc_op_elmat_size(const orc_patch* p, int op) {
    switch (op) {
    case ORC_UMAT: case ORC_UHMAT: case ORC_UTMAT: case ORC_UTMAT_H: return 4*p->n1e*p->n1e;
    case ORC_ROTMAT: return 2*p->n1e*p->n1e;
    case ORC_WMAT: case ORC_WHMAT: case ORC_WMATINV: case ORC_WHMATINV: return p->n2e*p->n2e;
    case ORC_PMAT: case ORC_PHMAT: return p->n0e*p->n0e;
    case ORC_WTQUMAT: case ORC_WTQDUDZ: case ORC_UTQWMAT: return 2*p->n2e*p->n1e;
    default: return -1;
    }
}

/* At(ni x mp12) . diag(c) . B(mp12 x nj) -> M, via the reference's two-step (FD then IP) */
static void triple(const orc_patch* p, int ni, int nj, double* At, double* c, double* B, double* tmp, double* M) {
    orc_la->mult_fd(ni, p->mp12, p->mp12, At, c, tmp);
    orc_la->mult(ni, nj, p->mp12, tmp, B, M);
}

int orc_op_elmats(const orc_patch* p, int op, int lev, double scale, int flag,
                  const double* f1, double* out) {
    int ex, ey, ei, ii, mp1 = p->mp1, mp12 = p->mp12, n1e = p->n1e, n2e = p->n2e, n0e = p->n0e;
    int iq[128];
    int esz = orc_op_elmat_size(p, op);
    double *ca = (double*)malloc(sizeof(double)*mp12), *cb = (double*)malloc(sizeof(double)*mp12),
           *cc = (double*)malloc(sizeof(double)*mp12);
    double* tmp = (double*)malloc(sizeof(double)*128*mp12);
    double* tmp2 = (double*)malloc(sizeof(double)*128*128);
    const double* tI = p->thickInv + (size_t)lev*p->n0q;
    const double* tH = p->thick + (size_t)lev*p->n0q;
    if (esz < 0) return 1;

    for (ey = 0; ey < p->nElsX; ey++) for (ex = 0; ex < p->nElsX; ex++) {
        double* M = out + (size_t)(ey*p->nElsX + ex)*esz;
        ei = ey*p->nElsX + ex;
        orc_elindsq_l(p, ex, ey, iq);
        for (ii = 0; ii < mp12; ii++) {
            double det = p->det[(size_t)ei*mp12 + ii];
            const double* jac = &p->J[((size_t)ei*mp12 + ii)*4];
            double hi, ux[2], vort;
            switch (op) {
            case ORC_UMAT:      /* Umat::_assemble :99-113 */
                ca[ii] = (J00*J00 + J10*J10)*p->Q[ii]*(scale/det);
                cb[ii] = (J00*J01 + J10*J11)*p->Q[ii]*(scale/det);
                cc[ii] = (J01*J01 + J11*J11)*p->Q[ii]*(scale/det);
                if (flag) { ca[ii] *= tI[iq[ii]]; cb[ii] *= tI[iq[ii]]; cc[ii] *= tI[iq[ii]]; }
                break;
            case ORC_UHMAT:     /* Uhmat::assemble :432-448 */
                orc_interp2_g(p, ex, ey, ii%mp1, ii/mp1, f1, &hi);
                if (flag) hi *= tI[iq[ii]];
                ca[ii] = hi*(J00*J00 + J10*J10)*p->Q[ii]*(scale/det);
                cb[ii] = hi*(J00*J01 + J10*J11)*p->Q[ii]*(scale/det);
                cc[ii] = hi*(J01*J01 + J11*J11)*p->Q[ii]*(scale/det);
                ca[ii] *= tI[iq[ii]]; cb[ii] *= tI[iq[ii]]; cc[ii] *= tI[iq[ii]];
                break;
            case ORC_UTMAT:     /* Ut_mat::assemble :1352-1364 (interface between lev, lev+1) */
                ca[ii] = (J00*J00 + J10*J10)*p->Q[ii]*(scale/det);
                cb[ii] = (J00*J01 + J10*J11)*p->Q[ii]*(scale/det);
                cc[ii] = (J01*J01 + J11*J11)*p->Q[ii]*(scale/det);
                ca[ii] *= 0.5*(tH[iq[ii]] + tH[p->n0q + iq[ii]]);
                cb[ii] *= 0.5*(tH[iq[ii]] + tH[p->n0q + iq[ii]]);
                cc[ii] *= 0.5*(tH[iq[ii]] + tH[p->n0q + iq[ii]]);
                break;
            case ORC_UTMAT_H:   /* Ut_mat::assemble_h :1402-1413 */
                orc_interp2_g(p, ex, ey, ii%mp1, ii/mp1, f1, &hi);
                ca[ii] = hi*(J00*J00 + J10*J10)*p->Q[ii]*(scale/det);
                cb[ii] = hi*(J00*J01 + J10*J11)*p->Q[ii]*(scale/det);
                cc[ii] = hi*(J01*J01 + J11*J11)*p->Q[ii]*(scale/det);
                break;
            case ORC_ROTMAT:    /* RotMat::assemble :1051-1065 */
                orc_interp0(p, ex, ey, ii%mp1, ii/mp1, f1, &vort);
                vort *= tI[iq[ii]];
                ca[ii] = vort*(-J00*J11 + J01*J10)*p->Q[ii]*(scale/det);
                cb[ii] = vort*(+J00*J11 - J01*J10)*p->Q[ii]*(scale/det);
                ca[ii] *= tI[iq[ii]]; cb[ii] *= tI[iq[ii]];
                break;
            case ORC_WMAT:      /* Wmat::_assemble :346-352 */
                ca[ii] = p->Q[ii]*(scale/det);
                if (flag) ca[ii] *= tI[iq[ii]];
                break;
            case ORC_WMATINV:   /* WmatInv::assemble :1701-1705 */
                ca[ii] = p->Q[ii]*(scale/det);
                ca[ii] *= tI[iq[ii]];
                break;
            case ORC_WHMAT:     /* Whmat::assemble :1268-1281 */
                orc_interp2_g(p, ex, ey, ii%mp1, ii/mp1, f1, &hi);
                if (flag) hi *= tI[iq[ii]];
                ca[ii] = hi*p->Q[ii]*(scale/det);
                ca[ii] *= tI[iq[ii]];
                break;
            case ORC_WHMATINV:  /* WhmatInv::assemble :1775-1785 */
                orc_interp2_g(p, ex, ey, ii%mp1, ii/mp1, f1, &hi);
                hi *= tI[iq[ii]];
                ca[ii] = hi*p->Q[ii]*(scale/det);
                ca[ii] *= tI[iq[ii]];
                break;
            case ORC_PMAT:      /* Pmat::assemble :2029-2033 */
                ca[ii] = scale*p->Q[ii]*det;
                ca[ii] *= tI[iq[ii]];
                break;
            case ORC_PHMAT:     /* Pmat::assemble_h :2075-2083 */
                ca[ii] = scale*p->Q[ii]*det;
                ca[ii] *= tI[iq[ii]];
                orc_interp2_g(p, ex, ey, ii%mp1, ii/mp1, f1, &hi);
                hi *= tI[iq[ii]];
                ca[ii] *= hi;
                break;
            case ORC_WTQUMAT:   /* WtQUmat::assemble :951-966 */
                orc_interp1_g(p, ex, ey, ii%mp1, ii/mp1, f1, ux);
                ux[0] *= tI[iq[ii]]; ux[1] *= tI[iq[ii]];
                ca[ii] = 0.5*(ux[0]*J00 + ux[1]*J10)*p->Q[ii]*(scale/det);
                cb[ii] = 0.5*(ux[0]*J01 + ux[1]*J11)*p->Q[ii]*(scale/det);
                ca[ii] *= tI[iq[ii]]; cb[ii] *= tI[iq[ii]];
                break;
            case ORC_UTQWMAT:   /* UtQWmat::assemble :1504-1517 (interp1_g_t == interp1_g) */
            case ORC_WTQDUDZ:   /* WtQdUdz_mat::assemble :1599-1621 */
                orc_interp1_g(p, ex, ey, ii%mp1, ii/mp1, f1, ux);
                ca[ii] = (ux[0]*J00 + ux[1]*J10)*p->Q[ii]*(scale/det);
                cb[ii] = (ux[0]*J01 + ux[1]*J11)*p->Q[ii]*(scale/det);
                break;
            }
        }
        switch (op) {
        case ORC_UMAT: case ORC_UHMAT: case ORC_UTMAT: case ORC_UTMAT_H:
            /* :118-126  UtQU, UtQV, VtQU, VtQV */
            triple(p, n1e, n1e, p->Ut, ca, p->U, tmp, M + 0*n1e*n1e);
            triple(p, n1e, n1e, p->Ut, cb, p->V, tmp, M + 1*n1e*n1e);
            triple(p, n1e, n1e, p->Vt, cb, p->U, tmp, M + 2*n1e*n1e);
            triple(p, n1e, n1e, p->Vt, cc, p->V, tmp, M + 3*n1e*n1e);
            break;
        case ORC_ROTMAT:   /* :1067-1073 UtQV (x rows, y cols), VtQU (y rows, x cols) */
            triple(p, n1e, n1e, p->Ut, ca, p->V, tmp, M + 0*n1e*n1e);
            triple(p, n1e, n1e, p->Vt, cb, p->U, tmp, M + 1*n1e*n1e);
            break;
        case ORC_WMAT: case ORC_WHMAT:
            triple(p, n2e, n2e, p->Wt, ca, p->W, tmp, M);
            break;
        case ORC_WMATINV: case ORC_WHMATINV:
            triple(p, n2e, n2e, p->Wt, ca, p->W, tmp, tmp2);
            orc_la->inv(tmp2, M, n2e);
            break;
        case ORC_PMAT: case ORC_PHMAT:
            triple(p, n0e, n0e, p->Pt, ca, p->P, tmp, M);
            break;
        case ORC_WTQUMAT: case ORC_WTQDUDZ:   /* WtQU, WtQV */
            triple(p, n2e, n1e, p->Wt, ca, p->U, tmp, M);
            triple(p, n2e, n1e, p->Wt, cb, p->V, tmp, M + n2e*n1e);
            break;
        case ORC_UTQWMAT:                     /* UtQW, VtQW */
            triple(p, n1e, n2e, p->Ut, ca, p->W, tmp, M);
            triple(p, n1e, n2e, p->Vt, cb, p->W, tmp, M + n1e*n2e);
            break;
        }
    }
    free(ca); free(cb); free(cc); free(tmp); free(tmp2);
    return 0;
}

/* y[r[i]] += sum_j M[i][j] x[c[j]] */
static void blk_apply(int nr, const int* r, int nc, const int* c, const double* M, const double* x, double* y) {
    int i, j;
    for (i = 0; i < nr; i++) {
        double s = 0.0;
        for (j = 0; j < nc; j++) s += M[i*nc+j]*x[c[j]];
        y[r[i]] += s;
    }
}

int orc_op_apply(const orc_patch* p, int op, const double* elmats, const double* x, double* y) {
    int ex, ey, esz = orc_op_elmat_size(p, op), n1e = p->n1e, n2e = p->n2e, n0e = p->n0e;
    int i0[128], ix[128], iy[128], i2[128];
    if (esz < 0) return 1;
    for (ey = 0; ey < p->nElsX; ey++) for (ex = 0; ex < p->nElsX; ex++) {
        const double* M = elmats + (size_t)(ey*p->nElsX + ex)*esz;
        orc_elinds0_l(p, ex, ey, i0); orc_elinds1x_l(p, ex, ey, ix);
        orc_elinds1y_l(p, ex, ey, iy); orc_elinds2_l(p, ex, ey, i2);
        switch (op) {
        case ORC_UMAT: case ORC_UHMAT: case ORC_UTMAT: case ORC_UTMAT_H:
            blk_apply(n1e, ix, n1e, ix, M + 0*n1e*n1e, x, y);
            blk_apply(n1e, ix, n1e, iy, M + 1*n1e*n1e, x, y);
            blk_apply(n1e, iy, n1e, ix, M + 2*n1e*n1e, x, y);
            blk_apply(n1e, iy, n1e, iy, M + 3*n1e*n1e, x, y);
            break;
        case ORC_ROTMAT:
            blk_apply(n1e, ix, n1e, iy, M + 0*n1e*n1e, x, y);
            blk_apply(n1e, iy, n1e, ix, M + 1*n1e*n1e, x, y);
            break;
        case ORC_WMAT: case ORC_WHMAT: case ORC_WMATINV: case ORC_WHMATINV:
            blk_apply(n2e, i2, n2e, i2, M, x, y);
            break;
        case ORC_PMAT: case ORC_PHMAT:
            blk_apply(n0e, i0, n0e, i0, M, x, y);
            break;
        case ORC_WTQUMAT: case ORC_WTQDUDZ:
            blk_apply(n2e, i2, n1e, ix, M, x, y);
            blk_apply(n2e, i2, n1e, iy, M + n2e*n1e, x, y);
            break;
        case ORC_UTQWMAT:
            blk_apply(n1e, ix, n2e, i2, M, x, y);
            blk_apply(n1e, iy, n2e, i2, M + n1e*n2e, x, y);
            break;
        }
    }
    return 0;
}

/* ---- matrix-free vectors ---------------------------------------------------------------- */

/* Pvec::assemble :602-625 (local accumulation; the gtol_0 reduce/broadcast is the halo layer).
 * NB the reference indexes thickInv with the NODE grid map (topo->elInds0_l), valid because
 * quadrature and basis orders coincide there (F8); we use the quad-grid map, identical for m==n. */
void orc_pvec(const orc_patch* p, int lev, double scale, double* vl) {
    int ex, ey, ei, ii, i0[128], iq[128];
    const double* tI = p->thickInv + (size_t)lev*p->n0q;
    memset(vl, 0, sizeof(double)*p->n0);
    for (ey = 0; ey < p->nElsX; ey++) for (ex = 0; ex < p->nElsX; ex++) {
        ei = ey*p->nElsX + ex;
        orc_elinds0_l(p, ex, ey, i0); orc_elindsq_l(p, ex, ey, iq);
        for (ii = 0; ii < p->n0e; ii++) {
            double e = scale*p->Q[ii]*p->det[(size_t)ei*p->mp12 + ii];
            e *= tI[iq[ii]];
            vl[i0[ii]] += e;
        }
    }
}

/* Phvec::assemble :654-683 */
void orc_phvec(const orc_patch* p, int lev, double scale, const double* h2, double* vl) {
    int ex, ey, ei, ii, i0[128], iq[128];
    const double* tI = p->thickInv + (size_t)lev*p->n0q;
    memset(vl, 0, sizeof(double)*p->n0);
    for (ey = 0; ey < p->nElsX; ey++) for (ex = 0; ex < p->nElsX; ex++) {
        ei = ey*p->nElsX + ex;
        orc_elinds0_l(p, ex, ey, i0); orc_elindsq_l(p, ex, ey, iq);
        for (ii = 0; ii < p->n0e; ii++) {
            double hi, e = scale*p->Q[ii]*p->det[(size_t)ei*p->mp12 + ii];
            e *= tI[iq[ii]];
            orc_interp2_g(p, ex, ey, ii%p->np1, ii/p->np1, h2, &hi);
            hi *= tI[iq[ii]];
            e *= hi;
            vl[i0[ii]] += e;
        }
    }
}

/* shared body of Uvec::assemble :2124-2191, assemble_hu :2198-2273 (mode 1), assemble_wxu
 * :2375-2424 (mode 2): four (or two) Ax_b projections added into the local 1-form vector */
static void uvec_core(const orc_patch* p, int mode, int lev, double scale, const double* vel,
                      const double* f2, double fac, double* vl) {
    int ex, ey, ei, ii, k, mp1 = p->mp1, mp12 = p->mp12, n1e = p->n1e, ix[128], iy[128], iq[128];
    double Qaa[128], Qab[128], Qba[128], Qbb[128], rhs[128], u[2], r;
    const double* tI = p->thickInv + (size_t)lev*p->n0q;
    for (ey = 0; ey < p->nElsX; ey++) for (ex = 0; ex < p->nElsX; ex++) {
        ei = ey*p->nElsX + ex;
        orc_elindsq_l(p, ex, ey, iq);
        orc_elinds1x_l(p, ex, ey, ix); orc_elinds1y_l(p, ex, ey, iy);
        for (ii = 0; ii < mp12; ii++) {
            double det = p->det[(size_t)ei*mp12 + ii];
            const double* jac = &p->J[((size_t)ei*mp12 + ii)*4];
            if (mode == 2) {
                orc_interp0(p, ex, ey, ii%mp1, ii/mp1, f2, &r);
                r *= tI[iq[ii]];
                orc_interp1_l(p, ex, ey, ii%mp1, ii/mp1, vel, u);
                Qab[ii] = (-J00*J11 + J01*J10)*p->Q[ii]*(scale/det);
                Qba[ii] = (+J00*J11 - J01*J10)*p->Q[ii]*(scale/det);
                Qab[ii] *= (r*u[1])*tI[iq[ii]];
                Qba[ii] *= (r*u[0])*tI[iq[ii]];
                continue;
            }
            Qaa[ii] = (J00*J00 + J10*J10)*p->Q[ii]*(scale/det);
            Qab[ii] = (J00*J01 + J10*J11)*p->Q[ii]*(scale/det);
            Qbb[ii] = (J01*J01 + J11*J11)*p->Q[ii]*(scale/det);
            Qaa[ii] *= tI[iq[ii]]; Qab[ii] *= tI[iq[ii]]; Qbb[ii] *= tI[iq[ii]];
            Qba[ii] = Qab[ii];
            orc_interp1_l(p, ex, ey, ii%mp1, ii/mp1, vel, u);
            if (mode == 1) {
                orc_interp2_g(p, ex, ey, ii%mp1, ii/mp1, f2, &r);
                r *= tI[iq[ii]];
                r *= fac;
                Qaa[ii] *= (u[0]*r); Qba[ii] *= (u[0]*r); Qab[ii] *= (u[1]*r); Qbb[ii] *= (u[1]*r);
            } else {
                Qaa[ii] *= u[0]; Qba[ii] *= u[0]; Qab[ii] *= u[1]; Qbb[ii] *= u[1];
            }
        }
        if (mode != 2) {
            orc_la->axb(n1e, mp12, p->Ut, Qaa, rhs); for (k = 0; k < n1e; k++) vl[ix[k]] += rhs[k];
            orc_la->axb(n1e, mp12, p->Ut, Qab, rhs); for (k = 0; k < n1e; k++) vl[ix[k]] += rhs[k];
            orc_la->axb(n1e, mp12, p->Vt, Qba, rhs); for (k = 0; k < n1e; k++) vl[iy[k]] += rhs[k];
            orc_la->axb(n1e, mp12, p->Vt, Qbb, rhs); for (k = 0; k < n1e; k++) vl[iy[k]] += rhs[k];
        } else {
            orc_la->axb(n1e, mp12, p->Ut, Qab, rhs); for (k = 0; k < n1e; k++) vl[ix[k]] += rhs[k];
            orc_la->axb(n1e, mp12, p->Vt, Qba, rhs); for (k = 0; k < n1e; k++) vl[iy[k]] += rhs[k];
        }
    }
}

void orc_uvec(const orc_patch* p, int lev, double scale, int vert_scale, const double* vel, double* vl) {
    (void)vert_scale;   /* the reference ignores it too: thickInv is applied unconditionally :2151-2153 */
    memset(vl, 0, sizeof(double)*p->n1);
    uvec_core(p, 0, lev, scale, vel, NULL, 1.0, vl);
}
void orc_uvec_hu(const orc_patch* p, int lev, double scale, const double* vel, const double* rho, double fac, double* vl) {
    /* zero_and_scatter==true path; callers that accumulate pass a pre-filled vl via orc_uvec_hu_acc */
    memset(vl, 0, sizeof(double)*p->n1);
    uvec_core(p, 1, lev, scale, vel, rho, fac, vl);
}
void orc_uvec_wxu(const orc_patch* p, int lev, double scale, const double* vel, const double* vort, double* vl) {
    memset(vl, 0, sizeof(double)*p->n1);
    uvec_core(p, 2, lev, scale, vel, vort, 1.0, vl);
}

/* B18  Wvec::assemble :2457-2495 and Wvec::assemble_K :2497-2545 -- CORRECTED restatement.  The reference allocates Wt in the
 * constructor (:2452) and never fills it (every other class calls Tran_IP right after Alloc2D), and all its live uses are
 * commented out (eul/HorizSolve.cpp:222-223, 441-448): as written it projects with an uninitialised table.  Here Wt = W^T, the
 * evident intent; everything else follows the reference's loops line by line.  These two are therefore NOT pinned to reference
 * behaviour -- they pin that  Wvec::assemble == Wmat x rho  and  Wvec::assemble_K == WtQUmat(vel2) x vel1. */
void orc_wvec(const orc_patch* p, int lev, double scale, int vert_scale, const double* rho, double* vg) {
    int ex, ey, ei, ii, k, mp1 = p->mp1, mp12 = p->mp12, n2e = p->n2e, i2[128], iq[128];
    double Qaa[128], rhs[128], r;
    memset(vg, 0, sizeof(double)*p->n2);
    for (ey = 0; ey < p->nElsX; ey++) for (ex = 0; ex < p->nElsX; ex++) {
        ei = ey*p->nElsX + ex;
        orc_elinds2_l(p, ex, ey, i2); orc_elindsq_l(p, ex, ey, iq);
        for (ii = 0; ii < mp12; ii++) {
            double det = p->det[(size_t)ei*mp12 + ii];
            Qaa[ii] = p->Q[ii]*(scale/det);
            if (vert_scale) Qaa[ii] *= p->thickInv[(size_t)lev*p->n0q + iq[ii]];
            orc_interp2_l(p, ex, ey, ii%mp1, ii/mp1, rho, &r);
            Qaa[ii] *= r;
        }
        orc_la->axb(n2e, mp12, p->Wt, Qaa, rhs);
        for (k = 0; k < n2e; k++) vg[i2[k]] += rhs[k];
    }
}
void orc_wvec_K(const orc_patch* p, int lev, double scale, const double* vel1, const double* vel2, double* vg) {
    int ex, ey, ei, ii, k, mp1 = p->mp1, mp12 = p->mp12, n2e = p->n2e, i2[128], iq[128];
    double Qaa[128], Qab[128], rhs_a[128], rhs_b[128], uxg[2], uxl[2];
    const double* tI = p->thickInv + (size_t)lev*p->n0q;
    memset(vg, 0, sizeof(double)*p->n2);
    for (ey = 0; ey < p->nElsX; ey++) for (ex = 0; ex < p->nElsX; ex++) {
        ei = ey*p->nElsX + ex;
        orc_elinds2_l(p, ex, ey, i2); orc_elindsq_l(p, ex, ey, iq);
        for (ii = 0; ii < mp12; ii++) {
            double det = p->det[(size_t)ei*mp12 + ii];
            const double* jac = &p->J[((size_t)ei*mp12 + ii)*4];
            orc_interp1_l(p, ex, ey, ii%mp1, ii/mp1, vel1, uxl);
            orc_interp1_g(p, ex, ey, ii%mp1, ii/mp1, vel2, uxg);
            uxg[0] *= tI[iq[ii]];
            uxg[1] *= tI[iq[ii]];
            Qaa[ii] = 0.5*(uxg[0]*J00 + uxg[1]*J10)*p->Q[ii]*(scale/det);
            Qab[ii] = 0.5*(uxg[0]*J01 + uxg[1]*J11)*p->Q[ii]*(scale/det);
            Qaa[ii] *= (uxl[0]*tI[iq[ii]]);
            Qab[ii] *= (uxl[1]*tI[iq[ii]]);
        }
        orc_la->axb(n2e, mp12, p->Wt, Qaa, rhs_a);
        orc_la->axb(n2e, mp12, p->Wt, Qab, rhs_b);
        for (k = 0; k < n2e; k++) vg[i2[k]] += (rhs_a[k] + rhs_b[k]);
    }
}

/* E10mat ctor :1118-1147 -- rows of the element's own (west/south) edges, INSERT semantics */
void orc_e10_apply(const orc_patch* p, const double* x0, double* y1) {
    int ex, ey, ii, jj, kk, ll, nn = p->n, np1 = p->np1, i0[128], ix[128], iy[128];
    for (ey = 0; ey < p->nElsX; ey++) for (ex = 0; ex < p->nElsX; ex++) {
        orc_elinds0_l(p, ex, ey, i0); orc_elinds1x_l(p, ex, ey, ix); orc_elinds1y_l(p, ex, ey, iy);
        for (ii = 0; ii < nn; ii++) for (jj = 0; jj < nn; jj++) {
            kk = jj*np1 + ii; ll = jj*np1 + ii;
            y1[ix[kk]] = (+1.0)*x0[i0[ll]] + (-1.0)*x0[i0[ll+np1]];
            kk = jj*nn + ii;
            y1[iy[kk]] = (-1.0)*x0[i0[ll]] + (+1.0)*x0[i0[ll+1]];
        }
    }
}

/* E21mat ctor :1185-1205 */
void orc_e21_apply(const orc_patch* p, const double* x1, double* y2) {
    int ex, ey, ii, jj, nn = p->n, np1 = p->np1, i2[128], ix[128], iy[128];
    for (ey = 0; ey < p->nElsX; ey++) for (ex = 0; ex < p->nElsX; ex++) {
        orc_elinds2_l(p, ex, ey, i2); orc_elinds1x_l(p, ex, ey, ix); orc_elinds1y_l(p, ex, ey, iy);
        for (ii = 0; ii < nn; ii++) for (jj = 0; jj < nn; jj++)
            y2[i2[ii*nn+jj]] = (-1.0)*x1[ix[ii*np1+jj]] + (+1.0)*x1[ix[ii*np1+jj+1]]
                             + (-1.0)*x1[iy[ii*nn+jj]]  + (+1.0)*x1[iy[(ii+1)*nn+jj]];
    }
}

/* ---- CSR: MatSetValues(ADD_VALUES) + MatMult cost structure ----------------------------- */
static int cmp_int(const void* a, const void* b) { int x = *(const int*)a, y = *(const int*)b; return (x > y) - (x < y); }

orc_csr* orc_csr_create(int nrows, int ncols, int nEl, int nr, const int* rows, int nc, const int* cols) {
    orc_csr* A = (orc_csr*)calloc(1, sizeof(orc_csr));
    int e, i, j, r;
    int* cnt = (int*)calloc(nrows + 1, sizeof(int));
    int *fill, *buf;
    A->nrows = nrows; A->ncols = ncols;
    for (e = 0; e < nEl; e++) for (i = 0; i < nr; i++) cnt[rows[e*nr+i] + 1] += nc;
    for (r = 0; r < nrows; r++) cnt[r+1] += cnt[r];
    buf = (int*)malloc(sizeof(int)*(cnt[nrows] ? cnt[nrows] : 1));
    fill = (int*)calloc(nrows, sizeof(int));
    for (e = 0; e < nEl; e++) for (i = 0; i < nr; i++) {
        r = rows[e*nr+i];
        for (j = 0; j < nc; j++) buf[cnt[r] + fill[r]++] = cols[e*nc+j];
    }
    A->rowptr = (int*)calloc(nrows + 1, sizeof(int));
    for (r = 0; r < nrows; r++) {
        int n = fill[r], u = 0;
        qsort(buf + cnt[r], n, sizeof(int), cmp_int);
        for (j = 0; j < n; j++) if (j == 0 || buf[cnt[r]+j] != buf[cnt[r]+j-1]) buf[cnt[r] + u++] = buf[cnt[r]+j];
        fill[r] = u;
        A->rowptr[r+1] = A->rowptr[r] + u;
    }
    A->nnz = A->rowptr[nrows];
    A->col = (int*)malloc(sizeof(int)*(A->nnz ? A->nnz : 1));
    A->val = (double*)calloc(A->nnz ? A->nnz : 1, sizeof(double));
    for (r = 0; r < nrows; r++) memcpy(A->col + A->rowptr[r], buf + cnt[r], sizeof(int)*fill[r]);
    free(cnt); free(buf); free(fill);
    return A;
}
void orc_csr_destroy(orc_csr* A) { if (!A) return; free(A->rowptr); free(A->col); free(A->val); free(A); }
void orc_csr_zero(orc_csr* A) { memset(A->val, 0, sizeof(double)*A->nnz); }
void orc_csr_add(orc_csr* A, int nr, const int* rows, int nc, const int* cols, const double* vals) {
    int i, j;
    for (i = 0; i < nr; i++) {
        int r = rows[i], lo0 = A->rowptr[r], hi0 = A->rowptr[r+1];
        for (j = 0; j < nc; j++) {
            int lo = lo0, hi = hi0, c = cols[j];
            while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (A->col[mid] > c) hi = mid; else lo = mid; }
            A->val[lo] += vals[i*nc+j];
        }
    }
}
void orc_csr_mult(const orc_csr* A, const double* x, double* y) {
    int r, k;
    for (r = 0; r < A->nrows; r++) {
        double s = 0.0;
        for (k = A->rowptr[r]; k < A->rowptr[r+1]; k++) s += A->val[k]*x[A->col[k]];
        y[r] = s;
    }
}

/* ---- timed CPU baseline: the reference's per-call cost structure (SURVEY 8(a) "today's cost", BASELINE.md
 * variant A): per call  rebuild nothing cached -> per-element coefficient loop + Mult_FD_IP/Mult_IP triple
 * products -> MatSetValues(ADD)-style insertion with per-entry column search -> SpMV.  The CSR pattern is
 * preallocated outside the timed loop (PETSc preallocation happens in the operator constructors).
 * Returns elapsed seconds for `reps` (assemble + MatMult) calls of `op` at level `lev` on this patch. */
#include <time.h>
double orc_bench_assemble_mult(const orc_patch* p, int op, int lev, double scale, int flag,
                               const double* f1, const double* x, double* y, int reps) {
    int e, r, n1e = p->n1e, n2e = p->n2e, n0e = p->n0e, nEl = p->nEl, esz = orc_op_elmat_size(p, op);
    int *i0 = (int*)malloc(sizeof(int)*nEl*n0e), *i1 = (int*)malloc(sizeof(int)*nEl*2*n1e), *i2 = (int*)malloc(sizeof(int)*nEl*n2e);
    double* em = (double*)malloc(sizeof(double)*(size_t)nEl*esz);
    orc_csr* A = NULL;
    struct timespec t0, t1;
    int rows_n = 0, cols_n = 0, nr = 0, nc = 0; const int *rows = NULL, *cols = NULL;
    for (e = 0; e < nEl; e++) {
        int ex = e%p->nElsX, ey = e/p->nElsX;
        orc_elinds0_l(p, ex, ey, i0 + e*n0e);
        orc_elinds1x_l(p, ex, ey, i1 + e*2*n1e); orc_elinds1y_l(p, ex, ey, i1 + e*2*n1e + n1e);
        orc_elinds2_l(p, ex, ey, i2 + e*n2e);
    }
    switch (op) {
    case ORC_UMAT: case ORC_UHMAT: case ORC_UTMAT: case ORC_UTMAT_H: case ORC_ROTMAT:
        rows = cols = i1; nr = nc = 2*n1e; rows_n = cols_n = p->n1; break;
    case ORC_WMAT: case ORC_WHMAT: rows = cols = i2; nr = nc = n2e; rows_n = cols_n = p->n2; break;
    case ORC_PMAT: case ORC_PHMAT: rows = cols = i0; nr = nc = n0e; rows_n = cols_n = p->n0; break;
    case ORC_WTQUMAT: case ORC_WTQDUDZ: rows = i2; nr = n2e; rows_n = p->n2; cols = i1; nc = 2*n1e; cols_n = p->n1; break;
    case ORC_UTQWMAT: rows = i1; nr = 2*n1e; rows_n = p->n1; cols = i2; nc = n2e; cols_n = p->n2; break;
    default: free(i0); free(i1); free(i2); free(em); return -1.0;
    }
    A = orc_csr_create(rows_n, cols_n, nEl, nr, rows, nc, cols);
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (r = 0; r < reps; r++) {
        orc_csr_zero(A);                                           /* MatZeroEntries */
        orc_op_elmats(p, op, lev, scale, flag, f1, em);            /* coefficient loops + triple products */
        for (e = 0; e < nEl; e++) {                                /* MatSetValues(ADD_VALUES) per block */
            const double* M = em + (size_t)e*esz;
            const int *ix = i1 + e*2*n1e, *iy = ix + n1e, *j2 = i2 + e*n2e, *j0 = i0 + e*n0e;
            switch (op) {
            case ORC_UMAT: case ORC_UHMAT: case ORC_UTMAT: case ORC_UTMAT_H:
                orc_csr_add(A, n1e, ix, n1e, ix, M); orc_csr_add(A, n1e, ix, n1e, iy, M + n1e*n1e);
                orc_csr_add(A, n1e, iy, n1e, ix, M + 2*n1e*n1e); orc_csr_add(A, n1e, iy, n1e, iy, M + 3*n1e*n1e); break;
            case ORC_ROTMAT:
                orc_csr_add(A, n1e, ix, n1e, iy, M); orc_csr_add(A, n1e, iy, n1e, ix, M + n1e*n1e); break;
            case ORC_WMAT: case ORC_WHMAT: orc_csr_add(A, n2e, j2, n2e, j2, M); break;
            case ORC_PMAT: case ORC_PHMAT: orc_csr_add(A, n0e, j0, n0e, j0, M); break;
            case ORC_WTQUMAT: case ORC_WTQDUDZ:
                orc_csr_add(A, n2e, j2, n1e, ix, M); orc_csr_add(A, n2e, j2, n1e, iy, M + n2e*n1e); break;
            case ORC_UTQWMAT:
                orc_csr_add(A, n1e, ix, n2e, j2, M); orc_csr_add(A, n1e, iy, n2e, j2, M + n1e*n2e); break;
            }
        }
        orc_csr_mult(A, x, y);                                     /* MatMult */
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    orc_csr_destroy(A); free(i0); free(i1); free(i2); free(em);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9*(double)(t1.tv_nsec - t0.tv_nsec);
}

/* ---- upwinded operators of the shallow-water stack (src/ flavour: scale = 1, no thickness) ------------
 * ORC_PHMAT_UP : Phmat::assemble_up(ul, hl, fac, dt)   src/Assembly.cpp:499-567  out [nEl][n0e][n0e]
 *                trial functions evaluated at the departure points x_q - tau*u_local, tau = 1/(1/(fac*dt))
 * ORC_ROTMAT_UP: RotMat_up::assemble(q0, ul, fac, dt)  src/Assembly.cpp:1784-1853 out [nEl][2][n1e][n1e]
 *                the vorticity is interpolated at the departure points
 * f1 = hl (2-form) resp. q0 (0-form), ul = local 1-form velocity.                                      */
int orc_op_elmats_up(const orc_patch* p, int which, double fac, double dt, const double* f1, const double* ul, double* out) {
    int ex, ey, ei, ii, jj, mp1 = p->mp1, mp12 = p->mp12, np1 = p->np1, np12 = p->n0e, n1e = p->n1e, i0[128];
    double lx[16], ly[16], ux[2], ux2[2], hx, tau, vort;
    double *QP = (double*)malloc(sizeof(double)*mp12*np12), *ca = (double*)malloc(sizeof(double)*mp12),
           *cb = (double*)malloc(sizeof(double)*mp12), *tmp = (double*)malloc(sizeof(double)*128*mp12);
    for (ey = 0; ey < p->nElsX; ey++) for (ex = 0; ex < p->nElsX; ex++) {
        ei = ey*p->nElsX + ex;
        orc_elinds0_l(p, ex, ey, i0);
        for (ii = 0; ii < mp12; ii++) {
            double det = p->det[(size_t)ei*mp12 + ii];
            const double* jac = &p->J[((size_t)ei*mp12 + ii)*4];
            orc_interp1_g(p, ex, ey, ii%mp1, ii/mp1, ul, ux);
            ux2[0] = +J11*ux[0]/det - J01*ux[1]/det;
            ux2[1] = -J10*ux[0]/det + J00*ux[1]/det;
            tau = 1.0/(1.0/(fac*dt));
            for (jj = 0; jj < np1; jj++) {
                lx[jj] = orc_node_eval(p->n, p->nx, p->qx[ii%mp1] - tau*ux2[0], jj);
                ly[jj] = orc_node_eval(p->n, p->nx, p->qx[ii/mp1] - tau*ux2[1], jj);
            }
            if (which == 0) {                                   /* Phmat::assemble_up :538-548 */
                orc_interp2_l(p, ex, ey, ii%mp1, ii/mp1, f1, &hx);
                for (jj = 0; jj < np12; jj++) QP[ii*np12 + jj] = hx*p->Q[ii]*lx[jj%np1]*ly[jj/np1];
            } else {                                            /* RotMat_up::assemble :1816-1826 */
                vort = 0.0;
                for (jj = 0; jj < np12; jj++) vort += f1[i0[jj]]*lx[jj%np1]*ly[jj/np1];
                ca[ii] = vort*(-J00*J11 + J01*J10)*p->Q[ii]/det;
                cb[ii] = vort*(+J00*J11 - J01*J10)*p->Q[ii]/det;
            }
        }
        if (which == 0) {
            orc_la->mult(np12, np12, mp12, p->Pt, QP, out + (size_t)ei*np12*np12);     /* Pt . QP :551 */
        } else {
            double* M = out + (size_t)ei*2*n1e*n1e;
            triple(p, n1e, n1e, p->Ut, ca, p->V, tmp, M);
            triple(p, n1e, n1e, p->Vt, cb, p->U, tmp, M + n1e*n1e);
        }
    }
    free(QP); free(ca); free(cb); free(tmp);
    return 0;
}


/* ---- B7: projections from the quadrature-point grid, built once for initial conditions / Coriolis -------------
 * WtQmat::assemble :707-751   y2 = W^T (Q xq)                     xq[n0q]
 * PtQmat::assemble :766-808   y0 = P^T (Q det xq)                 (added into the local 0-form vector)
 * UtQmat::assemble :824-902   y1 = [U^T (J00 Q vx + J10 Q vy) ; V^T (J01 Q vx + J11 Q vy)], xq interleaved [n0q][2]
 * which = 0, 1, 2.  The element blocks WtQ / PtQ / UtQ are formed with Mult_FD_IP exactly as the reference does and
 * applied to the quad-grid values gathered through Geom::elInds0_l. */
int orc_project_from_quad(const orc_patch* p, int which, const double* xq, double* y) {
    int ex, ey, ei, ii, k, mp12 = p->mp12, n1e = p->n1e, n2e = p->n2e, n0e = p->n0e;
    int iq[128], i0[128], ix[128], iy[128], i2[128];
    double c0[128], c1[128], v0[128], v1[128], rhs[128];
    double* BtQ = (double*)malloc(sizeof(double)*128*mp12);
    for (ey = 0; ey < p->nElsX; ey++) for (ex = 0; ex < p->nElsX; ex++) {
        ei = ey*p->nElsX + ex;
        orc_elindsq_l(p, ex, ey, iq); orc_elinds0_l(p, ex, ey, i0); orc_elinds1x_l(p, ex, ey, ix);
        orc_elinds1y_l(p, ex, ey, iy); orc_elinds2_l(p, ex, ey, i2);
        if (which == 0) {
            for (ii = 0; ii < mp12; ii++) { c0[ii] = p->Q[ii]; v0[ii] = xq[iq[ii]]; }
            orc_la->mult_fd(n2e, mp12, mp12, p->Wt, c0, BtQ);
            orc_la->axb(n2e, mp12, BtQ, v0, rhs);
            for (k = 0; k < n2e; k++) y[i2[k]] += rhs[k];
        } else if (which == 1) {
            for (ii = 0; ii < mp12; ii++) { c0[ii] = p->Q[ii]*p->det[(size_t)ei*mp12 + ii]; v0[ii] = xq[iq[ii]]; }
            orc_la->mult_fd(n0e, mp12, mp12, p->Pt, c0, BtQ);
            orc_la->axb(n0e, mp12, BtQ, v0, rhs);
            for (k = 0; k < n0e; k++) y[i0[k]] += rhs[k];
        } else {
            for (ii = 0; ii < mp12; ii++) { v0[ii] = xq[2*iq[ii]]; v1[ii] = xq[2*iq[ii] + 1]; }
            for (ii = 0; ii < mp12; ii++) { const double* jac = &p->J[((size_t)ei*mp12 + ii)*4]; c0[ii] = J00*p->Q[ii]; c1[ii] = J10*p->Q[ii]; }
            orc_la->mult_fd(n1e, mp12, mp12, p->Ut, c0, BtQ); orc_la->axb(n1e, mp12, BtQ, v0, rhs);
            for (k = 0; k < n1e; k++) y[ix[k]] += rhs[k];
            orc_la->mult_fd(n1e, mp12, mp12, p->Ut, c1, BtQ); orc_la->axb(n1e, mp12, BtQ, v1, rhs);
            for (k = 0; k < n1e; k++) y[ix[k]] += rhs[k];
            for (ii = 0; ii < mp12; ii++) { const double* jac = &p->J[((size_t)ei*mp12 + ii)*4]; c0[ii] = J01*p->Q[ii]; c1[ii] = J11*p->Q[ii]; }
            orc_la->mult_fd(n1e, mp12, mp12, p->Vt, c0, BtQ); orc_la->axb(n1e, mp12, BtQ, v0, rhs);
            for (k = 0; k < n1e; k++) y[iy[k]] += rhs[k];
            orc_la->mult_fd(n1e, mp12, mp12, p->Vt, c1, BtQ); orc_la->axb(n1e, mp12, BtQ, v1, rhs);
            for (k = 0; k < n1e; k++) y[iy[k]] += rhs[k];
        }
    }
    free(BtQ);
    return 0;
}

/* ---- eul-flavour operators with upwinded TEST functions (rows B2, B4, B17) ---------------------------------
 * which 0: Umat::assemble_up(lev, scale, tau, ui, uj)          eul/Assembly.cpp:156-279   f1 = ui, f2 = uj (local 1-forms)
 * which 1: Uhmat::assemble_up(h2, lev, scale, dt, u1)          :477-560                   f1 = h2 (2-form), f2 = u1
 * Element blocks [nEl][4][n1e][n1e] (UtQU UtQV VtQU VtQV) with the row tables evaluated at the departure points. */
int orc_op_elmats_testup(const orc_patch* p, int which, int lev, double scale, double tau,
                         const double* f1, const double* f2, double* out) {
    int ex, ey, ei, ii, jj, mp1 = p->mp1, mp12 = p->mp12, np1 = p->np1, nn = p->n, n1e = p->n1e, iq[128];
    double lx[16], ly[16], _ex[16], _ey[16], uil[2], ujl[2], ug[2], ul[2], hi;
    double *Ut = (double*)malloc(sizeof(double)*n1e*mp12), *Vt = (double*)malloc(sizeof(double)*n1e*mp12);
    double *ca = (double*)malloc(sizeof(double)*mp12), *cb = (double*)malloc(sizeof(double)*mp12), *cc = (double*)malloc(sizeof(double)*mp12);
    double* tmp = (double*)malloc(sizeof(double)*128*mp12);
    const double* tI = p->thickInv + (size_t)lev*p->n0q;
    for (ey = 0; ey < p->nElsX; ey++) for (ex = 0; ex < p->nElsX; ex++) {
        double* M = out + (size_t)(ey*p->nElsX + ex)*4*n1e*n1e;
        ei = ey*p->nElsX + ex;
        orc_elindsq_l(p, ex, ey, iq);
        for (ii = 0; ii < mp12; ii++) {
            double det = p->det[(size_t)ei*mp12 + ii];
            const double* jac = &p->J[((size_t)ei*mp12 + ii)*4];
            if (which == 0) {                                   /* :202-223 */
                orc_interp1_l(p, ex, ey, ii%mp1, ii/mp1, f1, uil);
                orc_interp1_l(p, ex, ey, ii%mp1, ii/mp1, f2, ujl);
                uil[0] *= tI[iq[ii]]/det; uil[1] *= tI[iq[ii]]/det;
                ujl[0] *= tI[iq[ii]]/det; ujl[1] *= tI[iq[ii]]/det;
                for (jj = 0; jj < np1; jj++) {
                    lx[jj] = orc_node_eval(nn, p->nx, p->qx[ii%mp1] + 0.5*tau*uil[0] + 0.5*tau*ujl[0], jj);
                    ly[jj] = orc_node_eval(nn, p->nx, p->qx[ii/mp1] + 0.5*tau*uil[1] + 0.5*tau*ujl[1], jj);
                }
                for (jj = 0; jj < nn*mp1; jj++) {
                    Ut[jj*mp12 + ii] = lx[jj%mp1]*p->ejxi[(ii/mp1)*nn + jj/np1];
                    Vt[jj*mp12 + ii] = p->ejxi[(ii%mp1)*nn + jj%nn]*ly[jj/nn];
                }
                ca[ii] = (J00*J00 + J10*J10)*p->Q[ii]*(scale/det);
                cb[ii] = (J00*J01 + J10*J11)*p->Q[ii]*(scale/det);
                cc[ii] = (J01*J01 + J11*J11)*p->Q[ii]*(scale/det);
                ca[ii] *= tI[iq[ii]]; cb[ii] *= tI[iq[ii]]; cc[ii] *= tI[iq[ii]];
            } else {                                            /* :499-530 */
                orc_interp2_g(p, ex, ey, ii%mp1, ii/mp1, f1, &hi);
                orc_interp1_g(p, ex, ey, ii%mp1, ii/mp1, f2, ug);
                ug[0] *= tI[iq[ii]]; ug[1] *= tI[iq[ii]];
                ul[0] = (+J11*ug[0] - J01*ug[1])/det;
                ul[1] = (-J10*ug[0] + J00*ug[1])/det;
                for (jj = 0; jj < mp1; jj++) {
                    lx[jj] = orc_node_eval(nn, p->nx, p->qx[ii%mp1] + tau*ul[0], jj);
                    ly[jj] = orc_node_eval(nn, p->nx, p->qx[ii/mp1] + tau*ul[1], jj);
                }
                for (jj = 0; jj < nn; jj++) {
                    _ex[jj] = orc_edge_eval(nn, p->nx, p->qx[ii%mp1] + tau*ul[0], jj);
                    _ey[jj] = orc_edge_eval(nn, p->nx, p->qx[ii/mp1] + tau*ul[1], jj);
                }
                for (jj = 0; jj < nn*mp1; jj++) {
                    Ut[jj*mp12 + ii] = lx[jj%mp1]*_ey[jj/mp1];
                    Vt[jj*mp12 + ii] = _ex[jj%nn]*ly[jj/nn];
                }
                ca[ii] = hi*(J00*J00 + J10*J10)*p->Q[ii]*(scale/det);
                cb[ii] = hi*(J00*J01 + J10*J11)*p->Q[ii]*(scale/det);
                cc[ii] = hi*(J01*J01 + J11*J11)*p->Q[ii]*(scale/det);
                ca[ii] *= tI[iq[ii]]; cb[ii] *= tI[iq[ii]]; cc[ii] *= tI[iq[ii]];
            }
        }
        triple(p, n1e, n1e, Ut, ca, p->U, tmp, M + 0*n1e*n1e);
        triple(p, n1e, n1e, Ut, cb, p->V, tmp, M + 1*n1e*n1e);
        triple(p, n1e, n1e, Vt, cb, p->U, tmp, M + 2*n1e*n1e);
        triple(p, n1e, n1e, Vt, cc, p->V, tmp, M + 3*n1e*n1e);
    }
    free(Ut); free(Vt); free(ca); free(cb); free(cc); free(tmp);
    return 0;
}

/* Uvec::assemble_hu_up(lev, scale, vel, rho, fac, tau, vel2)  eul/Assembly.cpp:2281-2373 -- accumulates into vl */
void orc_uvec_hu_up(const orc_patch* p, int lev, double scale, const double* vel, const double* rho, double fac,
                    double tau, const double* vel2, double* vl) {
    int ex, ey, ei, ii, jj, k, mp1 = p->mp1, mp12 = p->mp12, nn = p->n, np1 = p->np1, nj = p->n1e, ix[128], iy[128], iq[128];
    double Qaa[128], Qab[128], Qba[128], Qbb[128], rhs[128], u[2], uh[2], r, lx[16], ly[16];
    double *MUt = (double*)malloc(sizeof(double)*nj*mp12), *MVt = (double*)malloc(sizeof(double)*nj*mp12);
    const double* tI = p->thickInv + (size_t)lev*p->n0q;
    for (ey = 0; ey < p->nElsX; ey++) for (ex = 0; ex < p->nElsX; ex++) {
        ei = ey*p->nElsX + ex;
        orc_elindsq_l(p, ex, ey, iq);
        orc_elinds1x_l(p, ex, ey, ix); orc_elinds1y_l(p, ex, ey, iy);
        for (ii = 0; ii < mp12; ii++) {
            double det = p->det[(size_t)ei*mp12 + ii];
            const double* jac = &p->J[((size_t)ei*mp12 + ii)*4];
            Qaa[ii] = (J00*J00 + J10*J10)*p->Q[ii]*(scale/det);
            Qab[ii] = (J00*J01 + J10*J11)*p->Q[ii]*(scale/det);
            Qbb[ii] = (J01*J01 + J11*J11)*p->Q[ii]*(scale/det);
            Qaa[ii] *= tI[iq[ii]]; Qab[ii] *= tI[iq[ii]]; Qbb[ii] *= tI[iq[ii]];
            Qba[ii] = Qab[ii];
            orc_interp1_l(p, ex, ey, ii%mp1, ii/mp1, vel, u);
            orc_interp2_g(p, ex, ey, ii%mp1, ii/mp1, rho, &r);
            r *= tI[iq[ii]];
            r *= fac;
            Qaa[ii] *= (u[0]*r); Qba[ii] *= (u[0]*r); Qab[ii] *= (u[1]*r); Qbb[ii] *= (u[1]*r);
            orc_interp1_l(p, ex, ey, ii%mp1, ii/mp1, vel2, uh);
            uh[0] += u[0]; uh[1] += u[1];
            uh[0] *= 0.5*tI[iq[ii]]/det; uh[1] *= 0.5*tI[iq[ii]]/det;
            for (jj = 0; jj < np1; jj++) {
                lx[jj] = orc_node_eval(nn, p->nx, p->qx[ii%mp1] + 0.5*tau*uh[0], jj);
                ly[jj] = orc_node_eval(nn, p->nx, p->qx[ii/mp1] + 0.5*tau*uh[1], jj);
            }
            for (jj = 0; jj < nj; jj++) {
                MUt[jj*mp12 + ii] = lx[jj%np1]*p->ejxi[(ii/mp1)*nn + jj/np1];
                MVt[jj*mp12 + ii] = p->ejxi[(ii%mp1)*nn + jj%nn]*ly[jj/nn];
            }
        }
        orc_la->axb(nj, mp12, MUt, Qaa, rhs); for (k = 0; k < nj; k++) vl[ix[k]] += rhs[k];
        orc_la->axb(nj, mp12, MUt, Qab, rhs); for (k = 0; k < nj; k++) vl[ix[k]] += rhs[k];
        orc_la->axb(nj, mp12, MVt, Qba, rhs); for (k = 0; k < nj; k++) vl[iy[k]] += rhs[k];
        orc_la->axb(nj, mp12, MVt, Qbb, rhs); for (k = 0; k < nj; k++) vl[iy[k]] += rhs[k];
    }
    free(MUt); free(MVt);
}

/* compute_k_v  eul/Assembly.cpp:1845-1856 */
static double hs_k_v(double exner, double exner_s) {
    double p = pow(exner/1004.5, 1004.5/287.0);
    double ps = pow(exner_s/1004.5, 1004.5/287.0);
    double sigma = p/ps;
    double sigma_b = 0.7;
    double k_f = 1.1574074074074073e-05;
    if (sigma < sigma_b) return 0.0;
    return k_f*(sigma - sigma_b)/(1.0 - sigma_b);
}

/* B16 Umat_ray::assemble(lev, scale, dt, exner_k, exner_s)  eul/Assembly.cpp:1876-1979 ; out like ORC_UMAT */
int orc_umat_ray_elmats(const orc_patch* p, int lev, double scale, double dt, const double* exner_k,
                        const double* exner_s, double* out) {
    int ex, ey, ei, ii, mp1 = p->mp1, mp12 = p->mp12, n1e = p->n1e, iq[128];
    double *ca = (double*)malloc(sizeof(double)*mp12), *cb = (double*)malloc(sizeof(double)*mp12), *cc = (double*)malloc(sizeof(double)*mp12);
    double* tmp = (double*)malloc(sizeof(double)*128*mp12);
    double _e, _es, k_v;
    const double* tI = p->thickInv + (size_t)lev*p->n0q;
    const double* tI0 = p->thickInv;
    for (ey = 0; ey < p->nElsX; ey++) for (ex = 0; ex < p->nElsX; ex++) {
        double* M = out + (size_t)(ey*p->nElsX + ex)*4*n1e*n1e;
        ei = ey*p->nElsX + ex;
        orc_elindsq_l(p, ex, ey, iq);
        for (ii = 0; ii < mp12; ii++) {
            double det = p->det[(size_t)ei*mp12 + ii];
            const double* jac = &p->J[((size_t)ei*mp12 + ii)*4];
            ca[ii] = (J00*J00 + J10*J10)*p->Q[ii]*(scale/det);
            cb[ii] = (J00*J01 + J10*J11)*p->Q[ii]*(scale/det);
            cc[ii] = (J01*J01 + J11*J11)*p->Q[ii]*(scale/det);
            orc_interp2_g(p, ex, ey, ii%mp1, ii/mp1, exner_k, &_e);
            orc_interp2_g(p, ex, ey, ii%mp1, ii/mp1, exner_s, &_es);
            _e *= tI[iq[ii]];
            _es *= tI0[iq[ii]];
            k_v = hs_k_v(_e, _es);
            k_v *= dt;
            ca[ii] *= k_v*tI[iq[ii]]; cb[ii] *= k_v*tI[iq[ii]]; cc[ii] *= k_v*tI[iq[ii]];
        }
        triple(p, n1e, n1e, p->Ut, ca, p->U, tmp, M + 0*n1e*n1e);
        triple(p, n1e, n1e, p->Ut, cb, p->V, tmp, M + 1*n1e*n1e);
        triple(p, n1e, n1e, p->Vt, cb, p->U, tmp, M + 2*n1e*n1e);
        triple(p, n1e, n1e, p->Vt, cc, p->V, tmp, M + 3*n1e*n1e);
    }
    free(ca); free(cb); free(cc); free(tmp);
    return 0;
}
