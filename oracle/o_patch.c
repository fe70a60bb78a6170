/* oracle/o_patch.c -- TEST INFRASTRUCTURE (see oracle.h).
 * One "patch" = the data one reference MPI rank holds: element->local index formulas
 * (eul/Topo.cpp:200-251), the quad-point grid and sphere Jacobians (eul/Geom.cpp:245-326,
 * 682-764) and the point interpolators (eul/Geom.cpp:328-417). */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

extern const orc_linalg* orc_la;

static double* dnew(size_t n) { return (double*)calloc(n ? n : 1, sizeof(double)); }

orc_patch* orc_patch_create(int n, int m, int nElsX, int nk) {
    orc_patch* p = (orc_patch*)calloc(1, sizeof(orc_patch));
    p->n = n; p->m = m; p->np1 = n+1; p->mp1 = m+1; p->mp12 = p->mp1*p->mp1;
    p->n0e = p->np1*p->np1; p->n1e = p->np1*n; p->n2e = n*n;
    p->nElsX = nElsX; p->nEl = nElsX*nElsX; p->nDofsX = n*nElsX; p->nk = nk;
    p->n0 = (p->nDofsX+1)*(p->nDofsX+1);
    p->n1x = (p->nDofsX+1)*p->nDofsX; p->n1y = p->n1x; p->n1 = p->n1x + p->n1y;
    p->n2 = p->nDofsX*p->nDofsX;
    p->nqX = m*nElsX; p->n0q = (p->nqX+1)*(p->nqX+1);
    p->qx = dnew(p->mp1); p->qw = dnew(p->mp1); p->nx = dnew(p->np1);
    { double wtmp[16]; orc_gll(m, p->qx, p->qw); orc_gll(n, p->nx, wtmp); }
    p->ljxi = dnew(p->mp1*p->np1); p->ejxi = dnew(p->mp1*n);
    orc_node_table(n, m, p->ljxi); orc_edge_table(n, m, p->ejxi);
    p->P = dnew(p->mp12*p->n0e); p->U = dnew(p->mp12*p->n1e); p->V = dnew(p->mp12*p->n1e);
    p->W = dnew(p->mp12*p->n2e); p->Q = dnew(p->mp12);
    orc_tab_P(n, m, p->P); orc_tab_U(n, m, p->U); orc_tab_V(n, m, p->V); orc_tab_W(n, m, p->W); orc_tab_Q(m, p->Q);
    p->Pt = dnew(p->mp12*p->n0e); p->Ut = dnew(p->mp12*p->n1e); p->Vt = dnew(p->mp12*p->n1e); p->Wt = dnew(p->mp12*p->n2e);
    orc_la->tran(p->mp12, p->n0e, p->P, p->Pt);
    orc_la->tran(p->mp12, p->n1e, p->U, p->Ut);
    orc_la->tran(p->mp12, p->n1e, p->V, p->Vt);
    orc_la->tran(p->mp12, p->n2e, p->W, p->Wt);
    p->det = dnew((size_t)p->nEl*p->mp12);
    p->J = dnew((size_t)p->nEl*p->mp12*4);
    p->thick = dnew((size_t)nk*p->n0q); p->thickInv = dnew((size_t)nk*p->n0q);
    p->xq = dnew((size_t)p->n0q*3); p->sq = dnew((size_t)p->n0q*2);
    return p;
}

void orc_patch_destroy(orc_patch* p) {
    if (!p) return;
    free(p->qx); free(p->qw); free(p->nx); free(p->ljxi); free(p->ejxi);
    free(p->P); free(p->U); free(p->V); free(p->W); free(p->Q);
    free(p->Pt); free(p->Ut); free(p->Vt); free(p->Wt);
    free(p->det); free(p->J); free(p->thick); free(p->thickInv); free(p->xq); free(p->sq);
    free(p);
}

/* Topo::elInds0_l :200-212 -- nodes, row-major over the (nDofsX+1)^2 patch grid */
void orc_elinds0_l(const orc_patch* p, int ex, int ey, int* out) {
    int ix, iy, k = 0;
    for (iy = 0; iy < p->np1; iy++)
        for (ix = 0; ix < p->np1; ix++)
            out[k++] = (ey*p->n + iy)*(p->nDofsX + 1) + ex*p->n + ix;
}
/* Topo::elInds1x_l :214-226 -- x-normal edges live at even slots of the interleaved vector */
void orc_elinds1x_l(const orc_patch* p, int ex, int ey, int* out) {
    int ix, iy, k = 0;
    for (iy = 0; iy < p->n; iy++)
        for (ix = 0; ix < p->np1; ix++)
            out[k++] = 2*((ey*p->n + iy)*(p->nDofsX + 1) + ex*p->n + ix) + 0;
}
/* Topo::elInds1y_l :228-240 -- y-normal edges at odd slots */
void orc_elinds1y_l(const orc_patch* p, int ex, int ey, int* out) {
    int ix, iy, k = 0;
    for (iy = 0; iy < p->np1; iy++)
        for (ix = 0; ix < p->n; ix++)
            out[k++] = 2*((ey*p->n + iy)*(p->nDofsX) + ex*p->n + ix) + 1;
}
/* Topo::elInds2_l :242-251 -- faces are element-contiguous */
void orc_elinds2_l(const orc_patch* p, int ex, int ey, int* out) {
    int k, off = (ey*p->nElsX + ex)*p->n2e;
    for (k = 0; k < p->n2e; k++) out[k] = off + k;
}
/* Geom::elInds0_l :799-811 -- quadrature-point grid */
void orc_elindsq_l(const orc_patch* p, int ex, int ey, int* out) {
    int ix, iy, k = 0;
    for (iy = 0; iy < p->mp1; iy++)
        for (ix = 0; ix < p->mp1; ix++)
            out[k++] = (ey*p->m + iy)*(p->nqX + 1) + ex*p->m + ix;
}

/* bilinear blend of the four element corners at (x1,x2) -- shared by :259-263 and :707-709 */
static void corner_blend(const double* c1, const double* c2, const double* c3, const double* c4,
                         double x1, double x2, double* r) {
    int d;
    for (d = 0; d < 3; d++)
        r[d] = 0.25*((1.0-x1)*(1.0-x2)*c1[d] + (1.0+x1)*(1.0-x2)*c2[d] + (1.0+x1)*(1.0+x2)*c3[d] + (1.0-x1)*(1.0+x2)*c4[d]);
}

/* Geom::jacobian :245-319 (Guba et al. 2014): J = A.B.C.D * R/(4|r~|) */
static void sphere_jacobian(const orc_patch* p, const int* iq, int px, int py, double radius, double* jac) {
    int mp1 = p->mp1, i, j, k;
    const double* c1 = &p->xq[3*iq[0]];
    const double* c2 = &p->xq[3*iq[mp1-1]];
    const double* c3 = &p->xq[3*iq[mp1*mp1-1]];
    const double* c4 = &p->xq[3*iq[(mp1-1)*mp1]];
    const double* ss = &p->sq[2*iq[py*mp1+px]];
    double x1 = p->qx[px], x2 = p->qx[py], rt[3], rinv;
    double A[2][3], B[3][3], C[3][4], D[4][2], AB[2][3], ABC[2][4], out[2][2];

    corner_blend(c1, c2, c3, c4, x1, x2, rt);
    rinv = 1.0/sqrt(rt[0]*rt[0] + rt[1]*rt[1] + rt[2]*rt[2]);

    A[0][0] = -sin(ss[0]); A[0][1] = +cos(ss[0]); A[0][2] = 0.0;
    A[1][0] = 0.0;         A[1][1] = 0.0;         A[1][2] = 1.0;

    B[0][0] = +sin(ss[0])*sin(ss[0])*cos(ss[1])*cos(ss[1]) + sin(ss[1])*sin(ss[1]);
    B[0][1] = -0.5*sin(2.0*ss[0])*cos(ss[1])*cos(ss[1]);
    B[0][2] = -0.5*cos(ss[0])*sin(2.0*ss[1]);
    B[1][0] = -0.5*sin(2.0*ss[0])*cos(ss[1])*cos(ss[1]);
    B[1][1] = +cos(ss[0])*cos(ss[0])*cos(ss[1])*cos(ss[1]) + sin(ss[1])*sin(ss[1]);
    B[1][2] = -0.5*sin(ss[0])*sin(2.0*ss[1]);
    B[2][0] = -cos(ss[0])*sin(ss[1]);
    B[2][1] = -sin(ss[0])*sin(ss[1]);
    B[2][2] = +cos(ss[1]);

    for (i = 0; i < 3; i++) { C[i][0] = c1[i]; C[i][1] = c2[i]; C[i][2] = c3[i]; C[i][3] = c4[i]; }

    D[0][0] = -1.0 + x2; D[0][1] = -1.0 + x1;
    D[1][0] = +1.0 - x2; D[1][1] = -1.0 - x1;
    D[2][0] = +1.0 + x2; D[2][1] = +1.0 + x1;
    D[3][0] = -1.0 - x2; D[3][1] = +1.0 - x1;

    for (i = 0; i < 2; i++) for (j = 0; j < 3; j++) { AB[i][j] = 0.0; for (k = 0; k < 3; k++) AB[i][j] += A[i][k]*B[k][j]; }
    for (i = 0; i < 2; i++) for (j = 0; j < 4; j++) { ABC[i][j] = 0.0; for (k = 0; k < 3; k++) ABC[i][j] += AB[i][k]*C[k][j]; }
    for (i = 0; i < 2; i++) for (j = 0; j < 2; j++) { out[i][j] = 0.0; for (k = 0; k < 4; k++) out[i][j] += ABC[i][k]*D[k][j]; }

    jac[0] = out[0][0]*(0.25*radius*rinv);
    jac[1] = out[0][1]*(0.25*radius*rinv);
    jac[2] = out[1][0]*(0.25*radius*rinv);
    jac[3] = out[1][1]*(0.25*radius*rinv);
}

void orc_patch_set_sphere_geometry(orc_patch* p, const double* coords, double radius, int abs_det) {
    int ex, ey, ii, jj, el, iq[128], mp1 = p->mp1, mp12 = p->mp12;
    double rt[3], mag, dj;
    /* Geom ctor :80-96: cartesian as read; (lon,lat) by atan2/asin */
    memcpy(p->xq, coords, sizeof(double)*3*p->n0q);
    for (ii = 0; ii < p->n0q; ii++) {
        p->sq[2*ii+0] = atan2(p->xq[3*ii+1], p->xq[3*ii+0]);
        p->sq[2*ii+1] = asin(p->xq[3*ii+2]/radius);
    }
    /* updateGlobalCoords :682-724: re-project element-interior points through the corner map */
    for (ey = 0; ey < p->nElsX; ey++) for (ex = 0; ex < p->nElsX; ex++) {
        const double *c1, *c2, *c3, *c4;
        orc_elindsq_l(p, ex, ey, iq);
        c1 = &p->xq[3*iq[0]]; c2 = &p->xq[3*iq[mp1-1]]; c3 = &p->xq[3*iq[mp12-1]]; c4 = &p->xq[3*iq[(mp1-1)*mp1]];
        for (ii = 0; ii < mp12; ii++) {
            if (ii == 0 || ii == mp1-1 || ii == mp12-1 || ii == mp1*(mp1-1)) continue;
            jj = iq[ii];
            corner_blend(c1, c2, c3, c4, p->qx[ii%mp1], p->qx[ii/mp1], rt);
            mag = sqrt(rt[0]*rt[0] + rt[1]*rt[1] + rt[2]*rt[2]);
            p->xq[3*jj+0] = radius*rt[0]/mag;
            p->xq[3*jj+1] = radius*rt[1]/mag;
            p->xq[3*jj+2] = radius*rt[2]/mag;
            p->sq[2*jj+0] = atan2(p->xq[3*jj+1], p->xq[3*jj+0]);
            p->sq[2*jj+1] = asin(p->xq[3*jj+2]/radius);
        }
    }
    /* initJacobians :726-741 + jacDet :321-326 */
    for (ey = 0; ey < p->nElsX; ey++) for (ex = 0; ex < p->nElsX; ex++) {
        el = ey*p->nElsX + ex;
        orc_elindsq_l(p, ex, ey, iq);
        for (ii = 0; ii < mp12; ii++) {
            double* jac = &p->J[((size_t)el*mp12 + ii)*4];
            sphere_jacobian(p, iq, ii%mp1, ii/mp1, radius, jac);
            dj = jac[0]*jac[3] - jac[1]*jac[2];
            p->det[(size_t)el*mp12 + ii] = abs_det ? fabs(dj) : dj;
        }
    }
}

/* Geom::initTopog :758-763 (levs are supplied by the caller) */
void orc_patch_set_levels(orc_patch* p, const double* levs) {
    int k, j;
    for (k = 0; k < p->nk; k++)
        for (j = 0; j < p->n0q; j++) {
            p->thick[(size_t)k*p->n0q + j] = levs[(size_t)(k+1)*p->n0q + j] - levs[(size_t)k*p->n0q + j];
            p->thickInv[(size_t)k*p->n0q + j] = 1.0/p->thick[(size_t)k*p->n0q + j];
        }
}

void orc_patch_set_metric(orc_patch* p, const double* det, const double* J) {
    memcpy(p->det, det, sizeof(double)*(size_t)p->nEl*p->mp12);
    memcpy(p->J, J, sizeof(double)*(size_t)p->nEl*p->mp12*4);
}

/* ---- Geom::interp* :328-417 ----------------------------------------------------------- */
void orc_interp0(const orc_patch* p, int ex, int ey, int px, int py, const double* vec, double* val) {
    int inds[128], j, pxy = py*p->mp1 + px;
    orc_elinds0_l(p, ex, ey, inds);
    val[0] = 0.0;
    for (j = 0; j < p->n0e; j++) val[0] += vec[inds[j]]*p->P[pxy*p->n0e + j];
}
void orc_interp1_l(const orc_patch* p, int ex, int ey, int px, int py, const double* vec, double* val) {
    int ix[128], iy[128], j, pxy = py*p->mp1 + px;
    orc_elinds1x_l(p, ex, ey, ix);
    orc_elinds1y_l(p, ex, ey, iy);
    val[0] = 0.0; val[1] = 0.0;
    for (j = 0; j < p->n1e; j++) {
        val[0] += vec[ix[j]]*p->U[pxy*p->n1e + j];
        val[1] += vec[iy[j]]*p->V[pxy*p->n1e + j];
    }
}
void orc_interp2_l(const orc_patch* p, int ex, int ey, int px, int py, const double* vec, double* val) {
    int inds[128], j, pxy = py*p->mp1 + px;
    orc_elinds2_l(p, ex, ey, inds);
    val[0] = 0.0;
    for (j = 0; j < p->n2e; j++) val[0] += vec[inds[j]]*p->W[pxy*p->n2e + j];
}
void orc_interp1_g(const orc_patch* p, int ex, int ey, int px, int py, const double* vec, double* val) {
    int el = ey*p->nElsX + ex, pi = py*p->mp1 + px;
    double l[2], dj = p->det[(size_t)el*p->mp12 + pi];
    const double* jac = &p->J[((size_t)el*p->mp12 + pi)*4];
    orc_interp1_l(p, ex, ey, px, py, vec, l);
    val[0] = (jac[0]*l[0] + jac[1]*l[1])/dj;
    val[1] = (jac[2]*l[0] + jac[3]*l[1])/dj;
}
void orc_interp2_g(const orc_patch* p, int ex, int ey, int px, int py, const double* vec, double* val) {
    int el = ey*p->nElsX + ex, pi = py*p->mp1 + px;
    double l[1];
    orc_interp2_l(p, ex, ey, px, py, vec, l);
    val[0] = l[0]/p->det[(size_t)el*p->mp12 + pi];
}

/* ---- L2Vecs::HorizToVert / VertToHoriz eul/L2Vecs.cpp:55-101 ------------------------- */
void orc_horiz_to_vert(const orc_patch* p, const double* vh, double* vz) {
    int ex, ey, ei, kk, ii, inds[128];
    for (ey = 0; ey < p->nElsX; ey++) for (ex = 0; ex < p->nElsX; ex++) {
        ei = ey*p->nElsX + ex;
        orc_elinds2_l(p, ex, ey, inds);
        for (kk = 0; kk < p->nk; kk++)
            for (ii = 0; ii < p->n2e; ii++)
                vz[(size_t)ei*p->nk*p->n2e + kk*p->n2e + ii] = vh[(size_t)kk*p->n2 + inds[ii]];
    }
}
void orc_vert_to_horiz(const orc_patch* p, const double* vz, double* vh) {
    int ex, ey, ei, kk, ii, inds[128];
    for (ey = 0; ey < p->nElsX; ey++) for (ex = 0; ex < p->nElsX; ex++) {
        ei = ey*p->nElsX + ex;
        orc_elinds2_l(p, ex, ey, inds);
        for (kk = 0; kk < p->nk; kk++)
            for (ii = 0; ii < p->n2e; ii++)
                vh[(size_t)kk*p->n2 + inds[ii]] = vz[(size_t)ei*p->nk*p->n2e + kk*p->n2e + ii];
    }
}
