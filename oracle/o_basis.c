/* oracle/o_basis.c -- TEST INFRASTRUCTURE (see oracle.h).
 * CPU restatement of eul/Basis.cpp (GLL rule, Lagrange nodal basis, edge/histopolant basis)
 * and eul/ElMats.cpp (tensor-product evaluation tables). */
#include <math.h>
#include <stdlib.h>
#include "oracle.h"

/* Gauss-Lobatto-Legendre points/weights, closed forms n=1..6, literals n=7.
 * Restates GaussLobatto::GaussLobatto eul/Basis.cpp:22-98 (same expressions so that the
 * doubles agree bit-for-bit), including the sum(w)==2 self check :91-97. */
int orc_gll(int n, double* x, double* w) {
    double a, s;
    int i;
    switch (n) {
    case 1:
        x[0] = -1.0; x[1] = +1.0;
        w[0] = 1.0;  w[1] = 1.0;
        break;
    case 2:
        x[0] = -1.0; x[1] = 0.0; x[2] = +1.0;
        w[0] = 1.0/3.0; w[1] = 4.0/3.0; w[2] = 1.0/3.0;
        break;
    case 3:
        x[0] = -1.0; x[1] = -sqrt(0.2); x[2] = +sqrt(0.2); x[3] = +1.0;
        w[0] = w[3] = 1.0/6.0; w[1] = w[2] = 5.0/6.0;
        break;
    case 4:
        x[0] = -1.0; x[1] = -sqrt(3.0/7.0); x[2] = 0.0; x[3] = +sqrt(3.0/7.0); x[4] = +1.0;
        w[0] = w[4] = 0.1; w[1] = w[3] = 49.0/90.0; w[2] = 64.0/90.0;
        break;
    case 5:
        a = 2.0*sqrt(7.0)/21.0;
        x[0] = -1.0; x[1] = -sqrt(1.0/3.0+a); x[2] = -sqrt(1.0/3.0-a);
        x[3] = +sqrt(1.0/3.0-a); x[4] = +sqrt(1.0/3.0+a); x[5] = +1.0;
        w[0] = w[5] = 1.0/15.0;
        w[1] = w[4] = (14.0-sqrt(7.0))/30.0;
        w[2] = w[3] = (14.0+sqrt(7.0))/30.0;
        break;
    case 6:
        a = 2.0*sqrt(5.0/3.0)/11.0;
        x[0] = -1.0; x[1] = -sqrt(5.0/11.0+a); x[2] = -sqrt(5.0/11.0-a); x[3] = 0.0;
        x[4] = +sqrt(5.0/11.0-a); x[5] = +sqrt(5.0/11.0+a); x[6] = +1.0;
        w[0] = w[6] = 1.0/21.0;
        w[1] = w[5] = (124.0-7.0*sqrt(15.0))/350.0;
        w[2] = w[4] = (124.0+7.0*sqrt(15.0))/350.0;
        w[3] = 256.0/525.0;
        break;
    case 7:
        x[0] = -1.0; x[1] = -0.871740148509607; x[2] = -0.591700181433142; x[3] = -0.209299217902479;
        x[4] = +0.209299217902479; x[5] = +0.591700181433142; x[6] = +0.871740148509607; x[7] = +1.0;
        w[0] = w[7] = 0.035714285714286; w[1] = w[6] = 0.210704227143506;
        w[2] = w[5] = 0.341122692483504; w[3] = w[4] = 0.412458794658704;
        break;
    default:
        return 1;
    }
    s = 0.0;
    for (i = 0; i <= n; i++) s += w[i];
    return (fabs(s - 2.0) > 1.0e-8) ? 2 : 0;
}

/* l_i(x) by the product formula on the nodal points xn[0..n] -- LagrangeNode::eval_q :180-187 */
double orc_node_eval(int n, const double* xn, double x, int i) {
    double y = 1.0;
    int j;
    for (j = 0; j <= n; j++) {
        if (j == i) continue;
        y *= (x - xn[j])/(xn[i] - xn[j]);
    }
    return y;
}

/* l_i'(x) -- LagrangeNode::evalDeriv :189-210 */
double orc_node_deriv(int n, const double* xn, double x, int i) {
    double aa, bb = 0.0;
    int j, k;
    for (j = 0; j <= n; j++) {
        if (j == i) continue;
        aa = 1.0;
        for (k = 0; k <= n; k++) {
            if (k == i || k == j) continue;
            aa *= (x - xn[k])/(xn[i] - xn[k]);
        }
        bb += aa/(xn[i] - xn[j]);
    }
    return bb;
}

/* e_i(x) = -sum_{j<=i} l_j'(x) -- LagrangeEdge::eval :274-283 */
double orc_edge_eval(int n, const double* xn, double x, int i) {
    double c = 0.0;
    int j;
    for (j = 0; j <= i; j++) c -= orc_node_deriv(n, xn, x, j);
    return c;
}

/* ljxi[q][j] = l_j(x_q): basis on GLL(n) nodes, evaluated at GLL(m) points. ctor :124-131 */
void orc_node_table(int n, int m, double* ljxi) {
    double xn[16], wn[16], xq[16], wq[16];
    int q, j;
    orc_gll(n, xn, wn);
    orc_gll(m, xq, wq);
    for (q = 0; q <= m; q++)
        for (j = 0; j <= n; j++)
            ljxi[q*(n+1)+j] = orc_node_eval(n, xn, xq[q], j);
}

/* ejxi[q][j] = e_j(x_q). ctor :241-248 */
void orc_edge_table(int n, int m, double* ejxi) {
    double xn[16], wn[16], xq[16], wq[16];
    int q, j;
    orc_gll(n, xn, wn);
    orc_gll(m, xq, wq);
    for (q = 0; q <= m; q++)
        for (j = 0; j < n; j++)
            ejxi[q*n+j] = orc_edge_eval(n, xn, xq[q], j);
}

/* ---- ElMats: row = quad point q=qy*mp1+qx, column = dof; flat row-major A[q*nj+j] ---- */

/* P[q][j] = l_{j%np1}(x_qx) l_{j/np1}(x_qy) -- M0_j_xy_i eul/ElMats.cpp:120-142 */
void orc_tab_P(int n, int m, double* A) {
    int np1 = n+1, mp1 = m+1, mi = mp1*mp1, nj = np1*np1, i, j;
    double* l = (double*)malloc(sizeof(double)*mp1*np1);
    orc_node_table(n, m, l);
    for (j = 0; j < nj; j++)
        for (i = 0; i < mi; i++)
            A[i*nj+j] = l[(i%mp1)*np1 + j%np1]*l[(i/mp1)*np1 + j/np1];
    free(l);
}

/* U[q][j] = l_{j%np1}(x_qx) e_{j/np1}(x_qy) -- M1x_j_xy_i :20-45 */
void orc_tab_U(int n, int m, double* A) {
    int np1 = n+1, mp1 = m+1, mi = mp1*mp1, nj = np1*n, i, j;
    double* l = (double*)malloc(sizeof(double)*mp1*np1);
    double* e = (double*)malloc(sizeof(double)*mp1*n);
    orc_node_table(n, m, l);
    orc_edge_table(n, m, e);
    for (j = 0; j < nj; j++)
        for (i = 0; i < mi; i++)
            A[i*nj+j] = l[(i%mp1)*np1 + j%np1]*e[(i/mp1)*n + j/np1];
    free(l); free(e);
}

/* V[q][j] = e_{j%n}(x_qx) l_{j/n}(x_qy) -- M1y_j_xy_i :55-80 */
void orc_tab_V(int n, int m, double* A) {
    int np1 = n+1, mp1 = m+1, mi = mp1*mp1, nj = np1*n, i, j;
    double* l = (double*)malloc(sizeof(double)*mp1*np1);
    double* e = (double*)malloc(sizeof(double)*mp1*n);
    orc_node_table(n, m, l);
    orc_edge_table(n, m, e);
    for (j = 0; j < nj; j++)
        for (i = 0; i < mi; i++)
            A[i*nj+j] = e[(i%mp1)*n + j%n]*l[(i/mp1)*np1 + j/n];
    free(l); free(e);
}

/* W[q][j] = e_{j%n}(x_qx) e_{j/n}(x_qy) -- M2_j_xy_i :90-112 */
void orc_tab_W(int n, int m, double* A) {
    int mp1 = m+1, mi = mp1*mp1, nj = n*n, i, j;
    double* e = (double*)malloc(sizeof(double)*mp1*n);
    orc_edge_table(n, m, e);
    for (j = 0; j < nj; j++)
        for (i = 0; i < mi; i++)
            A[i*nj+j] = e[(i%mp1)*n + j%n]*e[(i/mp1)*n + j/n];
    free(e);
}

/* Q[q] = w_{q%mp1} w_{q/mp1} -- Wii::assemble :167-177 */
void orc_tab_Q(int m, double* A) {
    double xq[16], wq[16];
    int mp1 = m+1, i;
    orc_gll(m, xq, wq);
    for (i = 0; i < mp1*mp1; i++) A[i] = wq[i%mp1]*wq[i/mp1];
}
