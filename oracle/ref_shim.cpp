// oracle/ref_shim.cpp -- TEST INFRASTRUCTURE.
// extern "C" doorway into the REFERENCE's own eul/LinAlg.cpp and eul/Basis.cpp, compiled in place
// from /root/reference by oracle/Makefile (target `ref`) into oracle/_ref/libmimsem_ref.so.
// No reference source is copied: this file only includes the reference headers and forwards calls,
// so tests can (a) pin the oracle's A1-A5 restatements bit-for-bit and (b) re-run every assembly
// restatement with the reference's compiled dense kernels plugged in (orc_set_linalg).
// eul/ElMats.cpp and everything above it include <petsc.h> (absent here) -> unbuildable, not attempted.
#include "LinAlg.h"   // eul/LinAlg.h
#include "Basis.h"    // eul/Basis.h

extern "C" {

void ref_Mult_IP(int ni, int nj, int nk, double* A, double* B, double* C) { Mult_IP(ni, nj, nk, A, B, C); }
void ref_Mult_FD_IP(int ni, int nj, int nk, double* A, double* B, double* C) { Mult_FD_IP(ni, nj, nk, A, B, C); }
void ref_Mult_DF_IP(int ni, int nj, int nk, double* A, double* B, double* C) { Mult_DF_IP(ni, nj, nk, A, B, C); }
void ref_Tran_IP(int ni, int nj, double* A, double* B) { Tran_IP(ni, nj, A, B); }
void ref_Ax_b(int ni, int nj, double* A, double* x, double* b) { Ax_b(ni, nj, A, x, b); }
int  ref_Inv(double* A, double* Ainv, int n) { return Inv(A, Ainv, n); }

// GaussLobatto(n): x[n+1], w[n+1]
void ref_gll(int n, double* x, double* w) {
    GaussLobatto q(n);
    for (int i = 0; i <= n; i++) { x[i] = q.x[i]; w[i] = q.w[i]; }
}
// LagrangeNode(n, GaussLobatto(m))::ljxi -> [m+1][n+1] ; LagrangeEdge::ejxi -> [m+1][n]
void ref_tables(int n, int m, double* ljxi, double* ejxi) {
    GaussLobatto q(m);
    LagrangeNode l(n, &q);
    LagrangeEdge e(n, &l);
    for (int i = 0; i <= m; i++) {
        for (int j = 0; j <= n; j++) ljxi[i*(n+1)+j] = l.ljxi[i][j];
        for (int j = 0; j < n; j++)  ejxi[i*n+j] = e.ejxi[i][j];
    }
}
// point evaluations used by the upwinded assemblies
double ref_node_eval_q(int n, int m, double x, int i) { GaussLobatto q(m); LagrangeNode l(n, &q); return l.eval_q(x, i); }
double ref_node_deriv(int n, int m, double x, int i)  { GaussLobatto q(m); LagrangeNode l(n, &q); return l.evalDeriv(x, i); }
double ref_edge_eval(int n, int m, double x, int i)   { GaussLobatto q(m); LagrangeNode l(n, &q); LagrangeEdge e(n, &l); return e.eval(x, i); }

}
