/*
 * oracle/oracle.h -- CPU restatement of the MiMSEM element-assembly / column hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only
 * as the checker / the timed CPU baseline.  The shipped path is mimsem_amd/csrc (HIP).
 *
 * Every function cites the reference file:line it restates (paths relative to the
 * reference checkout, eul/ flavour unless noted).  Arithmetic is IEEE FP64, indices int32,
 * and the operation ORDER inside each function follows the reference so that results are
 * bit-reproducible against the compiled reference kernels (oracle/_ref, see Makefile).
 *
 * Pinning status: A1-A5 (GLL, Lagrange node/edge tables, dense helpers, Gauss-Jordan) are
 * pinned bit-for-bit against the reference's own eul/Basis.cpp + eul/LinAlg.cpp compiled
 * in place (oracle/_ref) and against the tests/golden fixtures generated from them; topology is
 * pinned against scr/Proc2.py fixtures.  The B/C rows (Assembly.cpp / VertOps.cpp /
 * VertSolve.cpp restatements) cannot be pinned by a reference run (PETSc is absent, the
 * reference holds no golden vectors): they are composed exclusively from the pinned
 * primitives, in the reference's order -- "parity unpinned" beyond that, see DESIGN.md.
 */
#ifndef MIMSEM_ORACLE_H
#define MIMSEM_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* ---- dense helpers (eul/LinAlg.cpp) -------------------------------------------------- */
typedef struct orc_linalg {
    void (*mult)(int ni, int nj, int nk, double* A, double* B, double* C);    /* Mult_IP    :87-97   */
    void (*mult_fd)(int ni, int nj, int nk, double* A, double* d, double* C); /* Mult_FD_IP :115-132 */
    void (*mult_df)(int ni, int nj, int nk, double* d, double* B, double* C); /* Mult_DF_IP :104-112 */
    void (*tran)(int ni, int nj, double* A, double* B);                       /* Tran_IP    :151-159 */
    void (*axb)(int ni, int nj, double* A, double* x, double* b);             /* Ax_b       :162-170 */
    int  (*inv)(double* A, double* Ainv, int n);                              /* Inv        :186-269 */
} orc_linalg;

void orc_mult(int ni, int nj, int nk, double* A, double* B, double* C);
void orc_mult_fd(int ni, int nj, int nk, double* A, double* d, double* C);
void orc_mult_df(int ni, int nj, int nk, double* d, double* B, double* C);
void orc_tran(int ni, int nj, double* A, double* B);
void orc_axb(int ni, int nj, double* A, double* x, double* b);
int  orc_inv(double* A, double* Ainv, int n);
/* swap the dense helpers used by every assembly routine (tests plug the compiled
 * reference kernels from oracle/_ref in here); NULL restores the built-in restatement. */
void orc_set_linalg(const orc_linalg* la);

/* ---- basis (eul/Basis.cpp) ----------------------------------------------------------- */
int    orc_gll(int n, double* x, double* w);                          /* :22-98  (returns !=0 on bad n) */
double orc_node_eval(int n, const double* xn, double x, int i);       /* LagrangeNode::eval_q   :180-187 */
double orc_node_deriv(int n, const double* xn, double x, int i);      /* LagrangeNode::evalDeriv:189-210 */
double orc_edge_eval(int n, const double* xn, double x, int i);       /* LagrangeEdge::eval     :274-283 */
void   orc_node_table(int n, int m, double* ljxi);   /* [m+1][n+1]  ctor :124-131 */
void   orc_edge_table(int n, int m, double* ejxi);   /* [m+1][n]    ctor :241-248 */

/* ---- reference-element tables (eul/ElMats.cpp) ---------------------------------------- */
void orc_tab_P(int n, int m, double* A);  /* M0_j_xy_i  :120-142  [mp12][np1*np1] */
void orc_tab_U(int n, int m, double* A);  /* M1x_j_xy_i :20-45    [mp12][np1*n]   */
void orc_tab_V(int n, int m, double* A);  /* M1y_j_xy_i :55-80    [mp12][np1*n]   */
void orc_tab_W(int n, int m, double* A);  /* M2_j_xy_i  :90-112   [mp12][n*n]     */
void orc_tab_Q(int m, double* A);         /* Wii::assemble :167-177 [mp12]        */

/* ---- one patch = what one reference MPI rank holds (eul/Topo + eul/Geom) -------------- */
typedef struct orc_patch {
    int n, m, np1, mp1, mp12, n0e, n1e, n2e;  /* element sizes                      */
    int nElsX, nEl, nDofsX, nk;               /* Topo::nElsX, nDofsX ; Geom::nk     */
    int n0, n1x, n1y, n1, n2;                 /* local (ghosted) sizes Topo.cpp:39-97 */
    int nqX, n0q;                             /* quad-point grid: (m*nElsX+1)^2     */
    double *qx, *qw, *nx;                     /* quadrature x,w ; nodal points      */
    double *ljxi, *ejxi;                      /* [mp1][np1], [mp1][n]               */
    double *P, *U, *V, *W, *Q;                /* tables (ElMats)                    */
    double *Pt, *Ut, *Vt, *Wt;                /* transposes (Tran_IP)               */
    double *det;                              /* [nEl][mp12]      Geom::det         */
    double *J;                                /* [nEl][mp12][4]   J00 J01 J10 J11   */
    double *thick, *thickInv;                 /* [nk][n0q]        Geom.cpp:743-764  */
    double *xq, *sq;                          /* [n0q][3], [n0q][2] coords          */
} orc_patch;

orc_patch* orc_patch_create(int n, int m, int nElsX, int nk);
void       orc_patch_destroy(orc_patch* p);
/* element -> local index maps, eul/Topo.cpp:200-251 and eul/Geom.cpp:799-811 */
void orc_elinds0_l(const orc_patch* p, int ex, int ey, int* out);
void orc_elinds1x_l(const orc_patch* p, int ex, int ey, int* out);
void orc_elinds1y_l(const orc_patch* p, int ex, int ey, int* out);
void orc_elinds2_l(const orc_patch* p, int ex, int ey, int* out);
void orc_elindsq_l(const orc_patch* p, int ex, int ey, int* out);
/* geometry: coords[n0q][3] as read from geom_%04u.txt; radius = RAD_SPHERE.
 * restates Geom ctor :80-96, updateGlobalCoords :682-724, initJacobians :726-741,
 * jacobian :245-319, jacDet :321-326 (abs_det=1: eul/box fabs; 0: src signed, F12) */
void orc_patch_set_sphere_geometry(orc_patch* p, const double* coords, double radius, int abs_det);
/* levels: levs[nk+1][n0q] -> thick, thickInv (Geom::initTopog :758-763) */
void orc_patch_set_levels(orc_patch* p, const double* levs);
/* direct injection of synthetic geometry (tests) */
void orc_patch_set_metric(orc_patch* p, const double* det, const double* J);

/* interpolation at one quad point, eul/Geom.cpp:328-417 */
void orc_interp0(const orc_patch* p, int ex, int ey, int px, int py, const double* vec, double* val);
void orc_interp1_l(const orc_patch* p, int ex, int ey, int px, int py, const double* vec, double* val);
void orc_interp2_l(const orc_patch* p, int ex, int ey, int px, int py, const double* vec, double* val);
void orc_interp1_g(const orc_patch* p, int ex, int ey, int px, int py, const double* vec, double* val);
void orc_interp2_g(const orc_patch* p, int ex, int ey, int px, int py, const double* vec, double* val);

/* ---- horizontal operator classes (eul/Assembly.cpp), SURVEY 8(a) rows B1..B17 --------- */
enum orc_op {
    ORC_UMAT = 0,     /* B1  Umat::_assemble        :66-153    out [nEl][4][n1e][n1e] xx xy yx yy */
    ORC_WMAT = 1,     /* B3  Wmat::_assemble        :324-373   out [nEl][n2e][n2e]                */
    ORC_UHMAT = 2,    /* B4  Uhmat::assemble        :416-474   f1=h2 (local 2-form)               */
    ORC_PMAT = 3,     /* B6  Pmat::assemble         :2004-2046 out [nEl][n0e][n0e]                */
    ORC_PHMAT = 4,    /* B6  Pmat::assemble_h       :2048-2098 f1=h2                              */
    ORC_WTQUMAT = 5,  /* B8  WtQUmat::assemble      :933-986   f1=u1 ; out [nEl][2][n2e][n1e]     */
    ORC_ROTMAT = 6,   /* B9  RotMat::assemble       :1030-1083 f1=q0 ; out [nEl][2][n1e][n1e] xy yx */
    ORC_WHMAT = 7,    /* B11 Whmat::assemble        :1243-1299 f1=rho, flag=vert_scale_rho        */
    ORC_UTMAT = 8,    /* B12 Ut_mat::assemble       :1338-1386                                    */
    ORC_UTMAT_H = 9,  /* B12 Ut_mat::assemble_h     :1388-1438 f1=rho                             */
    ORC_UTQWMAT = 10, /* B13 UtQWmat::assemble      :1490-1538 f1=u1 ; out [nEl][2][n1e][n2e]     */
    ORC_WTQDUDZ = 11, /* B14 WtQdUdz_mat::assemble  :1581-1640 f1=u1 ; out [nEl][2][n2e][n1e]     */
    ORC_WMATINV = 12, /* B15 WmatInv::assemble      :1673-1722                                    */
    ORC_WHMATINV = 13,/* B15 WhmatInv::assemble     :1744-1802 f1=rho                             */
    ORC_NOPS
};
/* number of doubles one element contributes to `out` for this op */
int orc_op_elmat_size(const orc_patch* p, int op);
/* flag: vert_scale (UMAT/WMAT), const_vert (UHMAT), vert_scale_rho (WHMAT); f1: see enum */
int orc_op_elmats(const orc_patch* p, int op, int lev, double scale, int flag,
                  const double* f1, double* out);
/* y += sum_e P_e^T M_e P_e x on LOCAL (ghosted) vectors with elInds*_l -- the single-rank
 * content of MatSetValues(ADD)+MatMult.  x,y sized by the op's col/row space. */
int orc_op_apply(const orc_patch* p, int op, const double* elmats, const double* x, double* y);

/* upwinded shallow-water operators (src/ flavour): which 0 = Phmat::assemble_up src/Assembly.cpp:499-567
 * (f1 = hl, out [nEl][n0e][n0e], apply with ORC_PMAT), 1 = RotMat_up::assemble :1784-1853 (f1 = q0, out like
 * ORC_ROTMAT).  ul = local 1-form velocity; tau = 1/(1/(fac*dt)). */
int orc_op_elmats_up(const orc_patch* p, int which, double fac, double dt, const double* f1, const double* ul, double* out);

/* B7 projections from the quad-point grid: which 0 = WtQmat :707-751, 1 = PtQmat :766-808, 2 = UtQmat :824-902
 * (xq interleaved [n0q][2]); y is ACCUMULATED into (local 2/0/1-form vector). */
int orc_project_from_quad(const orc_patch* p, int which, const double* xq, double* y);

/* eul operators with upwinded TEST functions: which 0 = Umat::assemble_up :156-279 (f1=ui, f2=uj), 1 = Uhmat::assemble_up
 * :477-560 (f1=h2, f2=u1; tau = dt); out like ORC_UMAT.  orc_uvec_hu_up = Uvec::assemble_hu_up :2281-2373 (accumulates). */
int orc_op_elmats_testup(const orc_patch* p, int which, int lev, double scale, double tau,
                         const double* f1, const double* f2, double* out);
void orc_uvec_hu_up(const orc_patch* p, int lev, double scale, const double* vel, const double* rho, double fac,
                    double tau, const double* vel2, double* vl);

/* B16 Umat_ray::assemble(lev,scale,dt,exner_k,exner_s) :1876-1979 (Held-Suarez friction); out like ORC_UMAT */
int orc_umat_ray_elmats(const orc_patch* p, int lev, double scale, double dt, const double* exner_k,
                        const double* exner_s, double* out);

/* matrix-free vectors, eul/Assembly.cpp */
void orc_pvec(const orc_patch* p, int lev, double scale, double* vl);                       /* B5 Pvec  :602-632 (local part) */
void orc_phvec(const orc_patch* p, int lev, double scale, const double* h2, double* vl);    /* B5 Phvec :654-689 */
void orc_uvec(const orc_patch* p, int lev, double scale, int vert_scale, const double* vel, double* vl);            /* B17 :2124-2196 (pre-scatter) */
void orc_uvec_hu(const orc_patch* p, int lev, double scale, const double* vel, const double* rho, double fac, double* vl); /* :2198-2279 */
void orc_uvec_wxu(const orc_patch* p, int lev, double scale, const double* vel, const double* vort, double* vl);    /* :2375-2430 */
/* B18 Wvec::assemble :2457-2495 / assemble_K :2497-2545 with Wt = W^T (the reference never fills Wt: corrected restatement, see o_assembly.c) */
void orc_wvec(const orc_patch* p, int lev, double scale, int vert_scale, const double* rho, double* vg);
void orc_wvec_K(const orc_patch* p, int lev, double scale, const double* vel1, const double* vel2, double* vg);
/* incidence applies on local vectors (E10mat :1102-1162, E21mat :1170-1220): signs only */
void orc_e10_apply(const orc_patch* p, const double* x0, double* y1);  /* owned (non E/N boundary) edges */
void orc_e21_apply(const orc_patch* p, const double* x1, double* y2);

/* ---- CSR assembly + SpMV: the reference's MatSetValues(ADD)/MatMult cost structure ---- */
typedef struct orc_csr {
    int nrows, ncols, nnz;
    int* rowptr; int* col; double* val;
} orc_csr;
/* pattern from element index tables rows[nEl][nr], cols[nEl][nc] */
orc_csr* orc_csr_create(int nrows, int ncols, int nEl, int nr, const int* rows, int nc, const int* cols);
void orc_csr_destroy(orc_csr* A);
void orc_csr_zero(orc_csr* A);
void orc_csr_add(orc_csr* A, int nr, const int* rows, int nc, const int* cols, const double* vals);
void orc_csr_mult(const orc_csr* A, const double* x, double* y);
/* timed "assemble + MatMult" loop with the reference's cost structure; returns seconds (bench.py cpu_baseline) */
double orc_bench_assemble_mult(const orc_patch* p, int op, int lev, double scale, int flag,
                               const double* f1, const double* x, double* y, int reps);

/* ---- vertical (column) operators, eul/VertOps.cpp; vectors indexed k*n2e+i ------------ */
enum orc_colop {
    ORC_V_CONST = 0,          /* C2 AssembleConst           :188-222   blocks [nk][n2e][n2e]            */
    ORC_V_CONST_INV = 1,      /* C2 AssembleConstInv        :789-821                                    */
    ORC_V_CONST_RHO = 2,      /* C2 AssembleConstWithRho    :492-536   f1=rho[nk*n2e]                   */
    ORC_V_CONST_RHO_INV = 3,  /* C2 AssembleConstWithRhoInv :445-490                                    */
    ORC_V_CONST_THETA = 4,    /* C2 AssembleConstWithTheta  :930-975   f1=theta[(nk+1)*n2e]             */
    ORC_V_EOS_BLOCK = 5,      /* C2 Assemble_EOS_Block      :1144-1202 f1=rt                            */
    ORC_V_LINEAR = 6,         /* C3 AssembleLinear          :228-271   blocks [nk-1][n2e][n2e]          */
    ORC_V_LINEAR_INV = 7,     /* C3 AssembleLinearInv       :411-443                                    */
    ORC_V_LINEAR_RT = 8,      /* C3 AssembleLinearWithRT    :607-667   f1=rt, flag=do_internal          */
    ORC_V_LINEAR_THETA = 9,   /* C3 AssembleLinearWithTheta :669-730   f1=theta[(nk+1)*n2e]             */
    ORC_V_LINEAR_RHO2 = 10,   /* C3 AssembleLinearWithRho2  :361-409   blocks [nk+1]                    */
    ORC_V_RAYLEIGH = 11,      /* C3 AssembleRayleigh        :826-888   blocks [nk-1]                    */
    ORC_V_LINCON = 12,        /* C4 AssembleLinCon          :273-317   (nk-1) x nk  : [nk][2] (lower row k-1, upper row k) */
    ORC_V_LINCON2 = 13,       /* C4 AssembleLinCon2         :319-359   (nk+1) x nk                      */
    ORC_V_CONLIN = 14,        /* C4 AssembleConLin          :890-928   nk x (nk-1)                      */
    ORC_V_CONLIN_W = 15,      /* C4 AssembleConLinWithW     :538-605   f1=velz[(nk-1)*n2e]              */
    ORC_V_CONLIN_RHODPI = 16, /* C4 AssembleConLinWithRhodPi:1307-1378 f1=theta(rt)[nk*n2e] f2=dpi[(nk-1)*n2e] */
    ORC_V_LINEAR_RAYLEIGH_INV = 17, /* C3 AssembleLinearWithRayleighInv :1380-1413 param=dt_fric              */
    ORC_V_EOS_BLOCK_INV = 18, /* C2 Assemble_EOS_BlockInv   :1049-1142 f1=rt f2=theta[(nk+1)*n2e] or NULL         */
    ORC_V_LINEAR_RHO2_UP = 19,/* C3 AssembleLinearWithRho2_up :1415-1490 f1=rho param=dt uh=[nk][n1]             */
    ORC_V_LINCON2_UP = 20,    /* C4 AssembleLinCon2_up      :1492-1561 param=dt uh=[nk][n1]                       */
    ORC_V_NOPS
};
/* dense result: out is a row-major (rows x cols) matrix of the column operator, zeroed
 * then filled exactly as the reference's MatSetValues sequence does (INSERT vs ADD kept). */
int orc_colop_dims(const orc_patch* p, int colop, int* rows, int* cols);
int orc_colop_dense(const orc_patch* p, int colop, int ex, int ey, int flag,
                    const double* f1, const double* f2, double* out);
/* EOS vectors (C7) */
void orc_eos_residual(const orc_patch* p, int ex, int ey, const double* rt, const double* exner, double* out); /* :987-1047 */
void orc_eos_rhs(const orc_patch* p, int ex, int ey, const double* rt, double factor, double exponent, double* out); /* :732-787 */
void orc_const_log_theta_plus_eta(const orc_patch* p, int ex, int ey, const double* theta, const double* eta, double* out); /* :1204-1255 */
void orc_const_rho_exp_eta(const orc_patch* p, int ex, int ey, const double* rho, const double* eta, double* out);       /* :1257-1305 */
/* L2Vecs transposes (C9), eul/L2Vecs.cpp:55-101: vh [nk][n2], vz [nEl][nk*n2e] */
void orc_horiz_to_vert(const orc_patch* p, const double* vh, double* vz);
void orc_vert_to_horiz(const orc_patch* p, const double* vz, double* vh);
/* dense LU solve with partial pivoting standing in for PETSc PCLU on the tiny column systems */
int orc_dense_solve(int n, double* A, double* b, double* x);
/* C6 diagTheta_L2 eul/VertSolve.cpp:322-352 (one column) and diagTheta2 :289-319 */
int orc_diag_theta_L2(const orc_patch* p, int ex, int ey, const double* rho, const double* rt, double* theta);
int orc_diag_theta2(const orc_patch* p, int ex, int ey, const double* rho, const double* rt, double* theta);
/* C5 solve_schur_column_eta eul/VertSolve.cpp:677-823 (dense restatement of the Mat chain).
 * F_* are modified in place exactly as the reference does; d_* are outputs. */
int orc_solve_schur_column_eta(const orc_patch* p, int ex, int ey, double dt,
        const double* theta, const double* velz, const double* rho, const double* eta, const double* pi,
        double* F_u, double* F_rho, double* F_eta, double* F_pi,
        double* d_u, double* d_rho, double* d_eta, double* d_pi, double* Lpi_out);

/* Held-Suarez / Strang column rows: the colops that take a scalar parameter and/or the horizontal velocity */
int orc_colop_dims_ex(const orc_patch* p, int colop, int* rows, int* cols);
int orc_colop_dense_ex(const orc_patch* p, int colop, int ex, int ey, int flag, double param,
                       const double* f1, const double* f2, const double* uh, double* out);
/* C6 diagTheta_up eul/VertSolve.cpp:354-384 (one column); uh = [nk][n1] local horizontal 1-forms */
int orc_diag_theta_up(const orc_patch* p, int ex, int ey, double dt, const double* rho, const double* rt,
                      const double* uh, double* theta);
/* AssembleTempForcing_HS eul/VertOps.cpp:1589-1633 (latitude from the patch's quad-point coordinates) */
void orc_temp_forcing_hs(const orc_patch* p, int ex, int ey, const double* exner, const double* theta,
                         const double* rho, double* vec);
/* C5 solve_schur_column_3 eul/VertSolve.cpp:504-675 (dense restatement; Lrt_out optional N x N) */
int orc_solve_schur_column_3(const orc_patch* p, int ex, int ey, double dt, int flags /* 1|2 = box twin */,
        const double* theta, const double* velz, const double* rho, const double* rt, const double* pi,
        double* F_u, double* F_rho, double* F_rt, double* F_pi,
        double* d_u, double* d_rho, double* d_rt, double* d_pi, double* Lrt_out);

#ifdef __cplusplus
}
#endif
#endif
