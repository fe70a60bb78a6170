// mimsem_amd/csrc/column_kernels.hip -- vertical (column) operators, SURVEY 8(a) rows C1..C9
// (reference eul/VertOps.cpp, eul/VertSolve.cpp:289-352,677-823, eul/L2Vecs.cpp:55-101).
//
// Every reference VertOps::Assemble* builds, for ONE column, nk small blocks  W^T diag(c_q) W  and drops
// them on the (bi)diagonal of a MATSEQAIJ; VertSolve then chains MatMatMult/PCLU on those matrices.
// Here the block structure is kept explicit for ALL columns at once:
//   coefficient pass  c[e][slot][q]   (one thread per quadrature point, fields interpolated with the
//                                       collocated edge table -- no dense W table reads)
//   block pass        M[e][slot]      = W^T diag(c) W  (one thread per entry, 16/25-term sums)
//   batched Gauss-Jordan (LinAlg.cpp:186-269 pivoting rules) for the *Inv operators
//   block (bi)diagonal mat-vec / mat-mat kernels, and a block-Thomas sweep for the Helmholtz solve.
// The Schur complement of solve_schur_column_eta is assembled ANALYTICALLY from these factors: every
// factor is block-diagonal or block-bidiagonal, so L_pi is block-tridiagonal -- no
// sparse mat-mat products, no symbolic phases, no per-column PETSc objects.
#include "ctx.hpp"

#define RD 287.0
#define CV 717.5
#define CP 1004.5
#define P0 100000.0
#define VSCALE 1.0e+8      /* eul/VertOps.cpp:21 */

namespace {

template <class F>
__global__ __launch_bounds__(256) void k_each(long long n, F f) {
    const long long i = (long long)blockIdx.x*256 + threadIdx.x;
    if (i < n) f(i);
}
template <class F>
int each(mimsem_ctx* c, long long n, F f) {
    if (n <= 0) return MIMSEM_OK;
    hipLaunchKernelGGL((k_each<F>), dim3((unsigned)((n + 255)/256)), dim3(256), 0, c->stream, n, f);
    MIMSEM_HIP_TRY(hipGetLastError());
    return MIMSEM_OK;
}

// geometry view handed to device lambdas
struct CG {
    int n, mp1, mp12, n2, nEl, nk;
    const double *det, *tI, *th, *E, *w;
};
CG make_cg(const mimsem_ctx* c) {
    CG g; g.n = c->es.n; g.mp1 = c->es.mp1; g.mp12 = c->es.mp12; g.n2 = c->es.n2e; g.nEl = c->nEl; g.nk = c->nk;
    g.det = c->d_det; g.tI = c->d_tI; g.th = c->d_th; g.E = c->d_E; g.w = c->d_w;
    return g;
}
__device__ __forceinline__ double g_th(const CG& g, int e, int k, int q) { return g.th[((size_t)k*g.nEl + e)*g.mp12 + q]; }
__device__ __forceinline__ double g_tI(const CG& g, int e, int k, int q) { return g.tI[((size_t)k*g.nEl + e)*g.mp12 + q]; }
__device__ __forceinline__ double g_Q(const CG& g, int q) { return g.w[q%g.mp1]*g.w[q/g.mp1]; }
__device__ __forceinline__ double g_W(const CG& g, int q, int j) { return g.E[(q%g.mp1)*g.n + j%g.n]*g.E[(q/g.mp1)*g.n + j/g.n]; }
// field f[e][slot k of nkv][j] interpolated to quad point q (the rk/tb/tt/wb loops of VertOps.cpp)
__device__ __forceinline__ double wint(const CG& g, const double* f, int nkv, int e, int k, int q) {
    const double* p = f + ((size_t)e*nkv + k)*g.n2;
    double r = 0.0;
    for (int j = 0; j < g.n2; j++) r += p[j]*g_W(g, q, j);
    return r;
}

// number of stored block rows / blocks per row of a column operator
void colop_shape(int colop, int nk, int* nr, int* nw, int* nx, int* ny) {
    *nw = 1;
    switch (colop) {
    case MIMSEM_V_CONST: case MIMSEM_V_CONST_INV: case MIMSEM_V_CONST_RHO: case MIMSEM_V_CONST_RHO_INV:
    case MIMSEM_V_CONST_THETA: case MIMSEM_V_EOS_BLOCK: *nr = nk; *nx = nk; *ny = nk; break;
    case MIMSEM_V_LINEAR: case MIMSEM_V_LINEAR_INV: case MIMSEM_V_LINEAR_RT: case MIMSEM_V_LINEAR_THETA:
    case MIMSEM_V_RAYLEIGH: *nr = nk - 1; *nx = nk - 1; *ny = nk - 1; break;
    case MIMSEM_V_LINEAR_RHO2: *nr = nk + 1; *nx = nk + 1; *ny = nk + 1; break;
    case MIMSEM_V_LINCON:  *nr = nk - 1; *nw = 2; *nx = nk; *ny = nk - 1; break;
    case MIMSEM_V_LINCON2: *nr = nk + 1; *nw = 2; *nx = nk; *ny = nk + 1; break;
    default: /* CONLIN family */ *nr = nk; *nw = 2; *nx = nk - 1; *ny = nk; break;
    }
}

// coefficient of stored block (r, w) at quad point q, BEFORE any inversion.  Restates the Q0/QB/QT loops.
__device__ double colop_coef(const CG& g, int colop, unsigned flags, int e, int r, int w, int q,
                             const double* f1, const double* f2) {
    const int nk = g.nk;
    const double det = g.det[(size_t)e*g.mp12 + q];
    const double q0 = g_Q(g, q)*(VSCALE/det);
    switch (colop) {
    case MIMSEM_V_CONST: case MIMSEM_V_CONST_INV:               // VertOps.cpp:201-209, :803-807
        return q0*g_tI(g, e, r, q);
    case MIMSEM_V_CONST_RHO: case MIMSEM_V_CONST_RHO_INV: {      // :508-521, :461-474
        double c = q0*g_tI(g, e, r, q);
        const double rk = wint(g, f1, nk, e, r, q);
        return c*(rk/(g_th(g, e, r, q)*det));
    }
    case MIMSEM_V_CONST_THETA: {                                 // :946-962 (theta on nk+1 interfaces)
        double c = q0*g_tI(g, e, r, q);
        const double tb = wint(g, f1, nk + 1, e, r, q), tt = wint(g, f1, nk + 1, e, r + 1, q);
        return c*(0.5*(tb + tt)/det);
    }
    case MIMSEM_V_LINEAR:                                        // :242-267: levels r and r+1 meet at interface r
        return q0*(0.5*g_th(g, e, r, q)) + q0*(0.5*g_th(g, e, r + 1, q));
    case MIMSEM_V_LINEAR_INV:                                    // :422-430
        return q0*(0.5*(g_th(g, e, r, q) + g_th(g, e, r + 1, q)));
    case MIMSEM_V_LINEAR_RT: {                                   // :621-662 ; flag = do_internal
        const bool internal = (flags & MIMSEM_FLAG_VERT) != 0;
        double acc = 0.0;
        for (int k = r; k <= r + 1; k++) {
            if (!internal && k > 0 && k < nk - 1) continue;
            double rk = wint(g, f1, nk, e, k, q);
            if (!internal) rk *= g_tI(g, e, k, q);
            acc += q0*(0.5*rk/det);
        }
        return acc;
    }
    case MIMSEM_V_LINEAR_THETA: {                                // :685-725: QT of level r + QB of level r+1
        const double tm = wint(g, f1, nk + 1, e, r + 1, q);
        return q0*(0.5*g_th(g, e, r, q))*(tm/det) + q0*(0.5*g_th(g, e, r + 1, q))*(tm/det);
    }
    case MIMSEM_V_LINEAR_RHO2: {                                 // :375-403
        double acc = 0.0;
        if (r > 0)  acc += q0*(0.5*wint(g, f1, nk, e, r - 1, q)/det);
        if (r < nk) acc += q0*(0.5*wint(g, f1, nk, e, r, q)/det);
        return acc;
    }
    case MIMSEM_V_RAYLEIGH: {                                    // :826-888: interfaces nk-2, nk-3, nk-4
        const int s = nk - 2 - r;
        if (s < 0 || s > 2) return 0.0;
        const double wgt = (s == 0) ? 0.5 : (s == 1 ? 0.25 : 0.125);
        return q0*(wgt*(g_th(g, e, r + 1, q) + g_th(g, e, r, q)));
    }
    case MIMSEM_V_LINCON:                                        // :285-313 (r,0)=(r,r) (r,1)=(r,r+1)
        return q0*0.5;
    case MIMSEM_V_LINCON2:                                       // :331-355 (r,0)=(r,r-1) (r,1)=(r,r)
        return (w == 0) ? (r > 0 ? q0*0.5 : 0.0) : (r < nk ? q0*0.5 : 0.0);
    case MIMSEM_V_CONLIN:                                        // :901-924 (k,0)=(k,k-1) (k,1)=(k,k)
        return (w == 0) ? (r > 0 ? q0*0.5 : 0.0) : (r < nk - 1 ? q0*0.5 : 0.0);
    case MIMSEM_V_CONLIN_W: {                                    // :551-600 ; f1 = velz on nk-1 interfaces
        const int j = r - 1 + w;
        if (j < 0 || j > nk - 2) return 0.0;
        return q0*(0.5*wint(g, f1, nk - 1, e, j, q)/det);
    }
    case MIMSEM_V_CONLIN_RHODPI: {                               // :1323-1373 ; f1 = theta (levels) f2 = dpi (interfaces)
        const int j = r - 1 + w;
        if (j < 0 || j > nk - 2) return 0.0;
        const double wb = wint(g, f2, nk - 1, e, j, q);
        double tb = wint(g, f1, nk, e, r, q);
        tb *= g_tI(g, e, r, q);
        return q0*(0.5*wb*tb/(det*det));
    }
    }
    return 0.0;
}

// coefficient pass for a whole operator: cq[e][r][w][q]
int coef_pass(mimsem_ctx* c, int colop, unsigned flags, const double* f1, const double* f2, double* cq, int nr, int nw) {
    const CG g = make_cg(c);
    const long long n = (long long)c->nEl*nr*nw*g.mp12;
    return each(c, n, [=] __device__(long long i) {
        const int q = (int)(i%g.mp12); long long t = i/g.mp12;
        const int w = (int)(t%nw); t /= nw;
        const int r = (int)(t%nr); const int e = (int)(t/nr);
        cq[i] = colop_coef(g, colop, flags, e, r, w, q, f1, f2);
    });
}

// block pass: M[b][i][j] = sum_q (W[q][i] c[b][q]) W[q][j]   (Mult_FD_IP then Mult_IP order)
int block_pass(mimsem_ctx* c, long long nb, const double* cq, double* M) {
    const CG g = make_cg(c);
    const int nn = g.n2*g.n2;
    return each(c, nb*nn, [=] __device__(long long i) {
        const long long b = i/nn; const int ij = (int)(i%nn), ii = ij/g.n2, jj = ij%g.n2;
        const double* cb = cq + b*g.mp12;
        double s = 0.0;
        for (int q = 0; q < g.mp12; q++) s += (g_W(g, q, ii)*cb[q])*g_W(g, q, jj);
        M[i] = s;
    });
}

// ---- batched Gauss-Jordan with full pivoting: one thread per block, private copy in LDS ------------
// (thread-minor layout => conflict-free; pivot search / swaps / elimination exactly as LinAlg.cpp:186-269)
__global__ __launch_bounds__(64) void k_block_inverse(long long nb, int n, int T, double* blocks, int* errcount) {
    extern __shared__ double lds[];
    const int t = threadIdx.x;
    const long long b = (long long)blockIdx.x*T + t;
    double* A = lds;                                   // A[(i*n+j)*T + t]
    int* ipiv = (int*)(lds + (size_t)n*n*T);           // [n][T]
    int* indxr = ipiv + n*T;
    int* indxc = indxr + n*T;
    if (t >= T || b >= nb) return;
    double* src = blocks + b*n*n;
    for (int k = 0; k < n*n; k++) A[k*T + t] = src[k];
    for (int j = 0; j < n; j++) ipiv[j*T + t] = 0;
    int err = 0, irow = 0, icol = 0;
    for (int i = 0; i < n; i++) {
        double big = 0.0;
        for (int j = 0; j < n; j++) {
            if (ipiv[j*T + t] == 1) continue;
            for (int k = 0; k < n; k++) {
                if (ipiv[k*T + t] == 0) {
                    const double v = fabs(A[(j*n + k)*T + t]);
                    if (v >= big) { big = v; irow = j; icol = k; }
                } else if (ipiv[k*T + t] > 1) err = 1;
            }
        }
        ++ipiv[icol*T + t];
        if (irow != icol)
            for (int l = 0; l < n; l++) {
                const double tmp = A[(irow*n + l)*T + t];
                A[(irow*n + l)*T + t] = A[(icol*n + l)*T + t];
                A[(icol*n + l)*T + t] = tmp;
            }
        indxr[i*T + t] = irow; indxc[i*T + t] = icol;
        if (fabs(A[(icol*n + icol)*T + t]) < 1.0e-12) err = 2;
        const double pivinv = 1.0/A[(icol*n + icol)*T + t];
        A[(icol*n + icol)*T + t] = 1.0;
        for (int l = 0; l < n; l++) A[(icol*n + l)*T + t] *= pivinv;
        for (int ll = 0; ll < n; ll++) {
            if (ll == icol) continue;
            const double dum = A[(ll*n + icol)*T + t];
            A[(ll*n + icol)*T + t] = 0.0;
            for (int l = 0; l < n; l++) A[(ll*n + l)*T + t] -= A[(icol*n + l)*T + t]*dum;
        }
    }
    for (int l = n - 1; l >= 0; l--) {
        const int ir = indxr[l*T + t], ic = indxc[l*T + t];
        if (ir == ic) continue;
        for (int k = 0; k < n; k++) {
            const double tmp = A[(k*n + ir)*T + t];
            A[(k*n + ir)*T + t] = A[(k*n + ic)*T + t];
            A[(k*n + ic)*T + t] = tmp;
        }
    }
    for (int k = 0; k < n*n; k++) src[k] = A[k*T + t];
    if (err && errcount) atomicAdd(errcount, 1);
}

}  // namespace

int mimsem_block_inverse_inplace(mimsem_ctx* c, long long nblocks, int n, double* blocks) {
    if (nblocks <= 0) return MIMSEM_OK;
    // matrices per workgroup: as many as fit a 144 KiB LDS budget (64 for n<=16, fewer for the 25..49-wide blocks of p>=5)
    int T = 64;
    auto need = [&](int t) { return (size_t)n*n*t*sizeof(double) + (size_t)3*n*t*sizeof(int); };
    while (T > 1 && need(T) > 144*1024) T >>= 1;
    const size_t lds = need(T);
    if (lds > 160*1024) return MIMSEM_ERR_UNSUPPORTED;
    if (lds > 64*1024)
        MIMSEM_HIP_TRY(hipFuncSetAttribute((const void*)k_block_inverse, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_block_inverse, dim3((unsigned)((nblocks + T - 1)/T)), dim3(64), lds, c->stream,
                       nblocks, n, T, blocks, (int*)nullptr);
    MIMSEM_HIP_TRY(hipGetLastError());
    return MIMSEM_OK;
}

namespace {

// workspace carving
struct WS {
    mimsem_ctx* c; double* base; long long used, cap;
    double* take(long long n) { double* p = base + used; used += n; return p; }
};

bool colop_is_inverse(int colop) {
    return colop == MIMSEM_V_CONST_INV || colop == MIMSEM_V_CONST_RHO_INV || colop == MIMSEM_V_LINEAR_INV;
}

// blocks of a column operator into M ([nEl][nr][nw][n2][n2]); cq scratch [nEl][nr][nw][mp12] (x2 for EOS)
int colop_blocks_into(mimsem_ctx* c, int colop, unsigned flags, const double* f1, const double* f2,
                      double* M, double* cq, double* tmpM) {
    int nr, nw, nx, ny;
    colop_shape(colop, c->nk, &nr, &nw, &nx, &ny);
    const long long nb = (long long)c->nEl*nr*nw;
    const int n2 = c->es.n2e, nn = n2*n2;
    int rc;
    if (colop == MIMSEM_V_EOS_BLOCK) {        // B . B(rt)^-1 . B   VertOps.cpp:1162-1196
        if ((rc = coef_pass(c, MIMSEM_V_CONST_RHO, 0, f1, nullptr, cq, nr, 1))) return rc;
        if ((rc = block_pass(c, nb, cq, tmpM))) return rc;                    // B(rt)
        if ((rc = mimsem_block_inverse_inplace(c, nb, n2, tmpM))) return rc;
        if ((rc = coef_pass(c, MIMSEM_V_CONST, 0, nullptr, nullptr, cq, nr, 1))) return rc;
        double* Bm = tmpM + nb*nn;
        if ((rc = block_pass(c, nb, cq, Bm))) return rc;                      // B
        const double* Binv = tmpM;
        return each(c, nb*nn, [=] __device__(long long i) {                   // B (Binv B)
            const long long b = i/nn; const int ij = (int)(i%nn), ii = ij/n2, jj = ij%n2;
            const double *B = Bm + b*nn, *Bi = Binv + b*nn;
            double s = 0.0;
            for (int k = 0; k < n2; k++) {
                double t = 0.0;
                for (int l = 0; l < n2; l++) t += Bi[k*n2 + l]*B[l*n2 + jj];
                s += B[ii*n2 + k]*t;
            }
            M[i] = s;
        });
    }
    if ((rc = coef_pass(c, colop, flags, f1, f2, cq, nr, nw))) return rc;
    if ((rc = block_pass(c, nb, cq, M))) return rc;
    if (colop_is_inverse(colop)) return mimsem_block_inverse_inplace(c, nb, n2, M);
    return MIMSEM_OK;
}

long long colop_ws_doubles(const mimsem_ctx* c) {
    const long long nbmax = (long long)c->nEl*(c->nk + 1)*2;
    return nbmax*(c->es.mp12 + 3LL*c->es.n2e*c->es.n2e);
}

// y = A x (or A^T x) with A given by its stored blocks.
//   nw == 1: block diagonal, y_r = M_r x_r
//   nw == 2: stored block (r,w) sits at block column col(r,w) = r + off + w, off = 0 (LINCON) or -1
//            (LINCON2 and the CONLIN family):   y_r = sum_w M(r,w) x_col(r,w)
//            transposed:                        y_j = sum_w M(r,w)^T x_r  with r = j - off - w
int stored_apply(mimsem_ctx* c, int colop, int transpose, const double* M, const double* x, double* y) {
    int nr, nw, nx, ny;
    colop_shape(colop, c->nk, &nr, &nw, &nx, &ny);
    const int n2 = c->es.n2e, nn = n2*n2, nEl = c->nEl;
    const int off = (colop == MIMSEM_V_LINCON) ? 0 : -1;
    const int nyy = transpose ? nx : ny, nxx = transpose ? ny : nx;
    return each(c, (long long)nEl*nyy*n2, [=] __device__(long long i) {
        const int a = (int)(i%n2); long long t = i/n2;
        const int ry = (int)(t%nyy), e = (int)(t/nyy);
        double s = 0.0;
        for (int w = 0; w < nw; w++) {
            int r, cx;            // stored block row, input slot
            if (nw == 1)         { r = ry; cx = ry; }
            else if (!transpose) { r = ry; cx = ry + off + w; }
            else                 { r = ry - off - w; cx = r; }
            if (r < 0 || r >= nr || cx < 0 || cx >= nxx) continue;
            const double* B = M + (((size_t)e*nr + r)*nw + w)*nn;
            const double* xv = x + ((size_t)e*nxx + cx)*n2;
            if (!transpose) { for (int k = 0; k < n2; k++) s += B[a*n2 + k]*xv[k]; }
            else            { for (int k = 0; k < n2; k++) s += B[k*n2 + a]*xv[k]; }
        }
        y[i] = s;
    });
}

}  // namespace

extern "C" {

int mimsem_colop_nblocks(const mimsem_ctx* c, int colop) {
    if (!c || colop < 0 || colop >= MIMSEM_V_COUNT) return MIMSEM_ERR_ARG;
    int nr, nw, nx, ny;
    colop_shape(colop, c->nk, &nr, &nw, &nx, &ny);
    return nr*nw;
}

int mimsem_colop_blocks(mimsem_ctx* c, int colop, unsigned flags, const double* f1, const double* f2, double* out) {
    if (!c || !out || colop < 0 || colop >= MIMSEM_V_COUNT) return MIMSEM_ERR_ARG;
    if (c->nk < 2 && colop >= MIMSEM_V_LINEAR) return MIMSEM_ERR_ARG;
    if (colop == MIMSEM_V_RAYLEIGH && c->nk < 4) return MIMSEM_ERR_ARG;
    int rc = c->ensure_col(colop_ws_doubles(c));
    if (rc) return rc;
    const long long nbmax = (long long)c->nEl*(c->nk + 1)*2;
    double* cq = c->d_col; double* tmpM = cq + nbmax*c->es.mp12;
    return colop_blocks_into(c, colop, flags, f1, f2, out, cq, tmpM);
}

int mimsem_colop_apply(mimsem_ctx* c, int colop, unsigned flags, int transpose,
                       const double* f1, const double* f2, const double* x, double* y) {
    if (!c || !x || !y || colop < 0 || colop >= MIMSEM_V_COUNT) return MIMSEM_ERR_ARG;
    int rc = c->ensure_col(colop_ws_doubles(c));
    if (rc) return rc;
    const long long nbmax = (long long)c->nEl*(c->nk + 1)*2;
    const int nn = c->es.n2e*c->es.n2e;
    double* cq = c->d_col; double* tmpM = cq + nbmax*c->es.mp12; double* M = tmpM + 2*nbmax*nn;
    if ((rc = colop_blocks_into(c, colop, flags, f1, f2, M, cq, tmpM))) return rc;
    return stored_apply(c, colop, transpose, M, x, y);
}

// L2Vecs::HorizToVert / VertToHoriz, eul/L2Vecs.cpp:55-101 ([k][e][i] <-> [e][k][i], faces element-contiguous)
int mimsem_l2_transpose(mimsem_ctx* c, int dir, int nkv, double* vh, long long hs, double* vz) {
    if (!c || !vh || !vz || nkv < 0) return MIMSEM_ERR_ARG;
    const int n2 = c->es.n2e, nEl = c->nEl;
    const int* i2 = c->d_i2;
    // index by the vertical layout so that the strided side is the read for dir=0 and the write for dir=1
    return each(c, (long long)nEl*nkv*n2, [=] __device__(long long i) {
        const int j = (int)(i%n2); long long t = i/n2;
        const int k = (int)(t%nkv), e = (int)(t/nkv);
        const size_t h = (size_t)k*hs + (i2 ? i2[e*n2 + j] : e*n2 + j);
        if (dir == 0) vz[i] = vh[h]; else vh[h] = vz[i];
    });
}

// Pvec / Phvec: diagonal 0-form mass as a vector (Assembly.cpp:602-689)
int mimsem_pvec(mimsem_ctx* c, int geom_lev0, int nlev, double scale,
                const double* h2, long long hs, double* y, long long ys) {
    if (!c || !y || nlev < 0 || geom_lev0 < 0 || geom_lev0 + nlev > c->nk) return MIMSEM_ERR_ARG;
    const ElemSizes es = c->es;
    const long long per = (long long)c->nEl*es.n0e;
    int rc = c->ensure_ye(per*nlev);
    if (rc) return rc;
    const CG g = make_cg(c);
    double* ye = c->d_ye;
    const int* i2 = c->d_i2;
    if ((rc = each(c, per*nlev, [=] __device__(long long i) {
        const int q = (int)(i%g.mp12); long long t = i/g.mp12;
        const int e = (int)(t%g.nEl), lev = (int)(t/g.nEl);
        const double det = g.det[(size_t)e*g.mp12 + q];
        const double tI = g_tI(g, e, geom_lev0 + lev, q);
        double v = scale*g_Q(g, q)*det;
        v *= tI;
        if (h2) {
            const double* hv = h2 + (size_t)lev*hs;
            double hi = 0.0;
            for (int j = 0; j < g.n2; j++) hi += hv[i2 ? i2[e*g.n2 + j] : e*g.n2 + j]*g_W(g, q, j);
            hi /= det;
            hi *= tI;
            v *= hi;
        }
        ye[i] = v;
    }))) return rc;
    return launch_gather_sum(c, 0, nlev, ye, per, 0, y, ys);
}

// EOS / log / exp vectors per (e,k), VertOps.cpp:732-787, :987-1047, :1204-1305
int mimsem_column_eos(mimsem_ctx* c, int which, const double* a, const double* b, double p0, double p1, double* out) {
    if (!c || !a || !out || which < 0 || which > 3) return MIMSEM_ERR_ARG;
    if ((which == 0 || which == 3) && !b) return MIMSEM_ERR_ARG;
    const CG g = make_cg(c);
    const int nk = c->nk;
    int rc = c->ensure_col((long long)c->nEl*nk*g.mp12);
    if (rc) return rc;
    double* rtq = c->d_col;
    if ((rc = each(c, (long long)c->nEl*nk*g.mp12, [=] __device__(long long i) {
        const int q = (int)(i%g.mp12); long long t = i/g.mp12;
        const int k = (int)(t%nk), e = (int)(t/nk);
        const double det = g.det[(size_t)e*g.mp12 + q], th = g_th(g, e, k, q);
        double v;
        if (which == 0) {          // Assemble_EOS_Residual
            double rk = wint(g, a, nk, e, k, q), ek = wint(g, b, nk, e, k, q);
            rk *= 1.0/(det*th); ek *= 1.0/(det*th);
            v = log(ek) - (RD/CV)*log(rk) - log(CP) - (RD/CV)*log(RD/P0);
            v *= 0.5*g_Q(g, q)*VSCALE;                    // WtQ = Wt diag(0.5 w SCALE); the x2 is applied below
        } else if (which == 1) {   // Assemble_EOS_RHS
            double rk = wint(g, a, nk, e, k, q);
            rk *= 1.0/(det*th);
            v = p0*pow(rk, p1);
            v *= 0.5*g_Q(g, q)*VSCALE;
        } else if (which == 2) {   // AssembleConstWithLogThetaPlusEta
            const double tb = wint(g, a, nk, e, k, q);
            double fac = log(tb/(th*det));
            if (b) fac += wint(g, b, nk, e, k, q)/(th*det);
            v = g_Q(g, q)*(VSCALE*fac);
        } else {                   // AssembleConstWithRhoExpEta
            double rk = wint(g, a, nk, e, k, q), ek = wint(g, b, nk, e, k, q);
            rk *= 1.0/(th*det); ek *= 1.0/(th*det);
            v = g_Q(g, q)*(VSCALE*rk*exp(ek));
        }
        rtq[i] = v;
    }))) return rc;
    return each(c, (long long)c->nEl*nk*g.n2, [=] __device__(long long i) {
        const int j = (int)(i%g.n2); const long long ek = i/g.n2;
        const double* r = rtq + ek*g.mp12;
        double s = 0.0;
        for (int q = 0; q < g.mp12; q++) s += g_W(g, q, j)*r[q];
        if (which < 2) s *= 2.0;
        out[i] = s;
    });
}

}  // extern "C"

// ---- WmatInv / WhmatInv applied: y = (W^T c W)^-1 x per element (Assembly.cpp:1673-1802) ------------
int mimsem_colop_block_inverse_apply(mimsem_ctx* c, int op, int geom_lev0, int nlev, double scale, unsigned flags,
                                     const double* f, long long fs, const double* x, long long xs,
                                     double* y, long long ys, double alpha) {
    const CG g = make_cg(c);
    const int n2 = g.n2, nn = n2*n2, nEl = c->nEl;
    const long long nb = (long long)nEl*nlev;
    int rc = c->ensure_col(nb*(g.mp12 + nn));
    if (rc) return rc;
    double* cq = c->d_col; double* M = cq + nb*g.mp12;
    const int* i2 = c->d_i2;
    const bool hmat = (op == MIMSEM_OP_WHMATINV);
    if ((rc = each(c, nb*g.mp12, [=] __device__(long long i) {
        const int q = (int)(i%g.mp12); long long t = i/g.mp12;
        const int e = (int)(t%nEl), lev = (int)(t/nEl);
        const double det = g.det[(size_t)e*g.mp12 + q], tI = g_tI(g, e, geom_lev0 + lev, q);
        double cv;
        if (hmat) {
            const double* hv = f + (size_t)lev*fs;
            double p = 0.0;
            for (int j = 0; j < n2; j++) p += hv[i2 ? i2[e*n2 + j] : e*n2 + j]*g_W(g, q, j);
            p /= det; p *= tI;
            cv = p*g_Q(g, q)*(scale/det);
        } else cv = g_Q(g, q)*(scale/det);
        cv *= tI;
        cq[i] = cv;
    }))) return rc;
    if ((rc = block_pass(c, nb, cq, M))) return rc;
    if ((rc = mimsem_block_inverse_inplace(c, nb, n2, M))) return rc;
    const bool accum = (flags & MIMSEM_FLAG_ACCUM) != 0;
    return each(c, nb*n2, [=] __device__(long long i) {
        const int a = (int)(i%n2); long long t = i/n2;
        const int e = (int)(t%nEl), lev = (int)(t/nEl);
        const double* B = M + t*nn;
        const double* xv = x + (size_t)lev*xs;
        double s = 0.0;
        for (int k = 0; k < n2; k++) s += B[a*n2 + k]*xv[i2 ? i2[e*n2 + k] : e*n2 + k];
        double* o = y + (size_t)lev*ys + (i2 ? i2[e*n2 + a] : e*n2 + a);
        if (accum) *o += alpha*s; else *o = alpha*s;
    });
}

// ---- C6 / C5: theta diagnosis and the column Schur solve ---------------------------------------------
namespace {

// block-array view: X[e][slot][n2*n2] with ns slots per column
struct BA { double* p; int ns; };

// C[e][r] (+)= alpha * A[e][r+da] . B[e][r+db]   for r in [0,nr); out-of-range operand slots contribute 0
int bmm(mimsem_ctx* c, int nr, BA C, BA A, int da, BA B, int db, double alpha, int accum) {
    const int n2 = c->es.n2e, nn = n2*n2, nEl = c->nEl;
    return each(c, (long long)nEl*nr*nn, [=] __device__(long long i) {
        const int ij = (int)(i%nn), ii = ij/n2, jj = ij%n2; long long t = i/nn;
        const int r = (int)(t%nr), e = (int)(t/nr);
        double* out = C.p + ((size_t)e*C.ns + r)*nn + ij;
        const int ra = r + da, rb = r + db;
        double s = 0.0;
        if (ra >= 0 && ra < A.ns && rb >= 0 && rb < B.ns) {
            const double* a = A.p + ((size_t)e*A.ns + ra)*nn;
            const double* b = B.p + ((size_t)e*B.ns + rb)*nn;
            for (int k = 0; k < n2; k++) s += a[ii*n2 + k]*b[k*n2 + jj];
        }
        if (accum) *out += alpha*s; else *out = alpha*s;
    });
}
// y[e][r] (+)= alpha * A[e][r+da] x[e][r+dx]   (vectors with nsy / nsx slots)
int bmv(mimsem_ctx* c, int nr, double* y, int nsy, BA A, int da, const double* x, int nsx, int dx, double alpha, int accum) {
    const int n2 = c->es.n2e, nn = n2*n2, nEl = c->nEl;
    return each(c, (long long)nEl*nr*n2, [=] __device__(long long i) {
        const int a = (int)(i%n2); long long t = i/n2;
        const int r = (int)(t%nr), e = (int)(t/nr);
        double* out = y + ((size_t)e*nsy + r)*n2 + a;
        const int ra = r + da, rx = r + dx;
        double s = 0.0;
        if (ra >= 0 && ra < A.ns && rx >= 0 && rx < nsx) {
            const double* m = A.p + ((size_t)e*A.ns + ra)*nn;
            const double* xv = x + ((size_t)e*nsx + rx)*n2;
            for (int k = 0; k < n2; k++) s += m[a*n2 + k]*xv[k];
        }
        if (accum) *out += alpha*s; else *out = alpha*s;
    });
}

// block Thomas for  L d = f,  L block-tridiagonal [nEl][nk][3] (sub, diag, super).
// One workgroup per column; entries of the working blocks live in LDS; Gauss-Jordan (partial pivoting
// by column-max) on the running diagonal block, cooperative across the threads of the workgroup.
__global__ void k_block_thomas(int nk, int n2, const double* __restrict__ L, const double* __restrict__ f,
                               double* __restrict__ d, double* __restrict__ Gws, double* __restrict__ yws) {
    extern __shared__ double sm[];
    const int nn = n2*n2, tid = threadIdx.x, e = blockIdx.x, nt = blockDim.x;
    double* D = sm;            // running diagonal block  [n2][n2]
    double* Di = D + nn;       // its inverse
    double* T = Di + nn;       // scratch block
    double* v = T + nn;        // running rhs [n2]
    double* u = v + n2;        // scratch vec
    __shared__ int piv;
    const double* Le = L + (size_t)e*nk*3*nn;
    double* G = Gws + (size_t)e*nk*nn;      // G_k = Dk'^-1 . super_k
    double* yv = yws + (size_t)e*nk*n2;     // y_k = Dk'^-1 . rhs_k'
    for (int k = 0; k < nk; k++) {
        const double* sub = Le + ((size_t)k*3 + 0)*nn;
        const double* dia = Le + ((size_t)k*3 + 1)*nn;
        const double* sup = Le + ((size_t)k*3 + 2)*nn;
        // D = diag_k - sub_k G_{k-1} ; v = f_k - sub_k y_{k-1}
        for (int t = tid; t < nn; t += nt) {
            double s = dia[t];
            if (k > 0) { const int i = t/n2, j = t%n2; const double* Gp = G + (size_t)(k - 1)*nn;
                         for (int m = 0; m < n2; m++) s -= sub[i*n2 + m]*Gp[m*n2 + j]; }
            D[t] = s; Di[t] = (t/n2 == t%n2) ? 1.0 : 0.0;
        }
        for (int t = tid; t < n2; t += nt) {
            double s = f[((size_t)e*nk + k)*n2 + t];
            if (k > 0) { const double* yp = yv + (size_t)(k - 1)*n2; for (int m = 0; m < n2; m++) s -= sub[t*n2 + m]*yp[m]; }
            v[t] = s;
        }
        __syncthreads();
        // Gauss-Jordan on [D | Di] with partial pivoting
        for (int col = 0; col < n2; col++) {
            if (tid == 0) {
                int p = col; double big = fabs(D[col*n2 + col]);
                for (int r = col + 1; r < n2; r++) if (fabs(D[r*n2 + col]) > big) { big = fabs(D[r*n2 + col]); p = r; }
                piv = p;
            }
            __syncthreads();
            const int p = piv;
            if (p != col) {
                for (int t = tid; t < 2*n2; t += nt) {
                    double* M = (t < n2) ? D : Di; const int j = t%n2;
                    const double tmp = M[col*n2 + j]; M[col*n2 + j] = M[p*n2 + j]; M[p*n2 + j] = tmp;
                }
                __syncthreads();
            }
            const double pinv = 1.0/D[col*n2 + col];
            __syncthreads();
            for (int t = tid; t < 2*n2; t += nt) { double* M = (t < n2) ? D : Di; M[col*n2 + t%n2] *= pinv; }
            __syncthreads();
            // eliminate column `col` from every other row
            for (int t = tid; t < n2; t += nt) u[t] = (t == col) ? 0.0 : D[t*n2 + col];
            __syncthreads();
            for (int t = tid; t < 2*nn; t += nt) {
                double* M = (t < nn) ? D : Di; const int ij = t%nn, i = ij/n2, j = ij%n2;
                if (i != col) M[ij] -= u[i]*M[col*n2 + j];
            }
            __syncthreads();
        }
        // G_k = Di . super_k ; y_k = Di . v
        for (int t = tid; t < nn; t += nt) {
            const int i = t/n2, j = t%n2; double s = 0.0;
            if (k < nk - 1) for (int m = 0; m < n2; m++) s += Di[i*n2 + m]*sup[m*n2 + j];
            G[(size_t)k*nn + t] = s;
        }
        for (int t = tid; t < n2; t += nt) {
            double s = 0.0;
            for (int m = 0; m < n2; m++) s += Di[t*n2 + m]*v[m];
            yv[(size_t)k*n2 + t] = s;
        }
        __syncthreads();
    }
    // back substitution: d_k = y_k - G_k d_{k+1}
    for (int k = nk - 1; k >= 0; k--) {
        for (int t = tid; t < n2; t += nt) {
            double s = yv[(size_t)k*n2 + t];
            if (k < nk - 1) { const double* Gk = G + (size_t)k*nn; const double* dn = d + ((size_t)e*nk + k + 1)*n2;
                              for (int m = 0; m < n2; m++) s -= Gk[t*n2 + m]*dn[m]; }
            d[((size_t)e*nk + k)*n2 + t] = s;
        }
        __syncthreads();
    }
}

int block_thomas(mimsem_ctx* c, const double* L, const double* f, double* d, double* Gws, double* yws) {
    const int n2 = c->es.n2e, nn = n2*n2;
    const int nt = std::min(256, ((nn + 63)/64)*64);
    const size_t lds = (size_t)(3*nn + 2*n2)*sizeof(double);
    hipLaunchKernelGGL(k_block_thomas, dim3(c->nEl), dim3(nt), lds, c->stream, c->nk, n2, L, f, d, Gws, yws);
    MIMSEM_HIP_TRY(hipGetLastError());
    return MIMSEM_OK;
}

struct Schur {
    // block arrays (all [nEl][ns][nn])
    BA B, Binv, Ainv, T, Rr, X, Npi, Nrho, R2, C2, DIVl, DIVu, Gl, Gu, M1, L, G;
    double *gpi, *geta, *rlump, *tA, *tB;
};

// assemble every factor of the Helmholtz operator; see the derivation in DESIGN.md ("C5")
int schur_assemble(mimsem_ctx* c, double dt, const double* theta, const double* rho, const double* eta,
                   const double* pi, Schur& S) {
    const int nk = c->nk, nm = nk - 1, n2 = c->es.n2e, nn = n2*n2, nEl = c->nEl, mp12 = c->es.mp12;
    int rc;
    if ((rc = c->ensure_col((long long)nEl*(nk + 1)*(26LL*nn + 8LL*n2 + 4LL*mp12) + colop_ws_doubles(c)))) return rc;
    WS w{c, c->d_col, 0, c->col_doubles};
    double* cq = w.take((long long)nEl*(nk + 1)*2*mp12);
    double* tmpM = w.take((long long)nEl*(nk + 1)*2*nn*2);
    auto ba = [&](int ns) { BA b; b.ns = ns; b.p = w.take((long long)nEl*ns*nn); return b; };
    S.B = ba(nk); S.Binv = ba(nk); S.Ainv = ba(nm); S.T = ba(nm); S.Rr = ba(nm); S.X = ba(nm);
    S.Npi = ba(nk); S.Nrho = ba(nk); S.M1 = ba(nm); S.G = ba(nk);
    S.R2.ns = 2*nk; S.R2.p = w.take((long long)nEl*2*nk*nn);   // RHODPI stored [k][2]
    S.C2.ns = 2*nk; S.C2.p = w.take((long long)nEl*2*nk*nn);   // CONLIN_W stored [k][2]
    BA R2 = S.R2, C2 = S.C2;
    S.DIVl = ba(nk); S.DIVu = ba(nk); S.Gl = ba(nm); S.Gu = ba(nm);
    S.L.ns = 3*nk; S.L.p = w.take((long long)nEl*3*nk*nn);
    S.gpi = w.take((long long)nEl*nm*n2); S.geta = w.take((long long)nEl*nm*n2);
    S.rlump = w.take((long long)nEl*nm*n2); S.tA = w.take((long long)nEl*nk*n2); S.tB = w.take((long long)nEl*nk*n2);
    if (w.used > w.cap) return MIMSEM_ERR_STATE;

    // VB, VB_inv, VA_inv   (VertSolve.cpp:690-692)
    if ((rc = colop_blocks_into(c, MIMSEM_V_CONST, 0, nullptr, nullptr, S.B.p, cq, tmpM))) return rc;
    if ((rc = colop_blocks_into(c, MIMSEM_V_CONST_INV, 0, nullptr, nullptr, S.Binv.p, cq, tmpM))) return rc;
    if ((rc = colop_blocks_into(c, MIMSEM_V_LINEAR_INV, 0, nullptr, nullptr, S.Ainv.p, cq, tmpM))) return rc;
    // grad: g_i = Ainv_i (B_{i+1} f_{i+1} - B_i f_i)       (:694-695, :700, :729)
    auto grad = [&](const double* fld, double* out) -> int {
        int r;
        if ((r = bmv(c, nk, S.tA, nk, S.B, 0, fld, nk, 0, 1.0, 0))) return r;            // tA_k = B_k f_k
        double* tA = S.tA; double* tB = S.tB;
        if ((r = each(c, (long long)nEl*nm*n2, [=] __device__(long long i) {
            const int a = (int)(i%n2); long long t = i/n2; const int ii = (int)(t%nm), e = (int)(t/nm);
            tB[((size_t)e*nk + ii)*n2 + a] = tA[((size_t)e*nk + ii + 1)*n2 + a] - tA[((size_t)e*nk + ii)*n2 + a];
        }))) return r;
        return bmv(c, nm, out, nm, S.Ainv, 0, tB, nk, 0, 1.0, 0);
    };
    if ((rc = grad(pi, S.gpi))) return rc;
    // VBA(theta, grad pi) -> R2 [k][0]=(k,k-1) [k][1]=(k,k)        (:701)
    if ((rc = colop_blocks_into(c, MIMSEM_V_CONLIN_RHODPI, 0, theta, S.gpi, R2.p, cq, tmpM))) return rc;
    // VA(theta), VA(rho) with do_internal                           (:709, :716)
    if ((rc = colop_blocks_into(c, MIMSEM_V_LINEAR_RT, MIMSEM_FLAG_VERT, theta, nullptr, S.T.p, cq, tmpM))) return rc;
    if ((rc = colop_blocks_into(c, MIMSEM_V_LINEAR_RT, MIMSEM_FLAG_VERT, rho, nullptr, S.Rr.p, cq, tmpM))) return rc;
    if ((rc = bmm(c, nm, S.X, S.Ainv, 0, S.Rr, 0, 1.0, 0))) return rc;                    // X = VA_inv VA(rho)  (:717)
    // entropy gradient and A_eta = 0.5dt CONLIN_W                   (:729-731)
    if ((rc = grad(eta, S.geta))) return rc;
    if ((rc = colop_blocks_into(c, MIMSEM_V_CONLIN_W, 0, S.geta, nullptr, C2.p, cq, tmpM))) return rc;
    // EOS blocks                                                     (:736, :739)
    if ((rc = colop_blocks_into(c, MIMSEM_V_EOS_BLOCK, 0, pi, nullptr, S.Npi.p, cq, tmpM))) return rc;
    if ((rc = colop_blocks_into(c, MIMSEM_V_EOS_BLOCK, 0, rho, nullptr, S.Nrho.p, cq, tmpM))) return rc;
    return MIMSEM_OK;
}

}  // namespace

extern "C" {

int mimsem_column_diag_theta(mimsem_ctx* c, int which, const double* rho, const double* rt, double* theta) {
    if (!c || !rho || !rt || !theta || which < 0 || which > 1) return MIMSEM_ERR_ARG;
    const int nk = c->nk, n2 = c->es.n2e, nn = n2*n2, nEl = c->nEl;
    int rc = c->ensure_col(colop_ws_doubles(c) + (long long)nEl*(nk + 1)*(3LL*nn + 2LL*n2));
    if (rc) return rc;
    const long long nbmax = (long long)nEl*(nk + 1)*2;
    double* cq = c->d_col; double* tmpM = cq + nbmax*c->es.mp12; double* M = tmpM + 2*nbmax*nn;
    double* frt = M + (long long)nEl*(nk + 1)*2*nn;
    if (which == 0) {   // diagTheta_L2: VB(rho) theta = VB rt, block diagonal => exact block solves (VertSolve.cpp:339-349)
        if ((rc = colop_blocks_into(c, MIMSEM_V_CONST, 0, nullptr, nullptr, M, cq, tmpM))) return rc;
        if ((rc = stored_apply(c, MIMSEM_V_CONST, 0, M, rt, frt))) return rc;
        if ((rc = colop_blocks_into(c, MIMSEM_V_CONST_RHO_INV, 0, rho, nullptr, M, cq, tmpM))) return rc;
        return stored_apply(c, MIMSEM_V_CONST_RHO_INV, 0, M, frt, theta);
    }
    // diagTheta2: VA2(rho) theta = VAB2 rt on nk+1 interfaces; VA2 is block diagonal (lumped) (VertSolve.cpp:306-315)
    if ((rc = colop_blocks_into(c, MIMSEM_V_LINCON2, 0, nullptr, nullptr, M, cq, tmpM))) return rc;
    if ((rc = stored_apply(c, MIMSEM_V_LINCON2, 0, M, rt, frt))) return rc;
    if ((rc = colop_blocks_into(c, MIMSEM_V_LINEAR_RHO2, 0, rho, nullptr, M, cq, tmpM))) return rc;
    if ((rc = mimsem_block_inverse_inplace(c, (long long)nEl*(nk + 1), n2, M))) return rc;
    return stored_apply(c, MIMSEM_V_LINEAR_RHO2, 0, M, frt, theta);
}

}  // extern "C"
namespace {
int schur_operator(mimsem_ctx* c, double dt, const double* theta, const double* rho, const double* eta,
                          const double* pi, Schur& S) {
    const int nk = c->nk, nm = nk - 1, n2 = c->es.n2e, nn = n2*n2, nEl = c->nEl;
    const double hdt = 0.5*dt, gam = RD/CV;
    int rc;
    if ((rc = schur_assemble(c, dt, theta, rho, eta, pi, S))) return rc;
    BA R2 = S.R2, C2 = S.C2;
    // M1_i = 0.5dt (R^t_i Binv_i + R^b_{i+1} Binv_{i+1})   = rows of G_rt VB_inv   (:702-703, :742)
    //   R^t_i = R2[2i+1], R^b_{i+1} = R2[2(i+1)+0]; handled by an explicit kernel (strided slots)
    {
        const double* R = R2.p; const double* Bi = S.Binv.p; double* M1 = S.M1.p;
        if ((rc = each(c, (long long)nEl*nm*nn, [=] __device__(long long x) {
            const int ij = (int)(x%nn), ii = ij/n2, jj = ij%n2; long long t = x/nn;
            const int i = (int)(t%nm), e = (int)(t/nm);
            const double* Rt = R + ((size_t)e*2*nk + 2*i + 1)*nn;
            const double* Rb = R + ((size_t)e*2*nk + 2*(i + 1))*nn;
            const double* B0 = Bi + ((size_t)e*nk + i)*nn;
            const double* B1 = Bi + ((size_t)e*nk + i + 1)*nn;
            double s = 0.0;
            for (int k = 0; k < n2; k++) s += Rt[ii*n2 + k]*B0[k*n2 + jj] + Rb[ii*n2 + k]*B1[k*n2 + jj];
            M1[x] = hdt*s;
        }))) return rc;
    }
    // G_rt VB_inv as two bidiagonal pieces is needed again for the residual: keep GVl_i = 0.5dt R^t_i Binv_i,
    // GVu_i = 0.5dt R^b_{i+1} Binv_{i+1} implicitly (recomputed in the residual kernel from R2, Binv).
    // lumped inverse of L_eta = VA - (G_rt VB_inv) A_eta : only its scalar diagonal (:744-751)
    {
        // Alin diagonal entries: computed on the fly from the LINEAR coefficient
        const CG g = make_cg(c);
        const double* M1 = S.M1.p; const double* Cw = C2.p; double* rl = S.rlump;
        if ((rc = each(c, (long long)nEl*nm*n2, [=] __device__(long long x) {
            const int a = (int)(x%n2); long long t = x/n2;
            const int i = (int)(t%nm), e = (int)(t/nm);
            // VA(i)[a][a]
            double va = 0.0;
            for (int q = 0; q < g.mp12; q++) {
                const double cqv = colop_coef(g, MIMSEM_V_LINEAR, 0, e, i, 0, q, nullptr, nullptr);
                const double wq = g_W(g, q, a);
                va += (wq*cqv)*wq;
            }
            // (GV A_eta)[i][i] = M1_i . (0.5dt C_i),  C_i = CONLIN_W block of column i = stored (k=i, w=1)
            const double* m = M1 + ((size_t)e*nm + i)*nn;
            const double* Ci = Cw + ((size_t)e*2*nk + 2*i + 1)*nn;
            double s = 0.0;
            for (int k = 0; k < n2; k++) s += m[a*n2 + k]*(hdt*Ci[k*n2 + a]);
            rl[x] = 1.0/(-1.0*s + va);
        }))) return rc;
    }
    // DIV (N x Nm), row k:  DIVl_k = (k,k-1),  DIVu_k = (k,k)                           (:754-761)
    //   = 0.5dt ( -+ N_rho_k X_j ) + 0.5dt C_j , then column-scaled by rlump_j
    {
        const double *Nr = S.Nrho.p, *Bi = S.Binv.p, *B = S.B.p, *X = S.X.p, *Cw = C2.p, *rl = S.rlump;
        double *Dl = S.DIVl.p, *Du = S.DIVu.p;
        if ((rc = each(c, (long long)nEl*nk*2*nn, [=] __device__(long long x) {
            const int ij = (int)(x%nn), ii = ij/n2, jj = ij%n2; long long t = x/nn;
            const int w = (int)(t%2); t /= 2; const int k = (int)(t%nk), e = (int)(t/nk);
            const int j = k - 1 + w;
            double* out = (w ? Du : Dl) + ((size_t)e*nk + k)*nn + ij;
            if (j < 0 || j > nm - 1) { *out = 0.0; return; }
            // CM_k = N_rho_k Binv_k ; D_rho(k,j) = +-0.5dt B_k X_j  => CM_k D_rho = +-0.5dt N_rho_k Binv_k B_k X_j
            const double* nr = Nr + ((size_t)e*nk + k)*nn; const double* bi = Bi + ((size_t)e*nk + k)*nn;
            const double* bk = B + ((size_t)e*nk + k)*nn;  const double* xj = X + ((size_t)e*nm + j)*nn;
            double s = 0.0;
            for (int p = 0; p < n2; p++) {            // (N_rho Binv)[ii][p]
                double cm = 0.0;
                for (int l = 0; l < n2; l++) cm += nr[ii*n2 + l]*bi[l*n2 + p];
                double bx = 0.0;                       // (B_k X_j)[p][jj]
                for (int l = 0; l < n2; l++) bx += bk[p*n2 + l]*xj[l*n2 + jj];
                s += cm*bx;
            }
            s *= (w ? +hdt : -hdt);
            s += hdt*Cw[((size_t)e*2*nk + 2*k + w)*nn + ij];
            *out = s*rl[((size_t)e*nm + j)*n2 + jj];
        }))) return rc;
    }
    // G_pi (Nm x N), row i: Gl_i = (i,i) = -0.5dt T_i Ainv_i B_i ; Gu_i = (i,i+1) = +0.5dt T_i Ainv_i B_{i+1}  (:710-711)
    {
        const double *T = S.T.p, *Ai = S.Ainv.p, *B = S.B.p; double *Gl = S.Gl.p, *Gu = S.Gu.p;
        if ((rc = each(c, (long long)nEl*nm*2*nn, [=] __device__(long long x) {
            const int ij = (int)(x%nn), ii = ij/n2, jj = ij%n2; long long t = x/nn;
            const int w = (int)(t%2); t /= 2; const int i = (int)(t%nm), e = (int)(t/nm);
            const double* tt = T + ((size_t)e*nm + i)*nn; const double* ai = Ai + ((size_t)e*nm + i)*nn;
            const double* bk = B + ((size_t)e*nk + i + w)*nn;
            double s = 0.0;
            for (int p = 0; p < n2; p++) {
                double ab = 0.0;                        // (Ainv_i B)[p][jj]
                for (int l = 0; l < n2; l++) ab += ai[p*n2 + l]*bk[l*n2 + jj];
                s += tt[ii*n2 + p]*ab;
            }
            ((w ? Gu : Gl) + ((size_t)e*nm + i)*nn)[ij] = (w ? +hdt : -hdt)*s;
        }))) return rc;
    }
    // L_pi = N_pi - gam DIV G_pi : block tridiagonal [k][3]                                (:766-767)
    {
        const double *Dl = S.DIVl.p, *Du = S.DIVu.p, *Gl = S.Gl.p, *Gu = S.Gu.p, *Np = S.Npi.p; double* L = S.L.p;
        if ((rc = each(c, (long long)nEl*nk*3*nn, [=] __device__(long long x) {
            const int ij = (int)(x%nn), ii = ij/n2, jj = ij%n2; long long t = x/nn;
            const int w = (int)(t%3); t /= 3; const int k = (int)(t%nk), e = (int)(t/nk);
            const double* dl = Dl + ((size_t)e*nk + k)*nn; const double* du = Du + ((size_t)e*nk + k)*nn;
            double s = 0.0;
            if (w == 0) {            // (k,k-1): DIV(k,k-1) G_pi(k-1,k-1)
                if (k > 0) { const double* g = Gl + ((size_t)e*nm + k - 1)*nn;
                             for (int p = 0; p < n2; p++) s += dl[ii*n2 + p]*g[p*n2 + jj]; }
                s = (-1.0*gam)*s;
            } else if (w == 2) {     // (k,k+1): DIV(k,k) G_pi(k,k+1)
                if (k < nk - 1) { const double* g = Gu + ((size_t)e*nm + k)*nn;
                                  for (int p = 0; p < n2; p++) s += du[ii*n2 + p]*g[p*n2 + jj]; }
                s = (-1.0*gam)*s;
            } else {                 // (k,k): DIV(k,k-1) G_pi(k-1,k) + DIV(k,k) G_pi(k,k)
                if (k > 0) { const double* g = Gu + ((size_t)e*nm + k - 1)*nn;
                             for (int p = 0; p < n2; p++) s += dl[ii*n2 + p]*g[p*n2 + jj]; }
                if (k < nk - 1) { const double* g = Gl + ((size_t)e*nm + k)*nn;
                                  for (int p = 0; p < n2; p++) s += du[ii*n2 + p]*g[p*n2 + jj]; }
                s = (-1.0*gam)*s + Np[((size_t)e*nk + k)*nn + ij];
            }
            L[x] = s;
        }))) return rc;
    }
    return MIMSEM_OK;
}

}  // namespace
extern "C" {
int mimsem_column_helmholtz_blocks(mimsem_ctx* c, double dt, const double* theta, const double* rho,
                                   const double* eta, const double* pi, double* out) {
    if (!c || !theta || !rho || !eta || !pi || !out || c->nk < 2) return MIMSEM_ERR_ARG;
    Schur S;
    int rc = schur_operator(c, dt, theta, rho, eta, pi, S);
    if (rc) return rc;
    MIMSEM_HIP_TRY(hipMemcpyAsync(out, S.L.p, (size_t)c->nEl*c->nk*3*c->es.n2e*c->es.n2e*sizeof(double),
                                  hipMemcpyDeviceToDevice, c->stream));
    return MIMSEM_OK;
}

int mimsem_column_solve_schur_eta(mimsem_ctx* c, double dt,
        const double* theta, const double* rho, const double* eta, const double* pi,
        double* F_u, double* F_rho, double* F_eta, double* F_pi,
        double* d_u, double* d_rho, double* d_eta, double* d_pi) {
    if (!c || !theta || !rho || !eta || !pi || !F_u || !F_rho || !F_eta || !F_pi || !d_u || !d_rho || !d_eta || !d_pi)
        return MIMSEM_ERR_ARG;
    if (c->nk < 2) return MIMSEM_ERR_ARG;
    const int nk = c->nk, nm = nk - 1, n2 = c->es.n2e, nn = n2*n2, nEl = c->nEl;
    const double hdt = 0.5*dt, gam = RD/CV;
    Schur S;
    int rc = schur_operator(c, dt, theta, rho, eta, pi, S);
    if (rc) return rc;
    BA R2 = S.R2, C2 = S.C2;
    // F_u -= (G_rt VB_inv) F_eta                                                         (:772-773)
    {
        const double *R = R2.p, *Bi = S.Binv.p; double* tA = S.tA;
        if ((rc = bmv(c, nk, tA, nk, S.Binv, 0, F_eta, nk, 0, 1.0, 0))) return rc;        // tA_k = Binv_k F_eta_k
        if ((rc = each(c, (long long)nEl*nm*n2, [=] __device__(long long x) {
            const int a = (int)(x%n2); long long t = x/n2; const int i = (int)(t%nm), e = (int)(t/nm);
            const double* Rt = R + ((size_t)e*2*nk + 2*i + 1)*nn; const double* Rb = R + ((size_t)e*2*nk + 2*(i + 1))*nn;
            const double* v0 = tA + ((size_t)e*nk + i)*n2; const double* v1 = tA + ((size_t)e*nk + i + 1)*n2;
            double s = 0.0;
            for (int k = 0; k < n2; k++) s += Rt[a*n2 + k]*v0[k] + Rb[a*n2 + k]*v1[k];
            F_u[x] += -1.0*(hdt*s);
        }))) return rc;
        (void)Bi;
    }
    // F_pi = -F_pi + gam DIV F_u - gam CM F_rho - gam F_eta                               (:775-780)
    {
        const double *Dl = S.DIVl.p, *Du = S.DIVu.p, *Nr = S.Nrho.p; double* tA = S.tA;
        if ((rc = bmv(c, nk, tA, nk, S.Binv, 0, F_rho, nk, 0, 1.0, 0))) return rc;        // Binv F_rho
        if ((rc = each(c, (long long)nEl*nk*n2, [=] __device__(long long x) {
            const int a = (int)(x%n2); long long t = x/n2; const int k = (int)(t%nk), e = (int)(t/nk);
            double div = 0.0;
            if (k > 0) { const double* m = Dl + ((size_t)e*nk + k)*nn; const double* v = F_u + ((size_t)e*nm + k - 1)*n2;
                         for (int p = 0; p < n2; p++) div += m[a*n2 + p]*v[p]; }
            if (k < nk - 1) { const double* m = Du + ((size_t)e*nk + k)*nn; const double* v = F_u + ((size_t)e*nm + k)*n2;
                              for (int p = 0; p < n2; p++) div += m[a*n2 + p]*v[p]; }
            double cm = 0.0;
            { const double* m = Nr + ((size_t)e*nk + k)*nn; const double* v = tA + ((size_t)e*nk + k)*n2;
              for (int p = 0; p < n2; p++) cm += m[a*n2 + p]*v[p]; }
            double f = -1.0*F_pi[x];
            f += (+1.0*gam)*div;
            f += (-1.0*gam)*cm;
            f += (-1.0*gam)*F_eta[x];
            F_pi[x] = f;
        }))) return rc;
    }
    // Helmholtz solve                                                                     (:783-789)
    if ((rc = block_thomas(c, S.L.p, F_pi, d_pi, S.G.p, S.tB))) return rc;
    // back substitution                                                                   (:792-815)
    {
        const double *Gl = S.Gl.p, *Gu = S.Gu.p, *rl = S.rlump;
        if ((rc = each(c, (long long)nEl*nm*n2, [=] __device__(long long x) {
            const int a = (int)(x%n2); long long t = x/n2; const int i = (int)(t%nm), e = (int)(t/nm);
            const double* gl = Gl + ((size_t)e*nm + i)*nn; const double* gu = Gu + ((size_t)e*nm + i)*nn;
            const double* p0 = d_pi + ((size_t)e*nk + i)*n2; const double* p1 = d_pi + ((size_t)e*nk + i + 1)*n2;
            double s = 0.0;
            for (int k = 0; k < n2; k++) s += gl[a*n2 + k]*p0[k] + gu[a*n2 + k]*p1[k];
            double f = F_u[x] + s;
            f *= -1.0;
            F_u[x] = f;
            d_u[x] = rl[x]*f;
        }))) return rc;
    }
    {
        const double *Cw = C2.p, *X = S.X.p; double* tA = S.tA;
        // F_eta += A_eta d_u ; F_eta = -F_eta ; d_eta = Binv F_eta
        if ((rc = each(c, (long long)nEl*nk*n2, [=] __device__(long long x) {
            const int a = (int)(x%n2); long long t = x/n2; const int k = (int)(t%nk), e = (int)(t/nk);
            double s = 0.0;
            for (int w = 0; w < 2; w++) {
                const int j = k - 1 + w;
                if (j < 0 || j > nm - 1) continue;
                const double* m = Cw + ((size_t)e*2*nk + 2*k + w)*nn; const double* v = d_u + ((size_t)e*nm + j)*n2;
                for (int p = 0; p < n2; p++) s += (hdt*m[a*n2 + p])*v[p];
            }
            double f = F_eta[x] + s;
            f *= -1.0;
            F_eta[x] = f;
        }))) return rc;
        if ((rc = bmv(c, nk, d_eta, nk, S.Binv, 0, F_eta, nk, 0, 1.0, 0))) return rc;
        // F_rho += D_rho d_u,  D_rho = 0.5dt VB (V10 X)   (:812-815)
        if ((rc = each(c, (long long)nEl*nm*n2, [=] __device__(long long x) {       // tA_j = X_j d_u_j  (interfaces)
            const int a = (int)(x%n2); long long t = x/n2; const int j = (int)(t%nm), e = (int)(t/nm);
            const double* m = X + ((size_t)e*nm + j)*nn; const double* v = d_u + ((size_t)e*nm + j)*n2;
            double s = 0.0;
            for (int p = 0; p < n2; p++) s += m[a*n2 + p]*v[p];
            tA[((size_t)e*nk + j)*n2 + a] = s;
        }))) return rc;
        const double* B = S.B.p;
        if ((rc = each(c, (long long)nEl*nk*n2, [=] __device__(long long x) {
            const int a = (int)(x%n2); long long t = x/n2; const int k = (int)(t%nk), e = (int)(t/nk);
            const double* m = B + ((size_t)e*nk + k)*nn;
            double s = 0.0;
            for (int p = 0; p < n2; p++) {
                double dx = 0.0;                        // (V10 X d_u)_k = X_k du_k - X_{k-1} du_{k-1}
                if (k < nk - 1) dx += tA[((size_t)e*nk + k)*n2 + p];
                if (k > 0)      dx -= tA[((size_t)e*nk + k - 1)*n2 + p];
                s += m[a*n2 + p]*dx;
            }
            double f = F_rho[x] + hdt*s;
            f *= -1.0;
            F_rho[x] = f;
        }))) return rc;
        if ((rc = bmv(c, nk, d_rho, nk, S.Binv, 0, F_rho, nk, 0, 1.0, 0))) return rc;
    }
    return MIMSEM_OK;
}

}  // extern "C"
