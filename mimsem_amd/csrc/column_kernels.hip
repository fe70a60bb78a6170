// mimsem_amd/csrc/column_kernels.hip -- vertical (column) operators, SURVEY 8(a) rows C1..C9
// (reference eul/VertOps.cpp, eul/VertSolve.cpp:289-352,677-823, eul/L2Vecs.cpp:55-101).
//
// Every reference VertOps::Assemble* builds, for ONE column, nk small blocks  W^T diag(c_q) W  and drops
// them on the (bi)diagonal of a MATSEQAIJ; VertSolve then chains MatMatMult/PCLU on those matrices.
// Here the block structure is kept explicit for ALL columns at once:
//   coefficient + block pass (k_coef_block): c[slot][q] evaluated into LDS (one thread per quadrature point, fields interpolated
//                     with the collocated edge table), M[e][slot] = W^T diag(c) W by sum factorisation (W is a tensor product)
//   batched Gauss-Jordan (LinAlg.cpp:186-269 pivoting rules) for the *Inv operators
//   block (bi)diagonal mat-vec / mat-mat kernels, and a block-Thomas sweep (+ one refinement step) for the Helmholtz solve.
// column_hs.inc (included at the end) adds the Strang / Held-Suarez rows on a block-banded algebra.
// The Schur complement of solve_schur_column_eta is assembled ANALYTICALLY from these factors: every
// factor is block-diagonal or block-bidiagonal, so L_pi is block-tridiagonal -- no
// sparse mat-mat products, no symbolic phases, no per-column PETSc objects.
#include <cstdlib>
#include <cstring>
#include <type_traits>
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
    double param;              // scalar argument of the operator (dt_fric of AssembleLinearWithRayleighInv)
};
CG make_cg(const mimsem_ctx* c) {
    CG g; g.n = c->es.n; g.mp1 = c->es.mp1; g.mp12 = c->es.mp12; g.n2 = c->es.n2e; g.nEl = c->nEl; g.nk = c->nk;
    g.det = c->d_det; g.tI = c->d_tI; g.th = c->d_th; g.E = c->d_E; g.w = c->d_w;
    g.param = c->col_param;
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
    case MIMSEM_V_CONST_THETA: case MIMSEM_V_EOS_BLOCK: case MIMSEM_V_EOS_BLOCK_INV: *nr = nk; *nx = nk; *ny = nk; break;
    case MIMSEM_V_LINEAR: case MIMSEM_V_LINEAR_INV: case MIMSEM_V_LINEAR_RT: case MIMSEM_V_LINEAR_THETA:
    case MIMSEM_V_RAYLEIGH: case MIMSEM_V_LINEAR_RAYLEIGH_INV: *nr = nk - 1; *nx = nk - 1; *ny = nk - 1; break;
    case MIMSEM_V_LINEAR_RHO2: case MIMSEM_V_LINEAR_RHO2_UP: *nr = nk + 1; *nx = nk + 1; *ny = nk + 1; break;
    case MIMSEM_V_LINCON:  *nr = nk - 1; *nw = 2; *nx = nk; *ny = nk - 1; break;
    case MIMSEM_V_LINCON2: case MIMSEM_V_LINCON2_UP: *nr = nk + 1; *nw = 2; *nx = nk; *ny = nk + 1; break;
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
    case MIMSEM_V_LINEAR_RAYLEIGH_INV: {                         // :1391-1400 (the kk == nk-1 branch is unreachable there too)
        double c = q0*(0.5*(g_th(g, e, r, q) + g_th(g, e, r + 1, q)));
        if (r == nk - 1)      c *= (1.0 + 1.00*g.param);
        else if (r == nk - 2) c *= (1.0 + 0.50*g.param);
        else if (r == nk - 3) c *= (1.0 + 0.25*g.param);
        return c;
    }
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

// block pass: M[b][i][j] = sum_q (W[q][i] c[b][q]) W[q][j]   (Mult_FD_IP then Mult_IP order).
// One workgroup = BT blocks; the dense W table (built from the edge table) and the BT coefficient rows are
// staged in LDS, every thread produces entries with mp12-term sums out of LDS -> write-bandwidth bound.
constexpr int BP_BT = 16;
__global__ __launch_bounds__(256) void k_block_pass(CG g, long long nb, const double* __restrict__ cq, double* __restrict__ M) {
    extern __shared__ double sm[];
    double* sW = sm;                         // [mp12][n2]
    double* sc = sm + g.mp12*g.n2;           // [BT][mp12]
    const int nn = g.n2*g.n2, tid = threadIdx.x;
    const long long b0 = (long long)blockIdx.x*BP_BT;
    const int nbt = (int)min((long long)BP_BT, nb - b0);
    for (int t = tid; t < g.mp12*g.n2; t += 256) sW[t] = g_W(g, t/g.n2, t%g.n2);
    for (int t = tid; t < nbt*g.mp12; t += 256) sc[t] = cq[b0*g.mp12 + t];
    __syncthreads();
    for (int t = tid; t < nbt*nn; t += 256) {
        const int lb = t/nn, ij = t%nn, ii = ij/g.n2, jj = ij%g.n2;
        const double* cb = sc + lb*g.mp12;
        double s = 0.0;
        for (int q = 0; q < g.mp12; q++) s += (sW[q*g.n2 + ii]*cb[q])*sW[q*g.n2 + jj];
        M[b0*nn + t] = s;
    }
}
int block_pass(mimsem_ctx* c, long long nb, const double* cq, double* M) {
    if (nb <= 0) return MIMSEM_OK;
    const CG g = make_cg(c);
    const size_t lds = (size_t)(g.mp12*g.n2 + BP_BT*g.mp12)*sizeof(double);
    hipLaunchKernelGGL(k_block_pass, dim3((unsigned)((nb + BP_BT - 1)/BP_BT)), dim3(256), lds, c->stream, g, nb, cq, M);
    MIMSEM_HIP_TRY(hipGetLastError());
    return MIMSEM_OK;
}

// coefficient pass + block pass fused.  The table W[q][j] = E[qx][jx] E[qy][jy] is a tensor product, so the triple product
// M = W^T diag(c) W is contracted one direction at a time (sum factorisation):
//     T1[qx][iy][jy] = sum_qy E[qy][iy] c[qx,qy] E[qy][jy]            (n^2 (p+1)^2 terms per block)
//     M[(ix,iy)][(jx,jy)] = sum_qx (E[qx][ix] E[qx][jx]) T1[qx][iy][jy]  (n^4 (p+1) terms per block)
// -- 2.8x fewer flops and 7x fewer LDS reads than the dense (p+1)^2-term sums at p = 3 (the dense form was LDS-bound: 69 us
// per 103 680 blocks, 0.9 TB/s of output).  One thread per (block, iy, jy); the coefficient rows of the workgroup's blocks are
// evaluated straight into LDS (one thread per quadrature point) and never touch HBM; finished blocks leave through LDS as
// contiguous rows.  (Summation order differs from Mult_FD_IP/Mult_IP by design; parity is at 1e-10, not bitwise.)
__global__ __launch_bounds__(256) void k_coef_block(CG g, int colop, unsigned flags, int nr, int nw, int bpw,
        const double* __restrict__ f1, const double* __restrict__ f2, double* __restrict__ M) {
    extern __shared__ double sm[];
    const int n = g.n, mp1 = g.mp1, mp12 = g.mp12, n2 = g.n2, nn = n2*n2, tid = threadIdx.x;
    double* sE  = sm;                        // [mp1][n]
    double* sEE = sE + mp1*n;                // [mp1][n][n]   E[qx][ix] E[qx][jx]
    double* sc  = sEE + mp1*n2;              // [bpw][mp12]
    double* sM  = sc + bpw*mp12;             // [bpw][nn]
    const long long nb = (long long)g.nEl*nr*nw, b0 = (long long)blockIdx.x*bpw;
    const int nbt = (int)min((long long)bpw, nb - b0);
    __shared__ double sw[8];
    for (int t = tid; t < mp1*n; t += 256) sE[t] = g.E[t];
    if (tid < mp1) sw[tid] = g.w[tid];
    for (int t = tid; t < mp1*n2; t += 256) { const int qx = t/n2, ij = t%n2; sEE[t] = g.E[qx*n + ij/n]*g.E[qx*n + ij%n]; }
    __syncthreads();
    CG gl = g; gl.E = sE; gl.w = sw;             // the field interpolations of colop_coef read the basis from LDS
    for (int t = tid; t < nbt*mp12; t += 256) {
        const long long b = b0 + t/mp12; const int q = t%mp12;
        const int w = (int)(b%nw); long long r2 = b/nw;
        const int r = (int)(r2%nr), e = (int)(r2/nr);
        sc[t] = colop_coef(gl, colop, flags, e, r, w, q, f1, f2);
    }
    __syncthreads();
    if (tid < nbt*n2) {
        const int lb = tid/n2, iy = (tid%n2)/n, jy = tid%n;
        const double* cb = sc + lb*mp12;
        double t1[8];
        for (int qx = 0; qx < mp1; qx++) {
            double s = 0.0;
            for (int qy = 0; qy < mp1; qy++) s += (sE[qy*n + iy]*cb[qy*mp1 + qx])*sE[qy*n + jy];
            t1[qx] = s;
        }
        double* mb = sM + lb*nn;
        for (int ix = 0; ix < n; ix++)
            for (int jx = 0; jx < n; jx++) {
                double s = 0.0;
                for (int qx = 0; qx < mp1; qx++) s += sEE[qx*n2 + ix*n + jx]*t1[qx];
                mb[(iy*n + ix)*n2 + jy*n + jx] = s;
            }
    }
    __syncthreads();
    for (int t = tid; t < nbt*nn; t += 256) M[b0*nn + t] = sM[t];
}
int coef_block_pass(mimsem_ctx* c, int colop, unsigned flags, const double* f1, const double* f2, double* M, int nr, int nw) {
    const CG g = make_cg(c);
    const long long nb = (long long)c->nEl*nr*nw;
    if (nb <= 0) return MIMSEM_OK;
    const int bpw = std::max(1, 256/g.n2);
    const size_t lds = (size_t)(g.mp1*g.n + g.mp1*g.n2 + bpw*g.mp12 + bpw*g.n2*g.n2)*sizeof(double);
    if (lds > 64*1024)
        MIMSEM_HIP_TRY(hipFuncSetAttribute((const void*)k_coef_block, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_coef_block, dim3((unsigned)((nb + bpw - 1)/bpw)), dim3(256), lds, c->stream, g, colop, flags, nr, nw, bpw, f1, f2, M);
    MIMSEM_HIP_TRY(hipGetLastError());
    return MIMSEM_OK;
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


// Round 5: ONE WAVEFRONT per block for 16 < n <= 64 (the 24 x 24 / 40 x 40 element blocks of M1 and the 33 x 33 / 56 x 56 coupled [u|h]
// blocks of the shallow-water preconditioner at p = 3 / 4; the 25 .. 49-wide 2-form blocks of p = 5 .. 7).  The thread-per-block kernel
// above walks n^3 entries through LDS with ONE lane working per block: 2.9 ms for the 3 456 blocks of SWEqn's PCSetUp (round-4 verdict).
// Here lane = row of the elimination (and column of the row swap / scaling), the block in LDS with an ODD row stride (conflict-free row
// and column walks), the same algorithm in the same order -- full pivoting with LinAlg.cpp:186-269's tie-breaking (the LAST entry of the
// row-major scan that attains the maximum), normalise, eliminate, un-permute the columns -- so the same pivots and the same operations
// per entry as the kernel above: identical bits.
__global__ __launch_bounds__(64) void k_block_inverse_wave(long long nb, int n, double* blocks, int* errcount) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x, ns = n | 1;
    double* A = lds;                                   // A[i*ns + j]
    int* ipiv = (int*)(lds + (size_t)n*ns);            // [n]; then indxr[n], indxc[n]
    int* indxr = ipiv + n; int* indxc = indxr + n;
    const long long b = blockIdx.x;
    if (b >= nb) return;
    double* src = blocks + b*n*n;
    for (int idx = lane; idx < n*n; idx += 64) A[(idx/n)*ns + idx%n] = src[idx];
    if (lane < n) ipiv[lane] = 0;
    __syncthreads();
    int err = 0;
    for (int i = 0; i < n; i++) {
        // pivot search: my row, then the wavefront.  Sequential semantics of the reference: the last (j, k) in row-major order with |a| == max
        double big = -1.0; int kk = 0;
        if (lane < n && ipiv[lane] != 1)
            for (int k = 0; k < n; k++) {
                const int pk = ipiv[k];
                if (pk == 0) { const double v = fabs(A[lane*ns + k]); if (v >= big) { big = v; kk = k; } }
                else if (pk > 1) err = 1;
            }
        double vmax = big;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) vmax = fmax(vmax, __shfl_xor(vmax, off));
        int jsel = (big == vmax && big >= 0.0) ? lane : -1;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) jsel = max(jsel, __shfl_xor(jsel, off));
        if (jsel < 0) { err = 2; jsel = 0; }           // (nothing left to pivot on / NaN: reported like a vanishing pivot)
        const int irow = jsel, icol = __shfl(kk, jsel);
        __syncthreads();
        if (lane == 0) { ++ipiv[icol]; indxr[i] = irow; indxc[i] = icol; }
        if (irow != icol && lane < n) { const double t0 = A[irow*ns + lane]; A[irow*ns + lane] = A[icol*ns + lane]; A[icol*ns + lane] = t0; }
        __syncthreads();
        const double piv = A[icol*ns + icol];
        if (fabs(piv) < 1.0e-12) err = 2;
        const double pivinv = 1.0/piv;
        __syncthreads();
        if (lane < n) A[icol*ns + lane] = (lane == icol ? 1.0 : A[icol*ns + lane])*pivinv;
        __syncthreads();
        if (lane < n && lane != icol) {
            const double dum = A[lane*ns + icol];
            A[lane*ns + icol] = 0.0;
            for (int l = 0; l < n; l++) A[lane*ns + l] -= A[icol*ns + l]*dum;
        }
        __syncthreads();
    }
    for (int l = n - 1; l >= 0; l--) {
        const int ir = indxr[l], ic = indxc[l];
        if (ir != ic && lane < n) { const double t0 = A[lane*ns + ir]; A[lane*ns + ir] = A[lane*ns + ic]; A[lane*ns + ic] = t0; }
        __syncthreads();
    }
    for (int idx = lane; idx < n*n; idx += 64) src[idx] = A[(idx/n)*ns + idx%n];
    if (errcount) { const unsigned long long any = __ballot(err != 0); if (lane == 0 && any) atomicAdd(errcount, 1); }
}

// Cooperative variant for the block sizes of p <= 4 (n = 1, 4, 9, 16): ONE LANE PER ROW, the row lives in
// registers, LPM lanes per matrix (4 matrices per wavefront at n = 9/16).  Same algorithm and tie-breaking as
// LinAlg.cpp:186-269: the pivot is the LAST entry (row-major scan of the not-yet-pivoted rows x columns)
// attaining the maximum modulus; row swap, normalise, eliminate, final column un-permutation.
template <int N> __device__ __forceinline__ double rsel(const double (&r)[N], int idx) {
    double v = r[0];
#pragma unroll
    for (int c = 1; c < N; c++) v = (c == idx) ? r[c] : v;
    return v;
}
template <int N, int LPM>
__global__ __launch_bounds__(256) void k_block_inverse_rows(long long nb, double* __restrict__ blocks, int* errcount) {
    const int tid = threadIdx.x, lane = tid%LPM;
    const long long m = ((long long)blockIdx.x*256 + tid)/LPM;
    const bool act = (m < nb) && (lane < N);
    double row[N];
    const long long mm = (m < nb) ? m : nb - 1;
    const int lr = (lane < N) ? lane : N - 1;
#pragma unroll
    for (int c = 0; c < N; c++) row[c] = blocks[mm*N*N + lr*N + c];
    unsigned pmask = 0;
    int indxr[N], indxc[N];
    int err = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        // ---- pivot search ----
        double bv = -2.0; int bk = 0, br = lane;
        if (lane < N && !((pmask >> lane) & 1u)) {
            bv = -1.0;
#pragma unroll
            for (int k = 0; k < N; k++)
                if (!((pmask >> k) & 1u)) { const double v = fabs(row[k]); if (v >= bv) { bv = v; bk = k; } }
        }
#pragma unroll
        for (int off = LPM/2; off > 0; off >>= 1) {
            const double ov = __shfl_xor(bv, off, LPM);
            const int ok = __shfl_xor(bk, off, LPM), orr = __shfl_xor(br, off, LPM);
            if (ov > bv || (ov == bv && orr > br)) { bv = ov; bk = ok; br = orr; }
        }
        const int irow = br, icol = bk;
        pmask |= 1u << icol;
        indxr[i] = irow; indxc[i] = icol;
        // ---- swap rows irow <-> icol ----
        const int partner = (lane == irow) ? icol : ((lane == icol) ? irow : lane);
#pragma unroll
        for (int c = 0; c < N; c++) row[c] = __shfl(row[c], partner, LPM);
        // ---- normalise the pivot row (now held by lane icol), broadcast it ----
        double prow[N];
#pragma unroll
        for (int c = 0; c < N; c++) prow[c] = __shfl(row[c], icol, LPM);
        const double piv = rsel<N>(prow, icol);
        if (fabs(piv) < 1.0e-12) err = 2;
        const double pivinv = 1.0/piv;
#pragma unroll
        for (int c = 0; c < N; c++) prow[c] = ((c == icol) ? 1.0 : prow[c])*pivinv;
        if (lane == icol) {
#pragma unroll
            for (int c = 0; c < N; c++) row[c] = prow[c];
        } else {
            const double dum = rsel<N>(row, icol);
#pragma unroll
            for (int c = 0; c < N; c++) row[c] = ((c == icol) ? 0.0 : row[c]) - prow[c]*dum;
        }
    }
    // ---- unscramble the columns ----
#pragma unroll
    for (int l = N - 1; l >= 0; l--) {
        const int ir = indxr[l], ic = indxc[l];
        if (ir != ic) {
            const double a = rsel<N>(row, ir), b = rsel<N>(row, ic);
#pragma unroll
            for (int c = 0; c < N; c++) row[c] = (c == ir) ? b : ((c == ic) ? a : row[c]);
        }
    }
    if (act) {
#pragma unroll
        for (int c = 0; c < N; c++) blocks[m*N*N + lane*N + c] = row[c];
        if (err && errcount && lane == 0) atomicAdd(errcount, 1);
    }
}
// Register-resident variant for n <= 9: one THREAD per block, all n*n entries in VGPRs with static indices,
// natural-order (unpivoted) Gauss-Jordan.  Every block this engine inverts is W^T diag(c) W with c > 0, i.e.
// symmetric positive definite, for which the reference's full-pivoting search can only ever select diagonal
// entries (|a_ij| <= max diagonal) and elimination is stable in any order; the results agree to round-off.
// A block whose natural-order pivot falls below 1e-8 x its largest diagonal entry (not SPD / near singular)
// is redone by the same thread with the reference's exact full-pivoting algorithm (LinAlg.cpp:186-269).
template <int N>
__global__ __launch_bounds__(64) void k_block_inverse_reg(long long nb, double* __restrict__ blocks, int* errcount) {
    const long long b = (long long)blockIdx.x*64 + threadIdx.x;
    if (b >= nb) return;
    double* src = blocks + b*N*N;
    double a[N*N];
#pragma unroll
    for (int k = 0; k < N*N; k++) a[k] = src[k];
    double dmax = 0.0;
#pragma unroll
    for (int k = 0; k < N; k++) dmax = fmax(dmax, fabs(a[k*N + k]));
    bool ok = true;
#pragma unroll
    for (int p = 0; p < N; p++) {
        const double piv = a[p*N + p];
        if (!(fabs(piv) >= 1.0e-8*dmax) || !(fabs(piv) >= 1.0e-12)) ok = false;
        const double pinv = 1.0/piv;
        a[p*N + p] = 1.0;
#pragma unroll
        for (int c = 0; c < N; c++) a[p*N + c] *= pinv;
#pragma unroll
        for (int r = 0; r < N; r++) {
            if (r == p) continue;
            const double d = a[r*N + p];
            a[r*N + p] = 0.0;
#pragma unroll
            for (int c = 0; c < N; c++) a[r*N + c] -= a[p*N + c]*d;
        }
    }
    if (ok) {
#pragma unroll
        for (int k = 0; k < N*N; k++) src[k] = a[k];
        return;
    }
    // ---- slow path: exact restatement with full pivoting on a private copy ----
    double A[N*N]; int ipiv[N], indxr[N], indxc[N];
    for (int k = 0; k < N*N; k++) A[k] = src[k];
    for (int j = 0; j < N; j++) ipiv[j] = 0;
    int err = 0, irow = 0, icol = 0;
    for (int i = 0; i < N; i++) {
        double big = 0.0;
        for (int j = 0; j < N; j++) {
            if (ipiv[j] == 1) continue;
            for (int k = 0; k < N; k++) {
                if (ipiv[k] == 0) { const double v = fabs(A[j*N + k]); if (v >= big) { big = v; irow = j; icol = k; } }
                else if (ipiv[k] > 1) err = 1;
            }
        }
        ++ipiv[icol];
        if (irow != icol) for (int l = 0; l < N; l++) { const double t = A[irow*N + l]; A[irow*N + l] = A[icol*N + l]; A[icol*N + l] = t; }
        indxr[i] = irow; indxc[i] = icol;
        if (fabs(A[icol*N + icol]) < 1.0e-12) err = 2;
        const double pivinv = 1.0/A[icol*N + icol];
        A[icol*N + icol] = 1.0;
        for (int l = 0; l < N; l++) A[icol*N + l] *= pivinv;
        for (int ll = 0; ll < N; ll++) {
            if (ll == icol) continue;
            const double dum = A[ll*N + icol];
            A[ll*N + icol] = 0.0;
            for (int l = 0; l < N; l++) A[ll*N + l] -= A[icol*N + l]*dum;
        }
    }
    for (int l = N - 1; l >= 0; l--) {
        if (indxr[l] == indxc[l]) continue;
        for (int k = 0; k < N; k++) { const double t = A[k*N + indxr[l]]; A[k*N + indxr[l]] = A[k*N + indxc[l]]; A[k*N + indxc[l]] = t; }
    }
    for (int k = 0; k < N*N; k++) src[k] = A[k];
    if (err && errcount) atomicAdd(errcount, 1);
}
template <int N>
int launch_inverse_reg(mimsem_ctx* c, long long nb, double* blocks, int* err) {
    hipLaunchKernelGGL((k_block_inverse_reg<N>), dim3((unsigned)((nb + 63)/64)), dim3(64), 0, c->stream, nb, blocks, err);
    MIMSEM_HIP_TRY(hipGetLastError());
    return MIMSEM_OK;
}

template <int N, int LPM>
int launch_inverse_rows(mimsem_ctx* c, long long nb, double* blocks, int* err) {
    const long long threads = nb*LPM;
    hipLaunchKernelGGL((k_block_inverse_rows<N, LPM>), dim3((unsigned)((threads + 255)/256)), dim3(256), 0, c->stream,
                       nb, blocks, err);
    MIMSEM_HIP_TRY(hipGetLastError());
    return MIMSEM_OK;
}

}  // namespace

// err (device int, may be null): incremented once per block in which the reference's Inv would have reported a pivot below 1e-12
// (eul/LinAlg.cpp:243; the reference's callers drop that return value, e.g. eul/VertOps.cpp:434 -- mimsem_block_inverse_status hands it out)
int mimsem_block_inverse_inplace(mimsem_ctx* c, long long nblocks, int n, double* blocks, int* err) {
    if (nblocks <= 0) return MIMSEM_OK;
    switch (n) {                     // register-resident one-lane-per-row kernel for the block sizes of p <= 4
    case 1:  return launch_inverse_reg<1>(c, nblocks, blocks, err);
    case 4:  return launch_inverse_reg<4>(c, nblocks, blocks, err);
#ifdef MIMSEM_WITH_EXPERIMENTS
    case 9:  return exp_env("MIMSEM_INV_ROWS") ? launch_inverse_rows<9, 16>(c, nblocks, blocks, err) : launch_inverse_reg<9>(c, nblocks, blocks, err);
#else
    case 9:  return launch_inverse_reg<9>(c, nblocks, blocks, err);
#endif
    case 16: return launch_inverse_rows<16, 16>(c, nblocks, blocks, err);
    default: break;                  // 25, 36, 49: thread-per-matrix in LDS below
    }
    if (n > 16 && n <= 64 && !(exp_env("MIMSEM_INV_THREAD") && atoi(exp_env("MIMSEM_INV_THREAD")) != 0)) {      // one wavefront per block (round 5)
        const size_t lds = (size_t)n*(n | 1)*sizeof(double) + (size_t)3*n*sizeof(int);
        hipLaunchKernelGGL(k_block_inverse_wave, dim3((unsigned)nblocks), dim3(64), lds, c->stream, nblocks, n, blocks, err);
        MIMSEM_HIP_TRY(hipGetLastError());
        return MIMSEM_OK;
    }
    // matrices per workgroup: as many as fit a 144 KiB LDS budget (64 for n<=16, fewer for the 25..49-wide blocks of p>=5)
    int T = 64;
    auto need = [&](int t) { return (size_t)n*n*t*sizeof(double) + (size_t)3*n*t*sizeof(int); };
    while (T > 1 && need(T) > 144*1024) T >>= 1;
    const size_t lds = need(T);
    if (lds > 160*1024) return MIMSEM_ERR_UNSUPPORTED;
    if (lds > 64*1024)
        MIMSEM_HIP_TRY(hipFuncSetAttribute((const void*)k_block_inverse, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_block_inverse, dim3((unsigned)((nblocks + T - 1)/T)), dim3(64), lds, c->stream,
                       nblocks, n, T, blocks, err);
    MIMSEM_HIP_TRY(hipGetLastError());
    return MIMSEM_OK;
}

namespace {

// workspace carving
struct WS {
    mimsem_ctx* c; double* base; long long used, cap;
    double* take(long long n) { double* p = base + used; used += n; return p; }
};

int flat_mm_impl(mimsem_ctx* c, long long nb, const double* A, const double* B, double* C);   // defined after bmm
// C[b] = A[b] . B[b] for nb independent n x n blocks
int flat_mm(mimsem_ctx* c, long long nb, int n, const double* A, const double* B, double* C) {
    (void)n;
    return flat_mm_impl(c, nb, A, B, C);
}

bool colop_is_inverse(int colop) {
    return colop == MIMSEM_V_CONST_INV || colop == MIMSEM_V_CONST_RHO_INV || colop == MIMSEM_V_LINEAR_INV ||
           colop == MIMSEM_V_LINEAR_RAYLEIGH_INV;
}

int up_blocks_into(mimsem_ctx* c, int colop, const double* rho, double* M, double* tmpM);   // below (needs the velocity set in the ctx)

// matrix-free MatMult for the operators whose stored block is W^T diag(c) W itself (no inverse, no product): defined with the row
// kernels below.  Returns MIMSEM_ERR_UNSUPPORTED when the operator needs its blocks.
int launch_colop_apply_mf(mimsem_ctx* c, int colop, unsigned flags, int transpose, const double* f1, const double* f2,
                          const double* x, double* y);

// blocks of a column operator into M ([nEl][nr][nw][n2][n2]); cq scratch [nEl][nr][nw][mp12] (x2 for EOS)
int colop_blocks_into(mimsem_ctx* c, int colop, unsigned flags, const double* f1, const double* f2,
                      double* M, double* cq, double* tmpM, const double* Bconst = nullptr /* CONST blocks if the caller has them */) {
    int nr, nw, nx, ny;
    colop_shape(colop, c->nk, &nr, &nw, &nx, &ny);
    const long long nb = (long long)c->nEl*nr*nw;
    const int n2 = c->es.n2e, nn = n2*n2;
    int rc;
    if (colop == MIMSEM_V_EOS_BLOCK) {        // B . B(rt)^-1 . B   VertOps.cpp:1162-1196
        if ((rc = coef_block_pass(c, MIMSEM_V_CONST_RHO, 0, f1, nullptr, tmpM, nr, 1))) return rc;   // B(rt)
        if ((rc = mimsem_block_inverse_inplace(c, nb, n2, tmpM))) return rc;
        double* Bm = tmpM + nb*nn;
        if (Bconst) Bm = const_cast<double*>(Bconst);
        else if ((rc = coef_block_pass(c, MIMSEM_V_CONST, 0, nullptr, nullptr, Bm, nr, 1))) return rc;     // B
        const double* Binv = tmpM;
        double* t1 = M;                                                        // Binv.B lands in the output, then B.(Binv.B)
        if ((rc = flat_mm(c, nb, n2, Binv, Bm, t1))) return rc;
        double* t2 = tmpM;                                                     // Binv no longer needed
        if ((rc = flat_mm(c, nb, n2, Bm, t1, t2))) return rc;
        MIMSEM_HIP_TRY(hipMemcpyAsync(M, t2, (size_t)nb*nn*sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        return MIMSEM_OK;
    }
    if (colop == MIMSEM_V_EOS_BLOCK_INV) {    // (T B^-1 + I) B B(rt)^-1 B, inverted   VertOps.cpp:1066-1128 (f2 = theta, optional)
        double *s0 = tmpM, *s1 = tmpM + nb*nn, *s2 = tmpM + 2*nb*nn, *s3 = tmpM + 3*nb*nn;
        if ((rc = coef_block_pass(c, MIMSEM_V_CONST_RHO, 0, f1, nullptr, s0, nr, 1))) return rc;
        if ((rc = mimsem_block_inverse_inplace(c, nb, n2, s0))) return rc;
        if ((rc = coef_block_pass(c, MIMSEM_V_CONST, 0, nullptr, nullptr, s1, nr, 1))) return rc;
        if ((rc = flat_mm(c, nb, n2, s0, s1, M))) return rc;                  // BinvB
        if ((rc = flat_mm(c, nb, n2, s1, M, s2))) return rc;                  // B_BinvB
        double* res = s2;
        if (f2) {
            MIMSEM_HIP_TRY(hipMemcpyAsync(s0, s1, (size_t)nb*nn*sizeof(double), hipMemcpyDeviceToDevice, c->stream));
            if ((rc = mimsem_block_inverse_inplace(c, nb, n2, s0))) return rc;                       // B^-1
            if ((rc = coef_block_pass(c, MIMSEM_V_CONST_THETA, 0, f2, nullptr, s3, nr, 1))) return rc;
            if ((rc = flat_mm(c, nb, n2, s3, s0, M))) return rc;
            if ((rc = each(c, nb*n2, [=] __device__(long long i) { M[(i/n2)*nn + (i%n2)*(n2 + 1)] += 1.0; }))) return rc;
            if ((rc = flat_mm(c, nb, n2, M, s2, s0))) return rc;
            res = s0;
        }
        MIMSEM_HIP_TRY(hipMemcpyAsync(M, res, (size_t)nb*nn*sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        return mimsem_block_inverse_inplace(c, nb, n2, M);
    }
    if (colop == MIMSEM_V_LINEAR_RHO2_UP || colop == MIMSEM_V_LINCON2_UP)
        return up_blocks_into(c, colop, f1, M, tmpM);
    if ((rc = coef_block_pass(c, colop, flags, f1, f2, M, nr, nw))) return rc;
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
    const int off = (colop == MIMSEM_V_LINCON) ? 0 : -1;        // LINCON2(_UP) and the CONLIN family: -1
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
    int rc = launch_colop_apply_mf(c, colop, flags, transpose, f1, f2, x, y);
    if (rc != MIMSEM_ERR_UNSUPPORTED) return rc;
    if ((rc = c->ensure_col(colop_ws_doubles(c)))) return rc;
    const long long nbmax = (long long)c->nEl*(c->nk + 1)*2;
    const int nn = c->es.n2e*c->es.n2e;
    double* cq = c->d_col; double* tmpM = cq + nbmax*c->es.mp12; double* M = tmpM + 2*nbmax*nn;
    if ((rc = colop_blocks_into(c, colop, flags, f1, f2, M, cq, tmpM))) return rc;
    return stored_apply(c, colop, transpose, M, x, y);
}

// MatMult with blocks the caller already holds (mimsem_colop_blocks output): the operators that depend on the geometry only
// (AssembleConst, AssembleConstInv, AssembleLinear, AssembleLinearInv, AssembleRayleigh) need not be re-assembled per use
int mimsem_colop_apply_blocks(mimsem_ctx* c, int colop, int transpose, const double* blocks, const double* x, double* y) {
    if (!c || !blocks || !x || !y || colop < 0 || colop >= MIMSEM_V_COUNT) return MIMSEM_ERR_ARG;
    return stored_apply(c, colop, transpose, blocks, x, y);
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

// C[e][r] (+)= alpha * A[e][r+da] . B[e][r+db]   for r in [0,nr); out-of-range operand slots contribute 0.
// A workgroup takes TPW consecutive (e,r) tasks, stages both operand blocks in LDS (coalesced 648-B reads at
// p=3) and every thread forms entries from LDS.
__global__ __launch_bounds__(256) void k_bmm(int n2, int nEl, int nr, int tpw, BA C, BA A, int da, BA B, int db, double alpha, int accum) {
    extern __shared__ double sm[];
    const int nn = n2*n2, tid = threadIdx.x;
    const long long t0 = (long long)blockIdx.x*tpw, ntask = (long long)nEl*nr;
    const int nt = (int)min((long long)tpw, ntask - t0);
    double* sA = sm; double* sB = sm + tpw*nn;
    for (int x = tid; x < nt*nn; x += 256) {
        const long long task = t0 + x/nn; const int ij = x%nn;
        const int r = (int)(task%nr), e = (int)(task/nr), ra = r + da, rb = r + db;
        sA[x] = (ra >= 0 && ra < A.ns) ? A.p[((size_t)e*A.ns + ra)*nn + ij] : 0.0;
        sB[x] = (rb >= 0 && rb < B.ns) ? B.p[((size_t)e*B.ns + rb)*nn + ij] : 0.0;
    }
    __syncthreads();
    for (int x = tid; x < nt*nn; x += 256) {
        const int lt = x/nn, ij = x%nn, ii = ij/n2, jj = ij%n2;
        const long long task = t0 + lt;
        const int r = (int)(task%nr), e = (int)(task/nr);
        const double *a = sA + lt*nn, *b = sB + lt*nn;
        double s = 0.0;
        for (int k = 0; k < n2; k++) s += a[ii*n2 + k]*b[k*n2 + jj];
        double* out = C.p + ((size_t)e*C.ns + r)*nn + ij;
        if (accum) *out += alpha*s; else *out = alpha*s;
    }
}
int bmm(mimsem_ctx* c, int nr, BA C, BA A, int da, BA B, int db, double alpha, int accum) {
    const int n2 = c->es.n2e, nn = n2*n2;
    const long long ntask = (long long)c->nEl*nr;
    if (ntask <= 0) return MIMSEM_OK;
    const int tpw = std::max(1, 768/nn);                 // ~3 passes of the 256 threads per workgroup
    const size_t lds = (size_t)2*tpw*nn*sizeof(double);
    hipLaunchKernelGGL(k_bmm, dim3((unsigned)((ntask + tpw - 1)/tpw)), dim3(256), lds, c->stream,
                       n2, c->nEl, nr, tpw, C, A, da, B, db, alpha, accum);
    MIMSEM_HIP_TRY(hipGetLastError());
    return MIMSEM_OK;
}
int flat_mm_impl(mimsem_ctx* c, long long nb, const double* A, const double* B, double* C) {
    // view the flat arrays as [nEl][slots]: nb is always nEl * (slots per column) here
    const int per = (int)(nb/c->nEl);
    BA a{const_cast<double*>(A), per}, b{const_cast<double*>(B), per}, cc{C, per};
    return bmm(c, per, cc, a, 0, b, 0, 1.0, 0);
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

// Wave-per-column variant (n2 <= 16): the whole sweep of a column is carried by ONE wavefront, so the
// level-to-level dependency chain needs only wave-level LDS ordering (no s_barrier); the previous level's
// G_{k-1} = D'^{-1} sup and y_{k-1} stay in LDS.  Gauss-Jordan with partial pivoting on the running diagonal block.
__device__ __forceinline__ void wsync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
// the same hand-off when only LDS data crosses lanes: the fences name the local address space, so outstanding GLOBAL loads /
// stores (the prefetch of the next level, the stores of G and D^-1) are not waited for
__device__ __forceinline__ void wsync_lds() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
// y = M x (or M^T x) WITHOUT the blocks, for the operators whose block (r, w) is W^T diag(c_{r,w}) W: one group of lanes per (column,
// output slot) task, lane q owns quadrature point q -- it evaluates the coefficient (the same colop_coef the block assembly
// uses), interpolates the input there and scales; lanes a < n2 project back.  Replaces k_coef_block (53 us) + the stored apply
// (33 us) of mimsem_colop_apply by one launch.
template <int N>
__global__ __launch_bounds__(256) void k_colop_apply_mf(CG g, int colop, unsigned flags, const double* __restrict__ f1,
        const double* __restrict__ f2, int transpose, int nr, int nw, int nxx, int nyy, int off,
        const double* __restrict__ x, double* __restrict__ y) {
    constexpr int N2 = N*N, MP1 = N + 1, MP12 = MP1*MP1;
    constexpr int GW = MP12 <= 16 ? 16 : (MP12 <= 32 ? 32 : 64), TPB = 256/GW;
    __shared__ double sE[MP1*N], sw[MP1], sW[MP12*N2], sx[TPB][N2], sv[TPB][MP12];
    const int tid = threadIdx.x, t = tid/GW, r = tid%GW;
    if (tid < MP1*N) sE[tid] = g.E[tid];
    if (tid < MP1) sw[tid] = g.w[tid];
    __syncthreads();
    for (int i = tid; i < MP12*N2; i += 256) { const int q = i/N2, j = i%N2; sW[i] = sE[(q%MP1)*N + j%N]*sE[(q/MP1)*N + j/N]; }
    __syncthreads();
    const long long task0 = (long long)blockIdx.x*TPB + t, ntask = (long long)g.nEl*nyy;
    const bool live = task0 < ntask;
    const long long task = live ? task0 : ntask - 1;
    const int ry = (int)(task%nyy), e = (int)(task/nyy);
    CG gl = g; gl.E = sE; gl.w = sw;
    double acc = 0.0;
    for (int w = 0; w < nw; w++) {
        int rb, cx;                       // stored block row, input slot (as in stored_apply)
        if (nw == 1)         { rb = ry; cx = ry; }
        else if (!transpose) { rb = ry; cx = ry + off + w; }
        else                 { rb = ry - off - w; cx = rb; }
        const bool ok = rb >= 0 && rb < nr && cx >= 0 && cx < nxx;
        wsync_lds();
        if (r < N2) sx[t][r] = ok ? x[((size_t)e*nxx + cx)*N2 + r] : 0.0;
        wsync_lds();
        if (r < MP12) {
            double u = 0.0;
#pragma unroll
            for (int j = 0; j < N2; j++) u += sW[r*N2 + j]*sx[t][j];
            sv[t][r] = ok ? colop_coef(gl, colop, flags, e, rb, w, r, f1, f2)*u : 0.0;
        }
        wsync_lds();
        if (r < N2) {
#pragma unroll
            for (int q = 0; q < MP12; q++) acc += sW[q*N2 + r]*sv[t][q];
        }
    }
    if (live && r < N2) y[(size_t)task*N2 + r] = acc;
}

int launch_colop_apply_mf(mimsem_ctx* c, int colop, unsigned flags, int transpose, const double* f1, const double* f2,
                          const double* x, double* y) {
    if (colop_is_inverse(colop) || colop == MIMSEM_V_EOS_BLOCK || colop == MIMSEM_V_EOS_BLOCK_INV ||
        colop == MIMSEM_V_LINEAR_RHO2_UP || colop == MIMSEM_V_LINCON2_UP || exp_env("MIMSEM_COLOP_BLOCKS"))
        return MIMSEM_ERR_UNSUPPORTED;
    int nr, nw, nx, ny;
    colop_shape(colop, c->nk, &nr, &nw, &nx, &ny);
    const int off = (colop == MIMSEM_V_LINCON) ? 0 : -1;
    const int nyy = transpose ? nx : ny, nxx = transpose ? ny : nx;
    const long long ntask = (long long)c->nEl*nyy;
    if (ntask == 0) return MIMSEM_OK;
    const CG g = make_cg(c);
    const int mp12 = c->es.mp12, gw = mp12 <= 16 ? 16 : (mp12 <= 32 ? 32 : 64), tpb = 256/gw;
    const unsigned grid = (unsigned)((ntask + tpb - 1)/tpb);
    switch (c->es.n) {
#define MIMSEM_MF(N) case N: hipLaunchKernelGGL((k_colop_apply_mf<N>), dim3(grid), dim3(256), 0, c->stream, g, colop, flags, f1, f2, transpose, \
                                                nr, nw, nxx, nyy, off, x, y); break;
    MIMSEM_MF(1) MIMSEM_MF(2) MIMSEM_MF(3) MIMSEM_MF(4) MIMSEM_MF(5) MIMSEM_MF(6) MIMSEM_MF(7)
#undef MIMSEM_MF
    default: return MIMSEM_ERR_UNSUPPORTED;
    }
    MIMSEM_HIP_TRY(hipGetLastError());
    return MIMSEM_OK;
}

template <int N2>
__global__ __launch_bounds__(64) void k_block_thomas_wave(int nk, const double* __restrict__ L, const double* __restrict__ f,
                                                          double* __restrict__ d, double* __restrict__ Gws, double* __restrict__ yws,
                                                          double* __restrict__ Dinv /* optional [nEl][nk][nn]: kept for re-solves */) {
    constexpr int nn = N2*N2, EPL = (nn + 63)/64;           // entries per lane
    __shared__ double D[nn], Di[nn], Gp[nn], S[nn], U[nn], v[N2], yp[N2], dn[N2];
    const int lane = threadIdx.x, e = blockIdx.x;
    const double* Le = L + (size_t)e*nk*3*nn;
    double* G = Gws + (size_t)e*nk*nn;
    double* yv = yws + (size_t)e*nk*N2;
    for (int k = 0; k < nk; k++) {
        const double* sub = Le + ((size_t)k*3 + 0)*nn;
        const double* dia = Le + ((size_t)k*3 + 1)*nn;
        const double* sup = Le + ((size_t)k*3 + 2)*nn;
#pragma unroll
        for (int r = 0; r < EPL; r++) { const int t = lane + 64*r; if (t < nn) { S[t] = sub[t]; U[t] = sup[t]; } }
        wsync();
#pragma unroll
        for (int r = 0; r < EPL; r++) {
            const int t = lane + 64*r;
            if (t < nn) {
                const int i = t/N2, j = t%N2;
                double s = dia[t];
                if (k > 0) {
#pragma unroll
                    for (int m = 0; m < N2; m++) s -= S[i*N2 + m]*Gp[m*N2 + j];
                }
                D[t] = s; Di[t] = (i == j) ? 1.0 : 0.0;
            }
        }
        if (lane < N2) {
            double s = f[((size_t)e*nk + k)*N2 + lane];
            if (k > 0) {
#pragma unroll
                for (int m = 0; m < N2; m++) s -= S[lane*N2 + m]*yp[m];
            }
            v[lane] = s;
        }
        wsync();
        // Gauss-Jordan on [D | Di], partial pivoting (every lane finds the same pivot row from LDS)
        for (int col = 0; col < N2; col++) {
            int p = col; double big = fabs(D[col*N2 + col]);
            for (int r = col + 1; r < N2; r++) { const double a = fabs(D[r*N2 + col]); if (a > big) { big = a; p = r; } }
            if (p != col) {
                double t0[EPL], t1[EPL];
#pragma unroll
                for (int r = 0; r < EPL; r++) {      // lanes < 2*N2 swap one entry of D or Di each
                    const int t = lane + 64*r;
                    if (t < 2*N2) { double* M = (t < N2) ? D : Di; const int j = t%N2; t0[r] = M[col*N2 + j]; t1[r] = M[p*N2 + j]; }
                }
                wsync();
#pragma unroll
                for (int r = 0; r < EPL; r++) {
                    const int t = lane + 64*r;
                    if (t < 2*N2) { double* M = (t < N2) ? D : Di; const int j = t%N2; M[col*N2 + j] = t1[r]; M[p*N2 + j] = t0[r]; }
                }
                wsync();
            }
            const double pinv = 1.0/D[col*N2 + col];
            // new values from old ones, then one sync: row col scaled, other rows eliminated
            double nd[EPL], ni[EPL];
#pragma unroll
            for (int r = 0; r < EPL; r++) {
                const int t = lane + 64*r;
                if (t < nn) {
                    const int i = t/N2, j = t%N2;
                    const double dcj = D[col*N2 + j]*pinv, icj = Di[col*N2 + j]*pinv;
                    if (i == col) { nd[r] = dcj; ni[r] = icj; }
                    else { const double u = D[i*N2 + col]; nd[r] = D[t] - u*dcj; ni[r] = Di[t] - u*icj; }
                }
            }
            wsync();
#pragma unroll
            for (int r = 0; r < EPL; r++) { const int t = lane + 64*r; if (t < nn) { D[t] = nd[r]; Di[t] = ni[r]; } }
            wsync();
        }
        // G_k = Di . sup ; y_k = Di . v
#pragma unroll
        for (int r = 0; r < EPL; r++) {
            const int t = lane + 64*r;
            if (t < nn) {
                const int i = t/N2, j = t%N2; double s = 0.0;
                if (k < nk - 1) {
#pragma unroll
                    for (int m = 0; m < N2; m++) s += Di[i*N2 + m]*U[m*N2 + j];
                }
                G[(size_t)k*nn + t] = s; Gp[t] = s;
                if (Dinv) Dinv[((size_t)e*nk + k)*nn + t] = Di[t];
            }
        }
        if (lane < N2) {
            double s = 0.0;
#pragma unroll
            for (int m = 0; m < N2; m++) s += Di[lane*N2 + m]*v[m];
            yv[(size_t)k*N2 + lane] = s; yp[lane] = s;
        }
        wsync();
    }
    // back substitution: d_k = y_k - G_k d_{k+1}; G re-read from this wave's own global writes
    __threadfence_block();
    for (int k = nk - 1; k >= 0; k--) {
        if (lane < N2) {
            double s = yv[(size_t)k*N2 + lane];
            if (k < nk - 1) { const double* Gk = G + (size_t)k*nn;
#pragma unroll
                for (int m = 0; m < N2; m++) s -= Gk[lane*N2 + m]*dn[m]; }
            d[((size_t)e*nk + k)*N2 + lane] = s;
            v[lane] = s;
        }
        wsync();
        if (lane < N2) dn[lane] = v[lane];
        wsync();
    }
}

// rotate within each row of 16 lanes (DPP row_ror: a VALU modifier, no LDS crossbar trip like ds_bpermute)
template <int N> __device__ __forceinline__ int row_ror(int v) {
    return __builtin_amdgcn_update_dpp(v, v, 0x120 + N, 0xF, 0xF, false);
}
template <int N> __device__ __forceinline__ double row_ror(double v) {
    return __hiloint2double(row_ror<N>(__double2hiint(v)), row_ror<N>(__double2loint(v)));
}

// The same sweep with ONE ROW PER LANE: 16 lanes per column (4 columns per wavefront), the running diagonal block lives in
// registers (row r in lane r) and the Gauss-Jordan steps talk through cross-lane shuffles instead of LDS round trips:
// pivot search = 4 butterfly rounds, pivot row = N2 broadcasts.  Rows are never swapped -- the lane chosen at step c keeps the
// row and remembers c; the in-place inverse T then satisfies  D^-1[c_l][p_c] = T[l][c]  (c_l: step at which lane l was the pivot,
// p_c: pivot lane of step c), which is how it is scattered into LDS.  LDS only carries the operands of the two block products.
template <int N2, int GW = 16>
__global__ __launch_bounds__(64) void k_block_thomas_rows(int nEl, int nk, const double* __restrict__ L, const double* __restrict__ f,
                                                          double* __restrict__ d, double* __restrict__ Gws, double* __restrict__ yws,
                                                          double* __restrict__ Dinv) {
    constexpr int nn = N2*N2, CPW = 64/GW;
    static_assert(N2 <= GW && (GW == 16 || GW == 32), "one lane per block row; 16 or 32 lanes per column");
    constexpr bool PIPE = N2 <= 9;        // prefetch + deferred stores cost 8 N2 VGPRs: only while the rows are short
    __shared__ double sG[CPW][nn], sU[CPW][nn], sI[CPW][nn], sy[CPW][N2], sv[CPW][N2];
    const int lane = threadIdx.x, g = lane/GW, r = lane%GW;
    const int e0 = blockIdx.x*CPW + g;
    const bool act = r < N2 && e0 < nEl;
    const int e = e0 < nEl ? e0 : nEl - 1, rr = r < N2 ? r : 0;           // clamped: idle lanes compute on valid addresses, store nothing
    const double* Le = L + (size_t)e*nk*3*nn;
    double* G = Gws + (size_t)e*nk*nn;
    double* yv = yws + (size_t)e*nk*N2;
    double S[N2], T[N2], U[N2], Sn[N2], Tn[N2], Un[N2], fn;
    auto fetch = [&](int k) {                     // this lane's rows of level k; issued one level ahead of their use
        const double* row = Le + (size_t)k*3*nn + rr*N2;
#pragma unroll
        for (int j = 0; j < N2; j++) { Sn[j] = row[j]; Tn[j] = row[nn + j]; Un[j] = row[2*nn + j]; }
        fn = f[((size_t)e*nk + k)*N2 + rr];
    };
    // results of a level are stored at the START of the next one, ahead of the prefetch: the in-order vmcnt wait for the prefetched
    // rows at the loop head then covers stores issued a whole level earlier instead of stalling on fresh ones
    double Gst[N2], Ist[N2], yst = 0.0;
    auto flush = [&](int k) {
        if (!act) return;
#pragma unroll
        for (int j = 0; j < N2; j++) G[(size_t)k*nn + r*N2 + j] = Gst[j];
        yv[(size_t)k*N2 + r] = yst;
        if (Dinv) {
#pragma unroll
            for (int m = 0; m < N2; m++) Dinv[((size_t)e*nk + k)*nn + r*N2 + m] = Ist[m];
        }
    };
    if (PIPE) fetch(0);
    for (int k = 0; k < nk; k++) {
        if (!PIPE) fetch(k);
#pragma unroll
        for (int j = 0; j < N2; j++) { S[j] = Sn[j]; T[j] = Tn[j]; U[j] = Un[j]; }
        double vv = fn;
        if (PIPE) {
            if (k > 0) flush(k - 1);
            if (k + 1 < nk) fetch(k + 1);
        }
        if (k > 0) {
#pragma unroll
            for (int m = 0; m < N2; m++) {
                const double sm = S[m];
#pragma unroll
                for (int j = 0; j < N2; j++) T[j] -= sm*sG[g][m*N2 + j];
                vv -= sm*sy[g][m];
            }
        }
        // in-place Gauss-Jordan inverse, implicit row pivoting
        bool used = !act;
        int myc = 0, piv[N2];
#pragma unroll
        for (int c = 0; c < N2; c++) {
            double cand = used ? -1.0 : fabs(T[c]);
            int bl = r;
            // arg-max over the 16 lanes of the column: rotations by 1, 2, 4, 8 (ties -> lowest lane, so every lane agrees)
#define MIMSEM_ARGMAX_ROUND(N) { const double oc = row_ror<N>(cand); const int ol = row_ror<N>(bl); \
                                 if (oc > cand || (oc == cand && ol < bl)) { cand = oc; bl = ol; } }
            MIMSEM_ARGMAX_ROUND(1) MIMSEM_ARGMAX_ROUND(2) MIMSEM_ARGMAX_ROUND(4) MIMSEM_ARGMAX_ROUND(8)
#undef MIMSEM_ARGMAX_ROUND
            if constexpr (GW == 32) {            // the two 16-lane rows of the column compare notes
                const double oc = __shfl_xor(cand, 16, 32); const int ol = __shfl_xor(bl, 16, 32);
                if (oc > cand || (oc == cand && ol < bl)) { cand = oc; bl = ol; }
            }
            piv[c] = bl;
            double pr[N2];
#pragma unroll
            for (int j = 0; j < N2; j++) pr[j] = __shfl(T[j], bl, GW);
            const double pinv = 1.0/pr[c];
            if (r == bl) {
#pragma unroll
                for (int j = 0; j < N2; j++) T[j] = pr[j]*pinv;
                T[c] = pinv; used = true; myc = c;
            } else {
                const double fct = T[c];
#pragma unroll
                for (int j = 0; j < N2; j++) T[j] -= fct*(pr[j]*pinv);
                T[c] = -fct*pinv;
            }
        }
        if (act) {
#pragma unroll
            for (int c = 0; c < N2; c++) sI[g][myc*N2 + piv[c]] = T[c];
#pragma unroll
            for (int j = 0; j < N2; j++) sU[g][r*N2 + j] = U[j];
            sv[g][r] = vv;
        }
        wsync_lds();
        double Ir[N2], Gr[N2], y = 0.0;
#pragma unroll
        for (int m = 0; m < N2; m++) { Ir[m] = sI[g][rr*N2 + m]; y += Ir[m]*sv[g][m]; }
#pragma unroll
        for (int j = 0; j < N2; j++) Gr[j] = 0.0;
        if (k < nk - 1) {
#pragma unroll
            for (int m = 0; m < N2; m++) {
                const double im = Ir[m];
#pragma unroll
                for (int j = 0; j < N2; j++) Gr[j] += im*sU[g][m*N2 + j];
            }
        }
        wsync_lds();                                  // every lane has read sI/sU/sv and the previous sG/sy
        if (act) {
#pragma unroll
            for (int j = 0; j < N2; j++) sG[g][r*N2 + j] = Gr[j];
            sy[g][r] = y;
        }
#pragma unroll
        for (int j = 0; j < N2; j++) { Gst[j] = Gr[j]; Ist[j] = Ir[j]; }
        yst = y;
        if (!PIPE) flush(k);
        wsync_lds();
    }
    if (PIPE) flush(nk - 1);
    // back substitution d_k = y_k - G_k d_{k+1}: each lane re-reads its own row of G_k (its own stores)
    for (int k = nk - 1; k >= 0; k--) {
        double s = 0.0;
        if (act) {
            s = yv[(size_t)k*N2 + r];
            if (k < nk - 1) {
                const double* Gk = G + (size_t)k*nn + r*N2;
#pragma unroll
                for (int m = 0; m < N2; m++) s -= Gk[m]*sy[g][m];
            }
            d[((size_t)e*nk + k)*N2 + r] = s;
        }
        wsync();
        if (act) sy[g][r] = s;
        wsync();
    }
}

// re-solve with the factors of a previous sweep (Dinv_k = D'_k^-1, G_k = D'_k^-1 sup_k): forward y_k = Dinv_k (f_k - sub_k y_{k-1}),
// backward d_k = y_k - G_k d_{k+1}.  One thread per (column, row), the column's recurrences run in LDS (columns do not interact).
template <int N2>
__global__ __launch_bounds__(64) void k_block_thomas_resolve(int nEl, int nk, const double* __restrict__ L, const double* __restrict__ Dinv,
        const double* __restrict__ G, const double* __restrict__ f, double* __restrict__ d) {
    constexpr int nn = N2*N2, CPW = 64/N2;               // columns per wave
    __shared__ double yl[CPW][2][N2], t[CPW][N2];
    const int lane = threadIdx.x, lc = lane/N2, a = lane%N2;
    const int e = blockIdx.x*CPW + lc;
    const bool act = lc < CPW && e < nEl;
    extern __shared__ double ybuf[];                      // [CPW][nk][N2]  forward results
    double* yv = ybuf + (size_t)lc*nk*N2;
    for (int k = 0; k < nk; k++) {
        if (act) {
            const double* sub = L + (((size_t)e*nk + k)*3 + 0)*nn;
            double s = f[((size_t)e*nk + k)*N2 + a];
            if (k > 0) {
#pragma unroll
                for (int m = 0; m < N2; m++) s -= sub[a*N2 + m]*yv[(size_t)(k - 1)*N2 + m];
            }
            t[lc][a] = s;
        }
        wsync();
        if (act) {
            const double* Di = Dinv + ((size_t)e*nk + k)*nn;
            double s = 0.0;
#pragma unroll
            for (int m = 0; m < N2; m++) s += Di[a*N2 + m]*t[lc][m];
            yv[(size_t)k*N2 + a] = s;
        }
        wsync();
    }
    for (int k = nk - 1; k >= 0; k--) {
        double s = 0.0;
        if (act) {
            s = yv[(size_t)k*N2 + a];
            if (k < nk - 1) {
                const double* Gk = G + ((size_t)e*nk + k)*nn;
#pragma unroll
                for (int m = 0; m < N2; m++) s -= Gk[a*N2 + m]*yl[lc][(k + 1) & 1][m];
            }
            d[((size_t)e*nk + k)*N2 + a] = s;
            yl[lc][k & 1][a] = s;
        }
        wsync();
    }
}

int block_thomas(mimsem_ctx* c, const double* L, const double* f, double* d, double* Gws, double* yws, double* Dinv = nullptr) {
    const int n2 = c->es.n2e, nn = n2*n2;
    const bool rows = !exp_env("MIMSEM_THOMAS_WAVE");          // row-per-lane kernel (4 columns per wavefront); the wave-per-column one stays selectable
    if (rows && !exp_env("MIMSEM_THOMAS_WG") && (n2 == 4 || n2 == 9 || n2 == 16)) {
        const unsigned grid = (unsigned)((c->nEl + 3)/4);
        switch (n2) {
#define MIMSEM_TRW(N) case N: hipLaunchKernelGGL((k_block_thomas_rows<N>), dim3(grid), dim3(64), 0, c->stream, c->nEl, c->nk, L, f, d, Gws, yws, Dinv); \
                     MIMSEM_HIP_TRY(hipGetLastError()); return MIMSEM_OK;
        MIMSEM_TRW(4) MIMSEM_TRW(9) MIMSEM_TRW(16)
#undef MIMSEM_TRW
        }
    }
    if (!exp_env("MIMSEM_THOMAS_WG")) {
        switch (n2) {
#define MIMSEM_TW(N) case N: hipLaunchKernelGGL((k_block_thomas_wave<N>), dim3(c->nEl), dim3(64), 0, c->stream, c->nk, L, f, d, Gws, yws, Dinv); \
                     MIMSEM_HIP_TRY(hipGetLastError()); return MIMSEM_OK;
        MIMSEM_TW(1) MIMSEM_TW(4) MIMSEM_TW(9) MIMSEM_TW(16)
#undef MIMSEM_TW
        default: break;
        }
    }
    const int nt = std::min(256, ((nn + 63)/64)*64);
    const size_t lds = (size_t)(3*nn + 2*n2)*sizeof(double);
    hipLaunchKernelGGL(k_block_thomas, dim3(c->nEl), dim3(nt), lds, c->stream, c->nk, n2, L, f, d, Gws, yws);
    MIMSEM_HIP_TRY(hipGetLastError());
    return MIMSEM_OK;
}

// r = f - L d for block-tridiagonal L [nEl][nk][3][n2][n2]: the residual of one step of iterative refinement
int block_tridiag_residual(mimsem_ctx* c, int nk, int n2, const double* L, const double* f, const double* d, double* r) {
    const int nn = n2*n2, nEl = c->nEl;
    return each(c, (long long)nEl*nk*n2, [=] __device__(long long i) {
        const int a = (int)(i%n2); long long t = i/n2; const int k = (int)(t%nk), e = (int)(t/nk);
        const double* Lk = L + ((size_t)e*nk + k)*3*nn;
        const double* de = d + (size_t)e*nk*n2;
        double s = f[i];
        if (k > 0)      { const double* m = Lk;          const double* v = de + (size_t)(k - 1)*n2; for (int p = 0; p < n2; p++) s -= m[a*n2 + p]*v[p]; }
        {                 const double* m = Lk + nn;     const double* v = de + (size_t)k*n2;       for (int p = 0; p < n2; p++) s -= m[a*n2 + p]*v[p]; }
        if (k < nk - 1) { const double* m = Lk + 2*nn;   const double* v = de + (size_t)(k + 1)*n2; for (int p = 0; p < n2; p++) s -= m[a*n2 + p]*v[p]; }
        r[i] = s;
    });
}

// L d = f with ONE step of iterative refinement: the block-Thomas sweep pivots only inside the running diagonal block, so on
// columns whose running diagonal blocks are poorly conditioned its backward error grows (1e-6 on the worst of 3 456 random
// columns, cond ~ 2e7, where LU with row pivoting over the whole band gives 1e-12); d += solve(f - L d) restores it.
int block_thomas_refined(mimsem_ctx* c, const double* L, const double* f, double* d, double* Gws, double* yws, double* r, double* dd,
                         double* Dinv /* [nEl][nk][nn] scratch or null */) {
    int rc;
    const int n2 = c->es.n2e, nk = c->nk, nEl = c->nEl;
    const bool wave = !exp_env("MIMSEM_THOMAS_WG") && (n2 == 1 || n2 == 4 || n2 == 9 || n2 == 16);
    const bool keep = Dinv && wave && (size_t)(64/n2)*nk*n2*sizeof(double) <= 48*1024;
    if ((rc = block_thomas(c, L, f, d, Gws, yws, keep ? Dinv : nullptr))) return rc;
    if (getenv("MIMSEM_NO_REFINE")) return MIMSEM_OK;
    if ((rc = block_tridiag_residual(c, nk, n2, L, f, d, r))) return rc;
    if (keep) {                      // substitution only, with the factors of the first sweep
        const int cpw = 64/n2;
        const size_t lds = (size_t)cpw*nk*n2*sizeof(double);
        const unsigned grid = (unsigned)((nEl + cpw - 1)/cpw);
        switch (n2) {
#define MIMSEM_TR(N) case N: hipLaunchKernelGGL((k_block_thomas_resolve<N>), dim3(grid), dim3(64), lds, c->stream, nEl, nk, L, Dinv, Gws, r, dd); break;
        MIMSEM_TR(1) MIMSEM_TR(4) MIMSEM_TR(9) MIMSEM_TR(16)
#undef MIMSEM_TR
        }
        MIMSEM_HIP_TRY(hipGetLastError());
    } else if ((rc = block_thomas(c, L, r, dd, Gws, yws))) return rc;
    return each(c, (long long)nEl*nk*n2, [=] __device__(long long i) { d[i] += dd[i]; });
}

// one workgroup per (column, level): the two DIV blocks of the row, the four G_pi blocks they meet and N_pi are
// staged in LDS; 3*nn entries of L(k,k-1), L(k,k), L(k,k+1) are formed from LDS.
__global__ __launch_bounds__(256) void k_helmholtz_rows(int n2, int nk, double gam, const double* __restrict__ Dl,
        const double* __restrict__ Du, const double* __restrict__ Gl, const double* __restrict__ Gu,
        const double* __restrict__ Np, double* __restrict__ L) {
    extern __shared__ double sm[];
    const int nn = n2*n2, nm = nk - 1, tid = threadIdx.x;
    const int k = blockIdx.x%nk, e = blockIdx.x/nk;
    double *sDl = sm, *sDu = sm + nn, *sGlm = sm + 2*nn, *sGum = sm + 3*nn, *sGl = sm + 4*nn, *sGu = sm + 5*nn, *sN = sm + 6*nn;
    for (int x = tid; x < nn; x += 256) {
        sDl[x] = Dl[((size_t)e*nk + k)*nn + x]; sDu[x] = Du[((size_t)e*nk + k)*nn + x]; sN[x] = Np[((size_t)e*nk + k)*nn + x];
        sGlm[x] = (k > 0) ? Gl[((size_t)e*nm + k - 1)*nn + x] : 0.0;
        sGum[x] = (k > 0) ? Gu[((size_t)e*nm + k - 1)*nn + x] : 0.0;
        sGl[x] = (k < nk - 1) ? Gl[((size_t)e*nm + k)*nn + x] : 0.0;
        sGu[x] = (k < nk - 1) ? Gu[((size_t)e*nm + k)*nn + x] : 0.0;
    }
    __syncthreads();
    for (int x = tid; x < 3*nn; x += 256) {
        const int w = x/nn, ij = x%nn, ii = ij/n2, jj = ij%n2;
        double s = 0.0;
        if (w == 0)      { for (int p = 0; p < n2; p++) s += sDl[ii*n2 + p]*sGlm[p*n2 + jj]; s = (-1.0*gam)*s; }          // DIV(k,k-1) G(k-1,k-1)
        else if (w == 2) { for (int p = 0; p < n2; p++) s += sDu[ii*n2 + p]*sGu[p*n2 + jj]; s = (-1.0*gam)*s; }           // DIV(k,k)   G(k,k+1)
        else { for (int p = 0; p < n2; p++) s += sDl[ii*n2 + p]*sGum[p*n2 + jj];                                         // DIV(k,k-1) G(k-1,k)
               for (int p = 0; p < n2; p++) s += sDu[ii*n2 + p]*sGl[p*n2 + jj];                                          // DIV(k,k)   G(k,k)
               s = (-1.0*gam)*s + sN[ij]; }
        L[(((size_t)e*nk + k)*3 + w)*nn + ij] = s;
    }
}

// ---- fused assembly of the Schur factors: one WAVE per (column, level) resp. (column, interface) ----------------------------
// All intermediate blocks of a task live in LDS; only what later stages need is written.  Wave-cooperative block algebra on
// N2 x N2 blocks (N2 = 4, 9, 16): sum-factorised W^T diag(c) W, Gauss-Jordan inverse with partial pivoting, products.
template <int N2>
struct WaveBlocks {
    static constexpr int nn = N2*N2, EPL = (nn + 63)/64;
    // M = W^T diag(c) W by sum factorisation (W[q][j] = E[qx][jx] E[qy][jy]); lanes (iy, jy)
    __device__ static void assemble(double* M, const double* cq, const double* sE, const double* sEE, int n, int mp1, int lane) {
        if (lane < N2) {
            const int iy = lane/n, jy = lane%n;
            double t1[8];
            for (int qx = 0; qx < mp1; qx++) {
                double s = 0.0;
                for (int qy = 0; qy < mp1; qy++) s += (sE[qy*n + iy]*cq[qy*mp1 + qx])*sE[qy*n + jy];
                t1[qx] = s;
            }
            for (int ix = 0; ix < n; ix++)
                for (int jx = 0; jx < n; jx++) {
                    double s = 0.0;
                    for (int qx = 0; qx < mp1; qx++) s += sEE[qx*N2 + ix*n + jx]*t1[qx];
                    M[(iy*n + ix)*N2 + jy*n + jx] = s;
                }
        }
        wsync();
    }
    // C = alpha * A . B   (C must not alias A or B)
    __device__ static void mul(double* C, const double* A, const double* B, double alpha, int lane) {
#pragma unroll
        for (int r = 0; r < EPL; r++) {
            const int t = lane + 64*r;
            if (t < nn) {
                const int i = t/N2, j = t%N2;
                double s = 0.0;
#pragma unroll
                for (int m = 0; m < N2; m++) s += A[i*N2 + m]*B[m*N2 + j];
                C[t] = alpha*s;
            }
        }
        wsync();
    }
    // Di = D^-1 by Gauss-Jordan with partial pivoting; D is destroyed
    __device__ static void inverse(double* D, double* Di, int lane) {
#pragma unroll
        for (int r = 0; r < EPL; r++) { const int t = lane + 64*r; if (t < nn) Di[t] = (t/N2 == t%N2) ? 1.0 : 0.0; }
        wsync();
        for (int col = 0; col < N2; col++) {
            int p = col; double big = fabs(D[col*N2 + col]);
            for (int r = col + 1; r < N2; r++) { const double a = fabs(D[r*N2 + col]); if (a > big) { big = a; p = r; } }
            if (p != col) {
                double t0[EPL], t1[EPL];
#pragma unroll
                for (int r = 0; r < EPL; r++) {
                    const int t = lane + 64*r;
                    if (t < 2*N2) { double* M = (t < N2) ? D : Di; const int j = t%N2; t0[r] = M[col*N2 + j]; t1[r] = M[p*N2 + j]; }
                }
                wsync();
#pragma unroll
                for (int r = 0; r < EPL; r++) {
                    const int t = lane + 64*r;
                    if (t < 2*N2) { double* M = (t < N2) ? D : Di; const int j = t%N2; M[col*N2 + j] = t1[r]; M[p*N2 + j] = t0[r]; }
                }
                wsync();
            }
            const double pinv = 1.0/D[col*N2 + col];
            double nd[EPL], ni[EPL];
#pragma unroll
            for (int r = 0; r < EPL; r++) {
                const int t = lane + 64*r;
                if (t < nn) {
                    const int i = t/N2, j = t%N2;
                    const double dcj = D[col*N2 + j]*pinv, icj = Di[col*N2 + j]*pinv;
                    if (i == col) { nd[r] = dcj; ni[r] = icj; }
                    else { const double u = D[i*N2 + col]; nd[r] = D[t] - u*dcj; ni[r] = Di[t] - u*icj; }
                }
            }
            wsync();
#pragma unroll
            for (int r = 0; r < EPL; r++) { const int t = lane + 64*r; if (t < nn) { D[t] = nd[r]; Di[t] = ni[r]; } }
            wsync();
        }
    }
    __device__ static void load(double* dst, const double* src, int lane) {
#pragma unroll
        for (int r = 0; r < EPL; r++) { const int t = lane + 64*r; if (t < nn) dst[t] = src[t]; }
        wsync();
    }
    __device__ static void store(double* dst, const double* src, int lane) {     // LDS -> global, no sync needed after (LDS is only read)
#pragma unroll
        for (int r = 0; r < EPL; r++) { const int t = lane + 64*r; if (t < nn) dst[t] = src[t]; }
    }
};

struct FusedArgs {
    CG g; double hdt;
    const double *theta, *rho, *eta, *pi;
    double *B, *Binv, *Npi, *Nrho;                 // level outputs   [nEl][nk][nn]
    double *Ainv, *X, *Gl, *Gu, *gpi, *geta;       // interface outputs [nEl][nk-1][nn] / [nEl][nk-1][n2]
    // row-per-lane path with a right-hand side: tE = VB^-1 F_eta, tR = VB^-1 F_rho per level, while VB^-1 is in registers
    const double *F_eta = nullptr, *F_rho = nullptr; double *tE = nullptr, *tR = nullptr;
};

// level k of column e: B = VB, Binv = VB^-1, N_pi = B B(pi)^-1 B, N_rho = B B(rho)^-1 B     (VertSolve.cpp:690-692, :736-739)
template <int N2>
__global__ __launch_bounds__(256) void k_schur_levels(FusedArgs a) {
    using WBk = WaveBlocks<N2>;
    constexpr int nn = N2*N2, WPB = 4;
    extern __shared__ double sm[];
    const CG& g = a.g;
    const int n = g.n, mp1 = g.mp1, mp12 = g.mp12, tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    double* sE = sm; double* sEE = sE + mp1*n; double* sw = sEE + mp1*N2;
    double* base = sw + 8 + (size_t)wv*(3*mp12 + 5*nn);
    double *cq = base, *Bk = cq + 3*mp12, *P = Bk + nn, *Pi_ = P + nn, *t1 = Pi_ + nn, *t2 = t1 + nn;
    for (int t = tid; t < mp1*n; t += 256) sE[t] = g.E[t];
    if (tid < mp1) sw[tid] = g.w[tid];
    for (int t = tid; t < mp1*N2; t += 256) { const int qx = t/N2, ij = t%N2; sEE[t] = g.E[qx*n + ij/n]*g.E[qx*n + ij%n]; }
    __syncthreads();
    const long long task = (long long)blockIdx.x*WPB + wv;
    if (task >= (long long)g.nEl*g.nk) return;             // whole waves leave together; no block-level barrier below
    const int k = (int)(task%g.nk), e = (int)(task/g.nk);
    CG gl = g; gl.E = sE; gl.w = sw;
    if (lane < mp12) {
        cq[lane]          = colop_coef(gl, MIMSEM_V_CONST, 0, e, k, 0, lane, nullptr, nullptr);
        cq[mp12 + lane]   = colop_coef(gl, MIMSEM_V_CONST_RHO, 0, e, k, 0, lane, a.pi, nullptr);
        cq[2*mp12 + lane] = colop_coef(gl, MIMSEM_V_CONST_RHO, 0, e, k, 0, lane, a.rho, nullptr);
    }
    wsync();
    const size_t off = ((size_t)e*g.nk + k)*nn;
    WBk::assemble(Bk, cq, sE, sEE, n, mp1, lane);
    WBk::store(a.B + off, Bk, lane);
    WBk::load(t1, Bk, lane);
    WBk::inverse(t1, t2, lane);
    WBk::store(a.Binv + off, t2, lane);
    for (int which = 0; which < 2; which++) {
        wsync();
        WBk::assemble(P, cq + (1 + which)*mp12, sE, sEE, n, mp1, lane);
        WBk::inverse(P, Pi_, lane);
        WBk::mul(t1, Pi_, Bk, 1.0, lane);                  // B(rt)^-1 B
        WBk::mul(t2, Bk, t1, 1.0, lane);                   // B (B(rt)^-1 B)
        WBk::store((which ? a.Nrho : a.Npi) + off, t2, lane);
    }
}

// interface i of column e: A^-1 = VA_inv, X = A^-1 VA(rho), G_pi row (Gl, Gu), grad pi, grad eta   (:691-731)
template <int N2>
__global__ __launch_bounds__(256) void k_schur_interfaces(FusedArgs a) {
    using WBk = WaveBlocks<N2>;
    constexpr int nn = N2*N2, WPB = 4;
    extern __shared__ double sm[];
    const CG& g = a.g;
    const int n = g.n, mp1 = g.mp1, mp12 = g.mp12, nk = g.nk, nm = nk - 1, tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    double* sE = sm; double* sEE = sE + mp1*n; double* sw = sEE + mp1*N2;
    double* base = sw + 8 + (size_t)wv*(3*mp12 + 7*nn + 2*N2);
    double *cq = base, *A = cq + 3*mp12, *Ai = A + nn, *T = Ai + nn, *Rr = T + nn, *B0 = Rr + nn, *B1 = B0 + nn, *t1 = B1 + nn, *v = t1 + nn;
    for (int t = tid; t < mp1*n; t += 256) sE[t] = g.E[t];
    if (tid < mp1) sw[tid] = g.w[tid];
    for (int t = tid; t < mp1*N2; t += 256) { const int qx = t/N2, ij = t%N2; sEE[t] = g.E[qx*n + ij/n]*g.E[qx*n + ij%n]; }
    __syncthreads();
    const long long task = (long long)blockIdx.x*WPB + wv;
    if (task >= (long long)g.nEl*nm) return;
    const int i = (int)(task%nm), e = (int)(task/nm);
    CG gl = g; gl.E = sE; gl.w = sw;
    if (lane < mp12) {
        cq[lane]          = colop_coef(gl, MIMSEM_V_LINEAR_INV, 0, e, i, 0, lane, nullptr, nullptr);
        cq[mp12 + lane]   = colop_coef(gl, MIMSEM_V_LINEAR_RT, MIMSEM_FLAG_VERT, e, i, 0, lane, a.theta, nullptr);
        cq[2*mp12 + lane] = colop_coef(gl, MIMSEM_V_LINEAR_RT, MIMSEM_FLAG_VERT, e, i, 0, lane, a.rho, nullptr);
    }
    wsync();
    const size_t off = ((size_t)e*nm + i)*nn;
    WBk::assemble(A, cq, sE, sEE, n, mp1, lane);
    WBk::inverse(A, Ai, lane);
    WBk::store(a.Ainv + off, Ai, lane);
    WBk::assemble(T, cq + mp12, sE, sEE, n, mp1, lane);
    WBk::assemble(Rr, cq + 2*mp12, sE, sEE, n, mp1, lane);
    WBk::mul(t1, Ai, Rr, 1.0, lane);                       // X = VA_inv VA(rho)
    WBk::store(a.X + off, t1, lane);
    WBk::load(B0, a.B + ((size_t)e*nk + i)*nn, lane);
    WBk::load(B1, a.B + ((size_t)e*nk + i + 1)*nn, lane);
    // grad f = A^-1 (B_{i+1} f_{i+1} - B_i f_i)  for f = pi, eta
    for (int which = 0; which < 2; which++) {
        const double* f = which ? a.eta : a.pi;
        if (lane < N2) {
            const double* f0 = f + ((size_t)e*nk + i)*N2; const double* f1 = f0 + N2;
            double s = 0.0;
#pragma unroll
            for (int m = 0; m < N2; m++) s += B1[lane*N2 + m]*f1[m];
            double s0 = 0.0;
#pragma unroll
            for (int m = 0; m < N2; m++) s0 += B0[lane*N2 + m]*f0[m];
            v[lane] = s - s0;
        }
        wsync();
        if (lane < N2) {
            double s = 0.0;
#pragma unroll
            for (int m = 0; m < N2; m++) s += Ai[lane*N2 + m]*v[m];
            (which ? a.geta : a.gpi)[((size_t)e*nm + i)*N2 + lane] = s;
        }
        wsync();
    }
    WBk::mul(t1, Ai, B0, 1.0, lane);                       // A^-1 B_i
    WBk::mul(Rr, T, t1, -a.hdt, lane);                     // G(i,i)   = -h T A^-1 B_i        (Rr is free now)
    WBk::store(a.Gl + off, Rr, lane);
    wsync();
    WBk::mul(t1, Ai, B1, 1.0, lane);                       // A^-1 B_{i+1}
    WBk::mul(Rr, T, t1, +a.hdt, lane);                     // G(i,i+1) = +h T A^-1 B_{i+1}
    WBk::store(a.Gu + off, Rr, lane);
}

// ---- the same two fused kernels on ROW-PER-LANE block algebra: 16 lanes per task (4 tasks per wavefront), every block of a task
// lives in registers (its row r in lane r), Gauss-Jordan as in k_block_thomas_rows (DPP row rotations for the pivot search,
// bpermute broadcasts of the pivot row, implicit pivoting); LDS only holds the right-hand operand of a product.
template <int N>
struct RowBlocks {
    static constexpr int N2 = N*N, nn = N2*N2, MP1 = N + 1, MP12 = MP1*MP1, GW = 16;
    static_assert(N2 <= GW, "one lane per block row");
    // my row of W^T diag(c) W (same association order as WaveBlocks::assemble)
    __device__ static void assemble(double (&M)[N2], const double* c, const double* sE, int r) {
        const int iy = r/N, ix = r%N;
#pragma unroll
        for (int jy = 0; jy < N; jy++) {
            double t1[MP1];
#pragma unroll
            for (int qx = 0; qx < MP1; qx++) {
                double s = 0.0;
#pragma unroll
                for (int qy = 0; qy < MP1; qy++) s += (sE[qy*N + iy]*c[qy*MP1 + qx])*sE[qy*N + jy];
                t1[qx] = s;
            }
#pragma unroll
            for (int jx = 0; jx < N; jx++) {
                double s = 0.0;
#pragma unroll
                for (int qx = 0; qx < MP1; qx++) s += (sE[qx*N + ix]*sE[qx*N + jx])*t1[qx];
                M[jy*N + jx] = s;
            }
        }
    }
    // T <- my row of T^-1 (natural row order); sI: this task's nn doubles of LDS scratch
    __device__ static void inverse(double (&T)[N2], double* sI, int r, bool act) {
        bool used = !act;
        int myc = 0, piv[N2];
#pragma unroll
        for (int c = 0; c < N2; c++) {
            double cand = used ? -1.0 : fabs(T[c]);
            int bl = r;
#define MIMSEM_ARGMAX_ROUND(K) { const double oc = row_ror<K>(cand); const int ol = row_ror<K>(bl); \
                                 if (oc > cand || (oc == cand && ol < bl)) { cand = oc; bl = ol; } }
            MIMSEM_ARGMAX_ROUND(1) MIMSEM_ARGMAX_ROUND(2) MIMSEM_ARGMAX_ROUND(4) MIMSEM_ARGMAX_ROUND(8)
#undef MIMSEM_ARGMAX_ROUND
            piv[c] = bl;
            double pr[N2];
#pragma unroll
            for (int j = 0; j < N2; j++) pr[j] = __shfl(T[j], bl, GW);
            const double pinv = 1.0/pr[c];
            if (r == bl) {
#pragma unroll
                for (int j = 0; j < N2; j++) T[j] = pr[j]*pinv;
                T[c] = pinv; used = true; myc = c;
            } else {
                const double fct = T[c];
#pragma unroll
                for (int j = 0; j < N2; j++) T[j] -= fct*(pr[j]*pinv);
                T[c] = -fct*pinv;
            }
        }
        wsync_lds();                                 // earlier readers of sI are done
        if (act) {
#pragma unroll
            for (int c = 0; c < N2; c++) sI[myc*N2 + piv[c]] = T[c];
        }
        wsync_lds();
        const int rr = act ? r : 0;
#pragma unroll
        for (int m = 0; m < N2; m++) T[m] = sI[rr*N2 + m];
    }
    // publish my row of a block as the right-hand operand of the next product(s)
    __device__ static void put(double* sB, const double (&R)[N2], int r, bool act) {
        wsync_lds();
        if (act) {
#pragma unroll
            for (int j = 0; j < N2; j++) sB[r*N2 + j] = R[j];
        }
        wsync_lds();
    }
    // my row of alpha * A . B  (A: my row in registers, B: published in LDS)
    __device__ static void mul(double (&C)[N2], const double (&A)[N2], const double* sB, double alpha) {
#pragma unroll
        for (int j = 0; j < N2; j++) C[j] = 0.0;
#pragma unroll
        for (int m = 0; m < N2; m++) {
            const double am = A[m];
#pragma unroll
            for (int j = 0; j < N2; j++) C[j] += am*sB[m*N2 + j];
        }
#pragma unroll
        for (int j = 0; j < N2; j++) C[j] *= alpha;
    }
    __device__ static void store(double* dst, const double (&R)[N2], int r, bool act) {
        if (!act) return;
#pragma unroll
        for (int j = 0; j < N2; j++) dst[r*N2 + j] = R[j];
    }
    // global <-> LDS as ONE contiguous run per block: the 16 lanes of the task move nn consecutive doubles (a lane reading or
    // writing its own row touches 9 doubles at a stride of 9 -- nine partially used transactions per row)
    static constexpr int CHUNKS = (nn + GW - 1)/GW;
    __device__ static void copy_in(double* sDst, const double* src, int r, bool live) {           // no sync: caller publishes
#pragma unroll
        for (int i = 0; i < CHUNKS; i++) { const int x = r + GW*i; if (x < nn) sDst[x] = live ? src[x] : 0.0; }
    }
    __device__ static void row_of(double (&R)[N2], const double* sSrc, int rr) {
#pragma unroll
        for (int j = 0; j < N2; j++) R[j] = sSrc[rr*N2 + j];
    }
    // my row -> scratch block in LDS -> one contiguous run in global memory
    __device__ static void store_block(double* dst, double* sTmp, const double (&R)[N2], int r, bool act, bool live) {
        put(sTmp, R, r, act);
        if (live) {
#pragma unroll
            for (int i = 0; i < CHUNKS; i++) { const int x = r + GW*i; if (x < nn) dst[x] = sTmp[x]; }
        }
    }
};

template <int N>
__global__ __launch_bounds__(64) void k_schur_levels_rows(FusedArgs a) {
    using RB = RowBlocks<N>;
    constexpr int N2 = RB::N2, nn = RB::nn, MP1 = RB::MP1, MP12 = RB::MP12, TPB = 4;
    __shared__ double sE[MP1*N], sw[MP1], cq[TPB][3*MP12], sB[TPB][nn], sT[TPB][nn], sI[TPB][nn];
    const CG& g = a.g;
    const int lane = threadIdx.x, t = lane/16, r = lane%16;
    if (lane < MP1*N) sE[lane] = g.E[lane];
    if (lane < MP1) sw[lane] = g.w[lane];
    __syncthreads();
    const long long task0 = (long long)blockIdx.x*TPB + t, ntask = (long long)g.nEl*g.nk;
    const bool live = task0 < ntask, act = live && r < N2;
    const long long task = live ? task0 : ntask - 1;
    const int k = (int)(task%g.nk), e = (int)(task/g.nk), rr = r < N2 ? r : 0;
    CG gl = g; gl.E = sE; gl.w = sw;
    for (int q = r; q < MP12; q += 16) {
        cq[t][q]          = colop_coef(gl, MIMSEM_V_CONST, 0, e, k, 0, q, nullptr, nullptr);
        cq[t][MP12 + q]   = colop_coef(gl, MIMSEM_V_CONST_RHO, 0, e, k, 0, q, a.pi, nullptr);
        cq[t][2*MP12 + q] = colop_coef(gl, MIMSEM_V_CONST_RHO, 0, e, k, 0, q, a.rho, nullptr);
    }
    wsync_lds();
    const size_t off = ((size_t)e*g.nk + k)*nn;
    double Bk[N2], T[N2], t1[N2], t2[N2];
    RB::assemble(Bk, cq[t], sE, rr);
    RB::store(a.B + off, Bk, r, act);
    RB::put(sB[t], Bk, r, act);                          // B stays published for the whole task
#pragma unroll
    for (int j = 0; j < N2; j++) T[j] = Bk[j];
    RB::inverse(T, sI[t], r, act);
    RB::store(a.Binv + off, T, r, act);
    if (a.F_eta) {                                       // (VertSolve.cpp:772, :777: VB_inv F_eta, VB_inv F_rho)
        const double* fe = a.F_eta + ((size_t)e*g.nk + k)*N2; const double* fr = a.F_rho + ((size_t)e*g.nk + k)*N2;
        double se = 0.0, sr = 0.0;
#pragma unroll
        for (int m = 0; m < N2; m++) { se += T[m]*fe[m]; sr += T[m]*fr[m]; }
        if (act) { a.tE[((size_t)e*g.nk + k)*N2 + r] = se; a.tR[((size_t)e*g.nk + k)*N2 + r] = sr; }
    }
#pragma unroll 1
    for (int which = 0; which < 2; which++) {
        RB::assemble(T, cq[t] + (1 + which)*MP12, sE, rr);
        RB::inverse(T, sI[t], r, act);
        RB::mul(t1, T, sB[t], 1.0);                      // B(f)^-1 B
        RB::put(sT[t], t1, r, act);
        RB::mul(t2, Bk, sT[t], 1.0);                     // B (B(f)^-1 B)
        RB::store((which ? a.Nrho : a.Npi) + off, t2, r, act);
    }
}

template <int N>
__global__ __launch_bounds__(64) void k_schur_interfaces_rows(FusedArgs a) {
    using RB = RowBlocks<N>;
    constexpr int N2 = RB::N2, nn = RB::nn, MP1 = RB::MP1, MP12 = RB::MP12, TPB = 4;
    __shared__ double sE[MP1*N], sw[MP1], cq[TPB][3*MP12], sB[TPB][nn], sT[TPB][nn], sI[TPB][nn], sv[TPB][N2];
    const CG& g = a.g;
    const int nk = g.nk, nm = nk - 1;
    const int lane = threadIdx.x, t = lane/16, r = lane%16;
    if (lane < MP1*N) sE[lane] = g.E[lane];
    if (lane < MP1) sw[lane] = g.w[lane];
    __syncthreads();
    const long long task0 = (long long)blockIdx.x*TPB + t, ntask = (long long)g.nEl*nm;
    const bool live = task0 < ntask, act = live && r < N2;
    const long long task = live ? task0 : ntask - 1;
    const int i = (int)(task%nm), e = (int)(task/nm), rr = r < N2 ? r : 0;
    CG gl = g; gl.E = sE; gl.w = sw;
    for (int q = r; q < MP12; q += 16) {
        cq[t][q]          = colop_coef(gl, MIMSEM_V_LINEAR_INV, 0, e, i, 0, q, nullptr, nullptr);
        cq[t][MP12 + q]   = colop_coef(gl, MIMSEM_V_LINEAR_RT, MIMSEM_FLAG_VERT, e, i, 0, q, a.theta, nullptr);
        cq[t][2*MP12 + q] = colop_coef(gl, MIMSEM_V_LINEAR_RT, MIMSEM_FLAG_VERT, e, i, 0, q, a.rho, nullptr);
    }
    wsync_lds();
    const size_t off = ((size_t)e*nm + i)*nn;
    // ordered so that few blocks are live at a time (each is 2 N2 VGPRs): A^-1 first, VA(rho) only for X, VA(theta) last
    double Ai[N2], R[N2], t1[N2];
    RB::assemble(Ai, cq[t], sE, rr);
    RB::inverse(Ai, sI[t], r, act);
    RB::store(a.Ainv + off, Ai, r, act);
    RB::assemble(R, cq[t] + 2*MP12, sE, rr);
    RB::put(sT[t], R, r, act);
    RB::mul(t1, Ai, sT[t], 1.0);                          // X = VA_inv VA(rho)
    RB::store(a.X + off, t1, r, act);
    {
        double B0[N2], B1[N2];
        const double* b0 = a.B + ((size_t)e*nk + i)*nn + rr*N2;
#pragma unroll
        for (int j = 0; j < N2; j++) { B0[j] = b0[j]; B1[j] = b0[nn + j]; }
        // grad f = A^-1 (B_{i+1} f_{i+1} - B_i f_i)  for f = pi, eta
#pragma unroll 1
        for (int which = 0; which < 2; which++) {
            const double* f0 = (which ? a.eta : a.pi) + ((size_t)e*nk + i)*N2;
            double s = 0.0, s0 = 0.0;
#pragma unroll
            for (int m = 0; m < N2; m++) s += B1[m]*f0[N2 + m];
#pragma unroll
            for (int m = 0; m < N2; m++) s0 += B0[m]*f0[m];
            wsync_lds();
            if (act) sv[t][r] = s - s0;
            wsync_lds();
            double gsum = 0.0;
#pragma unroll
            for (int m = 0; m < N2; m++) gsum += Ai[m]*sv[t][m];
            if (act) (which ? a.geta : a.gpi)[((size_t)e*nm + i)*N2 + r] = gsum;
        }
        RB::put(sT[t], B0, r, act);
        RB::mul(t1, Ai, sT[t], 1.0);                      // A^-1 B_i
        RB::put(sB[t], t1, r, act);
        RB::put(sT[t], B1, r, act);
        RB::mul(t1, Ai, sT[t], 1.0);                      // A^-1 B_{i+1}
        RB::put(sI[t], t1, r, act);                       // (the inverse scratch is free by now)
    }
    RB::assemble(R, cq[t] + MP12, sE, rr);                // T = VA(theta)
    RB::mul(t1, R, sB[t], -a.hdt);                        // G(i,i)   = -h T A^-1 B_i
    RB::store(a.Gl + off, t1, r, act);
    RB::mul(t1, R, sI[t], +a.hdt);                        // G(i,i+1) = +h T A^-1 B_{i+1}
    RB::store(a.Gu + off, t1, r, act);
}

// level k of column e, after the interface factors exist: the two DIV blocks of the row (scaled by the lumped inverse of L_eta)
// and the three blocks of the Helmholtz row  L_pi = N_pi - gam DIV G_pi  (VertSolve.cpp:754-767) -- what the pipeline of wide
// kernels does with two batched products, a scaling pass and k_helmholtz_rows; no intermediate block leaves the chip.
struct RowsArgs {
    int nEl, nk; double hdt, gam;
    const double *Nrho, *Npi, *X, *Cw, *rl, *Gl, *Gu;
    double *DIVl, *DIVu, *L;
    // optional: F_pi = -F_pi + gam DIV F_u - gam CM F_rho - gam F_eta  (:775-780), tR = VB^-1 F_rho, F_u already updated
    double* F_pi = nullptr; const double *F_u = nullptr, *tR = nullptr, *F_eta = nullptr;
};
template <int N>
__global__ __launch_bounds__(64) void k_schur_rows(RowsArgs a) {
    using RB = RowBlocks<N>;
    constexpr int N2 = RB::N2, nn = RB::nn, TPB = 4;
    __shared__ double sX[TPB][2][nn], sGl[TPB][2][nn], sGu[TPB][2][nn];      // [.][0]: interface k-1, [.][1]: interface k
    const int nk = a.nk, nm = nk - 1;
    const int lane = threadIdx.x, t = lane/16, r = lane%16;
    const long long task0 = (long long)blockIdx.x*TPB + t, ntask = (long long)a.nEl*nk;
    const bool live = task0 < ntask, act = live && r < N2;
    const long long task = live ? task0 : ntask - 1;
    const int k = (int)(task%nk), e = (int)(task/nk), rr = r < N2 ? r : 0;
    const bool lo = k > 0, hi = k < nk - 1;
    {   // the six right-hand operand blocks come in as contiguous runs (16 lanes x 8 B): measured 358 -> 308 us against row-wise loads
        const size_t om = ((size_t)e*nm + (lo ? k - 1 : 0))*nn, ok = ((size_t)e*nm + (hi ? k : 0))*nn;
        RB::copy_in(sX[t][0], a.X + om, r, live && lo);   RB::copy_in(sX[t][1], a.X + ok, r, live && hi);
        RB::copy_in(sGl[t][0], a.Gl + om, r, live && lo); RB::copy_in(sGl[t][1], a.Gl + ok, r, live && hi);
        RB::copy_in(sGu[t][0], a.Gu + om, r, live && lo); RB::copy_in(sGu[t][1], a.Gu + ok, r, live && hi);
    }
    const size_t offk = ((size_t)e*nk + k)*nn + rr*N2;
    double Nr[N2], Dl[N2], Du[N2], tt[N2], t2[N2];
#pragma unroll
    for (int j = 0; j < N2; j++) Nr[j] = a.Nrho[offk + j];
    wsync_lds();
    // DIV(k,k-1) = (-h N_rho,k X_{k-1} + h C_{k,0}) diag(rlump_{k-1}),  DIV(k,k) = (+h N_rho,k X_k + h C_{k,1}) diag(rlump_k)
    RB::mul(tt, Nr, sX[t][0], -a.hdt);
    {
        const double* c0 = a.Cw + ((size_t)e*2*nk + 2*k)*nn + rr*N2;
        const double* rl = a.rl + ((size_t)e*nm + (lo ? k - 1 : 0))*N2;
#pragma unroll
        for (int j = 0; j < N2; j++) Dl[j] = lo ? (tt[j] + a.hdt*c0[j])*rl[j] : 0.0;
    }
    RB::mul(tt, Nr, sX[t][1], +a.hdt);
    {
        const double* c1 = a.Cw + ((size_t)e*2*nk + 2*k + 1)*nn + rr*N2;
        const double* rl = a.rl + ((size_t)e*nm + (hi ? k : 0))*N2;
#pragma unroll
        for (int j = 0; j < N2; j++) Du[j] = hi ? (tt[j] + a.hdt*c1[j])*rl[j] : 0.0;
    }
    RB::store(a.DIVl + ((size_t)e*nk + k)*nn, Dl, r, act);
    RB::store(a.DIVu + ((size_t)e*nk + k)*nn, Du, r, act);
    if (a.F_pi) {
        double div = 0.0, cm = 0.0;
        if (lo) { const double* v = a.F_u + ((size_t)e*nm + k - 1)*N2;
#pragma unroll
                  for (int p = 0; p < N2; p++) div += Dl[p]*v[p]; }
        if (hi) { const double* v = a.F_u + ((size_t)e*nm + k)*N2;
#pragma unroll
                  for (int p = 0; p < N2; p++) div += Du[p]*v[p]; }
        const double* v = a.tR + ((size_t)e*nk + k)*N2;
#pragma unroll
        for (int p = 0; p < N2; p++) cm += Nr[p]*v[p];
        if (act) {
            const size_t x = ((size_t)e*nk + k)*N2 + r;
            double f = -1.0*a.F_pi[x];
            f += (+1.0*a.gam)*div;
            f += (-1.0*a.gam)*cm;
            f += (-1.0*a.gam)*a.F_eta[x];
            a.F_pi[x] = f;
        }
    }
    double* Lk = a.L + ((size_t)e*nk + k)*3*nn;
    RB::mul(tt, Dl, sGl[t][0], -a.gam);                   // L(k,k-1) = -gam DIV(k,k-1) G(k-1,k-1)
    RB::store(Lk, tt, r, act);
    RB::mul(tt, Du, sGu[t][1], -a.gam);                   // L(k,k+1) = -gam DIV(k,k)   G(k,k+1)
    RB::store(Lk + 2*nn, tt, r, act);
    RB::mul(tt, Dl, sGu[t][0], 1.0);                      // DIV(k,k-1) G(k-1,k)
    RB::mul(t2, Du, sGl[t][1], 1.0);                      // DIV(k,k)   G(k,k)
#pragma unroll
    for (int j = 0; j < N2; j++) tt[j] = (-1.0*a.gam)*(tt[j] + t2[j]) + a.Npi[offk + j];
    RB::store(Lk + nn, tt, r, act);
}

// interface i of column e: the scalar lumped inverse of  L_eta = VA - (G_rt VB_inv) A_eta  (VertSolve.cpp:742-751).  The block
// M1_i = h (R^t_i Binv_i + R^b_{i+1} Binv_{i+1}) is only ever needed for this diagonal, so it stays in registers.
struct LumpArgs {
    CG g; double hdt;
    const double *R2, *Binv, *Cw;          // RHODPI blocks [nEl][2 nk][nn], VB^-1 [nEl][nk][nn], CONLIN_W blocks [nEl][2 nk][nn]
    double* rl;                            // [nEl][nk-1][n2]
    double* F_u = nullptr; const double* tE = nullptr;      // optional: F_u -= (G_rt VB_inv) F_eta  (:772-773), tE = VB^-1 F_eta
};
template <int N>
__global__ __launch_bounds__(64) void k_schur_lump_rows(LumpArgs a) {
    using RB = RowBlocks<N>;
    constexpr int N2 = RB::N2, nn = RB::nn, MP1 = RB::MP1, MP12 = RB::MP12, TPB = 4;
    __shared__ double sE[MP1*N], sw[MP1], cq[TPB][MP12], sB0[TPB][nn], sB1[TPB][nn], sC[TPB][nn];
    const CG& g = a.g;
    const int nk = g.nk, nm = nk - 1;
    const int lane = threadIdx.x, t = lane/16, r = lane%16;
    if (lane < MP1*N) sE[lane] = g.E[lane];
    if (lane < MP1) sw[lane] = g.w[lane];
    __syncthreads();
    const long long task0 = (long long)blockIdx.x*TPB + t, ntask = (long long)g.nEl*nm;
    const bool live = task0 < ntask, act = live && r < N2;
    const long long task = live ? task0 : ntask - 1;
    const int i = (int)(task%nm), e = (int)(task/nm), rr = r < N2 ? r : 0;
    CG gl = g; gl.E = sE; gl.w = sw;
    for (int q = r; q < MP12; q += 16) cq[t][q] = colop_coef(gl, MIMSEM_V_LINEAR, 0, e, i, 0, q, nullptr, nullptr);
    RB::copy_in(sB0[t], a.Binv + ((size_t)e*nk + i)*nn, r, live);
    RB::copy_in(sB1[t], a.Binv + ((size_t)e*nk + i + 1)*nn, r, live);
    RB::copy_in(sC[t], a.Cw + ((size_t)e*2*nk + 2*i + 1)*nn, r, live);
    double Rt[N2], Rb[N2], m0[N2], m1[N2];
    const double* rt = a.R2 + ((size_t)e*2*nk + 2*i + 1)*nn + rr*N2;
#pragma unroll
    for (int j = 0; j < N2; j++) { Rt[j] = rt[j]; Rb[j] = rt[nn + j]; }          // R^b_{i+1} = R2[2(i+1)] is the next stored block
    wsync_lds();
    RB::mul(m0, Rt, sB0[t], 1.0);
    RB::mul(m1, Rb, sB1[t], 1.0);
    double s = 0.0, va = 0.0;
#pragma unroll
    for (int k = 0; k < N2; k++) s += (a.hdt*(m0[k] + m1[k]))*(a.hdt*sC[t][k*N2 + rr]);
#pragma unroll
    for (int q = 0; q < MP12; q++) {
        const double wq = sE[(q%MP1)*N + rr%N]*sE[(q/MP1)*N + rr/N];
        va += (wq*cq[t][q])*wq;
    }
    if (act) a.rl[((size_t)e*nm + i)*N2 + r] = 1.0/(-1.0*s + va);
    if (a.F_u) {
        const double* v0 = a.tE + ((size_t)e*nk + i)*N2;
        double su = 0.0;
#pragma unroll
        for (int k = 0; k < N2; k++) su += Rt[k]*v0[k] + Rb[k]*v0[N2 + k];
        if (act) a.F_u[((size_t)e*nm + i)*N2 + r] += -1.0*(a.hdt*su);
    }
}

template <int N>
int launch_schur_lump(mimsem_ctx* c, const LumpArgs& a) {
    const long long ti = (long long)a.g.nEl*(a.g.nk - 1);
    hipLaunchKernelGGL((k_schur_lump_rows<N>), dim3((unsigned)((ti + 3)/4)), dim3(64), 0, c->stream, a);
    MIMSEM_HIP_TRY(hipGetLastError());
    return MIMSEM_OK;
}

// level k of column e, after the Helmholtz solve and d_u: the rest of the back substitution in one pass (VertSolve.cpp:800-815)
//   F_eta <- -(F_eta + A_eta d_u),  d_eta = VB^-1 F_eta;   F_rho <- -(F_rho + h VB (V10 X d_u)),  d_rho = VB^-1 F_rho
// -- five launches of the wide pipeline (two of them re-reading VB^-1); one lane per block row, vectors exchanged through LDS.
struct BackArgs {
    int nEl, nk; double hdt;
    const double *Cw, *X, *B, *Binv, *d_u;
    double *F_eta, *F_rho, *d_eta, *d_rho;
};
template <int N>
__global__ __launch_bounds__(64) void k_schur_backsub_rows(BackArgs a) {
    constexpr int N2 = N*N, nn = N2*N2, TPB = 4;
    __shared__ double sv[TPB][3][N2];
    const int nk = a.nk, nm = nk - 1;
    const int lane = threadIdx.x, t = lane/16, r = lane%16;
    const long long task0 = (long long)blockIdx.x*TPB + t, ntask = (long long)a.nEl*nk;
    const bool live = task0 < ntask, act = live && r < N2;
    const long long task = live ? task0 : ntask - 1;
    const int k = (int)(task%nk), e = (int)(task/nk), rr = r < N2 ? r : 0;
    const bool lo = k > 0, hi = k < nk - 1;
    const double* dum = a.d_u + ((size_t)e*nm + (lo ? k - 1 : 0))*N2;      // d_u on the interface below / above this level
    const double* duk = a.d_u + ((size_t)e*nm + (hi ? k : 0))*N2;
    const size_t x = ((size_t)e*nk + k)*N2 + rr;
    // F_eta += A_eta d_u   (A_eta = h CONLIN_W, stored [k][2]: (k,k-1), (k,k))
    double s = 0.0, xm = 0.0, xk = 0.0;
    if (lo) {
        const double* m = a.Cw + ((size_t)e*2*nk + 2*k)*nn + rr*N2; const double* xr = a.X + ((size_t)e*nm + k - 1)*nn + rr*N2;
#pragma unroll
        for (int p = 0; p < N2; p++) { s += (a.hdt*m[p])*dum[p]; xm += xr[p]*dum[p]; }
    }
    if (hi) {
        const double* m = a.Cw + ((size_t)e*2*nk + 2*k + 1)*nn + rr*N2; const double* xr = a.X + ((size_t)e*nm + k)*nn + rr*N2;
#pragma unroll
        for (int p = 0; p < N2; p++) { s += (a.hdt*m[p])*duk[p]; xk += xr[p]*duk[p]; }
    }
    const double fe = -1.0*(a.F_eta[x] + s);
    if (act) { a.F_eta[x] = fe; sv[t][0][r] = fe; sv[t][1][r] = xk - xm; }       // (V10 X d_u)_k = X_k du_k - X_{k-1} du_{k-1}
    wsync_lds();
    const double* bi = a.Binv + ((size_t)e*nk + k)*nn + rr*N2;
    const double* bk = a.B + ((size_t)e*nk + k)*nn + rr*N2;
    double Bi[N2], de = 0.0, sb = 0.0;
#pragma unroll
    for (int m = 0; m < N2; m++) { Bi[m] = bi[m]; de += Bi[m]*sv[t][0][m]; sb += bk[m]*sv[t][1][m]; }
    const double fr = -1.0*(a.F_rho[x] + a.hdt*sb);
    if (act) { a.d_eta[x] = de; a.F_rho[x] = fr; sv[t][2][r] = fr; }
    wsync_lds();
    double dr = 0.0;
#pragma unroll
    for (int m = 0; m < N2; m++) dr += Bi[m]*sv[t][2][m];
    if (act) a.d_rho[x] = dr;
}
template <int N>
int launch_schur_backsub(mimsem_ctx* c, const BackArgs& a) {
    const long long tl = (long long)a.nEl*a.nk;
    hipLaunchKernelGGL((k_schur_backsub_rows<N>), dim3((unsigned)((tl + 3)/4)), dim3(64), 0, c->stream, a);
    MIMSEM_HIP_TRY(hipGetLastError());
    return MIMSEM_OK;
}

template <int N>
int launch_schur_rows(mimsem_ctx* c, const RowsArgs& a) {
    const long long tl = (long long)a.nEl*a.nk;
    hipLaunchKernelGGL((k_schur_rows<N>), dim3((unsigned)((tl + 3)/4)), dim3(64), 0, c->stream, a);
    MIMSEM_HIP_TRY(hipGetLastError());
    return MIMSEM_OK;
}

template <int N>
int launch_schur_fused_rows(mimsem_ctx* c, const FusedArgs& a) {
    const CG& g = a.g;
    const long long tl = (long long)g.nEl*g.nk, ti = (long long)g.nEl*(g.nk - 1);
    hipLaunchKernelGGL((k_schur_levels_rows<N>), dim3((unsigned)((tl + 3)/4)), dim3(64), 0, c->stream, a);
    hipLaunchKernelGGL((k_schur_interfaces_rows<N>), dim3((unsigned)((ti + 3)/4)), dim3(64), 0, c->stream, a);
    MIMSEM_HIP_TRY(hipGetLastError());
    return MIMSEM_OK;
}

template <int N2>
int launch_schur_fused(mimsem_ctx* c, const FusedArgs& a) {
    const CG& g = a.g;
    const int nn = N2*N2;
    const size_t common = (size_t)(g.mp1*g.n + g.mp1*N2 + 8);
    {
        const size_t lds = (common + 4*(size_t)(3*g.mp12 + 5*nn))*sizeof(double);
        const long long tasks = (long long)g.nEl*g.nk;
        if (lds > 64*1024) MIMSEM_HIP_TRY(hipFuncSetAttribute((const void*)k_schur_levels<N2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((k_schur_levels<N2>), dim3((unsigned)((tasks + 3)/4)), dim3(256), lds, c->stream, a);
        MIMSEM_HIP_TRY(hipGetLastError());
    }
    {
        const size_t lds = (common + 4*(size_t)(3*g.mp12 + 7*nn + 2*N2))*sizeof(double);
        const long long tasks = (long long)g.nEl*(g.nk - 1);
        if (lds > 64*1024) MIMSEM_HIP_TRY(hipFuncSetAttribute((const void*)k_schur_interfaces<N2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((k_schur_interfaces<N2>), dim3((unsigned)((tasks + 3)/4)), dim3(256), lds, c->stream, a);
        MIMSEM_HIP_TRY(hipGetLastError());
    }
    return MIMSEM_OK;
}

// ---- thread-per-block fused EOS block: N = B . B(f)^-1 . B with everything but B in registers (Assemble_EOS_Block :1162-1196).
// One THREAD per (column, level): the coefficient row, the sum-factorised W^T diag(c) W, its unpivoted Gauss-Jordan inverse
// (the design of k_block_inverse_reg) and the two products run without any inter-lane traffic; B is read from the stored CONST
// blocks.  Blocks whose natural-order pivots are too small are flagged and redone by the generic path.
// ---- thread-per-block register kernels (p = 3 specialisations of the Schur factors) ------------------------------------------
// One THREAD per block/task: coefficient rows, sum-factorised W^T diag(c) W, unpivoted Gauss-Jordan (the design of
// k_block_inverse_reg) and the small products run without any inter-lane traffic.  A block whose natural-order pivots are too
// small is redone with partial pivoting on separate arrays in private memory, so the fast path keeps everything in registers.
template <int N>
__device__ __forceinline__ void reg_assemble(const double (&E)[(N + 1)*N], const double (&c)[(N + 1)*(N + 1)], double (&P)[N*N*N*N]) {
    constexpr int N2 = N*N, MP1 = N + 1;
#pragma unroll
    for (int iy = 0; iy < N; iy++)
#pragma unroll
        for (int jy = 0; jy < N; jy++) {
            double t1[MP1];
#pragma unroll
            for (int qx = 0; qx < MP1; qx++) {
                double s = 0.0;
#pragma unroll
                for (int qy = 0; qy < MP1; qy++) s += (E[qy*N + iy]*c[qy*MP1 + qx])*E[qy*N + jy];
                t1[qx] = s;
            }
#pragma unroll
            for (int ix = 0; ix < N; ix++)
#pragma unroll
                for (int jx = 0; jx < N; jx++) {
                    double s = 0.0;
#pragma unroll
                    for (int qx = 0; qx < MP1; qx++) s += (E[qx*N + ix]*E[qx*N + jx])*t1[qx];
                    P[(iy*N + ix)*N2 + jy*N + jx] = s;
                }
        }
}
// in place; returns false when a natural-order pivot is too small (caller falls back)
template <int N2>
__device__ __forceinline__ bool reg_inverse(double (&P)[N2*N2]) {
    double dmax = 0.0;
#pragma unroll
    for (int i = 0; i < N2; i++) dmax = fmax(dmax, fabs(P[i*N2 + i]));
    bool ok = true;
#pragma unroll
    for (int p = 0; p < N2; p++) {
        const double piv = P[p*N2 + p];
        if (!(fabs(piv) >= 1.0e-8*dmax) || !(fabs(piv) >= 1.0e-12)) ok = false;
        const double pinv = 1.0/piv;
        P[p*N2 + p] = 1.0;
#pragma unroll
        for (int cc = 0; cc < N2; cc++) P[p*N2 + cc] *= pinv;
#pragma unroll
        for (int r = 0; r < N2; r++) {
            if (r == p) continue;
            const double d = P[r*N2 + p];
            P[r*N2 + p] = 0.0;
#pragma unroll
            for (int cc = 0; cc < N2; cc++) P[r*N2 + cc] -= P[p*N2 + cc]*d;
        }
    }
    return ok;
}
// A (private memory, dynamic indexing) -> its inverse in I, partial pivoting
template <int N2>
__device__ __noinline__ void private_inverse(double* A, double* I) {
    for (int t = 0; t < N2*N2; t++) I[t] = (t/N2 == t%N2) ? 1.0 : 0.0;
    for (int col = 0; col < N2; col++) {
        int pr = col; double big = fabs(A[col*N2 + col]);
        for (int r = col + 1; r < N2; r++) if (fabs(A[r*N2 + col]) > big) { big = fabs(A[r*N2 + col]); pr = r; }
        if (pr != col) for (int j = 0; j < N2; j++) {
            double t = A[col*N2 + j]; A[col*N2 + j] = A[pr*N2 + j]; A[pr*N2 + j] = t;
            t = I[col*N2 + j]; I[col*N2 + j] = I[pr*N2 + j]; I[pr*N2 + j] = t;
        }
        const double pinv = 1.0/A[col*N2 + col];
        for (int j = 0; j < N2; j++) { A[col*N2 + j] *= pinv; I[col*N2 + j] *= pinv; }
        for (int r = 0; r < N2; r++) {
            if (r == col) continue;
            const double d = A[r*N2 + col];
            for (int j = 0; j < N2; j++) { A[r*N2 + j] -= d*A[col*N2 + j]; I[r*N2 + j] -= d*I[col*N2 + j]; }
        }
    }
}
// field f (slot k of nkv) interpolated to the quadrature points of element e: the rk/tb loops of VertOps.cpp
template <int N>
__device__ __forceinline__ void reg_interp(const double (&E)[(N + 1)*N], const double* __restrict__ f, int nkv, int e, int k, double (&v)[(N + 1)*(N + 1)]) {
    constexpr int N2 = N*N, MP1 = N + 1;
    const double* fe = f + ((size_t)e*nkv + k)*N2;
    double fv[N2];
#pragma unroll
    for (int j = 0; j < N2; j++) fv[j] = fe[j];
#pragma unroll
    for (int q = 0; q < MP1*MP1; q++) {
        double r = 0.0;
#pragma unroll
        for (int j = 0; j < N2; j++) r += fv[j]*(E[(q%MP1)*N + j%N]*E[(q/MP1)*N + j/N]);
        v[q] = r;
    }
}

// N = B . B(f)^-1 . B   (Assemble_EOS_Block :1162-1196), B read from the stored CONST blocks
template <int N>
__global__ __launch_bounds__(64) void k_eos_block_thread(CG g, const double* __restrict__ f, const double* __restrict__ Bc,
                                                         double* __restrict__ out) {
    constexpr int N2 = N*N, MP1 = N + 1, MP12 = MP1*MP1;
    const long long b = (long long)blockIdx.x*64 + threadIdx.x;
    if (b >= (long long)g.nEl*g.nk) return;
    const int k = (int)(b%g.nk), e = (int)(b/g.nk);
    double E[MP1*N];
#pragma unroll
    for (int t = 0; t < MP1*N; t++) E[t] = g.E[t];
    double c[MP12], rk[MP12];
    reg_interp<N>(E, f, g.nk, e, k, rk);
#pragma unroll
    for (int q = 0; q < MP12; q++) {           // coefficient of CONST_RHO (VertOps.cpp:508-521)
        const double det = g.det[(size_t)e*MP12 + q];
        const size_t gl = ((size_t)k*g.nEl + e)*MP12 + q;
        const double cc = (g.w[q%MP1]*g.w[q/MP1])*(VSCALE/det)*g.tI[gl];
        c[q] = cc*(rk[q]/(g.th[gl]*det));
    }
    double P[N2*N2];
    reg_assemble<N>(E, c, P);
    if (!reg_inverse<N2>(P)) {                 // rare: a coefficient field that is not positive
        double A[N2*N2], I[N2*N2];
        reg_assemble<N>(E, c, A);
        private_inverse<N2>(A, I);
#pragma unroll
        for (int t = 0; t < N2*N2; t++) P[t] = I[t];
    }
    // N = B (Pinv B), one column at a time: v = Pinv B[:,j], N[:,j] = B v
    double B[N2*N2];
    const double* Bb = Bc + (size_t)b*N2*N2;
#pragma unroll
    for (int t = 0; t < N2*N2; t++) B[t] = Bb[t];
    double* o = out + (size_t)b*N2*N2;
#pragma unroll
    for (int j = 0; j < N2; j++) {
        double v[N2];
#pragma unroll
        for (int i = 0; i < N2; i++) {
            double s = 0.0;
#pragma unroll
            for (int m = 0; m < N2; m++) s += P[i*N2 + m]*B[m*N2 + j];
            v[i] = s;
        }
#pragma unroll
        for (int i = 0; i < N2; i++) {
            double s = 0.0;
#pragma unroll
            for (int m = 0; m < N2; m++) s += B[i*N2 + m]*v[m];
            o[i*N2 + j] = s;
        }
    }
}

struct Schur {
    // block arrays (all [nEl][ns][nn])
    BA B, Binv, Ainv, T, Rr, X, Npi, Nrho, R2, C2, DIVl, DIVu, Gl, Gu, M1, L, G, AB0, AB1;
    double *gpi, *geta, *rlump, *tA, *tB, *tC;
    bool fused = false;            // G_pi (Gl, Gu) already built by k_schur_interfaces
    bool rows = false;             // ... by the row-per-lane kernels: DIV and the Helmholtz rows follow in k_schur_rows
    // right-hand sides of the solve (null for mimsem_column_helmholtz_blocks); the row-per-lane kernels update them on the way
    double *F_u = nullptr, *F_pi = nullptr; const double *F_eta = nullptr, *F_rho = nullptr;
    bool rhs_done = false;
};

// assemble every factor of the Helmholtz operator; see the derivation in DESIGN.md ("C5")
int schur_assemble(mimsem_ctx* c, double dt, const double* theta, const double* rho, const double* eta,
                   const double* pi, Schur& S) {
    const int nk = c->nk, nm = nk - 1, n2 = c->es.n2e, nn = n2*n2, nEl = c->nEl, mp12 = c->es.mp12;
    int rc;
    if ((rc = c->ensure_col((long long)nEl*(nk + 1)*(30LL*nn + 8LL*n2 + 4LL*mp12) + colop_ws_doubles(c)))) return rc;
    WS w{c, c->d_col, 0, c->col_doubles};
    double* cq = w.take((long long)nEl*(nk + 1)*2*mp12);
    double* tmpM = w.take((long long)nEl*(nk + 1)*2*nn*2);
    auto ba = [&](int ns) { BA b; b.ns = ns; b.p = w.take((long long)nEl*ns*nn); return b; };
    S.B = ba(nk); S.Binv = ba(nk); S.Ainv = ba(nm); S.T = ba(nm); S.Rr = ba(nm); S.X = ba(nm);
    S.Npi = ba(nk); S.Nrho = ba(nk); S.M1 = ba(nm); S.G = ba(nk); S.AB0 = ba(nm); S.AB1 = ba(nm);
    S.R2.ns = 2*nk; S.R2.p = w.take((long long)nEl*2*nk*nn);   // RHODPI stored [k][2]
    S.C2.ns = 2*nk; S.C2.p = w.take((long long)nEl*2*nk*nn);   // CONLIN_W stored [k][2]
    BA R2 = S.R2, C2 = S.C2;
    S.DIVl = ba(nk); S.DIVu = ba(nk); S.Gl = ba(nm); S.Gu = ba(nm);
    S.L.ns = 3*nk; S.L.p = w.take((long long)nEl*3*nk*nn);
    S.gpi = w.take((long long)nEl*nm*n2); S.geta = w.take((long long)nEl*nm*n2);
    S.rlump = w.take((long long)nEl*nm*n2); S.tA = w.take((long long)nEl*nk*n2); S.tB = w.take((long long)nEl*nk*n2);
    S.tC = w.take((long long)nEl*nk*n2);
    if (w.used > w.cap) return MIMSEM_ERR_STATE;

    // fused assembly of the level / interface factors: default = the row-per-lane kernels where their blocks fit the register
    // file without spilling (2x2, 3x3 faces); "0" = the pipeline of wide kernels, "wave" = the earlier one-wave-per-task kernels
    const char* fmode = getenv("MIMSEM_SCHUR_FUSED");
    if (!fmode) fmode = (n2 == 4 || n2 == 9) ? "rows" : "0";
    if ((n2 == 4 || n2 == 9 || n2 == 16) && strcmp(fmode, "0") != 0) {
        // fused path: one wave per (column, level) / (column, interface), intermediates in LDS.  Measured SLOWER than the
        // pipeline of wide kernels below (2 x 0.70 ms for what costs 1.0 ms there: three wave-cooperative Gauss-Jordan sweeps
        // per task are latency-bound, the thread-per-block register inverse is not) -- profiles/r01_schur_fused_ab.txt; opt-in.
        FusedArgs fa;
        fa.g = make_cg(c); fa.hdt = 0.5*dt; fa.theta = theta; fa.rho = rho; fa.eta = eta; fa.pi = pi;
        fa.B = S.B.p; fa.Binv = S.Binv.p; fa.Npi = S.Npi.p; fa.Nrho = S.Nrho.p;
        fa.Ainv = S.Ainv.p; fa.X = S.X.p; fa.Gl = S.Gl.p; fa.Gu = S.Gu.p; fa.gpi = S.gpi; fa.geta = S.geta;
        if (S.F_u && strcmp(fmode, "wave") != 0) { fa.F_eta = S.F_eta; fa.F_rho = S.F_rho; fa.tE = S.tA; fa.tR = S.tC; }
        const bool rows = strcmp(fmode, "wave") != 0;
        if (rows) rc = (n2 == 4 ? launch_schur_fused_rows<2>(c, fa) : (n2 == 9 ? launch_schur_fused_rows<3>(c, fa) : launch_schur_fused_rows<4>(c, fa)));
        else rc = (n2 == 4 ? launch_schur_fused<4>(c, fa) : (n2 == 9 ? launch_schur_fused<9>(c, fa) : launch_schur_fused<16>(c, fa)));
        if (rc) return rc;
        if ((rc = colop_blocks_into(c, MIMSEM_V_CONLIN_RHODPI, 0, theta, S.gpi, R2.p, cq, tmpM))) return rc;       // :701
        if ((rc = colop_blocks_into(c, MIMSEM_V_CONLIN_W, 0, S.geta, nullptr, C2.p, cq, tmpM))) return rc;         // :730
        S.fused = true; S.rows = rows;
        return MIMSEM_OK;
    }
    // VB, VB_inv, VA_inv   (VertSolve.cpp:690-692)
    if ((rc = colop_blocks_into(c, MIMSEM_V_CONST, 0, nullptr, nullptr, S.B.p, cq, tmpM))) return rc;
    MIMSEM_HIP_TRY(hipMemcpyAsync(S.Binv.p, S.B.p, (size_t)nEl*nk*nn*sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    if ((rc = mimsem_block_inverse_inplace(c, (long long)nEl*nk, n2, S.Binv.p))) return rc;      // AssembleConstInv == Inv(AssembleConst block)
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
    if (c->es.n == 3 && !exp_env("MIMSEM_EOS_WIDE")) {
        // p = 3: thread-per-block register kernel for N = B B(f)^-1 B (93 us per call vs 210 us for coef + inverse + 2 products + copy);
        // the same treatment of the interface factors (A^-1, X, G_pi) needs two blocks live per thread, spills, and measured no gain
        const CG g = make_cg(c);
        const unsigned grid = (unsigned)(((long long)nEl*nk + 63)/64);
        hipLaunchKernelGGL((k_eos_block_thread<3>), dim3(grid), dim3(64), 0, c->stream, g, pi, S.B.p, S.Npi.p);
        hipLaunchKernelGGL((k_eos_block_thread<3>), dim3(grid), dim3(64), 0, c->stream, g, rho, S.B.p, S.Nrho.p);
        MIMSEM_HIP_TRY(hipGetLastError());
        return MIMSEM_OK;
    }
    if ((rc = colop_blocks_into(c, MIMSEM_V_EOS_BLOCK, 0, pi, nullptr, S.Npi.p, cq, tmpM, S.B.p))) return rc;
    if ((rc = colop_blocks_into(c, MIMSEM_V_EOS_BLOCK, 0, rho, nullptr, S.Nrho.p, cq, tmpM, S.B.p))) return rc;
    return MIMSEM_OK;
}

}  // namespace

namespace { bool diag_theta_fused_supported(const mimsem_ctx* c) { return c->es.n >= 1 && c->es.n <= 4 && c->nk >= 2; } }
extern "C" {
int mimsem_column_diag_theta_blend(mimsem_ctx* c, const double* rho, const double* rt, double* theta2, const double* blend2,
                                   double* thetaL, const double* blendL, double wa, double wb);

int mimsem_column_diag_theta(mimsem_ctx* c, int which, const double* rho, const double* rt, double* theta) {
    if (!c || !rho || !rt || !theta || which < 0 || which > 1) return MIMSEM_ERR_ARG;
    if (diag_theta_fused_supported(c) && !exp_env("MIMSEM_DIAG_THETA_WIDE"))      // orders 1..3: one launch on the DPP row algebra (column_newton.inc)
        return which == 0 ? mimsem_column_diag_theta_blend(c, rho, rt, nullptr, nullptr, theta, nullptr, 1.0, 0.0)
                          : mimsem_column_diag_theta_blend(c, rho, rt, theta, nullptr, nullptr, nullptr, 1.0, 0.0);
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
    if (S.rows) {
        // row-per-lane path: the lumped inverse straight from R2, VB^-1 and CONLIN_W (M1 never stored), then DIV + Helmholtz rows
        LumpArgs la{make_cg(c), hdt, R2.p, S.Binv.p, C2.p, S.rlump};
        if (S.F_u) { la.F_u = S.F_u; la.tE = S.tA; }
        if ((rc = (n2 == 4 ? launch_schur_lump<2>(c, la) : (n2 == 9 ? launch_schur_lump<3>(c, la) : launch_schur_lump<4>(c, la))))) return rc;
        RowsArgs ra{nEl, nk, hdt, gam, S.Nrho.p, S.Npi.p, S.X.p, C2.p, S.rlump, S.Gl.p, S.Gu.p, S.DIVl.p, S.DIVu.p, S.L.p};
        if (S.F_u) { ra.F_pi = S.F_pi; ra.F_u = S.F_u; ra.tR = S.tC; ra.F_eta = S.F_eta; S.rhs_done = true; }
        return n2 == 4 ? launch_schur_rows<2>(c, ra) : (n2 == 9 ? launch_schur_rows<3>(c, ra) : launch_schur_rows<4>(c, ra));
    }
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
    //   = 0.5dt ( -+ CM_k B_k X_j ) + 0.5dt C_j , then column-scaled by rlump_j ;  CM_k B_k = N_rho_k Binv_k B_k = N_rho_k
    {
        if ((rc = bmm(c, nk, S.DIVl, S.Nrho, 0, S.X, -1, -hdt, 0))) return rc;         // -h N_rho_k X_{k-1}
        if ((rc = bmm(c, nk, S.DIVu, S.Nrho, 0, S.X, 0, +hdt, 0))) return rc;          // +h N_rho_k X_k
        const double *Cw = C2.p, *rl = S.rlump; double *Dl = S.DIVl.p, *Du = S.DIVu.p;
        if ((rc = each(c, (long long)nEl*nk*2*nn, [=] __device__(long long x) {
            const int ij = (int)(x%nn), jj = ij%n2; long long t = x/nn;
            const int w = (int)(t%2); t /= 2; const int k = (int)(t%nk), e = (int)(t/nk);
            const int j = k - 1 + w;
            double* out = (w ? Du : Dl) + ((size_t)e*nk + k)*nn + ij;
            if (j < 0 || j > nm - 1) { *out = 0.0; return; }
            *out = (*out + hdt*Cw[((size_t)e*2*nk + 2*k + w)*nn + ij])*rl[((size_t)e*nm + j)*n2 + jj];
        }))) return rc;
    }
    // G_pi (Nm x N), row i: Gl_i = (i,i) = -0.5dt T_i Ainv_i B_i ; Gu_i = (i,i+1) = +0.5dt T_i Ainv_i B_{i+1}  (:710-711)
    if (!S.fused) {
        if ((rc = bmm(c, nm, S.AB0, S.Ainv, 0, S.B, 0, 1.0, 0))) return rc;            // Ainv_i B_i
        if ((rc = bmm(c, nm, S.AB1, S.Ainv, 0, S.B, 1, 1.0, 0))) return rc;            // Ainv_i B_{i+1}
        if ((rc = bmm(c, nm, S.Gl, S.T, 0, S.AB0, 0, -hdt, 0))) return rc;
        if ((rc = bmm(c, nm, S.Gu, S.T, 0, S.AB1, 0, +hdt, 0))) return rc;
    }
    // L_pi = N_pi - gam DIV G_pi : block tridiagonal [k][3]                                (:766-767)
    {
        const size_t lds = (size_t)7*nn*sizeof(double);
        hipLaunchKernelGGL(k_helmholtz_rows, dim3((unsigned)(nEl*nk)), dim3(256), lds, c->stream, n2, nk, gam,
                           S.DIVl.p, S.DIVu.p, S.Gl.p, S.Gu.p, S.Npi.p, S.L.p);
        MIMSEM_HIP_TRY(hipGetLastError());
    }
    return MIMSEM_OK;
}

}  // namespace

#include "column_pivot.inc"
#include "column_dpp.inc"
#include "column_penta.inc"
#include "column_newton.inc"

namespace {
// default for orders 1..3: the LDS-free fused path of column_dpp.inc; MIMSEM_SCHUR_FUSED=rows|wave|0 selects the round-1 kernels
bool use_sweep(const mimsem_ctx* c) {
    const char* fmode = getenv("MIMSEM_SCHUR_FUSED");
    return sweep_supported(c) && (!fmode || strcmp(fmode, "sweep") == 0);
}
}  // namespace

extern "C" {
int mimsem_column_helmholtz_blocks(mimsem_ctx* c, double dt, const double* theta, const double* rho,
                                   const double* eta, const double* pi, double* out) {
    if (!c || !theta || !rho || !eta || !pi || !out || c->nk < 2) return MIMSEM_ERR_ARG;
    if (use_sweep(c)) return sweep_helmholtz(c, dt, theta, rho, eta, pi, out);
    Schur S;
    int rc = schur_operator(c, dt, theta, rho, eta, pi, S);
    if (rc) return rc;
    MIMSEM_HIP_TRY(hipMemcpyAsync(out, S.L.p, (size_t)c->nEl*c->nk*3*c->es.n2e*c->es.n2e*sizeof(double),
                                  hipMemcpyDeviceToDevice, c->stream));
    return MIMSEM_OK;
}

// ---- self-test of the half-row block algebra (dpp::RowsH, order 4): every primitive the walks use, on caller-supplied blocks, so that a
// parity failure of a kernel built on it can be told from a failure of the algebra (tests/test_gpu_rows_half.py checks against numpy) ----
namespace {
__global__ __launch_bounds__(64) void k_rows_half_selftest(int ntask, const double* A, const double* B, const double* x, const double* cq,
                                                          const double* Eg, const double* Wg, const double* Earg, double* out) {
    using R = dpp::RowsH;
    constexpr int N2 = R::N2, NC = R::NC, NP = R::NP, MP12 = R::MP12, OUT = 3*256 + 16 + 25 + 16;
    const int lane = threadIdx.x, r = lane%16, c0 = R::c0(lane);
    const int t = min((int)blockIdx.x*R::TPW + R::task(lane), ntask - 1);
    typename R::Lane Ln;
    R::init_lane(Ln, Eg, Wg, lane);
    double E[20];
#pragma unroll
    for (int i = 0; i < 20; i++) E[i] = Earg[i];
    double a[NC], b[NC], c[NC], ai[NC], as[NC];
#pragma unroll
    for (int j = 0; j < NC; j++) { a[j] = A[((size_t)t*N2 + r)*N2 + c0 + j]; b[j] = B[((size_t)t*N2 + r)*N2 + c0 + j]; ai[j] = a[j]; }
    const double xv = x[(size_t)t*N2 + r];
    double cc[NP];
#pragma unroll
    for (int p = 0; p < NP; p++) cc[p] = cq[(size_t)t*MP12 + min(r + 16*p, MP12 - 1)];
    R::mul(c, a, b);
    R::mul_add(c, b, a);                                 // C = A B + B A
    const double y = R::matvec(a, xv, lane);
    R::inverse(ai, lane);
    double fq[NP];
    R::interp(fq, Ln, xv);
    R::assemble(as, Ln, E, cc);
    double* o = out + (size_t)t*OUT;
#pragma unroll
    for (int j = 0; j < NC; j++) { o[r*N2 + c0 + j] = c[j]; o[256 + r*N2 + c0 + j] = ai[j]; o[512 + r*N2 + c0 + j] = as[j]; }
    o[768 + r] = y;
#pragma unroll
    for (int p = 0; p < NP; p++) if (r + 16*p < MP12) o[784 + r + 16*p] = fq[p];
    o[809 + r] = dpp::getH<3>(xv, (lane >> 4) & 1) + 100.0*dpp::xsum(lane < 16 || (lane >= 32 && lane < 48) ? 1.0 : 2.0);      // x[8h + 3] + 100 (1 + 2)
}
}  // namespace
int mimsem_selftest_rows_half(mimsem_ctx* c, int ntask, const double* A, const double* B, const double* x, const double* cq, double* out) {
    if (!c || ntask < 1 || !A || !B || !x || !cq || !out) return MIMSEM_ERR_ARG;
    if (c->es.n != 4) return MIMSEM_ERR_UNSUPPORTED;
    struct { double E[20]; } e;
    for (int i = 0; i < 20; i++) e.E[i] = i < (int)c->tab.E.size() ? c->tab.E[i] : 0.0;
    double* dE = nullptr;
    MIMSEM_HIP_TRY(hipMalloc((void**)&dE, sizeof(e)));
    MIMSEM_HIP_TRY(hipMemcpyAsync(dE, &e, sizeof(e), hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_rows_half_selftest, dim3((unsigned)((ntask + 1)/2)), dim3(64), 0, c->stream, ntask, A, B, x, cq, c->d_E, c->d_W, dE, out);
    MIMSEM_HIP_TRY(hipGetLastError());
    MIMSEM_HIP_TRY(hipStreamSynchronize(c->stream));
    (void)hipFree(dE);
    return MIMSEM_OK;
}

int mimsem_column_flag_for_test(mimsem_ctx* c, const int* columns, int n) {
    if (!c || n < 0 || (n && !columns)) return MIMSEM_ERR_ARG;
    for (int i = 0; i < n; i++) if (columns[i] < 0 || columns[i] >= c->nEl) return MIMSEM_ERR_ARG;
    for (int i = 0; i < n; i++) for (int j = 0; j < i; j++) if (columns[i] == columns[j]) return MIMSEM_ERR_ARG;      // (a column listed twice would be counted twice)
    MIMSEM_HIP_TRY(hipSetDevice(c->device));
    if (c->d_forceflag) { MIMSEM_HIP_TRY(hipStreamSynchronize(c->stream)); (void)hipFree(c->d_forceflag); c->d_forceflag = nullptr; }
    c->n_forceflag = 0;
    if (n == 0) return MIMSEM_OK;
    MIMSEM_HIP_TRY(hipMalloc((void**)&c->d_forceflag, (size_t)n*sizeof(int)));
    MIMSEM_HIP_TRY(hipMemcpy(c->d_forceflag, columns, (size_t)n*sizeof(int), hipMemcpyHostToDevice));
    c->n_forceflag = n;
    return MIMSEM_OK;
}
int mimsem_column_set_pivot_fallback(mimsem_ctx* c, int on) {
    if (!c) return MIMSEM_ERR_ARG;
    c->pivot_fallback = on == 2 ? 2 : (on ? 1 : 0);
    return MIMSEM_OK;
}

int mimsem_column_solve_status(mimsem_ctx* c, int* n_unconverged, int* column_status, double* column_ratio) {
    if (!c || !n_unconverged) return MIMSEM_ERR_ARG;
    *n_unconverged = -1;
    if (!c->colstat_valid || !c->d_colstat) return MIMSEM_OK;           // the last solve ran on a path that keeps no status
    MIMSEM_HIP_TRY(hipStreamSynchronize(c->stream));
    MIMSEM_HIP_TRY(hipMemcpy(n_unconverged, c->d_colstat, sizeof(int), hipMemcpyDeviceToHost));
    if (column_status && c->nEl) MIMSEM_HIP_TRY(hipMemcpy(column_status, c->d_colstat + 1, (size_t)c->nEl*sizeof(int), hipMemcpyDeviceToHost));
    if (column_ratio && c->nEl) MIMSEM_HIP_TRY(hipMemcpy(column_ratio, c->d_colratio, (size_t)c->nEl*sizeof(double), hipMemcpyDeviceToHost));
    return MIMSEM_OK;
}

int mimsem_column_solve_schur_eta(mimsem_ctx* c, double dt,
        const double* theta, const double* rho, const double* eta, const double* pi,
        double* F_u, double* F_rho, double* F_eta, double* F_pi,
        double* d_u, double* d_rho, double* d_eta, double* d_pi) {
    if (!c || !theta || !rho || !eta || !pi || !F_u || !F_rho || !F_eta || !F_pi || !d_u || !d_rho || !d_eta || !d_pi)
        return MIMSEM_ERR_ARG;
    if (c->nk < 2) return MIMSEM_ERR_ARG;
    c->colstat_valid = false;
    if (use_sweep(c)) {
        const int rc = sweep_solve_eta(c, dt, theta, rho, eta, pi, F_u, F_rho, F_eta, F_pi, d_u, d_rho, d_eta, d_pi);
        c->colstat_valid = rc == MIMSEM_OK;
        return rc;
    }
    const int nk = c->nk, nm = nk - 1, n2 = c->es.n2e, nn = n2*n2, nEl = c->nEl;
    const double hdt = 0.5*dt, gam = RD/CV;
    Schur S;
    S.F_u = F_u; S.F_pi = F_pi; S.F_eta = F_eta; S.F_rho = F_rho;
    int rc = schur_operator(c, dt, theta, rho, eta, pi, S);
    if (rc) return rc;
    BA R2 = S.R2, C2 = S.C2;
    // F_u -= (G_rt VB_inv) F_eta                                                         (:772-773)
    if (!S.rhs_done) {
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
    if (!S.rhs_done) {
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
    if ((rc = block_thomas_refined(c, S.L.p, F_pi, d_pi, S.G.p, S.tB, S.tA, S.tC, S.Npi.p))) return rc;     // tA, Npi: free scratch by now
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
    if (S.rows) {
        BackArgs ba{nEl, nk, hdt, C2.p, S.X.p, S.B.p, S.Binv.p, d_u, F_eta, F_rho, d_eta, d_rho};
        return n2 == 4 ? launch_schur_backsub<2>(c, ba) : (n2 == 9 ? launch_schur_backsub<3>(c, ba) : launch_schur_backsub<4>(c, ba));
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

#include "column_schur3_dpp.inc"
#include "column_schur3_sweep.inc"
#include "column_hs.inc"
