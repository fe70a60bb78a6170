// krylov_kernels.hip -- the two dense building blocks of the device Krylov solvers: the classical
// Gram-Schmidt step of GMRES against a basis V [k][n] stored row-wise,
//     h = V w            (k simultaneous dot products, one pass over V)
//     w += alpha V^T h   (k simultaneous axpys, one pass over V)
// Both are HBM/L2-bandwidth bound (k*n*8 bytes); the generic BLAS routes for these tall-skinny shapes were 3-10x slower
// (profiles/r01_sw_kernel_stats.csv).  Deterministic: fixed partition, fixed reduction order, no atomics.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "ctx.hpp"

namespace {

// w[t] += alpha * sum_i h[i] V[i][t]
__global__ __launch_bounds__(256) void k_maxpy(int k, long long n, const double* __restrict__ V, long long ldv,
                                               const double* __restrict__ h, double alpha, double* __restrict__ w) {
    extern __shared__ double sh[];
    for (int i = threadIdx.x; i < k; i += 256) sh[i] = h[i];
    __syncthreads();
    const long long t = (long long)blockIdx.x*256 + threadIdx.x;
    if (t >= n) return;
    double s = 0.0;
    for (int i = 0; i < k; i++) s += sh[i]*V[(size_t)i*ldv + t];
    w[t] += alpha*s;
}

// One classical Gram-Schmidt pass in two launches: the partial sums of h = V w (k_rowdot_partial with w broadcast) are reduced
// by EVERY block of the update kernel (4 lanes per basis vector, fixed order: bitwise reproducible) instead of by a launch of
// their own; block 0 also stores h for the Hessenberg column.
__global__ __launch_bounds__(256) void k_maxpy_reduce(int k, int nb, long long n, const double* __restrict__ V, long long ldv,
                                                      const double* __restrict__ part, double alpha, double* __restrict__ w,
                                                      double* __restrict__ h_out, double* __restrict__ npart /* null, or [gridDim.x]: sum of the updated w^2 per block */) {
    extern __shared__ double sh[];
    __shared__ double red[4];
    for (int base = 0; base < k; base += 64) {
        const int i = base + (threadIdx.x >> 2), sub = threadIdx.x & 3;
        double s = 0.0;
        if (i < k) for (int b = sub; b < nb; b += 4) s += part[(size_t)i*nb + b];
        s += __shfl_xor(s, 1, 64);
        s += __shfl_xor(s, 2, 64);
        if (i < k && sub == 0) { sh[i] = s; if (blockIdx.x == 0) h_out[i] = s; }
    }
    __syncthreads();
    const long long t = (long long)blockIdx.x*256 + threadIdx.x;
    double wn = 0.0;
    if (t < n) {
        double s = 0.0;
        for (int i = 0; i < k; i++) s += sh[i]*V[(size_t)i*ldv + t];
        wn = w[t] + alpha*s;
        w[t] = wn;
    }
    if (npart) {                     // the norm of the result rides along: one launch less before the normalisation
        double q = wn*wn;
        for (int off = 32; off > 0; off >>= 1) q += __shfl_down(q, off, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = q;
        __syncthreads();
        if (threadIdx.x == 0) npart[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
    }
}

// v = w / |w| from the partial sums of w.w; block 0 also writes the finished Hessenberg column (h1 + h2, |w|) -- `col` may be
// pinned host memory (one store per entry, no separate copy node in the captured graph).
__global__ __launch_bounds__(256) void k_normalize(int nb, long long n, const double* __restrict__ part, const double* __restrict__ w,
                                                   double* __restrict__ v, int k, const double* __restrict__ h1,
                                                   const double* __restrict__ h2, double* __restrict__ col, int norm_slot) {
    __shared__ double s_nrm;
    if (threadIdx.x < 64) {
        double s = 0.0;
        for (int i = threadIdx.x; i < nb; i += 64) s += part[i];
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
        if (threadIdx.x == 0) s_nrm = sqrt(s);
    }
    __syncthreads();
    const double nrm = s_nrm;
    if (blockIdx.x == 0) {
        for (int i = threadIdx.x; i < k; i += 256) col[i] = h1[i] + (h2 ? h2[i] : 0.0);
        if (threadIdx.x == 0) col[norm_slot] = nrm;
    }
    const long long t = (long long)blockIdx.x*256 + threadIdx.x;
    if (t < n) v[t] = w[t]/nrm;
}

// ---- batched (one system per row / level) CG vector kernels: per-row scalars stay in device memory -----------------
constexpr int RD_BLOCKS = 32;       // partial sums per row
__global__ __launch_bounds__(256) void k_rowdot_partial(long long n, long long chunk, const double* __restrict__ A, long long lda,
                                                        const double* __restrict__ B, long long ldb, double* __restrict__ part) {
    __shared__ double red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, row = blockIdx.y;
    const long long lo = (long long)blockIdx.x*chunk, hi = min(n, lo + chunk);
    const double* a = A + (size_t)row*lda; const double* b = B + (size_t)row*ldb;
    double s = 0.0;
    for (long long t = lo + tid; t < hi; t += 256) s += a[t]*b[t];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    if (tid == 0) part[(size_t)row*gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
// Round 4: the dots of the first Gram-Schmidt pass with the 1-form GATHER of w folded in.  w's first n1 entries are still element-local
// (ze: what k_sw_blocks_apply left; plan: the two contributors of every edge slot, the order k_gather_sum adds them in): every block forms
// the entries of its chunk on the fly, the blocks of row 0 also store them.  Same products, same order as k_gather_sum + k_rowdot_partial.
__global__ __launch_bounds__(256) void k_rowdot_partial_gather(long long n, long long n1, long long chunk, const double* __restrict__ V, long long ldv,
                                                               const double* __restrict__ ze, const int* __restrict__ plan,
                                                               double* __restrict__ w, double* __restrict__ part) {
    __shared__ double red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, row = blockIdx.y;
    const long long lo = (long long)blockIdx.x*chunk, hi = min(n, lo + chunk);
    const double* a = V + (size_t)row*ldv;
    double s = 0.0;
    for (long long t = lo + tid; t < hi; t += 256) {
        double wv;
        if (t < n1) {
            const int p0 = plan[(size_t)t*2], p1 = plan[(size_t)t*2 + 1];
            double acc = 0.0;
            if (p0 >= 0) acc += ze[p0];
            if (p1 >= 0) acc += ze[p1];
            wv = acc;
            if (row == 0) w[t] = wv;
        } else wv = w[t];
        s += a[t]*wv;
    }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    if (tid == 0) part[(size_t)row*gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
// the k rows V_i . w of a Gram-Schmidt pass AND w . w (row k) in one launch: what the second pass needs to normalise without a
// reduction of its own (|w - V h|^2 = |w|^2 - |h|^2 for an orthonormal V)
__global__ __launch_bounds__(256) void k_mdot_self_partial(int k, long long n, long long chunk, const double* __restrict__ V, long long ldv,
                                                           const double* __restrict__ w, double* __restrict__ part) {
    __shared__ double red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, row = blockIdx.y;
    const long long lo = (long long)blockIdx.x*chunk, hi = min(n, lo + chunk);
    const double* a = row < k ? V + (size_t)row*ldv : w;
    double s = 0.0;
    for (long long t = lo + tid; t < hi; t += 256) s += a[t]*w[t];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    if (tid == 0) part[(size_t)row*gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
// second Gram-Schmidt pass, normalisation and the Hessenberg column in ONE launch: every block reduces the partial sums of
// h2 = V w and of w . w (fixed order), takes |w - V h2| = sqrt(w.w - h2.h2) -- exact for an orthonormal V, and h2 is the small
// correction of a re-orthogonalisation, so nothing cancels; if more than half of w.w went into h2 (the first pass had lost its
// orthogonality altogether) *flag is raised and the caller repeats the step with the three-launch form -- and writes
// w <- w - V^T h2, v = w/|w|; block 0 stores h2 and the column (col may be pinned host memory).
__global__ __launch_bounds__(256) void k_maxpy_reduce_normalize(int k, int nb, long long n, const double* __restrict__ V, long long ldv,
                                                                const double* __restrict__ part, double* __restrict__ w, double* __restrict__ v,
                                                                const double* __restrict__ h1, double* __restrict__ h2, double* __restrict__ col,
                                                                int norm_slot, int* __restrict__ flag) {
    extern __shared__ double sh[];                       // k + 1 reduced dots
    for (int base = 0; base <= k; base += 64) {
        const int i = base + (threadIdx.x >> 2), sub = threadIdx.x & 3;
        double s = 0.0;
        if (i <= k) for (int b = sub; b < nb; b += 4) s += part[(size_t)i*nb + b];
        s += __shfl_xor(s, 1, 64);
        s += __shfl_xor(s, 2, 64);
        if (i <= k && sub == 0) sh[i] = s;
    }
    __syncthreads();
    double hh = 0.0;
    for (int i = 0; i < k; i++) hh += sh[i]*sh[i];       // (every thread, same order: the same bits everywhere)
    const double ww = sh[k];
    const double n2 = ww - hh;
    const double nrm = sqrt(n2 > 0.0 ? n2 : 0.0);
    if (blockIdx.x == 0) {
        for (int i = threadIdx.x; i < k; i += 256) { h2[i] = sh[i]; col[i] = h1[i] + sh[i]; }
        // (a zero or non-finite w.w -- a look-ahead step past a breakdown -- is not a cancellation: the flag stays down)
        if (threadIdx.x == 0) { col[norm_slot] = nrm; if (flag && ww > 0.0 && hh > 0.5*ww && ww < 1.0e300) *flag = 1; }
    }
    const long long t = (long long)blockIdx.x*256 + threadIdx.x;
    if (t < n) {
        double s = 0.0;
        for (int i = 0; i < k; i++) s += sh[i]*V[(size_t)i*ldv + t];
        const double wn = w[t] - s;
        w[t] = wn;
        v[t] = nrm > 0.0 ? wn/nrm : 0.0;
    }
}
// Round 4: the UPDATE of the first Gram-Schmidt pass and the DOTS of the second in one launch.  A block reduces the partial sums of
// h1 = V w (as k_maxpy_reduce does), updates its EPB entries  w <- w - V^T h1  and -- the dots of the second pass being sums over entries,
// and each entry final as soon as its own update is done -- accumulates V_i . w and w . w over those entries.  With k_rowdot_partial before it
// and k_maxpy_reduce_normalize after it a whole CGS2 step with normalisation and Hessenberg column is THREE launches (rounds 2-3: four).
constexpr int GS_EPB = 512;         // entries per block of k_gs_update_dots
__global__ __launch_bounds__(256) void k_gs_update_dots(int k, int nb, long long n, const double* __restrict__ V, long long ldv,
                                                        const double* __restrict__ part1, double* __restrict__ w, double* __restrict__ h_out,
                                                        double* __restrict__ part2) {
    extern __shared__ double sh[];                       // [k] h1, then [(k + 1)][4] per-wave partial dots
    double* red = sh + k;
    for (int base = 0; base < k; base += 64) {
        const int i = base + (threadIdx.x >> 2), sub = threadIdx.x & 3;
        double s = 0.0;
        if (i < k) for (int b = sub; b < nb; b += 4) s += part1[(size_t)i*nb + b];
        s += __shfl_xor(s, 1, 64);
        s += __shfl_xor(s, 2, 64);
        if (i < k && sub == 0) { sh[i] = s; if (blockIdx.x == 0) h_out[i] = s; }
    }
    __syncthreads();
    const long long t0 = (long long)blockIdx.x*GS_EPB + threadIdx.x, t1 = t0 + 256;
    const bool a0 = t0 < n, a1 = t1 < n;
    double w0 = 0.0, w1 = 0.0;
    // (entries past n read entry n-1 and contribute 0: no exec-masked loads, and the loops below are unrolled by 8 so that eight rows' loads
    //  are in flight together -- rolled, every row waited for its own L2 round trip: 8 us for k = 30)
    const long long u0 = a0 ? t0 : n - 1, u1 = a1 ? t1 : n - 1;
    {
        double s0 = 0.0, s1 = 0.0;
#pragma unroll 8
        for (int i = 0; i < k; i++) {
            const double hi = sh[i];
            s0 += hi*V[(size_t)i*ldv + u0];
            s1 += hi*V[(size_t)i*ldv + u1];
        }
        if (a0) { w0 = w[t0] - s0; w[t0] = w0; }
        if (a1) { w1 = w[t1] - s1; w[t1] = w1; }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll 8
    for (int i = 0; i < k; i++) {
        double d = V[(size_t)i*ldv + u0]*w0 + V[(size_t)i*ldv + u1]*w1;
        for (int off = 32; off > 0; off >>= 1) d += __shfl_down(d, off, 64);
        if (lane == 0) red[i*4 + wave] = d;
    }
    {
        double d = w0*w0 + w1*w1;
        for (int off = 32; off > 0; off >>= 1) d += __shfl_down(d, off, 64);
        if (lane == 0) red[k*4 + wave] = d;
    }
    __syncthreads();
    for (int i = threadIdx.x; i <= k; i += 256) part2[(size_t)i*gridDim.x + blockIdx.x] = (red[i*4] + red[i*4 + 1]) + (red[i*4 + 2] + red[i*4 + 3]);
}
// rowdot in ONE launch (round 6): every block leaves its partial sum as k_rowdot_partial does; the block that arrives LAST at the row's counter
// (device scope, behind a fence) reduces the row's partial sums in the order and with the tree of k_rowdot_final -- the same bits as the two
// launches -- and clears the counter for the next call.  Worth a launch (~5 us) per inner product where the vectors are short: the check
// norms of the shallow-water iteration (12 launches -> 6 per Picard iteration), the CG of the library's KSP objects.
__global__ __launch_bounds__(256) void k_rowdot_fused(long long n, long long chunk, const double* __restrict__ A, long long lda,
                                                      const double* __restrict__ B, long long ldb, double* part, unsigned* counters, double* __restrict__ out) {
    __shared__ double red[4];
    __shared__ int last;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, row = blockIdx.y, nb = gridDim.x;
    const long long lo = (long long)blockIdx.x*chunk, hi = min(n, lo + chunk);
    const double* a = A + (size_t)row*lda; const double* b = B + (size_t)row*ldb;
    double s = 0.0;
    for (long long t = lo + tid; t < hi; t += 256) s += a[t]*b[t];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    if (tid == 0) {
        __hip_atomic_store(part + (size_t)row*nb + blockIdx.x, (red[0] + red[1]) + (red[2] + red[3]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // The partial sum went out as a write-through store (sc1: coherent at device scope once acknowledged); waiting for that acknowledgement
        // and then counting the arrival with a RELAXED atomic orders the two without a release fence.  A release here is `buffer_wbl2 sc1` on
        // gfx950 -- a write-back of the XCD's whole L2, which holds the previous kernel's output: 48 us per call for 60 rows of HorizSolve's
        // check norms where the two launches took 12.  The reader below uses sc1 loads for the same reason (no acquire / buffer_inv).
        __builtin_amdgcn_s_waitcnt(0);
        last = __hip_atomic_fetch_add(counters + row, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)nb - 1;
    }
    __syncthreads();
    if (!last || wave != 0) return;
    double v = 0.0;
    for (int i = lane; i < nb; i += 64) v += __hip_atomic_load(part + (size_t)row*nb + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (nb <= 64: one term per lane)
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if (lane == 0) { out[row] = v; __hip_atomic_store(counters + row, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
}
__global__ __launch_bounds__(64) void k_rowdot_final(int nb, const double* __restrict__ part, double* __restrict__ out) {
    const int row = blockIdx.x, lane = threadIdx.x;
    double s = 0.0;
    for (int i = lane; i < nb; i += 64) s += part[(size_t)row*nb + i];      // (nb <= 64: one term per lane, as before rows longer than 262 144 got more blocks)
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (lane == 0) out[row] = s;
}
// x += alpha p ; r -= alpha Ap with alpha_row = num[row]/den[row]
__global__ __launch_bounds__(256) void k_cg_update(long long n, const double* __restrict__ num, const double* __restrict__ den,
        const double* __restrict__ p, long long ldp, const double* __restrict__ Ap, long long ldap,
        double* __restrict__ x, long long ldx, double* __restrict__ r, long long ldr) {
    const int row = blockIdx.y;
    const long long t = (long long)blockIdx.x*256 + threadIdx.x;
    if (t >= n) return;
    const double d = den[row];
    const double alpha = num[row]/(fabs(d) < 1e-300 ? 1e-300 : d);
    x[(size_t)row*ldx + t] += alpha*p[(size_t)row*ldp + t];
    r[(size_t)row*ldr + t] -= alpha*Ap[(size_t)row*ldap + t];
}
// p = z + beta p with beta_row = num[row]/den[row]
__global__ __launch_bounds__(256) void k_cg_direction(long long n, const double* __restrict__ num, const double* __restrict__ den,
        const double* __restrict__ z, long long ldz, double* __restrict__ p, long long ldp) {
    const int row = blockIdx.y;
    const long long t = (long long)blockIdx.x*256 + threadIdx.x;
    if (t >= n) return;
    const double d = den[row];
    const double beta = num[row]/(fabs(d) < 1e-300 ? 1e-300 : d);
    p[(size_t)row*ldp + t] = z[(size_t)row*ldz + t] + beta*p[(size_t)row*ldp + t];
}

}  // namespace

// out[r][j] = alpha * (a[r][j] (op) b[r][j]) + beta * c[r][j]   (op 0: a alone, 1: a*b, 2: a/b); rows r at their own strides.
// The level-wise field algebra between operator calls of HorizSolve (averages of two states, M0^-1 = 1/diagonal, accumulations)
// as ONE library launch each instead of chains of framework elementwise kernels.
__global__ __launch_bounds__(256) void k_vec_combine(int nrows, long long n, double alpha, const double* __restrict__ a, long long as_,
                                                     int op, const double* __restrict__ b, long long bs, double beta,
                                                     const double* c, long long cs, double* out, long long os) {
    const long long j = (long long)blockIdx.x*256 + threadIdx.x;
    if (j >= n) return;
    for (int r = blockIdx.y; r < nrows; r += gridDim.y) {
        double t = a[(size_t)r*as_ + j];
        if (op == 1) t *= b[(size_t)r*bs + j];
        else if (op == 2) t /= b[(size_t)r*bs + j];
        t *= alpha;
        if (c) t += beta*c[(size_t)r*cs + j];
        out[(size_t)r*os + j] = t;
    }
}
// out[k] = 0.5 a[k-1] + 0.5 a[k] over the nk levels of a field given on the nk-1 interfaces (the missing boundary interfaces left out):
// HorizSolve::diagnose_Phi eul/HorizSolve.cpp:451-459
__global__ __launch_bounds__(256) void k_interface_average(int nk, long long n, const double* __restrict__ a, long long as_, double* __restrict__ out, long long os) {
    const long long j = (long long)blockIdx.x*256 + threadIdx.x;
    if (j >= n) return;
    for (int k = blockIdx.y; k < nk; k += gridDim.y) {
        double t = 0.0;
        if (k > 0) t += 0.5*a[(size_t)(k - 1)*as_ + j];
        if (k < nk - 1) t += 0.5*a[(size_t)k*as_ + j];
        out[(size_t)k*os + j] = t;
    }
}

extern "C" {

int mimsem_vec_combine(mimsem_ctx* c, int nrows, long long n, double alpha, const double* a, long long as_, int op, const double* b, long long bs,
                       double beta, const double* cc, long long cs, double* out, long long os) {
    if (!c || nrows < 0 || n < 0 || op < 0 || op > 2) return MIMSEM_ERR_ARG;
    if (nrows == 0 || n == 0) return MIMSEM_OK;
    if (!a || !out || (op && !b)) return MIMSEM_ERR_ARG;
    hipLaunchKernelGGL(k_vec_combine, dim3((unsigned)((n + 255)/256), (unsigned)std::min(nrows, 64)), dim3(256), 0, c->stream,
                       nrows, n, alpha, a, as_, op, b, bs, beta, cc, cs, out, os);
    MIMSEM_HIP_TRY(hipGetLastError());
    return MIMSEM_OK;
}
int mimsem_interface_average(mimsem_ctx* c, int nk, long long n, const double* a, long long as_, double* out, long long os) {
    if (!c || nk < 1 || n < 0 || !out || (nk > 1 && !a)) return MIMSEM_ERR_ARG;
    if (n == 0) return MIMSEM_OK;
    hipLaunchKernelGGL(k_interface_average, dim3((unsigned)((n + 255)/256), (unsigned)std::min(nk, 64)), dim3(256), 0, c->stream, nk, n, a, as_, out, os);
    MIMSEM_HIP_TRY(hipGetLastError());
    return MIMSEM_OK;
}

int mimsem_krylov_rowdot(mimsem_ctx* c, int nrows, long long n, const double* A, long long lda, const double* B, long long ldb, double* out) {
    if (!c || !A || !B || !out || nrows < 0 || n < 0) return MIMSEM_ERR_ARG;
    if (nrows == 0) return MIMSEM_OK;
    // up to 32 blocks per row for rows up to 262 144 entries (the 1-form vectors of a level: unchanged since round 1, the same bits); longer rows
    // -- a whole [levels x slots] array as ONE row: the C++ HorizSolve's check norms, 2 rows of 1.87 M -- get a block per 8 192 entries
    // (round 6: 64 blocks on 256 CUs took 66 us per call, 0.5 ms of a 5.7 ms evaluation)
    const int nb = (int)std::max<long long>(1, std::min<long long>(1024, std::max<long long>(std::min<long long>(RD_BLOCKS, (n + 1023)/1024), (n + 8191)/8192)));
    const long long chunk = (n + nb - 1)/nb;
    int rc = c->ensure_kry((long long)nb*nrows);
    if (rc) return rc;
    // one launch where a launch is what the call costs (few rows: the check norms of the shallow-water iteration, the CG of the KSP objects): the
    // last block of a row reduces it, same bits as the two launches below.  Many rows stay on two launches: their arrivals (rows x 32 atomics
    // on neighbouring words) serialise -- 17 us against 7.4 + 4.8 for the 60 rows x 62 208 of HorizSolve's check norms
    if (nrows <= MIMSEM_RD_COUNTERS && c->d_rdcnt && !c->rd_two) {
        hipLaunchKernelGGL(k_rowdot_fused, dim3(nb, nrows), dim3(256), 0, c->stream, n, chunk, A, lda, B, ldb, c->d_kry, c->d_rdcnt, out);
        MIMSEM_HIP_TRY(hipGetLastError());
        return MIMSEM_OK;
    }
    hipLaunchKernelGGL(k_rowdot_partial, dim3(nb, nrows), dim3(256), 0, c->stream, n, chunk, A, lda, B, ldb, c->d_kry);
    hipLaunchKernelGGL(k_rowdot_final, dim3(nrows), dim3(64), 0, c->stream, nb, c->d_kry, out);
    MIMSEM_HIP_TRY(hipGetLastError());
    return MIMSEM_OK;
}

int mimsem_krylov_cg_update(mimsem_ctx* c, int nrows, long long n, const double* num, const double* den,
                            const double* p, long long ldp, const double* Ap, long long ldap,
                            double* x, long long ldx, double* r, long long ldr) {
    if (!c || !num || !den || !p || !Ap || !x || !r || nrows < 0 || n < 0) return MIMSEM_ERR_ARG;
    if (nrows == 0 || n == 0) return MIMSEM_OK;
    hipLaunchKernelGGL(k_cg_update, dim3((unsigned)((n + 255)/256), nrows), dim3(256), 0, c->stream, n, num, den, p, ldp, Ap, ldap, x, ldx, r, ldr);
    MIMSEM_HIP_TRY(hipGetLastError());
    return MIMSEM_OK;
}

// one step of the Chebyshev semi-iteration's vector algebra in ONE pass: x += d;  r -= Bd;  d = a d + b r (the updated r)
namespace {
__global__ __launch_bounds__(256) void k_chebyshev_update(long long n, double a, double b, const double* __restrict__ Bd, long long ldb,
                                                          double* __restrict__ x, long long ldx, double* __restrict__ r, long long ldr,
                                                          double* __restrict__ d, long long ldd) {
    const long long i = (long long)blockIdx.x*256 + threadIdx.x;
    if (i >= n) return;
    const size_t row = blockIdx.y;
    const double dv = d[row*ldd + i];
    x[row*ldx + i] += dv;
    const double rv = r[row*ldr + i] - Bd[row*ldb + i];
    r[row*ldr + i] = rv;
    d[row*ldd + i] = fma(a, dv, b*rv);
}
// the start of a Chebyshev solve from x = 0 with the preconditioned right-hand side c = P b (or s = -1: b = -f, c = P f):
// r = s c;  d = r / theta;  x = 0 -- one launch for what were a negation, a copy, a clear and a scaling
__global__ __launch_bounds__(256) void k_chebyshev_start(long long n, double s, double inv_theta, const double* cv, long long ldc, double* r, long long ldr,
                                                         double* d, long long ldd, double* x, long long ldx) {
    const long long i = (long long)blockIdx.x*256 + threadIdx.x;
    if (i >= n) return;
    const size_t row = blockIdx.y;
    const double rv = s*cv[row*ldc + i];
    r[row*ldr + i] = rv;
    d[row*ldd + i] = rv*inv_theta;
    x[row*ldx + i] = 0.0;
}
// the vector algebra of a Chebyshev step whose operator result had to be COMPLETED over a halo first (sharded meshes: the fused sweeps cannot run
// across the exchange): z = dinv (b - y) (or z = y when dinv is null: y already is the preconditioned residual); p = z + beta p; x += alpha p;
// upd = z if given -- one launch for what were up to five element-wise kernels; the arithmetic of k_gather_epilogue's modes 3 / 5
__global__ __launch_bounds__(256) void k_chebyshev_px(long long n, double alpha, double beta, const double* __restrict__ y, long long ldy,
                                                      const double* __restrict__ b, long long ldb, const double* __restrict__ dinv, long long ldd,
                                                      double* __restrict__ p, long long ldp, double* __restrict__ x, long long ldx, double* __restrict__ upd, long long ldu) {
    const long long i = (long long)blockIdx.x*256 + threadIdx.x;
    if (i >= n) return;
    const size_t row = blockIdx.y;
    const double yv = y[row*ldy + i];
    const double z = dinv ? dinv[row*ldd + i]*(b[row*ldb + i] - yv) : yv;
    const double pn = fma(beta, p[row*ldp + i], z);
    p[row*ldp + i] = pn;
    x[row*ldx + i] = fma(alpha, pn, x[row*ldx + i]);
    if (upd) upd[row*ldu + i] = z;
}
// the end of a Picard iteration: x += dx, out = {dx . dx, x . x (the new x)} -- the update and both norms of the stopping test in ONE launch (the
// partial sums and their reduction as mimsem_krylov_rowdot's one-launch form: the same bits as the update followed by two rowdot calls)
__global__ __launch_bounds__(256) void k_axpy_dots(long long n, long long chunk, const double* __restrict__ dx, double* __restrict__ x, double* part,
                                                   unsigned* counters, double* __restrict__ out) {
    __shared__ double red[2][4];
    __shared__ int last;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nb = gridDim.x;
    const long long lo = (long long)blockIdx.x*chunk, hi = min(n, lo + chunk);
    double s0 = 0.0, s1 = 0.0;
    for (long long t = lo + tid; t < hi; t += 256) {
        const double dv = dx[t], xn = x[t] + dv;
        x[t] = xn;
        s0 += dv*dv; s1 += xn*xn;
    }
    for (int off = 32; off > 0; off >>= 1) { s0 += __shfl_down(s0, off, 64); s1 += __shfl_down(s1, off, 64); }
    if (lane == 0) { red[0][wave] = s0; red[1][wave] = s1; }
    __syncthreads();
    if (tid == 0) {
        __hip_atomic_store(part + blockIdx.x, (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(part + nb + blockIdx.x, (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_s_waitcnt(0);                               // (write-through stores acknowledged before the arrival is counted: k_rowdot_fused)
        last = __hip_atomic_fetch_add(counters, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)nb - 1;
    }
    __syncthreads();
    if (!last || wave > 1) return;
    double v = 0.0;
    for (int i = lane; i < nb; i += 64) v += __hip_atomic_load(part + (size_t)wave*nb + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if (lane == 0) out[wave] = v;
    if (tid == 0) __hip_atomic_store(counters, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
}  // namespace
int mimsem_krylov_chebyshev_start(mimsem_ctx* c, int nrows, long long n, double s, double theta, const double* cv, long long ldc,
                                  double* r, long long ldr, double* d, long long ldd, double* x, long long ldx) {
    if (!c || !cv || !x || !r || !d || nrows < 0 || n < 0 || !(theta != 0.0) || x == r || x == d || r == d || x == cv || d == cv) return MIMSEM_ERR_ARG;
    if (nrows == 0 || n == 0) return MIMSEM_OK;
    hipLaunchKernelGGL(k_chebyshev_start, dim3((unsigned)((n + 255)/256), nrows), dim3(256), 0, c->stream, n, s, 1.0/theta, cv, ldc, r, ldr, d, ldd, x, ldx);
    MIMSEM_HIP_TRY(hipGetLastError());
    return MIMSEM_OK;
}
int mimsem_krylov_chebyshev_px(mimsem_ctx* c, int nrows, long long n, double alpha, double beta, const double* y, long long ldy,
                               const double* b, long long ldb, const double* dinv, long long ldd, double* p, long long ldp, double* x, long long ldx,
                               double* upd, long long ldu) {
    if (!c || !y || !p || !x || nrows < 0 || n < 0 || (dinv && !b) || p == x || y == p || y == x) return MIMSEM_ERR_ARG;
    if (nrows == 0 || n == 0) return MIMSEM_OK;
    hipLaunchKernelGGL(k_chebyshev_px, dim3((unsigned)((n + 255)/256), nrows), dim3(256), 0, c->stream, n, alpha, beta, y, ldy, b, ldb, dinv, ldd, p, ldp, x, ldx, upd, ldu);
    MIMSEM_HIP_TRY(hipGetLastError());
    return MIMSEM_OK;
}
int mimsem_krylov_axpy_dots(mimsem_ctx* c, long long n, const double* dx, double* x, double* out) {
    if (!c || !dx || !x || !out || n < 0 || dx == x) return MIMSEM_ERR_ARG;
    if (n == 0) return mimsem_memset(c, out, 0, 2*sizeof(double));
    const int nb = (int)std::max<long long>(1, std::min<long long>(1024, std::max<long long>(std::min<long long>(RD_BLOCKS, (n + 1023)/1024), (n + 8191)/8192)));      // (mimsem_krylov_rowdot's blocks)
    const long long chunk = (n + nb - 1)/nb;
    int rc = c->ensure_kry(2LL*nb);
    if (rc) return rc;
    if (!c->d_rdcnt) return MIMSEM_ERR_STATE;
    hipLaunchKernelGGL(k_axpy_dots, dim3(nb), dim3(256), 0, c->stream, n, chunk, dx, x, c->d_kry, c->d_rdcnt, out);
    MIMSEM_HIP_TRY(hipGetLastError());
    return MIMSEM_OK;
}
int mimsem_krylov_chebyshev_update(mimsem_ctx* c, int nrows, long long n, double a, double b, const double* Bd, long long ldb,
                                   double* x, long long ldx, double* r, long long ldr, double* d, long long ldd) {
    if (!c || !Bd || !x || !r || !d || nrows < 0 || n < 0) return MIMSEM_ERR_ARG;
    if (nrows == 0 || n == 0) return MIMSEM_OK;
    hipLaunchKernelGGL(k_chebyshev_update, dim3((unsigned)((n + 255)/256), nrows), dim3(256), 0, c->stream, n, a, b, Bd, ldb, x, ldx, r, ldr, d, ldd);
    MIMSEM_HIP_TRY(hipGetLastError());
    return MIMSEM_OK;
}

int mimsem_krylov_cg_direction(mimsem_ctx* c, int nrows, long long n, const double* num, const double* den,
                               const double* z, long long ldz, double* p, long long ldp) {
    if (!c || !num || !den || !z || !p || nrows < 0 || n < 0) return MIMSEM_ERR_ARG;
    if (nrows == 0 || n == 0) return MIMSEM_OK;
    hipLaunchKernelGGL(k_cg_direction, dim3((unsigned)((n + 255)/256), nrows), dim3(256), 0, c->stream, n, num, den, z, ldz, p, ldp);
    MIMSEM_HIP_TRY(hipGetLastError());
    return MIMSEM_OK;
}


int mimsem_krylov_rowdot(mimsem_ctx* c, int nrows, long long n, const double* A, long long lda, const double* B, long long ldb, double* out);

// h = V w: the k dot products are k rows of a row-dot with a broadcast second operand (ldb = 0): grid = chunks x rows
int mimsem_krylov_mdot(mimsem_ctx* c, int k, long long n, const double* V, long long ldv, const double* w, double* h) {
    if (!c || !V || !w || !h || k < 0 || n < 0 || ldv < n) return MIMSEM_ERR_ARG;
    return mimsem_krylov_rowdot(c, k, n, V, ldv, w, 0, h);
}

static int rowdot_partials(mimsem_ctx* c, int nrows, long long n, const double* A, long long lda, const double* B, long long ldb, int* nb_out) {
    const int nb = (int)std::max<long long>(1, std::min<long long>(RD_BLOCKS, (n + 1023)/1024));
    const long long chunk = (n + nb - 1)/nb;
    int rc = c->ensure_kry((long long)nb*nrows);
    if (rc) return rc;
    hipLaunchKernelGGL(k_rowdot_partial, dim3(nb, nrows), dim3(256), 0, c->stream, n, chunk, A, lda, B, ldb, c->d_kry);
    *nb_out = nb;
    return MIMSEM_OK;
}

// h = V w ; w += alpha V^T h  (one classical Gram-Schmidt pass, two launches)
int mimsem_krylov_orthogonalize(mimsem_ctx* c, int k, long long n, const double* V, long long ldv, double alpha, double* w, double* h) {
    if (!c || !V || !w || !h || k < 0 || n < 0 || ldv < n) return MIMSEM_ERR_ARG;
    if (k == 0 || n == 0) return MIMSEM_OK;
    int nb = 0;
    int rc = rowdot_partials(c, k, n, V, ldv, w, 0, &nb);
    if (rc) return rc;
    hipLaunchKernelGGL(k_maxpy_reduce, dim3((unsigned)((n + 255)/256)), dim3(256), (size_t)k*sizeof(double), c->stream,
                       k, nb, n, V, ldv, c->d_kry, alpha, w, h, (double*)nullptr);
    MIMSEM_HIP_TRY(hipGetLastError());
    return MIMSEM_OK;
}

// w = P (A x) ; h = V w ; w += alpha V^T h : the Krylov body of the shallow-water solve and the first Gram-Schmidt pass in FOUR launches
// (mimsem_sw_operator_precond_apply + mimsem_krylov_orthogonalize: five) -- the 1-form gather of w rides in the dot pass.
int mimsem_sw_operator_precond_orthogonalize(mimsem_ctx* c, double a, double grav, double H, const double* f0, const double* blocks,
                                             const double* x, double* w, int k, const double* V, long long ldv, double alpha, double* h) {
    if (!c || !f0 || !blocks || !x || !w || x == w || k < 0 || (k > 0 && (!V || !h))) return MIMSEM_ERR_ARG;
    const long long n1 = c->n1, n = (long long)c->n1 + c->n2;
    if (c->nEl == 0 || n == 0) return MIMSEM_OK;
    if (k > 0 && ldv < n) return MIMSEM_ERR_ARG;
    if (k == 0) return launch_sw_operator_precond(c, 1, a, grav, H, f0, 0, blocks, x, n, w, n);
    const double* ze = nullptr;
    int rc = launch_sw_operator_precond(c, 1, a, grav, H, f0, 0, blocks, x, n, w, n, &ze);
    if (rc) return rc;
    if (!ze) return MIMSEM_ERR_STATE;
    const int nb = (int)std::max<long long>(1, std::min<long long>(RD_BLOCKS, (n + 1023)/1024));
    const long long chunk = (n + nb - 1)/nb;
    if ((rc = c->ensure_kry((long long)nb*k))) return rc;
    hipLaunchKernelGGL(k_rowdot_partial_gather, dim3(nb, k), dim3(256), 0, c->stream, n, n1, chunk, V, ldv, ze, c->d_g1, w, c->d_kry);
    hipLaunchKernelGGL(k_maxpy_reduce, dim3((unsigned)((n + 255)/256)), dim3(256), (size_t)k*sizeof(double), c->stream,
                       k, nb, n, V, ldv, c->d_kry, alpha, w, h, (double*)nullptr);
    MIMSEM_HIP_TRY(hipGetLastError());
    return MIMSEM_OK;
}

// second Gram-Schmidt pass and normalisation in three launches: h2 = V w ; w -= V^T h2 (the norm of the result accumulated by the
// same kernel) ; v = w/|w|, col[0..k) = h1 + h2, col[norm_slot] = |w|
int mimsem_krylov_reorthonormalize_ex(mimsem_ctx* c, int k, long long n, const double* V, long long ldv, double* w, double* v,
                                      const double* h1, double* h2, double* col, int norm_slot, int fused, int* flag) {
    if (!c || !V || !w || !v || !h1 || !h2 || !col || k <= 0 || n <= 0 || ldv < n || norm_slot < 0) return MIMSEM_ERR_ARG;
    const unsigned gb = (unsigned)((n + 255)/256);
    int nb = 0;
    int rc = c->ensure_kry((long long)RD_BLOCKS*(k + 1) + gb);
    if (rc) return rc;
    if (fused) {
        // round 3: two launches -- the dots of the pass and w . w together, then update + normalisation + column in one kernel
        nb = (int)std::max<long long>(1, std::min<long long>(RD_BLOCKS, (n + 1023)/1024));
        const long long chunk = (n + nb - 1)/nb;
        hipLaunchKernelGGL(k_mdot_self_partial, dim3(nb, k + 1), dim3(256), 0, c->stream, k, n, chunk, V, ldv, w, c->d_kry);
        hipLaunchKernelGGL(k_maxpy_reduce_normalize, dim3(gb), dim3(256), (size_t)(k + 1)*sizeof(double), c->stream, k, nb, n, V, ldv, c->d_kry,
                           w, v, h1, h2, col, norm_slot, flag);
        MIMSEM_HIP_TRY(hipGetLastError());
        return MIMSEM_OK;
    }
    if ((rc = rowdot_partials(c, k, n, V, ldv, w, 0, &nb))) return rc;
    double* npart = c->d_kry + (size_t)RD_BLOCKS*k;
    hipLaunchKernelGGL(k_maxpy_reduce, dim3(gb), dim3(256), (size_t)k*sizeof(double), c->stream, k, nb, n, V, ldv, c->d_kry, -1.0, w, h2, npart);
    hipLaunchKernelGGL(k_normalize, dim3(gb), dim3(256), 0, c->stream, (int)gb, n, npart, w, v, k, h1, h2, col, norm_slot);
    MIMSEM_HIP_TRY(hipGetLastError());
    return MIMSEM_OK;
}
// One whole Arnoldi orthogonalisation (two classical Gram-Schmidt passes, normalisation, Hessenberg column) in THREE launches:
// dots of pass 1; update of pass 1 + dots of pass 2 (k_gs_update_dots); update of pass 2 + norm + column (k_maxpy_reduce_normalize, with the
// norm from w.w - h2.h2 and the flag word of the two-launch re-orthonormalisation).  h1, h2: device [k].
int mimsem_krylov_cgs2(mimsem_ctx* c, int k, long long n, const double* V, long long ldv, double* w, double* v,
                       double* h1, double* h2, double* col, int norm_slot, int* flag) {
    if (!c || !V || !w || !v || !h1 || !h2 || !col || k <= 0 || n <= 0 || ldv < n || norm_slot < 0) return MIMSEM_ERR_ARG;
    const int nb1 = (int)std::max<long long>(1, std::min<long long>(RD_BLOCKS, (n + 1023)/1024));
    const long long chunk = (n + nb1 - 1)/nb1;
    const unsigned gb2 = (unsigned)((n + GS_EPB - 1)/GS_EPB), gb = (unsigned)((n + 255)/256);
    int rc = c->ensure_kry((long long)RD_BLOCKS*(k + 1) + (long long)(k + 1)*gb2);
    if (rc) return rc;
    double* part1 = c->d_kry; double* part2 = c->d_kry + (size_t)RD_BLOCKS*(k + 1);
    hipLaunchKernelGGL(k_rowdot_partial, dim3(nb1, k), dim3(256), 0, c->stream, n, chunk, V, ldv, w, 0, part1);
    hipLaunchKernelGGL(k_gs_update_dots, dim3(gb2), dim3(256), (size_t)(k + 4*(k + 1))*sizeof(double), c->stream, k, nb1, n, V, ldv, part1, w, h1, part2);
    hipLaunchKernelGGL(k_maxpy_reduce_normalize, dim3(gb), dim3(256), (size_t)(k + 1)*sizeof(double), c->stream, k, (int)gb2, n, V, ldv, part2,
                       w, v, h1, h2, col, norm_slot, flag);
    MIMSEM_HIP_TRY(hipGetLastError());
    return MIMSEM_OK;
}

// the context-state form (round 3): form and flag word from mimsem_krylov_gs_control / MIMSEM_GS_FUSED_NORM
int mimsem_krylov_reorthonormalize(mimsem_ctx* c, int k, long long n, const double* V, long long ldv, double* w, double* v,
                                   const double* h1, double* h2, double* col, int norm_slot) {
    if (!c) return MIMSEM_ERR_ARG;
    if (c->gs_fused < 0) c->gs_fused = !(exp_env("MIMSEM_GS_FUSED_NORM") && atoi(exp_env("MIMSEM_GS_FUSED_NORM")) == 0);
    return mimsem_krylov_reorthonormalize_ex(c, k, n, V, ldv, w, v, h1, h2, col, norm_slot, c->gs_fused, c->gs_flag);
}

int mimsem_krylov_gs_control(mimsem_ctx* c, int fused, int* flag) {
    if (!c) return MIMSEM_ERR_ARG;
    if (fused >= 0) c->gs_fused = fused != 0;
    c->gs_flag = flag;
    return MIMSEM_OK;
}

// v = w/|w| ; col[0..k) = h1 + h2 (h2 may be null) ; col[norm_slot] = |w|   (two launches; col: device or pinned host memory)
int mimsem_krylov_normalize(mimsem_ctx* c, long long n, const double* w, double* v, int k, const double* h1, const double* h2,
                            double* col, int norm_slot) {
    if (!c || !w || !v || !col || k < 0 || n <= 0 || norm_slot < 0 || (k > 0 && !h1)) return MIMSEM_ERR_ARG;
    int nb = 0;
    int rc = rowdot_partials(c, 1, n, w, n, w, n, &nb);
    if (rc) return rc;
    hipLaunchKernelGGL(k_normalize, dim3((unsigned)((n + 255)/256)), dim3(256), 0, c->stream, nb, n, c->d_kry, w, v, k, h1, h2, col, norm_slot);
    MIMSEM_HIP_TRY(hipGetLastError());
    return MIMSEM_OK;
}

int mimsem_krylov_maxpy(mimsem_ctx* c, int k, long long n, const double* V, long long ldv, const double* h, double alpha, double* w) {
    if (!c || !V || !w || !h || k < 0 || n < 0 || ldv < n) return MIMSEM_ERR_ARG;
    if (k == 0 || n == 0) return MIMSEM_OK;
    hipLaunchKernelGGL(k_maxpy, dim3((unsigned)((n + 255)/256)), dim3(256), (size_t)k*sizeof(double), c->stream, k, n, V, ldv, h, alpha, w);
    MIMSEM_HIP_TRY(hipGetLastError());
    return MIMSEM_OK;
}

}  // extern "C"
