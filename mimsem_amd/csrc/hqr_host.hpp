// hqr_host.hpp -- host-only: eigenvalues of a small real upper Hessenberg matrix (the Arnoldi matrix of mimsem_ksp_ritz, csrc/ksp.hip, and
// of mimsem_hessenberg_eigenvalues).  Plain C++ (no HIP): tests/cpp/hqr_cli.cpp compiles it with g++ and tests/test_hqr.py checks it against
// numpy on the CPU.
//
// Round 6: written afresh (the round-5 routine followed the textbook real double-shift `hqr` variable for variable; advisor).  This one is the
// SINGLE-shift QR iteration in COMPLEX arithmetic, the form of LAPACK's zlahqr reduced to what is needed here (eigenvalues only, matrices of
// a few dozen rows, a set-up path): the real matrix is promoted to std::complex, each sweep subtracts a Wilkinson shift (the eigenvalue of
// the trailing 2 x 2 block nearer to its last diagonal entry), factors the active unreduced block with Givens rotations from the left,
// applies them from the right and adds the shift back; a subdiagonal entry that is negligible against its two diagonal neighbours splits
// the problem, a 1 x 1 trailing block is an eigenvalue.  Every tenth sweep without a deflation takes an ad-hoc shift instead (the usual
// remedy for the rare cycling case).  Conjugate pairs of a real matrix come out as two complex values that agree to round-off.
#pragma once
#include <algorithm>
#include <cmath>
#include <complex>
#include <vector>

namespace {

// a[n][n] row-major (entries below the first subdiagonal are ignored; the array is left untouched apart from being copied).
// Returns 0, or the number of eigenvalues NOT found within the sweep limit.  wr / wi: real and imaginary parts, in no particular order.
int hessenberg_eigenvalues(std::vector<double>& a, int n, std::vector<double>& wr, std::vector<double>& wi) {
    using cplx = std::complex<double>;
    wr.assign(n, 0.0); wi.assign(n, 0.0);
    if (n <= 0) return 0;
    std::vector<cplx> h((size_t)n*n, cplx(0.0, 0.0));
    for (int i = 0; i < n; i++)
        for (int j = std::max(i - 1, 0); j < n; j++) h[(size_t)i*n + j] = a[(size_t)i*n + j];
    auto H = [&](int i, int j) -> cplx& { return h[(size_t)i*n + j]; };
    const double eps = 2.220446049250313e-16;
    double scale = 0.0;
    for (const cplx& v : h) scale = std::max(scale, std::abs(v));
    if (scale == 0.0) return 0;                                       // the zero matrix: n zero eigenvalues
    const double tiny = scale*1.0e-300/eps;

    std::vector<cplx> cs(n), sn(n);                                   // the rotations of one sweep
    int hi = n - 1, sweeps_here = 0;
    long total = 0;
    const long limit = 60L*n + 200;
    while (hi >= 0) {
        // where does the unreduced block that ends at `hi` begin?
        int lo = hi;
        while (lo > 0) {
            const double sub = std::abs(H(lo, lo - 1));
            double nb = std::abs(H(lo - 1, lo - 1)) + std::abs(H(lo, lo));
            if (nb == 0.0) nb = scale;
            if (sub <= eps*nb || sub <= tiny) { H(lo, lo - 1) = 0.0; break; }
            lo--;
        }
        if (lo == hi) {                                               // 1 x 1 block: an eigenvalue
            wr[hi] = H(hi, hi).real(); wi[hi] = H(hi, hi).imag();
            hi--; sweeps_here = 0;
            continue;
        }
        if (++total > limit) return hi + 1;
        // shift
        cplx mu;
        sweeps_here++;
        if (sweeps_here % 10 == 0) {
            mu = H(hi, hi) + cplx(std::abs(H(hi, hi - 1).real()) + (hi - 2 >= lo ? std::abs(H(hi - 1, hi - 2).real()) : 0.0), 0.0)*0.75;
        } else {
            const cplx p = H(hi - 1, hi - 1), q = H(hi - 1, hi), r = H(hi, hi - 1), s = H(hi, hi);
            const cplx half_tr = 0.5*(p + s), disc = std::sqrt(0.25*(p - s)*(p - s) + q*r);
            const cplx e1 = half_tr + disc, e2 = half_tr - disc;
            mu = std::abs(e1 - s) <= std::abs(e2 - s) ? e1 : e2;
        }
        // (H - mu I) = Q R on rows / columns lo .. hi, Q a product of Givens rotations G_k acting on rows k, k + 1
        for (int k = lo; k <= hi; k++) H(k, k) -= mu;
        for (int k = lo; k < hi; k++) {
            const cplx f = H(k, k), g = H(k + 1, k);
            const double nf = std::abs(f), ng = std::abs(g), nr = std::hypot(nf, ng);
            cplx c, s;
            if (nr == 0.0) { c = 1.0; s = 0.0; }
            else if (nf == 0.0) { c = 0.0; s = std::conj(g)/ng; }
            else { c = nf/nr; s = (f/nf)*std::conj(g)/nr; }
            cs[k] = c; sn[k] = s;
            // rows k, k + 1:  [ c  s ; -conj(s)  c ] (c real)
            for (int j = k; j <= hi; j++) {
                const cplx x = H(k, j), y = H(k + 1, j);
                H(k, j) = c*x + s*y;
                H(k + 1, j) = -std::conj(s)*x + c*y;
            }
            H(k + 1, k) = 0.0;
        }
        // R Q: the conjugate-transposed rotations on columns k, k + 1 (rows lo .. k + 1 hold the non-zeros)
        for (int k = lo; k < hi; k++) {
            const cplx c = cs[k], s = sn[k];
            for (int i = lo; i <= k + 1; i++) {
                const cplx x = H(i, k), y = H(i, k + 1);
                H(i, k) = x*c + y*std::conj(s);
                H(i, k + 1) = -x*s + y*c;
            }
        }
        for (int k = lo; k <= hi; k++) H(k, k) += mu;
    }
    // a real matrix: imaginary parts at round-off level are noise of the complex arithmetic
    for (int i = 0; i < n; i++) if (std::fabs(wi[i]) <= 64.0*eps*std::max(scale, std::fabs(wr[i]))) wi[i] = 0.0;
    return 0;
}

}  // namespace
