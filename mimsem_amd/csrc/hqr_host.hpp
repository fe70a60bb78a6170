// hqr_host.hpp -- host-only: eigenvalues of a small real upper Hessenberg matrix (the Arnoldi matrix of mimsem_ksp_ritz, csrc/ksp.hip).
// Plain C++ (no HIP): tests/cpp/hqr_cli.cpp compiles it with g++ and tests/test_hqr.py checks it against numpy on the CPU.
#pragma once
#include <algorithm>
#include <cmath>
#include <vector>

namespace {
// eigenvalues of a real upper Hessenberg matrix a[n][n] (row-major, destroyed) by the shifted QR algorithm (EISPACK hqr as in the
// literature: deflation, exceptional shifts, double-shift Francis steps).  Returns 0, or the number of eigenvalues NOT found.
int hessenberg_eigenvalues(std::vector<double>& a, int n, std::vector<double>& wr, std::vector<double>& wi) {
    auto A = [&](int i, int j) -> double& { return a[(size_t)i*n + j]; };
    wr.assign(n, 0.0); wi.assign(n, 0.0);
    double anorm = 0.0;
    for (int i = 0; i < n; i++) for (int j = std::max(i - 1, 0); j < n; j++) anorm += std::fabs(A(i, j));
    int nn = n - 1; double t = 0.0;
    double p = 0, q = 0, r = 0, s = 0, w, x, y, z;
    while (nn >= 0) {
        int its = 0, l;
        do {
            for (l = nn; l >= 1; l--) {
                s = std::fabs(A(l - 1, l - 1)) + std::fabs(A(l, l));
                if (s == 0.0) s = anorm;
                if (std::fabs(A(l, l - 1)) + s == s) { A(l, l - 1) = 0.0; break; }
            }
            x = A(nn, nn);
            if (l == nn) { wr[nn] = x + t; wi[nn--] = 0.0; }
            else {
                y = A(nn - 1, nn - 1); w = A(nn, nn - 1)*A(nn - 1, nn);
                if (l == nn - 1) {
                    p = 0.5*(y - x); q = p*p + w; z = std::sqrt(std::fabs(q)); x += t;
                    if (q >= 0.0) { z = p + (p >= 0.0 ? std::fabs(z) : -std::fabs(z)); wr[nn - 1] = wr[nn] = x + z; if (z != 0.0) wr[nn] = x - w/z; wi[nn - 1] = wi[nn] = 0.0; }
                    else { wr[nn - 1] = wr[nn] = x + p; wi[nn - 1] = -(wi[nn] = z); }
                    nn -= 2;
                } else {
                    if (its == 60) return nn + 1;
                    if (its == 10 || its == 20) {
                        t += x;
                        for (int i = 0; i <= nn; i++) A(i, i) -= x;
                        s = std::fabs(A(nn, nn - 1)) + std::fabs(A(nn - 1, nn - 2));
                        y = x = 0.75*s; w = -0.4375*s*s;
                    }
                    ++its;
                    int m;
                    for (m = nn - 2; m >= l; m--) {
                        z = A(m, m); r = x - z; s = y - z;
                        p = (r*s - w)/A(m + 1, m) + A(m, m + 1); q = A(m + 1, m + 1) - z - r - s; r = A(m + 2, m + 1);
                        s = std::fabs(p) + std::fabs(q) + std::fabs(r);
                        p /= s; q /= s; r /= s;
                        if (m == l) break;
                        const double u = std::fabs(A(m, m - 1))*(std::fabs(q) + std::fabs(r));
                        const double v = std::fabs(p)*(std::fabs(A(m - 1, m - 1)) + std::fabs(z) + std::fabs(A(m + 1, m + 1)));
                        if (u + v == v) break;
                    }
                    for (int i = m + 2; i <= nn; i++) { A(i, i - 2) = 0.0; if (i != m + 2) A(i, i - 3) = 0.0; }
                    for (int k = m; k <= nn - 1; k++) {
                        if (k != m) {
                            p = A(k, k - 1); q = A(k + 1, k - 1); r = 0.0;
                            if (k != nn - 1) r = A(k + 2, k - 1);
                            if ((x = std::fabs(p) + std::fabs(q) + std::fabs(r)) != 0.0) { p /= x; q /= x; r /= x; }
                        }
                        const double sg = std::sqrt(p*p + q*q + r*r);
                        if ((s = (p >= 0.0 ? sg : -sg)) != 0.0) {
                            if (k == m) { if (l != m) A(k, k - 1) = -A(k, k - 1); }
                            else A(k, k - 1) = -s*x;
                            p += s; x = p/s; y = q/s; z = r/s; q /= p; r /= p;
                            for (int j = k; j <= nn; j++) {
                                p = A(k, j) + q*A(k + 1, j);
                                if (k != nn - 1) { p += r*A(k + 2, j); A(k + 2, j) -= p*z; }
                                A(k + 1, j) -= p*y; A(k, j) -= p*x;
                            }
                            const int mmin = nn < k + 3 ? nn : k + 3;
                            for (int i = l; i <= mmin; i++) {
                                p = x*A(i, k) + y*A(i, k + 1);
                                if (k != nn - 1) { p += z*A(i, k + 2); A(i, k + 2) -= p*r; }
                                A(i, k + 1) -= p*q; A(i, k) -= p;
                            }
                        }
                    }
                }
            }
        } while (l < nn - 1);
    }
    return 0;
}
}  // namespace
