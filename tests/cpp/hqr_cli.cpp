// tests/cpp/hqr_cli.cpp -- reads n and an n x n matrix (row-major; entries below the first subdiagonal are ignored / must be 0) from stdin,
// prints the eigenvalues found by csrc/hqr_host.hpp (the routine behind mimsem_ksp_ritz).  CPU only: tests/test_hqr.py.
#include <cstdio>
#include "../../mimsem_amd/csrc/hqr_host.hpp"
int main() {
    int n;
    if (std::scanf("%d", &n) != 1 || n < 1 || n > 400) return 2;
    std::vector<double> a((size_t)n*n), wr, wi;
    for (auto& v : a) if (std::scanf("%lf", &v) != 1) return 2;
    const int rc = hessenberg_eigenvalues(a, n, wr, wi);
    std::printf("%d\n", rc);
    for (int i = 0; i < n; i++) std::printf("%.17g %.17g\n", wr[i], wi[i]);
    return 0;
}
