// tests/cpp/test_horiz.cpp -- the right-hand sides of the horizontal dynamics (row N2) driven from C++: mimsem_host::HorizSolve
// (mimsem_amd/host/mimsem_horizsolve.hpp, the counterpart of eul/HorizSolve.cpp:208-786) on the mesh and fields the pytest wrapper wrote,
// every level in one call.  Results go back to the wrapper, which compares them with the dense restatement oracle/horiz_oracle.py
// (per-level dense matrices, LU for the KSP solves) -- the same check the Python host gets in tests/test_gpu_next_rows.py.
//   usage: test_horiz <in.arr> <out.bin> [ksp]
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
#include "../../mimsem_amd/host/mimsem_horizsolve.hpp"
#include "../../mimsem_amd/host/sw_io.hpp"

using namespace mimsem_host;

int main(int argc, char** argv) {
    if (argc < 3) { std::fprintf(stderr, "usage: test_horiz in.arr out.bin\n"); return 2; }
    try {
        const ArrayFile a = read_arrays(argv[1]);
        const mimsem_mesh_desc d = desc_of(a);
        Mesh mesh(d);
        const int nk = d.nk, n0 = d.n0, n1 = d.n1, n2 = d.n2;
        auto dev = [&](const char* k) { const auto& v = a.reals(k); return mesh.to_device(v.data(), v.size()); };
        double *fg = dev("fg"), *u1 = dev("u1"), *u2 = dev("u2"), *h1 = dev("h1"), *h2 = dev("h2"), *th = dev("theta"), *Pi = dev("Pi");
        double *velz = dev("velz1"), *velz2 = dev("velz2"), *dudz = dev("dudz1"), *dudz2 = dev("dudz2"), *Fz = dev("Fz");
        double *dwdx1 = dev("dwdx1"), *dwdx2 = dev("dwdx2");
        HorizSolve hs(&mesh, fg);
        if (argc > 3 && argv[3][0] == 'k') hs.use_fixed_length(false);          // "ksp": the CG of the reference's structure
        std::printf("del2 = %.12e   M1 solves: %s (%d steps)\n", hs.del2, hs.fixed_length ? "fixed-length Chebyshev" : "CG", hs.cheb_steps);
        const size_t s1 = (size_t)nk*n1, s2 = (size_t)nk*n2, s0 = (size_t)nk*n0;
        double *dF = mesh.device_alloc(s2), *dG = mesh.device_alloc(s2), *Fk = mesh.device_alloc(s1), *Gk = mesh.device_alloc(s1);
        double *Phi = mesh.device_alloc(s2), *q = mesh.device_alloc(s0), *fuA = mesh.device_alloc(s1), *fuB = mesh.device_alloc(s1), *fuC = mesh.device_alloc(s1);
        hs.advection_rhs_ec(u1, u2, h1, h2, th, dF, dG, Fk, Gk);
        std::printf("M1 solve: %d iterations\n", hs.last_its);
        hs.diagnose_Phi(u1, u2, velz, velz2, Phi);
        hs.diagnose_q(h1, u1, q);
        hs.momentum_rhs_ec(th, dudz, dudz2, velz, velz2, Pi, u1, u2, h1, h2, fuA, nullptr, nullptr, nullptr, nullptr, Fk);
        const double k2iA = hs.k2i();
        hs.momentum_rhs_ec(th, dudz, dudz2, velz, velz2, Pi, u1, u2, h1, h2, fuB, Fk, Fz, nullptr, nullptr, Fk);
        const double k2iB = hs.k2i();
        hs.momentum_rhs_ec(th, dudz, dudz2, velz, velz2, Pi, u1, u2, h1, h2, fuC, Fk, Fz, dwdx1, dwdx2, Fk);      // + the horizontal gradient of w (:704-712)
        // every fixed-length solve above logged its check norms on the device: ONE read for all of them (round 6: not just the first three)
        const bool fixed = hs.fixed_length;
        if (!hs.verify()) { std::printf("FAIL: a fixed-length 1-form mass solve missed its check (worst %.2e)\n", hs.worst_rel); return 1; }
        if (fixed && hs.solves_checked < 15) { std::printf("FAIL: %d solves checked, 15+ expected\n", hs.solves_checked); return 1; }
        std::printf("checked %d fixed-length solves, worst |P r| / |P b| = %.2e\n", hs.solves_checked, hs.worst_rel);
        // a sabotaged interval (half the steps) must be caught by the same log and hand the solves to the CG
        if (fixed) {
            HorizSolve bad(&mesh, fg);
            bad.rtol = 1.0e-14;
            bad.shorten_for_test(4);
            double *tA = mesh.device_alloc(s1), *tB = mesh.device_alloc(s1);
            bad.grad(Pi, tB);
            if (bad.verify() || bad.fixed_length || bad.solves_missed != 1) { std::printf("FAIL: a 4-step Chebyshev solve passed its check\n"); return 1; }
            bad.grad(Pi, tB);                                          // now the CG
            if (bad.last_its < 5) { std::printf("FAIL: the CG did not take over\n"); return 1; }
            hs.grad(Pi, tA);
            if (!hs.verify()) { std::printf("FAIL: check missed\n"); return 1; }
            std::vector<double> ha(s1), hb(s1);
            mesh.to_host(ha.data(), tA, s1); mesh.to_host(hb.data(), tB, s1);
            double e2 = 0.0, r2 = 0.0;
            for (size_t i = 0; i < s1; i++) { e2 += (ha[i] - hb[i])*(ha[i] - hb[i]); r2 += ha[i]*ha[i]; }
            if (!(std::sqrt(e2/r2) < 1.0e-11)) { std::printf("FAIL: CG after the fallback vs fixed-length: %.2e\n", std::sqrt(e2/r2)); return 1; }
            // the whole solve as one call (the default on one context) and the sweep calls give the same bits
            if (!hs.whole_solve) { std::printf("FAIL: the whole-solve entry is not in use\n"); return 1; }
            hs.whole_solve = false;
            hs.grad(Pi, tB);
            hs.whole_solve = true;
            if (!hs.verify()) { std::printf("FAIL: check missed (sweep calls)\n"); return 1; }
            mesh.to_host(hb.data(), tB, s1);
            if (std::memcmp(ha.data(), hb.data(), s1*sizeof(double)) != 0) {
                double worst = 0.0; for (size_t i = 0; i < s1; i++) worst = std::max(worst, std::fabs(ha[i] - hb[i]));
                if (worst != 0.0) { std::printf("FAIL: whole solve vs sweep calls: max difference %.3e (the same bits expected)\n", worst); return 1; }
            }
            mimsem_free(tA); mimsem_free(tB);
        }
        FILE* g = std::fopen(argv[2], "wb");
        if (!g) { std::perror(argv[2]); return 2; }
        auto put = [&](const double* p, size_t n) { std::vector<double> h(n); mesh.to_host(h.data(), p, n); std::fwrite(h.data(), 8, n, g); };
        put(dF, s2); put(dG, s2); put(Fk, s1); put(Gk, s1); put(Phi, s2); put(q, s0); put(fuA, s1); put(fuB, s1); put(fuC, s1);
        const double tail[3] = {k2iA, k2iB, hs.del2};
        std::fwrite(tail, 8, 3, g);
        std::fclose(g);
        for (double* p : {fg, u1, u2, h1, h2, th, Pi, velz, velz2, dudz, dudz2, Fz, dwdx1, dwdx2, dF, dG, Fk, Gk, Phi, q, fuA, fuB, fuC}) mimsem_free(p);
    } catch (const std::exception& e) { std::printf("FAIL: %s\n", e.what()); return 1; }
    std::printf("DONE\n");
    return 0;
}
