// tests/cpp/test_ksp.cpp -- C++ call sites of the solve loops behind the C ABI (mimsem_ksp_*, round 4), written like the reference's:
//   HorizSolve::grad             eul/HorizSolve.cpp:208-228   M2 phi -> E12 -> KSPSolve(ksp1)
//   HorizSolve::diagnose_fluxes  eul/HorizSolve.cpp:285-328   four Uvec::assemble_hu -> KSPSolve(ksp1) -> F->assemble; MatMult -> KSPSolve(ksp1)
//   SWEqn::solve, one linear solve  src/SWEqn_Picard.cpp:751-765   KSPSolve(kspA, b, x) on the packed [u|h] operator
// and checked against DENSE solves of the oracle's assembled matrices (oracle/oracle.h: test infrastructure; Gaussian elimination with
// partial pivoting below).  Built and run by tests/test_gpu_cpp_shim.py on the GPU box.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>
#include "../../mimsem_amd/host/mimsem_shim.hpp"
#include "../../oracle/oracle.h"

using namespace mimsem_host;

// x <- A^-1 b, A dense n x n row-major (destroyed)
static void dense_solve(std::vector<double>& A, std::vector<double>& b, int n) {
    for (int c = 0; c < n; c++) {
        int p = c;
        for (int r = c + 1; r < n; r++) if (std::fabs(A[(size_t)r*n + c]) > std::fabs(A[(size_t)p*n + c])) p = r;
        if (p != c) { for (int j = 0; j < n; j++) std::swap(A[(size_t)p*n + j], A[(size_t)c*n + j]); std::swap(b[p], b[c]); }
        const double piv = A[(size_t)c*n + c];
        for (int r = c + 1; r < n; r++) {
            const double f = A[(size_t)r*n + c]/piv;
            if (f == 0.0) continue;
            for (int j = c; j < n; j++) A[(size_t)r*n + j] -= f*A[(size_t)c*n + j];
            b[r] -= f*b[c];
        }
    }
    for (int r = n - 1; r >= 0; r--) {
        double s = b[r];
        for (int j = r + 1; j < n; j++) s -= A[(size_t)r*n + j]*b[j];
        b[r] = s/A[(size_t)r*n + r];
    }
}
static double rel_l2(const std::vector<double>& a, const std::vector<double>& b) {
    double num = 0, den = 0;
    for (size_t i = 0; i < a.size(); i++) { num += (a[i] - b[i])*(a[i] - b[i]); den += b[i]*b[i]; }
    return std::sqrt(num/den);
}

int main() {
    const int n = 3, nels = 4, nk = 2, lev = 1;
    const double SCALE = 1.0e8;
    std::mt19937_64 rng(11);
    std::uniform_real_distribution<double> U(0.5, 1.5), S(-1.0, 1.0);
    orc_patch* P = orc_patch_create(n, n, nels, nk);
    const int nEl = P->nEl, mp12 = P->mp12, n1 = P->n1, n2 = P->n2;
    std::vector<double> det((size_t)nEl*mp12), J((size_t)nEl*mp12*4), levs((size_t)(nk + 1)*P->n0q);
    for (auto& v : det) v = U(rng)*1e10;
    for (size_t i = 0; i < det.size(); i++) { J[4*i] = 1e5*U(rng); J[4*i + 1] = 1e4*S(rng); J[4*i + 2] = 1e4*S(rng); J[4*i + 3] = 1e5*U(rng); }
    for (int k = 0; k <= nk; k++) for (int j = 0; j < P->n0q; j++) levs[(size_t)k*P->n0q + j] = 1000.0*k*(1.0 + 0.01*S(rng));
    orc_patch_set_metric(P, det.data(), J.data());
    orc_patch_set_levels(P, levs.data());
    Topo topo(n, nels, nk);
    Geom geom; geom.nk = nk; geom.quad_n = n; geom.nDofsX = n*nels; geom.det = det; geom.J = J;
    geom.thick.assign(P->thick, P->thick + (size_t)nk*P->n0q);
    geom.thickInv.assign(P->thickInv, P->thickInv + (size_t)nk*P->n0q);
    GaussLobatto quad{n}; LagrangeNode node{n, &quad}; LagrangeEdge edge{n, &node};
    Mesh& mesh = *Mesh::of(&topo, &geom);
    int fails = 0;
    auto report = [&](const char* name, double err, double tol, int its) {
        std::printf("%-18s rel L2 = %.3e  (%d iterations)\n", name, err, its);
        if (!(err < tol)) fails++;
    };

    // dense operators of the oracle: column j = the operator applied to e_j
    auto dense_of = [&](int op, int flag, const double* f, int nn) {
        std::vector<double> em((size_t)nEl*orc_op_elmat_size(P, op)), A((size_t)nn*nn), e(nn), col(nn);
        orc_op_elmats(P, op, lev, SCALE, flag, f, em.data());
        for (int j = 0; j < nn; j++) {
            std::fill(e.begin(), e.end(), 0.0); e[j] = 1.0;
            std::fill(col.begin(), col.end(), 0.0);
            orc_op_apply(P, op, em.data(), e.data(), col.data());
            for (int i = 0; i < nn; i++) A[(size_t)i*nn + j] = col[i];
        }
        return A;
    };
    auto matvec = [](const std::vector<double>& A, const std::vector<double>& x, int rows, int cols) {
        std::vector<double> y(rows, 0.0);
        for (int i = 0; i < rows; i++) { double s = 0; for (int j = 0; j < cols; j++) s += A[(size_t)i*cols + j]*x[j]; y[i] = s; }
        return y;
    };
    // E12 = -E21^T (eul/Assembly.cpp:1170-1220)
    std::vector<double> E21((size_t)n2*n1);
    {
        std::vector<double> e(n1), col(n2);
        for (int j = 0; j < n1; j++) {
            std::fill(e.begin(), e.end(), 0.0); e[j] = 1.0;
            orc_e21_apply(P, e.data(), col.data());
            for (int i = 0; i < n2; i++) E21[(size_t)i*n1 + j] = col[i];
        }
    }

    Umat M1(&topo, &geom, &node, &edge);
    Wmat M2(&topo, &geom, &edge);
    Uhmat F(&topo, &geom, &node, &edge);
    E21mat EtoF(&topo);
    Uvec m1(&topo, &geom, &node, &edge);
    // HorizSolve::HorizSolve, eul/HorizSolve.cpp:77-96
    KSP ksp1(&mesh);
    ksp1.setTolerances(1.0e-16, 1.0e-50, 1000);
    ksp1.setType(KSP::GMRES);
    ksp1.setPCBJacobi();

    // ---- HorizSolve::grad(assemble = true, phi, &u, lev) ----
    {
        std::vector<double> phi(n2), u(n1);
        for (auto& v : phi) v = U(rng)*1e6;
        double *d_phi = mesh.to_device(phi.data(), n2), *d_Mphi = mesh.device_alloc(n2), *d_dMphi = mesh.device_alloc(n1), *d_u = mesh.device_alloc(n1);
        M1.assemble(lev, SCALE, true);
        M2.assemble(lev, SCALE, true);
        ksp1.setOperators(M1);
        M2.mult(d_phi, d_Mphi);
        EtoF.mult_E12(d_Mphi, d_dMphi);
        KSPSolve(ksp1, d_dMphi, d_u);
        mesh.to_host(u.data(), d_u, n1);
        // oracle: dense M2 phi, E12 = -E21^T, dense M1 solve
        std::vector<double> A1 = dense_of(ORC_UMAT, 1, nullptr, n1), A2 = dense_of(ORC_WMAT, 1, nullptr, n2);
        std::vector<double> Mphi = matvec(A2, phi, n2, n2), rhs(n1, 0.0);
        for (int j = 0; j < n1; j++) { double s = 0; for (int i = 0; i < n2; i++) s -= E21[(size_t)i*n1 + j]*Mphi[i]; rhs[j] = s; }
        dense_solve(A1, rhs, n1);
        report("grad (GMRES)", rel_l2(u, rhs), 1e-10, ksp1.iterations());
        // the same solve by CG (M1 is symmetric positive definite: the same solution)
        KSP cg(&mesh, KSP::CG);
        cg.setTolerances(1.0e-15, 1.0e-50, 1000);
        cg.setPCBJacobi();
        cg.setOperators(M1);
        cg.solve(d_dMphi, d_u);
        mesh.to_host(u.data(), d_u, n1);
        report("grad (CG)", rel_l2(u, rhs), 1e-10, cg.iterations());
        if (ksp1.convergedReason() <= 0 || cg.convergedReason() <= 0) { std::printf("not converged: %d %d\n", ksp1.convergedReason(), cg.convergedReason()); fails++; }
        mimsem_free(d_phi); mimsem_free(d_Mphi); mimsem_free(d_dMphi); mimsem_free(d_u);
    }
    // ---- the reference's own call ORDER (eul/HorizSolve.cpp:77-84: KSPSetOperators, then KSPSetType, then PCSetType(PCBJACOBI)) and the
    // 2-form / 0-form solves (ksp2 on M2: eul/HorizSolve.cpp:86-96; ksp0 on M0: src/SWEqn_Picard.cpp:84-96).  The preconditioner must be in
    // place whatever the order (same iteration count as ksp1 above), a type change must keep the operator, and the element blocks of a
    // 2-form (block-diagonal) and of a 0-form (diagonal, collocated) mass matrix invert it exactly: one or two iterations.
    {
        std::vector<double> b1(n1), x1(n1), b2(n2), x2(n2), b0(P->n0), x0(P->n0);
        for (auto& v : b1) v = S(rng); for (auto& v : b2) v = S(rng); for (auto& v : b0) v = S(rng);
        double *d_b = mesh.to_device(b1.data(), n1), *d_x = mesh.device_alloc(n1);
        M1.assemble(lev, SCALE, true);
        KSP k1(&mesh, KSP::CG);                              // created as something else on purpose
        k1.setOperators(M1);                                 // KSPSetOperators first ...
        k1.setTolerances(1.0e-16, 1.0e-50, 1000);
        k1.setType(KSP::GMRES);                              // ... then the type (re-creates the library object) ...
        k1.setPCBJacobi();                                   // ... then the preconditioner
        KSPSolve(k1, d_b, d_x);
        mesh.to_host(x1.data(), d_x, n1);
        ksp1.setOperators(M1);
        KSPSolve(ksp1, d_b, d_x);                            // the order the shim's own tests used so far
        std::vector<double> A1 = dense_of(ORC_UMAT, 1, nullptr, n1), w1 = b1;
        dense_solve(A1, w1, n1);
        report("M1, PETSc order", rel_l2(x1, w1), 1e-10, k1.iterations());
        if (k1.iterations() != ksp1.iterations() || k1.convergedReason() <= 0) { std::printf("order-dependent solve: %d vs %d iterations\n", k1.iterations(), ksp1.iterations()); fails++; }
        KSP kn(&mesh, KSP::GMRES);                           // the same solve unpreconditioned needs visibly more iterations
        kn.setTolerances(1.0e-16, 1.0e-50, 1000);
        kn.setOperators(M1);
        KSPSolve(kn, d_b, d_x);
        if (kn.iterations() <= k1.iterations()) { std::printf("PCBJACOBI did nothing: %d vs %d iterations\n", k1.iterations(), kn.iterations()); fails++; }
        mimsem_free(d_b); mimsem_free(d_x);

        M2.assemble(lev, SCALE, true);
        double *d_b2 = mesh.to_device(b2.data(), n2), *d_x2 = mesh.device_alloc(n2);
        KSP k2(&mesh);
        k2.setOperators(M2);
        k2.setTolerances(1.0e-16, 1.0e-50, 1000);
        k2.setType(KSP::GMRES);
        k2.setPCBJacobi();
        KSPSolve(k2, d_b2, d_x2);
        mesh.to_host(x2.data(), d_x2, n2);
        std::vector<double> A2 = dense_of(ORC_WMAT, 1, nullptr, n2), w2 = b2;
        dense_solve(A2, w2, n2);
        report("M2, BJACOBI", rel_l2(x2, w2), 1e-10, k2.iterations());
        if (k2.iterations() > 2 || k2.convergedReason() <= 0) { std::printf("2-form element blocks are not the exact inverse: %d iterations\n", k2.iterations()); fails++; }
        mimsem_free(d_b2); mimsem_free(d_x2);

        Pmat M0(&topo, &geom, &node);
        M0.assemble(lev, SCALE);
        double *d_b0 = mesh.to_device(b0.data(), P->n0), *d_x0 = mesh.device_alloc(P->n0);
        KSP k0(&mesh);
        k0.setOperators(M0);
        k0.setTolerances(1.0e-16, 1.0e-50, 1000);
        k0.setPCBJacobi();
        KSPSolve(k0, d_b0, d_x0);
        mesh.to_host(x0.data(), d_x0, P->n0);
        std::vector<double> A0 = dense_of(ORC_PMAT, 0, nullptr, P->n0), w0 = b0;
        dense_solve(A0, w0, P->n0);
        report("M0, BJACOBI", rel_l2(x0, w0), 1e-10, k0.iterations());
        if (k0.iterations() > 2 || k0.convergedReason() <= 0) { std::printf("0-form element blocks are not the exact inverse: %d iterations\n", k0.iterations()); fails++; }
        mimsem_free(d_b0); mimsem_free(d_x0);
    }
    // ---- HorizSolve::diagnose_fluxes(level, u1, u2, h1l, h2l, theta_l, _F, _G, u1l, u2l, theta_in_Wt = false) ----
    {
        std::vector<double> u1(n1), u2(n1), h1(n2), h2(n2), th(n2), Fh(n1), Gh(n1);
        for (auto& v : u1) v = S(rng); for (auto& v : u2) v = S(rng);
        for (auto& v : h1) v = U(rng)*1e6; for (auto& v : h2) v = U(rng)*1e6; for (auto& v : th) v = U(rng)*3e8;
        double *d_u1 = mesh.to_device(u1.data(), n1), *d_u2 = mesh.to_device(u2.data(), n1), *d_h1 = mesh.to_device(h1.data(), n2),
               *d_h2 = mesh.to_device(h2.data(), n2), *d_th = mesh.to_device(th.data(), n2);
        double *d_hu = mesh.device_alloc(n1), *d_F = mesh.device_alloc(n1), *d_G = mesh.device_alloc(n1);
        check(mimsem_memset(mesh.ctx, d_hu, 0, (long long)n1*8), "memset");
        m1.assemble_hu(lev, SCALE, d_u1, d_h1, false, 1.0/3.0, d_hu);
        m1.assemble_hu(lev, SCALE, d_u1, d_h2, false, 1.0/6.0, d_hu);
        m1.assemble_hu(lev, SCALE, d_u2, d_h1, false, 1.0/6.0, d_hu);
        m1.assemble_hu(lev, SCALE, d_u2, d_h2, false, 1.0/3.0, d_hu);
        M1.assemble(lev, SCALE, true);
        ksp1.setOperators(M1);
        KSPSolve(ksp1, d_hu, d_F);
        const int its_F = ksp1.iterations();
        F.assemble(d_th, lev, true, SCALE);
        F.mult(d_F, d_hu);
        KSPSolve(ksp1, d_hu, d_G);
        mesh.to_host(Fh.data(), d_F, n1); mesh.to_host(Gh.data(), d_G, n1);
        // oracle
        std::vector<double> hu(n1, 0.0);
        {   // (orc_uvec_hu is the zero_and_scatter form: it overwrites -- the four contributions are summed here)
            std::vector<double> part(n1);
            const double* us[4] = {u1.data(), u1.data(), u2.data(), u2.data()}; const double* hs[4] = {h1.data(), h2.data(), h1.data(), h2.data()};
            const double fac[4] = {1.0/3.0, 1.0/6.0, 1.0/6.0, 1.0/3.0};
            for (int t = 0; t < 4; t++) { orc_uvec_hu(P, lev, SCALE, us[t], hs[t], fac[t], part.data()); for (int i = 0; i < n1; i++) hu[i] += part[i]; }
        }
        std::vector<double> A1 = dense_of(ORC_UMAT, 1, nullptr, n1), A1b = A1, AF = dense_of(ORC_UHMAT, 1, th.data(), n1);
        dense_solve(A1, hu, n1);                       // hu <- F
        report("fluxes: F", rel_l2(Fh, hu), 1e-10, its_F);
        std::vector<double> g = matvec(AF, hu, n1, n1);
        dense_solve(A1b, g, n1);
        report("fluxes: G", rel_l2(Gh, g), 1e-10, ksp1.iterations());
        mimsem_free(d_u1); mimsem_free(d_u2); mimsem_free(d_h1); mimsem_free(d_h2); mimsem_free(d_th); mimsem_free(d_hu); mimsem_free(d_F); mimsem_free(d_G);
    }
    // ---- SWEqn::solve: KSPSolve(kspA, b, x) on the packed [u|h] operator (src/SWEqn_Picard.cpp:751-765).  The operator itself is checked
    // against the Python oracle elsewhere (tests/test_gpu_sweqn.py); here the LOOP: GMRES on the matrix-free operator against a dense solve of
    // the matrix that operator defines (columns = applies to unit vectors).  src/ flavour: unit scale, flat levels.
    {
        const int N = n1 + n2;
        const double a = 0.5*120.0, grav = 9.80616, H = 1.0e4;
        std::vector<double> f0(P->n0), b(N), x(N), col(N), e(N);
        for (auto& v : f0) v = 1.0e-4*S(rng);
        for (auto& v : b) v = S(rng);
        double *d_f0 = mesh.to_device(f0.data(), P->n0), *d_b = mesh.to_device(b.data(), N), *d_x = mesh.device_alloc(N),
               *d_e = mesh.device_alloc(N), *d_c = mesh.device_alloc(N);
        std::vector<double> A((size_t)N*N);
        for (int j = 0; j < N; j++) {
            std::fill(e.begin(), e.end(), 0.0); e[j] = 1.0;
            check(mimsem_memcpy_h2d(mesh.ctx, d_e, e.data(), (long long)N*8), "h2d");
            check(mimsem_sw_operator_apply(mesh.ctx, 1, a, grav, H, d_f0, 0, d_e, N, d_c, N), "sw_operator_apply");
            mesh.to_host(col.data(), d_c, N);
            for (int i = 0; i < N; i++) A[(size_t)i*N + j] = col[i];
        }
        KSP kspA(&mesh, KSP::GMRES);
        kspA.setTolerances(1.0e-14, 1.0e-50, 1000);
        kspA.setPCBJacobi();                                  // coupled [u|h] element blocks, built by the library from the operator
        kspA.setOperatorsSW(a, grav, H, d_f0);
        KSPSolve(kspA, d_b, d_x);
        mesh.to_host(x.data(), d_x, N);
        std::vector<double> want = b;
        dense_solve(A, want, N);
        report("kspA (GMRES)", rel_l2(x, want), 1e-9, kspA.iterations());
        if (kspA.convergedReason() <= 0) { std::printf("kspA not converged: reason %d, rnorm %.2e\n", kspA.convergedReason(), kspA.residualNorm()); fails++; }
        mimsem_free(d_f0); mimsem_free(d_b); mimsem_free(d_x); mimsem_free(d_e); mimsem_free(d_c);
    }
    orc_patch_destroy(P);
    Mesh::release_all();
    std::printf(fails ? "FAILED\n" : "OK\n");
    return fails;
}
