// tests/cpp/test_shim.cpp -- C++ caller of the drop-in boundary, written like a reference call site
// (eul/HorizSolve.cpp:216-221:  M1->assemble(lev, SCALE, true); MatMult(M1->M, u, Mu);) and checked against the
// CPU oracle (oracle/oracle.h, test infrastructure).  Built and run by tests/test_gpu_cpp_shim.py on the GPU box.
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>
#include "../../mimsem_amd/host/mimsem_shim.hpp"
#include "../../oracle/oracle.h"

using namespace mimsem_host;

int main() {
    const int n = 3, nels = 4, nk = 2;
    const double SCALE = 1.0e8;
    std::mt19937_64 rng(7);
    std::uniform_real_distribution<double> U(0.5, 1.5), S(-1.0, 1.0);
    orc_patch* P = orc_patch_create(n, n, nels, nk);
    const int nEl = P->nEl, mp12 = P->mp12;
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
    Mesh mesh(&topo, &geom, 0);

    std::vector<double> u(P->n1), h(P->n2), Mu(P->n1), want(P->n1, 0.0);
    for (auto& v : u) v = S(rng);
    for (auto& v : h) v = U(rng)*1e6;
    double* d_u = mesh.to_device(u.data(), u.size());
    double* d_h = mesh.to_device(h.data(), h.size());
    double* d_y = mesh.to_device(Mu.data(), Mu.size());

    int fails = 0;
    auto compare = [&](const char* name) {
        mesh.to_host(Mu.data(), d_y, Mu.size());
        double num = 0, den = 0;
        for (size_t i = 0; i < Mu.size(); i++) { num += (Mu[i] - want[i])*(Mu[i] - want[i]); den += want[i]*want[i]; }
        const double err = std::sqrt(num/den);
        std::printf("%-8s rel L2 = %.3e\n", name, err);
        if (!(err < 1e-10)) fails++;
    };
    std::vector<double> em((size_t)nEl*orc_op_elmat_size(P, ORC_UMAT));

    // M1->assemble(lev, SCALE, true); MatMult(M1->M, u, Mu);
    Umat M1(&mesh, &node, &edge);
    M1.assemble(1, SCALE, true);
    M1.mult(d_u, d_y);
    orc_op_elmats(P, ORC_UMAT, 1, SCALE, 1, nullptr, em.data());
    std::fill(want.begin(), want.end(), 0.0);
    orc_op_apply(P, ORC_UMAT, em.data(), u.data(), want.data());
    compare("Umat");

    // F->assemble(h, lev, true, SCALE); MatMult(F->M, u, hu);
    Uhmat F(&mesh, &node, &edge);
    F.assemble(d_h, 0, true, SCALE);
    F.mult(d_u, d_y);
    orc_op_elmats(P, ORC_UHMAT, 0, SCALE, 1, h.data(), em.data());
    std::fill(want.begin(), want.end(), 0.0);
    orc_op_apply(P, ORC_UHMAT, em.data(), u.data(), want.data());
    compare("Uhmat");

    mimsem_free(d_u); mimsem_free(d_h); mimsem_free(d_y);
    orc_patch_destroy(P);
    std::printf(fails ? "FAILED\n" : "OK\n");
    return fails;
}
