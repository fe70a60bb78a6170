// tests/cpp/test_shim.cpp -- C++ caller of the drop-in boundary, written like a reference call site
// (eul/HorizSolve.cpp:216-221:  M1->assemble(lev, SCALE, true); MatMult(M1->M, u, Mu);) and checked against the
// CPU oracle (oracle/oracle.h, test infrastructure).  Built and run by tests/test_gpu_cpp_shim.py on the GPU box.
#include <algorithm>
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
    Mesh& mesh = *Mesh::of(&topo, &geom);          // what every (Topo*, Geom*, ...) constructor below resolves to

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
    Umat M1(&topo, &geom, &node, &edge);                  // the reference's constructor signature (eul/Assembly.h:3)
    M1.assemble(1, SCALE, true);
    M1.mult(d_u, d_y);
    orc_op_elmats(P, ORC_UMAT, 1, SCALE, 1, nullptr, em.data());
    std::fill(want.begin(), want.end(), 0.0);
    orc_op_apply(P, ORC_UMAT, em.data(), u.data(), want.data());
    compare("Umat");

    // F->assemble(h, lev, true, SCALE); MatMult(F->M, u, hu);
    Uhmat F(&topo, &geom, &node, &edge);
    F.assemble(d_h, 0, true, SCALE);
    F.mult(d_u, d_y);
    orc_op_elmats(P, ORC_UHMAT, 0, SCALE, 1, h.data(), em.data());
    std::fill(want.begin(), want.end(), 0.0);
    orc_op_apply(P, ORC_UHMAT, em.data(), u.data(), want.data());
    compare("Uhmat");


    // ---- the reference's per-level loop recorded once (Graph::record -> mimsem_graph_*) and replayed: the replay must write the same bits as
    // the loop itself, see its inputs' CURRENT values, and a call that cannot be recorded must say so and leave the context usable
    {
        Wmat M2(&topo, &geom, &edge);
        std::vector<double> x2((size_t)nk*P->n2), y2((size_t)nk*P->n2), r2((size_t)nk*P->n2), y1((size_t)nk*P->n1), r1((size_t)nk*P->n1), uu((size_t)nk*P->n1);
        for (auto& v : x2) v = S(rng); for (auto& v : uu) v = S(rng);
        double *d_x2 = mesh.to_device(x2.data(), x2.size()), *d_y2 = mesh.device_alloc(x2.size()), *d_uu = mesh.to_device(uu.data(), uu.size()), *d_y1 = mesh.device_alloc(uu.size());
        auto loop = [&]() {
            for (int kk = 0; kk < nk; kk++) {
                M1.assemble(kk, SCALE, true); M1.mult(d_uu + (size_t)kk*P->n1, d_y1 + (size_t)kk*P->n1);
                M2.assemble(kk, SCALE, true); M2.mult(d_x2 + (size_t)kk*P->n2, d_y2 + (size_t)kk*P->n2);
            }
        };
        loop();
        mesh.to_host(r1.data(), d_y1, r1.size()); mesh.to_host(r2.data(), d_y2, r2.size());
        Graph g(&mesh);
        g.record(loop);
        check(mimsem_memset(mesh.ctx, d_y1, 0, (long long)(r1.size()*8)), "memset"); check(mimsem_memset(mesh.ctx, d_y2, 0, (long long)(r2.size()*8)), "memset");
        g.launch();
        mesh.to_host(y1.data(), d_y1, y1.size()); mesh.to_host(y2.data(), d_y2, y2.size());
        bool same = y1 == r1 && y2 == r2;
        // new values in the same arrays: the replay reads them
        for (auto& v : uu) v = S(rng);
        check(mimsem_memcpy_h2d(mesh.ctx, d_uu, uu.data(), (long long)(uu.size()*8)), "h2d");
        loop(); mesh.to_host(r1.data(), d_y1, r1.size());
        check(mimsem_memset(mesh.ctx, d_y1, 0, (long long)(r1.size()*8)), "memset");
        g.launch(); g.launch();
        mesh.to_host(y1.data(), d_y1, y1.size());
        same = same && y1 == r1 && g.nodes() >= 2*nk;
        std::printf("%-8s %s (%d nodes)\n", "Graph", same ? "bit-exact replay" : "MISMATCH", g.nodes()); if (!same) fails++;
        // what cannot be recorded is refused, and the context survives: a device-to-host copy inside a recording
        int rc_bad = MIMSEM_OK;
        check(mimsem_graph_begin(mesh.ctx), "graph_begin");
        if (mimsem_graph_begin(mesh.ctx) != MIMSEM_ERR_STATE) fails++;                      // (no nesting)
        rc_bad = mimsem_ctx_set_stream(mesh.ctx, nullptr);
        mimsem_graph* junk = nullptr;
        const int rc_end = mimsem_graph_end(mesh.ctx, &junk);
        mimsem_graph_destroy(junk);
        if (rc_bad != MIMSEM_ERR_STATE || rc_end != MIMSEM_OK || mimsem_graph_end(mesh.ctx, &junk) != MIMSEM_ERR_STATE) { std::printf("Graph    state codes %d %d\n", rc_bad, rc_end); fails++; }
        M1.assemble(1, SCALE, true);                                                     // eager calls work as before
        M1.mult(d_u, d_y);
        orc_op_elmats(P, ORC_UMAT, 1, SCALE, 1, nullptr, em.data());
        std::fill(want.begin(), want.end(), 0.0);
        orc_op_apply(P, ORC_UMAT, em.data(), u.data(), want.data());
        compare("Umat'");
        mimsem_free(d_x2); mimsem_free(d_y2); mimsem_free(d_uu); mimsem_free(d_y1);
    }

    // ---- upwinded test functions: M1->assemble_up(lev, SCALE, tau, ui, uj); MatMult(M1->M, ..) and MatMult(M1->MT, ..)  (eul/Assembly.cpp:156-279)
    {
        double tI = 0.0; for (int j = 0; j < P->n0q; j++) tI += P->thickInv[(size_t)1*P->n0q + j]; tI /= P->n0q;
        double dm = 0.0; for (double v : det) dm += v; dm /= det.size();
        const double tau = 75.0, vs = dm/tI*0.3/tau;                    // departure shift ~0.3 of the reference element
        std::vector<double> ui(P->n1), uj(P->n1);
        for (auto& v : ui) v = S(rng)*vs;
        for (auto& v : uj) v = S(rng)*vs;
        double *d_ui = mesh.to_device(ui.data(), ui.size()), *d_uj = mesh.to_device(uj.data(), uj.size());
        M1.assemble_up(1, SCALE, tau, d_ui, d_uj);
        M1.mult(d_u, d_y);
        orc_op_elmats_testup(P, 0, 1, SCALE, tau, ui.data(), uj.data(), em.data());
        std::fill(want.begin(), want.end(), 0.0);
        orc_op_apply(P, ORC_UMAT, em.data(), u.data(), want.data());
        compare("Umat_up");
        M1.mult_MT(d_u, d_y);
        {   // MT: transpose every element's 2x2 block matrix [UU UV; VU VV]
            const int n1e = P->n1e; std::vector<double> et(em.size());
            for (int e = 0; e < nEl; e++) for (int b = 0; b < 4; b++) for (int i = 0; i < n1e; i++) for (int j = 0; j < n1e; j++) {
                const int bt = (b == 1) ? 2 : (b == 2 ? 1 : b);
                et[((size_t)e*4 + bt)*n1e*n1e + (size_t)j*n1e + i] = em[((size_t)e*4 + b)*n1e*n1e + (size_t)i*n1e + j];
            }
            std::fill(want.begin(), want.end(), 0.0);
            orc_op_apply(P, ORC_UMAT, et.data(), u.data(), want.data());
        }
        compare("Umat MT");
        F.assemble_up(d_h, 1, SCALE, tau, d_ui);
        F.mult(d_u, d_y);
        orc_op_elmats_testup(P, 1, 1, SCALE, tau, h.data(), ui.data(), em.data());
        std::fill(want.begin(), want.end(), 0.0);
        orc_op_apply(P, ORC_UMAT, em.data(), u.data(), want.data());
        compare("Uhmat_up");
        // Uvec::assemble_hu_up(lev, scale, vel, rho, fac, tau, vel2)  (eul/Assembly.cpp:2281-2373; accumulates into vl)
        Uvec uv(&topo, &geom, &node, &edge);
        std::fill(want.begin(), want.end(), 0.0);
        check(mimsem_memset(mesh.ctx, d_y, 0, (long long)(Mu.size()*sizeof(double))), "memset");
        uv.assemble_hu_up(1, SCALE, d_u, d_h, 1.0/3.0, tau, d_uj, d_y);
        orc_uvec_hu_up(P, 1, SCALE, u.data(), h.data(), 1.0/3.0, tau, uj.data(), want.data());
        compare("hu_up");
        M1.assemble(1, SCALE, true);                                    // back to the plain operator for what follows
        mimsem_free(d_ui); mimsem_free(d_uj);
    }
    // ---- src/ flavour (src/Assembly.h): no lev / scale; Phmat::assemble_up, RotMat_up::assemble  (src/Assembly.cpp:499-567, 1784-1853)
    {
        double dm = 0.0; for (double v : det) dm += v; dm /= det.size();
        const double fac = 0.5, dts = 600.0;
        std::vector<double> ul(P->n1), x0(P->n0), q0(P->n0), y0(P->n0), w0(P->n0, 0.0), e0((size_t)nEl*orc_op_elmat_size(P, ORC_PMAT)), e1((size_t)nEl*orc_op_elmat_size(P, ORC_ROTMAT));
        for (auto& v : ul) v = S(rng)*dm*0.2/(fac*dts);
        for (auto& v : x0) v = S(rng);
        for (auto& v : q0) v = S(rng)*1e-4;
        double *d_ul = mesh.to_device(ul.data(), ul.size()), *d_x0 = mesh.to_device(x0.data(), x0.size()), *d_q0 = mesh.to_device(q0.data(), q0.size());
        double* d_y0 = mesh.device_alloc(P->n0);
        src::Phmat M0h(&topo, &geom, &node);
        M0h.assemble_up(d_ul, d_h, fac, dts);
        M0h.mult(d_x0, d_y0); mesh.to_host(y0.data(), d_y0, y0.size());
        orc_op_elmats_up(P, 0, fac, dts, h.data(), ul.data(), e0.data());
        orc_op_apply(P, ORC_PMAT, e0.data(), x0.data(), w0.data());
        double num = 0, den = 0; for (size_t i = 0; i < y0.size(); i++) { num += (y0[i] - w0[i])*(y0[i] - w0[i]); den += w0[i]*w0[i]; }
        std::printf("%-8s rel L2 = %.3e\n", "Phmat_up", std::sqrt(num/den)); if (!(std::sqrt(num/den) < 1e-10)) fails++;
        src::RotMat_up R(&topo, &geom, &node, &edge);
        R.assemble(d_q0, d_ul, fac, dts);
        R.mult(d_u, d_y);
        orc_op_elmats_up(P, 1, fac, dts, q0.data(), ul.data(), e1.data());
        std::fill(want.begin(), want.end(), 0.0);
        orc_op_apply(P, ORC_ROTMAT, e1.data(), u.data(), want.data());
        compare("RotMat_up");
        src::Umat M1s(&topo, &geom, &node, &edge);
        M1s.assemble();
        M1s.mult(d_u, d_y);
        orc_op_elmats(P, ORC_UMAT, 0, 1.0, 0, nullptr, em.data());
        std::fill(want.begin(), want.end(), 0.0);
        orc_op_apply(P, ORC_UMAT, em.data(), u.data(), want.data());
        compare("src Umat");
        mimsem_free(d_ul); mimsem_free(d_x0); mimsem_free(d_q0); mimsem_free(d_y0);
    }
    // ---- incidence, Pvec, projections -------------------------------------------------------------------
    {
        std::vector<double> x0(P->n0), y1(P->n1), w1(P->n1, 0.0), y2(P->n2), w2(P->n2, 0.0), pv(P->n0), wp(P->n0, 0.0);
        for (auto& v : x0) v = S(rng);
        double *d_x0 = mesh.to_device(x0.data(), x0.size()), *d_y1 = mesh.device_alloc(P->n1), *d_y2 = mesh.device_alloc(P->n2), *d_p = mesh.device_alloc(P->n0);
        E10mat NtoE(&topo); E21mat EtoF(&topo);
        NtoE.mult_E10(d_x0, d_y1); mesh.to_host(y1.data(), d_y1, y1.size()); orc_e10_apply(P, x0.data(), w1.data());
        EtoF.mult_E21(d_u, d_y2);  mesh.to_host(y2.data(), d_y2, y2.size()); orc_e21_apply(P, u.data(), w2.data());
        bool same = true;
        for (size_t i = 0; i < y1.size(); i++) same = same && y1[i] == w1[i];
        for (size_t i = 0; i < y2.size(); i++) same = same && y2[i] == w2[i];
        std::printf("%-8s %s\n", "E10/E21", same ? "bit-exact" : "MISMATCH"); if (!same) fails++;
        Pvec m0(&topo, &geom, &node); m0.assemble(1, SCALE, d_p); mesh.to_host(pv.data(), d_p, pv.size()); orc_pvec(P, 1, SCALE, wp.data());
        double num = 0, den = 0; for (size_t i = 0; i < pv.size(); i++) { num += (pv[i] - wp[i])*(pv[i] - wp[i]); den += wp[i]*wp[i]; }
        std::printf("%-8s rel L2 = %.3e\n", "Pvec", std::sqrt(num/den)); if (!(std::sqrt(num/den) < 1e-10)) fails++;
        std::vector<double> xq(P->n0q), pq(P->n0), wq(P->n0, 0.0);
        for (auto& v : xq) v = S(rng);
        double* d_xq = mesh.to_device(xq.data(), xq.size());
        PtQmat PtQ(&topo, &geom, &node); PtQ.mult(d_xq, d_p); mesh.to_host(pq.data(), d_p, pq.size()); orc_project_from_quad(P, 1, xq.data(), wq.data());
        num = den = 0; for (size_t i = 0; i < pq.size(); i++) { num += (pq[i] - wq[i])*(pq[i] - wq[i]); den += wq[i]*wq[i]; }
        std::printf("%-8s rel L2 = %.3e\n", "PtQmat", std::sqrt(num/den)); if (!(std::sqrt(num/den) < 1e-10)) fails++;
        mimsem_free(d_x0); mimsem_free(d_y1); mimsem_free(d_y2); mimsem_free(d_p); mimsem_free(d_xq);
    }
    // ---- geom->interp1_g(ex, ey, px, py, array_1, un) / interp2_g at every point (src/SWEqn_Picard.cpp:1101, 1168) -------------
    {
        std::vector<double> g1((size_t)nEl*mp12*2), g2((size_t)nEl*mp12);
        double *d_g1 = mesh.device_alloc(g1.size()), *d_g2 = mesh.device_alloc(g2.size());
        mesh.interp1_g(d_u, d_g1); mesh.interp2_g(d_h, d_g2);
        mesh.to_host(g1.data(), d_g1, g1.size()); mesh.to_host(g2.data(), d_g2, g2.size());
        double num = 0, den = 0;
        for (int e = 0; e < nEl; e++) for (int q = 0; q < mp12; q++) {
            double un[2], hn[1];
            orc_interp1_g(P, e%nels, e/nels, q%(n + 1), q/(n + 1), u.data(), un);
            orc_interp2_g(P, e%nels, e/nels, q%(n + 1), q/(n + 1), h.data(), hn);
            const double* a = &g1[((size_t)e*mp12 + q)*2];
            num += (a[0] - un[0])*(a[0] - un[0])/(un[0]*un[0] + un[1]*un[1]) + (a[1] - un[1])*(a[1] - un[1])/(un[0]*un[0] + un[1]*un[1]);
            num += (g2[(size_t)e*mp12 + q] - hn[0])*(g2[(size_t)e*mp12 + q] - hn[0])/(hn[0]*hn[0]);
            den += 2.0;
        }
        std::printf("%-8s rel L2 = %.3e\n", "interp_g", std::sqrt(num/den)); if (!(std::sqrt(num/den) < 1e-12)) fails++;
        mimsem_free(d_g1); mimsem_free(d_g2);
    }
    // ---- L2Vecs, VertOps, VertSolve: the column path as eul/VertSolve.cpp drives it ---------------------------
    {
        const int n2e = P->n2e, N = nk*n2e;
        L2Vecs rho(nk, &topo, &geom);
        std::vector<double> vh((size_t)nk*P->n2), vz((size_t)nEl*N), wz((size_t)nEl*N);
        for (auto& v : vh) v = U(rng)*1e9;
        rho.CopyFromHoriz(vh.data()); rho.HorizToVert();
        mesh.to_host(vz.data(), rho.vz, vz.size()); orc_horiz_to_vert(P, vh.data(), wz.data());
        bool same = true; for (size_t i = 0; i < vz.size(); i++) same = same && vz[i] == wz[i];
        std::printf("%-8s %s\n", "L2Vecs", same ? "bit-exact" : "MISMATCH"); if (!same) fails++;

        // vo->AssembleConstWithRho(ex, ey, rho, vo->VB); MatMult(vo->VB, a, b);   (every column at once)
        VertOps vo(&topo, &geom);
        std::vector<double> a((size_t)nEl*N), b((size_t)nEl*N), dense((size_t)N*N);
        for (auto& v : a) v = S(rng);
        double *d_a = mesh.to_device(a.data(), a.size()), *d_b = mesh.device_alloc(a.size());
        vo.AssembleConstWithRho(rho.vz); vo.mult(d_a, d_b); mesh.to_host(b.data(), d_b, b.size());
        double worst = 0.0;
        for (int e : {0, nEl - 1}) {
            orc_colop_dense(P, ORC_V_CONST_RHO, e%nels, e/nels, 0, &vz[(size_t)e*N], nullptr, dense.data());
            double num = 0, den = 0;
            for (int i = 0; i < N; i++) { double s = 0; for (int j = 0; j < N; j++) s += dense[(size_t)i*N + j]*a[(size_t)e*N + j];
                                          num += (b[(size_t)e*N + i] - s)*(b[(size_t)e*N + i] - s); den += s*s; }
            worst = std::max(worst, std::sqrt(num/den));
        }
        std::printf("%-8s rel L2 = %.3e\n", "VertOps", worst); if (!(worst < 1e-10)) fails++;
        mimsem_free(d_a); mimsem_free(d_b);

        // solve_schur_column_eta(ex, ey, theta, velz, rho, eta, pi, F_u, F_rho, F_eta, F_pi, d_u, d_rho, d_eta, d_pi)  eul/VertSolve.cpp:1868
        double area = 0.0, dz = 0.0;
        for (double v : det) area += v; area = area/det.size()*4.0/n2e;
        for (size_t i = 0; i < (size_t)nk*P->n0q; i++) dz += P->thick[i]; dz /= (double)nk*P->n0q;
        const int Nm = (nk - 1)*n2e;
        auto field = [&](int slots, double lo, double hi) { std::vector<double> f((size_t)nEl*slots*n2e); for (auto& v : f) v = (lo + (hi - lo)*(U(rng) - 0.5))*area*dz; return f; };
        std::vector<double> th = field(nk, 280, 320), rh = field(nk, 0.5, 1.2), et = field(nk, 5, 6), pi = field(nk, 700, 1000);
        std::vector<double> Fu((size_t)nEl*Nm), Fr((size_t)nEl*N), Fe((size_t)nEl*N), Fp((size_t)nEl*N);
        for (auto* F : {&Fu, &Fr, &Fe, &Fp}) for (auto& v : *F) v = S(rng)*1e8;
        std::vector<double*> dv;
        for (auto* f : {&th, &rh, &et, &pi, &Fu, &Fr, &Fe, &Fp}) dv.push_back(mesh.to_device(f->data(), f->size()));
        double *d_du = mesh.device_alloc((size_t)nEl*Nm), *d_dr = mesh.device_alloc((size_t)nEl*N), *d_de = mesh.device_alloc((size_t)nEl*N), *d_dp = mesh.device_alloc((size_t)nEl*N);
        VertSolve vert(&topo, &geom, 75.0);
        vert.solve_schur_column_eta(dv[0], nullptr, dv[1], dv[2], dv[3], dv[4], dv[5], dv[6], dv[7], d_du, d_dr, d_de, d_dp);
        std::vector<double> gp((size_t)nEl*N), gu((size_t)nEl*Nm);
        mesh.to_host(gp.data(), d_dp, gp.size()); mesh.to_host(gu.data(), d_du, gu.size());
        worst = 0.0;
        for (int e : {0, nEl - 1}) {
            std::vector<double> velz(Nm, 0.0), wu(Nm), wr(N), we(N), wp(N);
            std::vector<double> fu(Fu.begin() + (size_t)e*Nm, Fu.begin() + (size_t)(e + 1)*Nm), fr(Fr.begin() + (size_t)e*N, Fr.begin() + (size_t)(e + 1)*N),
                                fe(Fe.begin() + (size_t)e*N, Fe.begin() + (size_t)(e + 1)*N), fp(Fp.begin() + (size_t)e*N, Fp.begin() + (size_t)(e + 1)*N);
            orc_solve_schur_column_eta(P, e%nels, e/nels, 75.0, &th[(size_t)e*N], velz.data(), &rh[(size_t)e*N], &et[(size_t)e*N], &pi[(size_t)e*N],
                                       fu.data(), fr.data(), fe.data(), fp.data(), wu.data(), wr.data(), we.data(), wp.data(), nullptr);
            double num = 0, den = 0;
            for (int i = 0; i < N; i++) { num += (gp[(size_t)e*N + i] - wp[i])*(gp[(size_t)e*N + i] - wp[i]); den += wp[i]*wp[i]; }
            for (int i = 0; i < Nm; i++) { num += (gu[(size_t)e*Nm + i] - wu[i])*(gu[(size_t)e*Nm + i] - wu[i]); den += wu[i]*wu[i]; }
            worst = std::max(worst, std::sqrt(num/den));
        }
        std::printf("%-8s rel L2 = %.3e\n", "Schur", worst); if (!(worst < 1e-10)) fails++;
        {   // the status a caller reads instead of trusting a band-wide pivoted LU: every column of this solve converged
            std::vector<int> st(nEl, -7); std::vector<double> ratio(nEl, -1.0);
            const int nbad = vert.solve_status(st.data(), ratio.data());
            int ok = (nbad == 0 || nbad == -1);                       // -1: a path without status (orders >= 4)
            if (nbad == 0) for (int e = 0; e < nEl; e++) ok = ok && st[e] == 0 && ratio[e] >= 0.0 && ratio[e] <= 1e-10;
            std::printf("%-8s unconverged columns = %d\n", "Status", nbad); if (!ok) fails++;
        }
        {   // PCLU's pivoting itself (round 4): every column through the band LU with partial pivoting -- the same solutions, status 3
            std::vector<double*> dF;
            for (auto* f : {&Fu, &Fr, &Fe, &Fp}) dF.push_back(mesh.to_device(f->data(), f->size()));
            vert.set_pivot_fallback(2);
            vert.solve_schur_column_eta(dv[0], nullptr, dv[1], dv[2], dv[3], dF[0], dF[1], dF[2], dF[3], d_du, d_dr, d_de, d_dp);
            std::vector<int> st(nEl, -7);
            const int nbad = vert.solve_status(st.data(), nullptr);
            vert.set_pivot_fallback(0);
            std::vector<double> gp2((size_t)nEl*N);
            mesh.to_host(gp2.data(), d_dp, gp2.size());
            double num = 0, den = 0;
            for (size_t i = 0; i < gp2.size(); i++) { num += (gp2[i] - gp[i])*(gp2[i] - gp[i]); den += gp[i]*gp[i]; }
            int ok = nbad == 0 || nbad == -1;
            if (nbad == 0) for (int e = 0; e < nEl; e++) ok = ok && st[e] == 3;
            std::printf("%-8s rel L2 = %.3e (pivoted band LU for every column vs the block sweep)\n", "PivotLU", std::sqrt(num/den));
            if (!ok || !(std::sqrt(num/den) < 1e-10)) fails++;
            for (double* p : dF) mimsem_free(p);
        }
        // vert->diagTheta2(rho, rt, theta) / diagTheta_L2  (eul/VertSolve.cpp:289-352), every column
        {
            std::vector<double> rt = field(nk, 250, 400), t2((size_t)nEl*(nk + 1)*n2e), tl((size_t)nEl*N), w2((nk + 1)*n2e), wl(N);
            double *d_rt = mesh.to_device(rt.data(), rt.size()), *d_t2 = mesh.device_alloc(t2.size()), *d_tl = mesh.device_alloc(tl.size());
            vert.diagTheta2(dv[1], d_rt, d_t2); vert.diagTheta_L2(dv[1], d_rt, d_tl);
            mesh.to_host(t2.data(), d_t2, t2.size()); mesh.to_host(tl.data(), d_tl, tl.size());
            worst = 0.0;
            for (int e : {0, nEl - 1}) {
                orc_diag_theta2(P, e%nels, e/nels, &rh[(size_t)e*N], &rt[(size_t)e*N], w2.data());
                orc_diag_theta_L2(P, e%nels, e/nels, &rh[(size_t)e*N], &rt[(size_t)e*N], wl.data());
                double num = 0, den = 0;
                for (int i = 0; i < (nk + 1)*n2e; i++) { const double g = t2[(size_t)e*(nk + 1)*n2e + i]; num += (g - w2[i])*(g - w2[i]); den += w2[i]*w2[i]; }
                for (int i = 0; i < N; i++) { const double g = tl[(size_t)e*N + i]; num += (g - wl[i])*(g - wl[i]); den += wl[i]*wl[i]; }
                worst = std::max(worst, std::sqrt(num/den));
            }
            std::printf("%-8s rel L2 = %.3e\n", "diagTheta", worst); if (!(worst < 1e-10)) fails++;
            mimsem_free(d_rt); mimsem_free(d_t2); mimsem_free(d_tl);
        }
        for (double* q : dv) mimsem_free(q);
        mimsem_free(d_du); mimsem_free(d_dr); mimsem_free(d_de); mimsem_free(d_dp);
    }
    // VecScatterBegin/End behind the ABI (eul/Assembly.cpp:2194-2195), loop-back transport: the "ghost" edges of this single patch
    // are sent to "mirror" edges of the same patch and added there, with the operator split around the exchange
    {
        M1.assemble(0, SCALE, true);
        std::vector<int> ranks{0}, ghost, mirror, off;
        for (int i = 0; i < 12; i++) { ghost.push_back(3*i + 1); mirror.push_back(P->n1 - 2 - 5*i); }
        off = {0, 12};
        VecScatterHalo halo(&mesh, 1, ranks, ghost, off, mirror, off);
        halo.use_loopback();
        std::vector<double> whole(P->n1), split(P->n1);
        M1.mult(d_u, d_y); mesh.to_host(whole.data(), d_y, whole.size());
        for (int i = 0; i < 12; i++) whole[mirror[i]] += whole[ghost[i]];          // what REVERSE / ADD does
        for (int i = 0; i < 12; i++) whole[ghost[i]] = whole[mirror[i]];           // then FORWARD / INSERT
        M1.mult_part(d_u, d_y, MIMSEM_PART_BOUNDARY);
        halo.begin_reverse_add(d_y, 1, P->n1);
        M1.mult_part(d_u, d_y, MIMSEM_PART_INTERIOR);
        halo.end_reverse_add();
        halo.forward_insert(d_y, 1, P->n1);
        mesh.to_host(split.data(), d_y, split.size());
        double worst = 0.0;
        for (int i = 0; i < P->n1; i++) worst = std::max(worst, std::fabs(split[i] - whole[i]));
        std::printf("%-8s max abs diff = %.3e\n", "Halo", worst); if (!(worst == 0.0)) fails++;
        // the contract of the two parts: a second BOUNDARY part while one is pending is refused, reset_parts() forgets it
        M1.mult_part(d_u, d_y, MIMSEM_PART_BOUNDARY);
        const int rc2 = mimsem_op_apply_part(mesh.ctx, MIMSEM_OP_UMAT, 0, 1, 1.0e8, MIMSEM_FLAG_VERT, nullptr, 0, d_u, P->n1, d_y, P->n1, 1.0, MIMSEM_PART_BOUNDARY);
        M1.reset_parts();
        M1.mult_part(d_u, d_y, MIMSEM_PART_BOUNDARY); M1.mult_part(d_u, d_y, MIMSEM_PART_INTERIOR);
        int wst[5]; const int really_split = mimsem_op_wave_stats(mesh.ctx, 1, wst) == 1;       // (the two-pass form runs whole in the BOUNDARY part: nothing pending)
        std::printf("%-8s second BOUNDARY while pending -> %d (split in force: %d, MIMSEM_ERR_STATE = %d)\n", "Parts", rc2, really_split, MIMSEM_ERR_STATE);
        if (rc2 != (really_split ? MIMSEM_ERR_STATE : MIMSEM_OK)) fails++;
        if (VecScatterHalo::use_rccl_library(nullptr) != MIMSEM_ERR_ARG) fails++;
    }
    // the co-located local layout (Topo(..., paired = true)): the same operator on the permuted vector, now on the
    // wave-level kernels (mimsem_op_wave_stats reports a plan), against the oracle's result in the reference's layout
    {
        Topo topo_p(n, nels, nk, true);
        Mesh& mesh_p = *Mesh::of(&topo_p, &geom);
        int st[5] = {0, 0, 0, 0, 0}, st0[5] = {0, 0, 0, 0, 0};
        const int has_p = mimsem_op_wave_stats(mesh_p.ctx, nk, st), has_r = mimsem_op_wave_stats(mesh.ctx, nk, st0);
        std::printf("wave plan: paired layout %d (%d groups), reference layout %d\n", has_p, st[0], has_r);
        if (has_p != 1) fails++;            // (on a patch this small the reference layout can still have one; from 8 x 8 elements at p = 3 it has not)
        std::vector<int> slot(P->n1);                                  // reference slot -> paired slot
        const int D = n*nels;
        for (int ii = 0; ii < D*(D + 1); ii++) { slot[2*ii] = 2*ii; slot[2*ii + 1] = topo_p.slot_of_edge_y(ii); }
        std::vector<double> up(P->n1), yp(P->n1);
        for (int i = 0; i < P->n1; i++) up[slot[i]] = u[i];
        double* d_up = mesh_p.to_device(up.data(), up.size());
        double* d_yp = mesh_p.to_device(yp.data(), yp.size());
        Umat M1p(&topo_p, &geom, &node, &edge);
        M1p.assemble(1, SCALE, true);
        M1p.mult(d_up, d_yp);
        mesh_p.to_host(yp.data(), d_yp, yp.size());
        orc_op_elmats(P, ORC_UMAT, 1, SCALE, 1, nullptr, em.data());
        std::fill(want.begin(), want.end(), 0.0);
        orc_op_apply(P, ORC_UMAT, em.data(), u.data(), want.data());
        double num = 0, den = 0;
        for (int i = 0; i < P->n1; i++) { num += (yp[slot[i]] - want[i])*(yp[slot[i]] - want[i]); den += want[i]*want[i]; }
        std::printf("%-8s rel L2 = %.3e\n", "paired", std::sqrt(num/den)); if (!(std::sqrt(num/den) < 1e-10)) fails++;
        mimsem_free(d_up); mimsem_free(d_yp);
    }
    mimsem_free(d_u); mimsem_free(d_h); mimsem_free(d_y);
    orc_patch_destroy(P);
    Mesh::release_all();
    std::printf(fails ? "FAILED\n" : "OK\n");
    return fails;
}
