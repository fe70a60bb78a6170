// tests/cpp/test_vert.cpp -- the vertical implicit solve driven from C++: mimsem_host::VertSolveEta (mimsem_amd/host/mimsem_vertsolve.hpp, the
// Newton loop of VertSolve::solve_schur_eta, eul/VertSolve.cpp:1721-1973, on the library's fused entry points) on the patch, geopotential and
// state the pytest wrapper wrote: (1) three iterations without forcing, (2) two iterations with the Held-Suarez temperature forcing and the
// u dw/dx term.  States and max-norm histories go back to the wrapper, which compares them with oracle/vert_oracle.py.
//   usage: test_vert <in.arr> <out.bin>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <vector>
#include "../../mimsem_amd/host/mimsem_vertsolve.hpp"
#include "../../mimsem_amd/host/sw_io.hpp"

using namespace mimsem_host;

int main(int argc, char** argv) {
    if (argc < 3) { std::fprintf(stderr, "usage: test_vert in.arr out.bin\n"); return 2; }
    try {
        const ArrayFile a = read_arrays(argv[1]);
        const mimsem_mesh_desc d = desc_of(a);
        Mesh mesh(d);
        const double dt = a.reals("dt").at(0);
        auto dev = [&](const char* k) { const auto& v = a.reals(k); return mesh.to_device(v.data(), v.size()); };
        double *zv = dev("zv"), *lat = dev("lat"), *udwdx = dev("udwdx");
        FILE* g = std::fopen(argv[2], "wb");
        if (!g) { std::perror(argv[2]); return 2; }
        auto put = [&](const double* p, size_t n) { std::vector<double> h(n); mesh.to_host(h.data(), p, n); std::fwrite(h.data(), 8, n, g); };
        VertSolveEta vs(&mesh, dt);
        for (int run = 0; run < 2; run++) {
            double *velz = dev("velz"), *rho = dev("rho"), *rt = dev("rt"), *exner = dev("exner");
            const int its = run == 0 ? vs.solve_schur_eta(velz, rho, rt, exner, zv, 3, 0.0)
                                     : vs.solve_schur_eta(velz, rho, rt, exner, zv, 2, 0.0, udwdx, lat);
            std::printf("run %d: %d iterations, last norms exner %.3e w %.3e rho %.3e eta %.3e, k2i_z %.6e\n", run, its, vs.history.back().exner,
                        vs.history.back().w, vs.history.back().rho, vs.history.back().eta, vs.k2i_z);
            put(velz, a.reals("velz").size()); put(rho, a.reals("rho").size()); put(rt, a.reals("rt").size()); put(exner, a.reals("exner").size());
            for (const auto& h : vs.history) { const double v[4] = {h.exner, h.w, h.rho, h.eta}; std::fwrite(v, 8, 4, g); }
            for (double* p : {velz, rho, rt, exner}) mimsem_free(p);
        }
        std::fclose(g);
        // the reductions of a multi-rank host (round 6: MPI_Allreduce(MAX) of the four norms per iteration, eul/VertSolve.cpp:1915-1918, and the sum of
        // k2i_z): the callbacks are called once per iteration / once per solve, and what they return decides the stopping test -- a second "rank"
        // whose norms are still large keeps this one iterating
        {
            int nmax = 0, nsum = 0;
            vs.allreduce_max = [&](double* v, int n) { nmax++; for (int i = 0; i < n; i++) v[i] = std::max(v[i], 1.0e-3); };       // (the other rank has not converged)
            vs.allreduce_sum = [&](double* v, int n) { nsum++; for (int i = 0; i < n; i++) v[i] *= 2.0; };                          // (... and holds as much k2i_z)
            double *velz = dev("velz"), *rho = dev("rho"), *rt = dev("rt"), *exner = dev("exner");
            VertSolveEta one(&mesh, dt);
            double *v1 = dev("velz"), *r1 = dev("rho"), *t1 = dev("rt"), *e1 = dev("exner");
            const int its1 = one.solve_schur_eta(v1, r1, t1, e1, zv, 4, 1.0e-6);            // alone: stops as soon as its own norms are below 1e-6
            const int its = vs.solve_schur_eta(velz, rho, rt, exner, zv, 4, 1.0e-6);        // with the unconverged neighbour: all 4 iterations
            if (its1 > 4 || its != 4 || nmax != 4 || nsum != 1 || !(vs.history.back().exner >= 1.0e-3) || std::fabs(vs.k2i_z - 2.0*[&] { VertSolveEta w(&mesh, dt);
                    double *a2 = dev("velz"), *b2 = dev("rho"), *c2 = dev("rt"), *d2 = dev("exner"); w.solve_schur_eta(a2, b2, c2, d2, zv, 4, 0.0); const double k = w.k2i_z;
                    for (double* p : {a2, b2, c2, d2}) mimsem_free(p); return k; }()) > 1.0e-9*std::fabs(vs.k2i_z)) {
                std::printf("FAIL: multi-rank reductions: its alone %d, with neighbour %d, max calls %d, sum calls %d\n", its1, its, nmax, nsum); return 1;
            }
            std::printf("multi-rank reductions: alone %d iterations, with an unconverged neighbour %d; %d MAX reductions, %d SUM\n", its1, its, nmax, nsum);
            for (double* p : {velz, rho, rt, exner, v1, r1, t1, e1}) mimsem_free(p);
        }
        for (double* p : {zv, lat, udwdx}) mimsem_free(p);
    } catch (const std::exception& e) { std::printf("FAIL: %s\n", e.what()); return 1; }
    std::printf("DONE\n");
    return 0;
}
