// tests/cpp/test_vert.cpp -- the vertical implicit solve driven from C++: mimsem_host::VertSolveEta (mimsem_amd/host/mimsem_vertsolve.hpp, the
// Newton loop of VertSolve::solve_schur_eta, eul/VertSolve.cpp:1721-1973, on the library's fused entry points) on the patch, geopotential and
// state the pytest wrapper wrote: (1) three iterations without forcing, (2) two iterations with the Held-Suarez temperature forcing and the
// u dw/dx term.  States and max-norm histories go back to the wrapper, which compares them with oracle/vert_oracle.py.
//   usage: test_vert <in.arr> <out.bin>
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
        for (double* p : {zv, lat, udwdx}) mimsem_free(p);
    } catch (const std::exception& e) { std::printf("FAIL: %s\n", e.what()); return 1; }
    std::printf("DONE\n");
    return 0;
}
