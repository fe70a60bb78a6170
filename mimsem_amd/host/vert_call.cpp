// mimsem_amd/host/vert_call.cpp -- one Newton iteration of the vertical implicit solve with the HOST in C++: mimsem_host::VertSolveEta
// (mimsem_vertsolve.hpp: VertSolve::solve_schur_eta, eul/VertSolve.cpp:1721-1973, for every column at once) on the mesh, geopotential and
// state bench.py wrote (sw_io.hpp::read_arrays).  Built by __graft_entry__.build(), run as a child of bench.py.
//   usage: vert_call <case.arr> [iterations]      prints one JSON object
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include "mimsem_vertsolve.hpp"
#include "sw_io.hpp"

using namespace mimsem_host;
using clk = std::chrono::steady_clock;

int main(int argc, char** argv) {
    if (argc < 2) { std::fprintf(stderr, "usage: vert_call case.arr [iterations]\n"); return 2; }
    const int its = argc > 2 ? std::atoi(argv[2]) : 4;
    try {
        const ArrayFile a = read_arrays(argv[1]);
        const mimsem_mesh_desc d = desc_of(a);
        Mesh mesh(d);
        auto dev = [&](const char* k) { const auto& v = a.reals(k); return mesh.to_device(v.data(), v.size()); };
        double *zv = dev("zv"), *velz = dev("velz"), *rho = dev("rho"), *rt = dev("rt"), *exner = dev("exner");
        VertSolveEta vs(&mesh, a.reals("dt").at(0));
        vs.solve_schur_eta(velz, rho, rt, exner, zv, 2, 0.0);                 // warm-up: workspaces reach their size
        check(mimsem_ctx_sync(mesh.ctx), "sync");
        const auto t0 = clk::now();
        vs.solve_schur_eta(velz, rho, rt, exner, zv, its, 0.0);
        check(mimsem_ctx_sync(mesh.ctx), "sync");
        const double ms = std::chrono::duration<double, std::milli>(clk::now() - t0).count()/its;
        const auto& h = vs.history.back();
        std::printf("{\"ms_per_newton_iteration\": %.4f, \"iterations\": %d, \"norms_last\": {\"exner\": %.6e, \"w\": %.6e, \"rho\": %.6e, \"eta\": %.6e}}\n",
                    ms, its, h.exner, h.w, h.rho, h.eta);
        for (double* p : {zv, velz, rho, rt, exner}) mimsem_free(p);
    } catch (const std::exception& e) { std::fprintf(stderr, "vert_call: %s\n", e.what()); return 1; }
    return 0;
}
