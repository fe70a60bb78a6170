// mimsem_amd/host/horiz_call.cpp -- the right-hand sides of the horizontal dynamics with the HOST in C++: mimsem_host::HorizSolve
// (mimsem_horizsolve.hpp: advection_rhs_ec + momentum_rhs_ec of eul/HorizSolve.cpp:380-786, every level per call) on the mesh and fields
// bench.py wrote (sw_io.hpp::read_arrays).  Built by __graft_entry__.build(), run as a child of bench.py.
//   usage: horiz_call <case.arr> [evaluations]      prints one JSON object
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include "mimsem_horizsolve.hpp"
#include "sw_io.hpp"

using namespace mimsem_host;
using clk = std::chrono::steady_clock;

int main(int argc, char** argv) {
    if (argc < 2) { std::fprintf(stderr, "usage: horiz_call case.arr [evaluations]\n"); return 2; }
    const int reps = argc > 2 ? std::atoi(argv[2]) : 10;
    try {
        const ArrayFile a = read_arrays(argv[1]);
        const mimsem_mesh_desc d = desc_of(a);
        Mesh mesh(d);
        const size_t s1 = (size_t)d.nk*d.n1, s2 = (size_t)d.nk*d.n2;
        auto dev = [&](const char* k) { const auto& v = a.reals(k); return mesh.to_device(v.data(), v.size()); };
        double *fg = dev("fg"), *u1 = dev("u1"), *u2 = dev("u2"), *h1 = dev("h1"), *h2 = dev("h2"), *th = dev("theta"), *Pi = dev("Pi");
        double *vz = dev("velz"), *dudz = dev("dudz");
        double *dF = mesh.device_alloc(s2), *dG = mesh.device_alloc(s2), *Fk = mesh.device_alloc(s1), *Gk = mesh.device_alloc(s1), *fu = mesh.device_alloc(s1);
        HorizSolve hs(&mesh, fg);
        auto rhs = [&]() {
            hs.advection_rhs_ec(u1, u2, h1, h2, th, dF, dG, Fk, Gk);
            hs.momentum_rhs_ec(th, dudz, dudz, vz, vz, Pi, u1, u2, h1, h2, fu, Fk, nullptr, nullptr, nullptr, Fk);
        };
        // ... and with grad(theta) of advection_rhs_ec handed to momentum_rhs_ec (six mass solves instead of seven)
        auto rhs6 = [&]() {
            hs.advection_rhs_ec(u1, u2, h1, h2, th, dF, dG, Fk, Gk);
            hs.momentum_rhs_ec(th, dudz, dudz, vz, vz, Pi, u1, u2, h1, h2, fu, Fk, nullptr, nullptr, nullptr, Fk, hs.last_grad_theta());
        };
        rhs(); rhs();
        if (!hs.verify()) rhs();                       // (a missed check: the same evaluation again, on the CG)
        check(mimsem_ctx_sync(mesh.ctx), "sync");
        const auto t0 = clk::now();
        for (int i = 0; i < reps; i++) { rhs(); if (!hs.verify()) rhs(); }          // every solve checked: one read of the log per evaluation
        check(mimsem_ctx_sync(mesh.ctx), "sync");
        const double ms = std::chrono::duration<double, std::milli>(clk::now() - t0).count()/reps;
        // the whole evaluation recorded once (no solve needs the host after the first three have been verified)
        rhs6();
        check(mimsem_ctx_sync(mesh.ctx), "sync");
        const auto t6 = clk::now();
        for (int i = 0; i < reps; i++) { rhs6(); if (!hs.verify()) rhs6(); }
        check(mimsem_ctx_sync(mesh.ctx), "sync");
        const double ms6 = std::chrono::duration<double, std::milli>(clk::now() - t6).count()/reps;
        double ms_graph = -1.0, ms_graph6 = -1.0; int nodes = 0;
        if (hs.fixed_length) {
            Graph g(&mesh);
            g.record(rhs);
            nodes = g.nodes();
            g.launch();
            check(mimsem_ctx_sync(mesh.ctx), "sync");
            const auto t1 = clk::now();
            bool ok = true;
            for (int i = 0; i < reps; i++) { g.launch(); ok = hs.verify() && ok; }     // (the replay logs into the slots of the recording)
            check(mimsem_ctx_sync(mesh.ctx), "sync");
            if (!ok) ms_graph = -2.0;
            if (ms_graph != -2.0) ms_graph = std::chrono::duration<double, std::milli>(clk::now() - t1).count()/reps;
            Graph g6(&mesh);
            g6.record(rhs6);
            g6.launch();
            check(mimsem_ctx_sync(mesh.ctx), "sync");
            const auto t2 = clk::now();
            for (int i = 0; i < reps; i++) { g6.launch(); ok = hs.verify() && ok; }
            check(mimsem_ctx_sync(mesh.ctx), "sync");
            ms_graph6 = std::chrono::duration<double, std::milli>(clk::now() - t2).count()/reps;
        }
        std::vector<double> h(s1);
        mesh.to_host(h.data(), fu, s1);
        double n2 = 0.0;
        for (double v : h) n2 += v*v;
        std::printf("{\"ms_per_evaluation\": %.4f, \"ms_per_evaluation_recorded\": %.4f, \"ms_per_evaluation_reusing_grad_theta\": %.4f, \"ms_per_evaluation_recorded_reusing_grad_theta\": %.4f, \"graph_nodes\": %d, \"m1_steps\": %d, \"m1_fixed_length\": %s, \"m1_solves_checked\": %d, \"m1_solves_missed\": %d, \"m1_worst_check\": %.2e, \"fu_l2\": %.15e}\n",
                    ms, ms_graph, ms6, ms_graph6, nodes, hs.last_its, hs.fixed_length ? "true" : "false", hs.solves_checked, hs.solves_missed, hs.worst_rel, std::sqrt(n2));
        for (double* p : {fg, u1, u2, h1, h2, th, Pi, vz, dudz, dF, dG, Fk, Gk, fu}) mimsem_free(p);
    } catch (const std::exception& e) { std::fprintf(stderr, "horiz_call: %s\n", e.what()); return 1; }
    return 0;
}
