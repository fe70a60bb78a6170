// mimsem_amd/host/bench_call.cpp -- what an UNCHANGED reference rank pays per operator call, measured from a C++ host through the shim
// (bench.py's `reference_local_layout.cpp_host`; built by __graft_entry__.build(), run as a child process).  One 12 x 12-element p = 3
// patch x 30 levels in the reference's rank-local numbering (eul/Topo.cpp:200-251), synthetic metric; the reference's own call pattern
//     for (kk = 0; kk < nk; kk++) { M1->assemble(kk, SCALE, true); MatMult(M1->M, x[kk], y[kk]); }        (eul/Euler_2.cpp:1427-1457)
// (a) as written: 2 launches per level;  (b) the same loop recorded once in a hipGraph (Graph::record) and replayed;
// (c) one 30-level call of the engine (what a host that adopts the batched entry point gets).  Prints one JSON object.
#include <chrono>
#include <cstdio>
#include <random>
#include <vector>
#include "mimsem_shim.hpp"

using namespace mimsem_host;
using clk = std::chrono::steady_clock;

int main(int argc, char** argv) {
    const int n = 3, nels = argc > 1 ? atoi(argv[1]) : 12, nk = 30, reps = argc > 2 ? atoi(argv[2]) : 300;
    const double SCALE = 1.0e8;
    std::mt19937_64 rng(7);
    std::uniform_real_distribution<double> U(0.5, 1.5), S(-1.0, 1.0);
    const int nEl = nels*nels, mp12 = (n + 1)*(n + 1);
    Topo topo(n, nels, nk);
    Geom geom; geom.nk = nk; geom.quad_n = n; geom.nDofsX = n*nels;
    geom.det.resize((size_t)nEl*mp12); geom.J.resize((size_t)nEl*mp12*4);
    for (auto& v : geom.det) v = U(rng)*1e10;
    for (size_t i = 0; i < geom.det.size(); i++) { geom.J[4*i] = 1e5*U(rng); geom.J[4*i + 1] = 1e4*S(rng); geom.J[4*i + 2] = 1e4*S(rng); geom.J[4*i + 3] = 1e5*U(rng); }
    const size_t n0q = (size_t)(geom.nDofsX + 1)*(geom.nDofsX + 1);
    geom.thick.resize((size_t)nk*n0q); geom.thickInv.resize((size_t)nk*n0q);
    for (size_t i = 0; i < geom.thick.size(); i++) { geom.thick[i] = 1000.0*U(rng); geom.thickInv[i] = 1.0/geom.thick[i]; }
    GaussLobatto quad{n}; LagrangeNode node{n, &quad}; LagrangeEdge edge{n, &node};
    try {
        Mesh& mesh = *Mesh::of(&topo, &geom);
        Umat M1(&topo, &geom, &node, &edge);
        const int n1 = topo.n1;
        std::vector<double> x((size_t)nk*n1);
        for (auto& v : x) v = S(rng);
        double *d_x = mesh.to_device(x.data(), x.size()), *d_y = mesh.device_alloc((size_t)nk*n1);
        auto loop = [&]() {
            for (int kk = 0; kk < nk; kk++) { M1.assemble(kk, SCALE, true); M1.mult(d_x + (size_t)kk*n1, d_y + (size_t)kk*n1); }
        };
        auto timeit = [&](auto&& f, int r) {
            f(); check(mimsem_ctx_sync(mesh.ctx), "sync");
            const auto t0 = clk::now();
            for (int i = 0; i < r; i++) f();
            check(mimsem_ctx_sync(mesh.ctx), "sync");
            return std::chrono::duration<double, std::micro>(clk::now() - t0).count()/r;
        };
        const double us_loop = timeit(loop, reps);
        Graph g(&mesh);
        g.record(loop);
        const double us_graph = timeit([&]() { g.launch(); }, reps);
        const double us_batched = timeit([&]() { check(mimsem_op_apply(mesh.ctx, MIMSEM_OP_UMAT, 0, nk, SCALE, MIMSEM_FLAG_VERT, nullptr, 0, d_x, n1, d_y, n1, 1.0), "op_apply"); }, reps);
        // one single-level call, synchronised each time: the latency a caller that needs the result at once sees
        const double us_single_sync = timeit([&]() { M1.assemble(3, SCALE, true); M1.mult(d_x, d_y); check(mimsem_ctx_sync(mesh.ctx), "sync"); }, reps);
        std::printf("{\"elements\": %d, \"levels\": %d, \"per_level_calls_us_per_call\": %.3f, \"per_level_calls_in_a_graph_us_per_call\": %.3f, "
                    "\"graph_nodes\": %d, \"one_30_level_call_us_per_level\": %.3f, \"single_level_call_synchronised_us\": %.3f, "
                    "\"applies_per_s_per_level_calls\": %.4g, \"applies_per_s_graph\": %.4g, \"applies_per_s_batched\": %.4g}\n",
                    nEl, nk, us_loop/nk, us_graph/nk, g.nodes(), us_batched/nk, us_single_sync,
                    nEl*nk/(us_loop*1e-6), nEl*nk/(us_graph*1e-6), nEl*nk/(us_batched*1e-6));
        mimsem_free(d_x); mimsem_free(d_y);
        Mesh::release_all();
    } catch (const std::exception& ex) {
        std::printf("{\"error\": \"%s\"}\n", ex.what());
        return 1;
    }
    return 0;
}
