// tests/cpp/test_sw_sharded.cpp -- the shallow-water Picard step (row N3) on SEVERAL RANKS driven from C++: src::SWEqn of
// mimsem_amd/host/mimsem_sweqn.hpp over a Shard (the rank's halo plans + ownership weights + the host's all-reduce), the counterpart of the
// reference's distributed SWEqn::solve (src/SWEqn_Picard.cpp:751-765 with gtol_x :131-153, :341-400 and the ghost updates :422-425).
// The ranks of this test are THREADS of one process on the one GPU, each with its own Mesh (context), Shard and SWEqn -- what every MPI rank of
// a real host holds; the transport and the all-reduce are host callbacks that meet at a barrier and hand the messages over through host
// memory (a plain-MPI host would do the same with MPI_Sendrecv / MPI_Allreduce; RCCL refuses two ranks on one device).  The pytest wrapper
// (tests/test_gpu_cpp_shim.py) writes one array file per rank (mesh tables of the rank's patches, start state, slot lists) and compares the
// gathered result with the one-context run.
//   usage: test_sw_sharded <world> <case prefix> <out prefix> <nsteps>
#include <cstdio>
#include "../../mimsem_amd/host/mimsem_sweqn.hpp"
#include "../../mimsem_amd/host/sw_io.hpp"
#include "thread_ranks.hpp"

using namespace mimsem_host;

using namespace thread_ranks;

int main(int argc, char** argv) {
    if (argc < 5) { std::fprintf(stderr, "usage: test_sw_sharded world case_prefix out_prefix nsteps\n"); return 2; }
    const int world = std::atoi(argv[1]), nsteps = std::atoi(argv[4]);
    if (world < 2 || world > 6) return 2;
    World W(world);
    std::vector<int> status(world, 0);
    std::vector<std::string> report(world);
    std::mutex create;                               // (contexts are created one after the other: nothing in the test depends on concurrent creation)
    auto rank_main = [&](int rank) {
        try {
            const std::string in = std::string(argv[2]) + std::to_string(rank) + ".arr";
            const ArrayFile a = read_arrays(in.c_str());
            const mimsem_mesh_desc d = desc_of(a);
            std::unique_lock<std::mutex> lk(create);
            Mesh mesh(d);
            lk.unlock();
            RankCtx rc{&W, rank, mesh.ctx};
            Shard sh(&mesh, a.ints("ranks"), a.ints("ghost1"), a.ints("ghost1_off"), a.ints("mirror1"), a.ints("mirror1_off"),
                     a.ints("ghost0"), a.ints("ghost0_off"), a.ints("mirror0"), a.ints("mirror0_off"), a.reals("own0"), a.reals("own1"), &allreduce, &rc);
            sh.use_transport(&transport, &rc);
            if ((int)a.ints("ranks").size() != world - 1) throw std::runtime_error("this harness needs every rank to neighbour every other");
            double* fg = mesh.to_device(a.reals("fg").data(), a.reals("fg").size());
            double *un = mesh.to_device(a.reals("u").data(), a.reals("u").size()), *hn = mesh.to_device(a.reals("h").data(), a.reals("h").size());
            const auto& par = a.reals("params");              // dt, nits, q_exact
            src::SWEqn sw(&mesh, fg, &sh);
            long red_setup = 0, red_steps = 0, iters = 0;
            for (int s = 0; s < nsteps; s++) {
                const long r0 = rc.reductions;
                sw.solve(un, hn, par[0], false, (int)par[1], par[2] != 0.0);
                if (s == 0) red_setup = rc.reductions - r0;       // (the first step estimates the spectral regions: inner products, all-reduced)
                else { red_steps += rc.reductions - r0; iters += (long)sw.history.size(); }
            }
            std::vector<double> u(mesh.n1), h(mesh.n2);
            mesh.to_host(u.data(), un, u.size()); mesh.to_host(h.data(), hn, h.size());
            const std::string out = std::string(argv[3]) + std::to_string(rank) + ".bin";
            FILE* g = std::fopen(out.c_str(), "wb");
            if (!g) throw std::runtime_error("cannot write " + out);
            std::fwrite(u.data(), 8, u.size(), g); std::fwrite(h.data(), 8, h.size(), g);
            std::fclose(g);
            char buf[512];
            std::snprintf(buf, sizeof buf, "rank %d: chebyshev steps [%d, %d, %d], fallbacks %d, recalibrations %d, all-reduces in the set-up step %ld, "
                          "in the %ld Picard iterations after it %ld, exchanges %ld, |dx|/|x| last %.3e", rank, sw.steps_A, sw.steps_M1, sw.steps_q, sw.fallbacks,
                          sw.recalibrations, red_setup, iters, red_steps, sh.exchanges, sw.history.empty() ? 0.0 : sw.history.back());
            report[rank] = buf;
            // the contract of the sharded fixed-length mode: ONE all-reduce per Picard iteration once the regions are known, none inside a solve
            if (nsteps > 1 && red_steps != iters) { report[rank] += "  -- FAIL: all-reduces != Picard iterations"; status[rank] = 1; }
            if (sw.fallbacks != 0 || sw.recalibrations != 0) { report[rank] += "  -- FAIL: a check missed"; status[rank] = 1; }
            mimsem_free(un); mimsem_free(hn); mimsem_free(fg);
        } catch (const std::exception& e) { report[rank] = std::string("rank ") + std::to_string(rank) + " FAIL: " + e.what(); status[rank] = 1; std::fprintf(stderr, "%s\n", report[rank].c_str()); std::_Exit(1); }
    };
    std::vector<std::thread> th;
    for (int r = 0; r < world; r++) th.emplace_back(rank_main, r);
    for (auto& t : th) t.join();
    int bad = 0;
    for (int r = 0; r < world; r++) { std::printf("%s\n", report[r].c_str()); bad += status[r]; }
    std::printf(bad ? "FAIL\n" : "DONE\n");
    return bad ? 1 : 0;
}
