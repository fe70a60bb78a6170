// tests/cpp/test_horiz_sharded.cpp -- the right-hand sides of the horizontal dynamics (row N2) on SEVERAL RANKS driven from C++:
// mimsem_host::HorizSolve over a Shard (mimsem_horizsolve.hpp + mimsem_shard.hpp; the reference's distributed HorizSolve: every MatMult on the
// MPIAIJ matrices of eul/Assembly.cpp followed by the gtol scatters, the ksp1 solves on MPI_COMM_WORLD, eul/HorizSolve.cpp:208-786), the ranks as
// threads of one process (thread_ranks.hpp).  advection_rhs_ec + momentum_rhs_ec (viscosity on, grad(theta) handed over) on every rank's
// patches; the pytest wrapper gathers fu / dG / k2i and compares them with the one-context evaluation.
//   usage: test_horiz_sharded <world> <case prefix> <out prefix>
#include <cstdio>
#include "../../mimsem_amd/host/mimsem_horizsolve.hpp"
#include "../../mimsem_amd/host/sw_io.hpp"
#include "thread_ranks.hpp"

using namespace mimsem_host;
using namespace thread_ranks;

int main(int argc, char** argv) {
    if (argc < 4) { std::fprintf(stderr, "usage: test_horiz_sharded world case_prefix out_prefix\n"); return 2; }
    const int world = std::atoi(argv[1]);
    if (world < 2 || world > 6) return 2;
    World W(world);
    std::vector<int> status(world, 0);
    std::vector<std::string> report(world);
    std::mutex create;
    auto rank_main = [&](int rank) {
        try {
            const std::string in = std::string(argv[2]) + std::to_string(rank) + ".arr";
            const ArrayFile a = read_arrays(in.c_str());
            const mimsem_mesh_desc d = desc_of(a);
            std::unique_lock<std::mutex> lk(create);
            Mesh mesh(d);
            lk.unlock();
            RankCtx rc{&W, rank, mesh.ctx};
            if ((int)a.ints("ranks").size() != world - 1) throw std::runtime_error("this harness needs every rank to neighbour every other");
            Shard sh(&mesh, a.ints("ranks"), a.ints("ghost1"), a.ints("ghost1_off"), a.ints("mirror1"), a.ints("mirror1_off"),
                     a.ints("ghost0"), a.ints("ghost0_off"), a.ints("mirror0"), a.ints("mirror0_off"), a.reals("own0"), a.reals("own1"), &allreduce, &rc);
            sh.use_transport(&transport, &rc);
            const size_t s1 = (size_t)d.nk*d.n1, s2 = (size_t)d.nk*d.n2;
            auto dev = [&](const char* k) { const auto& v = a.reals(k); return mesh.to_device(v.data(), v.size()); };
            double *fg = dev("fg"), *u1 = dev("u1"), *u2 = dev("u2"), *h1 = dev("h1"), *h2 = dev("h2"), *th = dev("theta"), *Pi = dev("Pi"), *vz = dev("velz"), *dudz = dev("dudz");
            double *dF = mesh.device_alloc(s2), *dG = mesh.device_alloc(s2), *Fk = mesh.device_alloc(s1), *Gk = mesh.device_alloc(s1), *fu = mesh.device_alloc(s1);
            HorizSolve hs(&mesh, fg, (long long)a.reals("params").at(0), true, &sh);
            const long r_setup = rc.reductions;
            hs.advection_rhs_ec(u1, u2, h1, h2, th, dF, dG, Fk, Gk);
            hs.momentum_rhs_ec(th, dudz, dudz, vz, vz, Pi, u1, u2, h1, h2, fu, Fk, nullptr, nullptr, nullptr, Fk, hs.last_grad_theta());
            const long r_solves = rc.reductions - r_setup;               // all-reduces INSIDE the evaluation: must be none
            const bool ok = hs.verify();
            const double k2i = hs.k2i();
            std::vector<double> hf(s1), hg(s2);
            mesh.to_host(hf.data(), fu, s1); mesh.to_host(hg.data(), dG, s2);
            const std::string out = std::string(argv[3]) + std::to_string(rank) + ".bin";
            FILE* g = std::fopen(out.c_str(), "wb");
            if (!g) throw std::runtime_error("cannot write " + out);
            std::fwrite(hf.data(), 8, s1, g); std::fwrite(hg.data(), 8, s2, g); std::fwrite(&k2i, 8, 1, g);
            std::fclose(g);
            char buf[400];
            std::snprintf(buf, sizeof buf, "rank %d: %d Chebyshev steps per mass solve, %d solves checked (worst %.2e), all-reduces: set-up %ld, inside the evaluation %ld, "
                          "after it %ld; exchanges %ld; k2i %.12e", rank, hs.cheb_steps, hs.solves_checked, hs.worst_rel, r_setup, r_solves, rc.reductions - r_setup - r_solves,
                          sh.exchanges, k2i);
            report[rank] = buf;
            if (!ok || r_solves != 0 || hs.solves_checked < 6) { report[rank] += "  -- FAIL"; status[rank] = 1; }
            for (double* p : {fg, u1, u2, h1, h2, th, Pi, vz, dudz, dF, dG, Fk, Gk, fu}) mimsem_free(p);
        } catch (const std::exception& e) { std::fprintf(stderr, "rank %d FAIL: %s\n", rank, e.what()); std::_Exit(1); }
    };
    std::vector<std::thread> th;
    for (int r = 0; r < world; r++) th.emplace_back(rank_main, r);
    for (auto& t : th) t.join();
    int bad = 0;
    for (int r = 0; r < world; r++) { std::printf("%s\n", report[r].c_str()); bad += status[r]; }
    std::printf(bad ? "FAIL\n" : "DONE\n");
    return bad ? 1 : 0;
}
