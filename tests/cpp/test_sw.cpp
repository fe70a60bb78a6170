// tests/cpp/test_sw.cpp -- the shallow-water Picard step (row N3) driven from C++: src::SWEqn of mimsem_amd/host/mimsem_sweqn.hpp on a mesh,
// a Coriolis 0-form and a start state the pytest wrapper wrote (the cubed sphere of tests/test_gpu_sweqn.py, device-global numbering), in
// its three modes -- KSP objects (as the reference: src/SWEqn_Picard.cpp:727-791), fixed-length Chebyshev solves issued eagerly, and the same
// recorded as one hipGraph per Picard iteration.  The states after `nsteps` steps go back to the wrapper, which compares them with the numpy
// oracle's (oracle/sw_oracle.py); the three modes are compared with each other here.
//   usage: test_sw <in.bin> <out.bin>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../mimsem_amd/host/mimsem_sweqn.hpp"
#include "../../mimsem_amd/host/sw_io.hpp"

using namespace mimsem_host;

static double rel_l2(const std::vector<double>& a, const std::vector<double>& b) {
    double num = 0, den = 0;
    for (size_t i = 0; i < a.size(); i++) { num += (a[i] - b[i])*(a[i] - b[i]); den += b[i]*b[i]; }
    return std::sqrt(num/den);
}

int main(int argc, char** argv) {
    if (argc < 3) { std::fprintf(stderr, "usage: test_sw in.bin out.bin\n"); return 2; }
    int fails = 0;
    std::vector<double> out[4][2];
    int n1 = 0, n2 = 0;
    try {
        const SWCase cs = read_sw_case(argv[1]);
        const mimsem_mesh_desc d = cs.desc();
        const std::vector<double>&fg = cs.fg, &u0 = cs.u, &h0 = cs.h;
        const int n0 = cs.n0, nsteps = cs.nsteps, nits = cs.nits; const bool q_exact = cs.q_exact; const double dt = cs.dt;
        n1 = cs.n1; n2 = cs.n2;
        Mesh mesh(d);
        double* dfg = mesh.to_device(fg.data(), n0);
        for (int mode = 0; mode < 4; mode++) {              // 0: KSP objects; 1: fixed-length solves, eager; 2: recorded; 3: recorded, the [u|h] step in two launches
            src::SWEqn sw(&mesh, dfg);
            sw.fixed_length = mode > 0; sw.use_graph = mode >= 2; sw.two_launch_steps = mode == 3;
            double *un = mesh.to_device(u0.data(), n1), *hn = mesh.to_device(h0.data(), n2);
            double* bot = cs.bot.empty() ? nullptr : mesh.to_device(cs.bot.data(), n2);          // bottom topography (src/SWEqn_Picard.cpp:727 `bot`)
            for (int s = 0; s < nsteps; s++) {
                sw.solve(un, hn, dt, false, nits, q_exact, bot);
                std::printf("mode %d step %d:", mode, s);
                for (double v : sw.history) std::printf(" %.6e", v);
                std::printf("\n");
            }
            // ... and one step with HALF the time step: the [u|h] operator, its blocks, the spectral regions and the recorded graphs belong to a dt
            // (SWEqn::solve re-assembles A when dt changes, src/SWEqn_Picard.cpp:732-734) -- the set-up must notice
            sw.solve(un, hn, 0.5*dt, false, nits, q_exact, bot);
            out[mode][0].resize(n1); out[mode][1].resize(n2);
            mesh.to_host(out[mode][0].data(), un, n1); mesh.to_host(out[mode][1].data(), hn, n2);
            std::printf("mode %d: Chebyshev steps [u|h] %d, M1 %d, q %d; iterations handed to the KSP objects: %d\n", mode, sw.steps_A, sw.steps_M1, sw.steps_q, sw.fallbacks);
            if (mode > 0 && sw.fallbacks) { std::printf("FAIL: the fixed-length solves missed their tolerance\n"); fails++; }
            mimsem_free(un); mimsem_free(hn); if (bot) mimsem_free(bot);
        }
        mimsem_free(dfg);
    } catch (const std::exception& e) { std::printf("FAIL: %s\n", e.what()); return 1; }
    for (int mode = 1; mode < 4; mode++) {
        const double eu = rel_l2(out[mode][0], out[0][0]), eh = rel_l2(out[mode][1], out[0][1]);
        std::printf("mode %d vs KSP mode: u %.3e  h %.3e\n", mode, eu, eh);
        if (!(eu < 1e-11 && eh < 1e-11)) fails++;
    }
    // the two-launch step does the three-launch step's arithmetic in the same order: the same bits
    if (out[3][0] != out[2][0] || out[3][1] != out[2][1]) { std::printf("FAIL: two-launch and three-launch steps differ\n"); fails++; }
    FILE* g = std::fopen(argv[2], "wb");
    if (!g) { std::perror(argv[2]); return 2; }
    for (int mode = 0; mode < 4; mode++) { std::fwrite(out[mode][0].data(), 8, n1, g); std::fwrite(out[mode][1].data(), 8, n2, g); }
    std::fclose(g);
    std::printf(fails ? "FAILED (%d)\n" : "ALL OK\n", fails);
    return fails ? 1 : 0;
}
