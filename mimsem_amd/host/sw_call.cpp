// mimsem_amd/host/sw_call.cpp -- shallow-water time steps per second with the HOST in C++: src::SWEqn of mimsem_sweqn.hpp (the Picard step of
// src/SWEqn_Picard.cpp:727-791 over the C ABI) on a case bench.py wrote (sw_io.hpp), in its default mode (fixed-length solves, one hipGraph
// per Picard iteration) and with the KSP objects (the reference's structure).  Built by __graft_entry__.build(), run as a child of bench.py.
//   usage: sw_call <case.bin> [warm-up steps]      prints one JSON object
#include <chrono>
#include <csignal>
#include <cstdio>
#include <cstdlib>
#include <execinfo.h>
#include <unistd.h>
#include "mimsem_sweqn.hpp"
#include "sw_io.hpp"

using namespace mimsem_host;
using clk = std::chrono::steady_clock;

// MIMSEM_BACKTRACE=1: the frames of a fatal signal on stderr (module + offset), so that a crash under a profiler can be attributed to the tool's
// library or to this one from ONE run (round 5 left a SIGSEGV of this program under rocprofv3 --kernel-trace unexplained: scripts/prof_sw_cpp.sh)
static void on_fatal(int sig) {
    void* fr[64];
    const int n = backtrace(fr, 64);
    const char msg[] = "sw_call: fatal signal, frames of the faulting thread:\n";
    (void)!write(2, msg, sizeof msg - 1);
    backtrace_symbols_fd(fr, n, 2);
    signal(sig, SIG_DFL); raise(sig);
}

int main(int argc, char** argv) {
    if (argc < 2) { std::fprintf(stderr, "usage: sw_call case.bin [warm-up steps]\n"); return 2; }
    if (const char* e = std::getenv("MIMSEM_BACKTRACE")) if (std::atoi(e)) { signal(SIGSEGV, on_fatal); signal(SIGBUS, on_fatal); signal(SIGABRT, on_fatal); }
    const int warm = argc > 2 ? std::atoi(argv[2]) : 3;
    try {
        const SWCase cs = read_sw_case(argv[1]);
        const mimsem_mesh_desc d = cs.desc();
        Mesh mesh(d);
        if (std::getenv("MIMSEM_EXPERIMENTS") && std::atoi(std::getenv("MIMSEM_EXPERIMENTS")))        // (closed experiment; A/B: scripts/ab_sw_cpp.sh)
            if (const char* e = std::getenv("MIMSEM_SW_DEFAULT_STREAM")) if (std::atoi(e)) mesh.use_default_stream();
        double* fg = mesh.to_device(cs.fg.data(), cs.fg.size());
        std::printf("{");
        for (int mode = 0; mode < 2; mode++) {
            src::SWEqn sw(&mesh, fg);
            sw.fixed_length = mode == 0; sw.use_graph = mode == 0;
            double *un = mesh.to_device(cs.u.data(), cs.u.size()), *hn = mesh.to_device(cs.h.data(), cs.h.size());
            for (int s = 0; s < warm; s++) sw.solve(un, hn, cs.dt, false, cs.nits, cs.q_exact);
            check(mimsem_ctx_sync(mesh.ctx), "sync");
            const auto t0 = clk::now();
            size_t picard = 0;
            for (int s = 0; s < cs.nsteps; s++) { sw.solve(un, hn, cs.dt, false, cs.nits, cs.q_exact); picard += sw.history.size(); }
            check(mimsem_ctx_sync(mesh.ctx), "sync");
            const double el = std::chrono::duration<double>(clk::now() - t0).count();
            std::printf("%s\"%s\": {\"steps_per_s\": %.3f, \"ms_per_step\": %.4f, \"picard_iterations_per_step\": %.2f, \"chebyshev_steps\": [%d, %d, %d], "
                        "\"iterations_handed_to_ksp\": %d, \"graph_submit_us\": %.1f, \"wait_and_read_us\": %.1f, \"graph_nodes_first_iteration\": %d, \"graph_nodes_later_iterations\": %d}",
                        mode ? ", " : "", mode == 0 ? "graph" : "ksp_objects",
                        cs.nsteps/el, 1e3*el/cs.nsteps, (double)picard/cs.nsteps, sw.steps_A, sw.steps_M1, sw.steps_q, sw.fallbacks,
                        sw.replays ? sw.us_submit/sw.replays : 0.0, sw.replays ? sw.us_wait/sw.replays : 0.0, sw.graph_nodes(true), sw.graph_nodes(false));
            mimsem_free(un); mimsem_free(hn);
        }
        std::printf("}\n");
        mimsem_free(fg);
    } catch (const std::exception& e) { std::fprintf(stderr, "sw_call: %s\n", e.what()); return 1; }
    return 0;
}
