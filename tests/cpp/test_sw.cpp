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

using namespace mimsem_host;

template <class T> static std::vector<T> rd(FILE* f, size_t n) {
    std::vector<T> v(n);
    if (n && std::fread(v.data(), sizeof(T), n, f) != n) { std::fprintf(stderr, "short read\n"); std::exit(2); }
    return v;
}
static double rel_l2(const std::vector<double>& a, const std::vector<double>& b) {
    double num = 0, den = 0;
    for (size_t i = 0; i < a.size(); i++) { num += (a[i] - b[i])*(a[i] - b[i]); den += b[i]*b[i]; }
    return std::sqrt(num/den);
}

int main(int argc, char** argv) {
    if (argc < 3) { std::fprintf(stderr, "usage: test_sw in.bin out.bin\n"); return 2; }
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) { std::perror(argv[1]); return 2; }
    auto hd = rd<int>(f, 12);            // elOrd quadOrd nEl nk n0 n1 n2 nq nsteps nits q_exact reserved
    const int n = hd[0], m = hd[1], nEl = hd[2], n0 = hd[4], n1 = hd[5], n2 = hd[6], nsteps = hd[8], nits = hd[9];
    const bool q_exact = hd[10] != 0;
    const size_t mp12 = (size_t)(m + 1)*(m + 1);
    auto i0 = rd<int>(f, (size_t)nEl*(n + 1)*(n + 1)), ix = rd<int>(f, (size_t)nEl*(n + 1)*n), iy = rd<int>(f, (size_t)nEl*(n + 1)*n);
    auto i2 = rd<int>(f, (size_t)nEl*n*n), iq = rd<int>(f, nEl*mp12);
    auto det = rd<double>(f, nEl*mp12), J = rd<double>(f, nEl*mp12*4), th = rd<double>(f, (size_t)hd[3]*nEl*mp12), ti = rd<double>(f, (size_t)hd[3]*nEl*mp12);
    auto fg = rd<double>(f, n0), u0 = rd<double>(f, n1), h0 = rd<double>(f, n2), dtv = rd<double>(f, 1);
    std::fclose(f);
    const double dt = dtv[0];
    mimsem_mesh_desc d{};
    d.elOrd = n; d.quadOrd = m; d.nEl = nEl; d.nk = hd[3]; d.n0 = n0; d.n1 = n1; d.n2 = n2; d.nq = hd[7];
    d.inds0 = i0.data(); d.inds1x = ix.data(); d.inds1y = iy.data(); d.inds2 = i2.data(); d.indsq = iq.data();
    d.det = det.data(); d.J = J.data(); d.thick = th.data(); d.thickInv = ti.data();
    int fails = 0;
    std::vector<double> out[3][2];
    try {
        Mesh mesh(d);
        double* dfg = mesh.to_device(fg.data(), n0);
        for (int mode = 0; mode < 3; mode++) {
            src::SWEqn sw(&mesh, dfg);
            sw.fixed_length = mode > 0; sw.use_graph = mode == 2;
            double *un = mesh.to_device(u0.data(), n1), *hn = mesh.to_device(h0.data(), n2);
            for (int s = 0; s < nsteps; s++) {
                sw.solve(un, hn, dt, false, nits, q_exact);
                std::printf("mode %d step %d:", mode, s);
                for (double v : sw.history) std::printf(" %.6e", v);
                std::printf("\n");
            }
            out[mode][0].resize(n1); out[mode][1].resize(n2);
            mesh.to_host(out[mode][0].data(), un, n1); mesh.to_host(out[mode][1].data(), hn, n2);
            std::printf("mode %d: Chebyshev steps [u|h] %d, M1 %d, q %d; iterations handed to the KSP objects: %d\n", mode, sw.steps_A, sw.steps_M1, sw.steps_q, sw.fallbacks);
            if (mode > 0 && sw.fallbacks) { std::printf("FAIL: the fixed-length solves missed their tolerance\n"); fails++; }
            mimsem_free(un); mimsem_free(hn);
        }
        mimsem_free(dfg);
    } catch (const std::exception& e) { std::printf("FAIL: %s\n", e.what()); return 1; }
    for (int mode = 1; mode < 3; mode++) {
        const double eu = rel_l2(out[mode][0], out[0][0]), eh = rel_l2(out[mode][1], out[0][1]);
        std::printf("mode %d vs KSP mode: u %.3e  h %.3e\n", mode, eu, eh);
        if (!(eu < 1e-11 && eh < 1e-11)) fails++;
    }
    FILE* g = std::fopen(argv[2], "wb");
    if (!g) { std::perror(argv[2]); return 2; }
    for (int mode = 0; mode < 3; mode++) { std::fwrite(out[mode][0].data(), 8, n1, g); std::fwrite(out[mode][1].data(), 8, n2, g); }
    std::fclose(g);
    std::printf(fails ? "FAILED (%d)\n" : "ALL OK\n", fails);
    return fails ? 1 : 0;
}
