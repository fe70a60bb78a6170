// sw_io.hpp -- a shallow-water case on disk for the C++ hosts (tests/cpp/test_sw.cpp, mimsem_amd/host/sw_call.cpp): the mesh tables of
// mimsem_mesh_desc, the Coriolis 0-form and a start state, written by mimsem_amd/workloads.py::write_sw_case.
//   int32 [12]: elOrd quadOrd nEl nk n0 n1 n2 nq nsteps nits q_exact reserved;  int32 tables inds0 inds1x inds1y inds2 indsq;
//   float64: det J thick thickInv fg[n0] u[n1] h[n2] dt
#pragma once
#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>
#include "../../include/mimsem_hip.h"

namespace mimsem_host {
struct SWCase {
    int n = 0, m = 0, nEl = 0, nk = 1, n0 = 0, n1 = 0, n2 = 0, nq = 0, nsteps = 0, nits = 0; bool q_exact = false; double dt = 0.0;
    std::vector<int> i0, ix, iy, i2, iq;
    std::vector<double> det, J, th, ti, fg, u, h;
    mimsem_mesh_desc desc() const {
        mimsem_mesh_desc d{};
        d.elOrd = n; d.quadOrd = m; d.nEl = nEl; d.nk = nk; d.n0 = n0; d.n1 = n1; d.n2 = n2; d.nq = nq;
        d.inds0 = i0.data(); d.inds1x = ix.data(); d.inds1y = iy.data(); d.inds2 = i2.data(); d.indsq = iq.data();
        d.det = det.data(); d.J = J.data(); d.thick = th.data(); d.thickInv = ti.data();
        return d;
    }
};
inline SWCase read_sw_case(const char* path) {
    FILE* f = std::fopen(path, "rb");
    if (!f) throw std::runtime_error(std::string("cannot open ") + path);
    auto rd = [&](auto& v, size_t n) {
        v.resize(n);
        if (n && std::fread(v.data(), sizeof(v[0]), n, f) != n) { std::fclose(f); throw std::runtime_error(std::string("short read: ") + path); }
    };
    SWCase c; std::vector<int> hd; std::vector<double> dt;
    rd(hd, 12);
    c.n = hd[0]; c.m = hd[1]; c.nEl = hd[2]; c.nk = hd[3]; c.n0 = hd[4]; c.n1 = hd[5]; c.n2 = hd[6]; c.nq = hd[7]; c.nsteps = hd[8]; c.nits = hd[9];
    c.q_exact = hd[10] != 0;
    if (c.n < 1 || c.m != c.n || c.nEl < 1 || c.nk < 1 || c.n0 < 1 || c.n1 < 1 || c.n2 < 1) { std::fclose(f); throw std::runtime_error("bad header"); }
    const size_t np1 = (size_t)c.n + 1, mp12 = (size_t)(c.m + 1)*(c.m + 1), e = (size_t)c.nEl;
    rd(c.i0, e*np1*np1); rd(c.ix, e*np1*c.n); rd(c.iy, e*np1*c.n); rd(c.i2, e*c.n*c.n); rd(c.iq, e*mp12);
    rd(c.det, e*mp12); rd(c.J, e*mp12*4); rd(c.th, (size_t)c.nk*e*mp12); rd(c.ti, (size_t)c.nk*e*mp12);
    rd(c.fg, (size_t)c.n0); rd(c.u, (size_t)c.n1); rd(c.h, (size_t)c.n2); rd(dt, 1);
    c.dt = dt[0];
    std::fclose(f);
    return c;
}
}  // namespace mimsem_host
