// sw_io.hpp -- a shallow-water case on disk for the C++ hosts (tests/cpp/test_sw.cpp, mimsem_amd/host/sw_call.cpp): the mesh tables of
// mimsem_mesh_desc, the Coriolis 0-form and a start state, written by mimsem_amd/workloads.py::write_sw_case.
//   int32 [12]: elOrd quadOrd nEl nk n0 n1 n2 nq nsteps nits q_exact reserved;  int32 tables inds0 inds1x inds1y inds2 indsq;
//   float64: det J thick thickInv fg[n0] u[n1] h[n2] dt  [bot[n2] when the header's last entry is 1]
#pragma once
#include <cstdio>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>
#include "../../include/mimsem_hip.h"

namespace mimsem_host {
struct SWCase {
    int n = 0, m = 0, nEl = 0, nk = 1, n0 = 0, n1 = 0, n2 = 0, nq = 0, nsteps = 0, nits = 0; bool q_exact = false; double dt = 0.0;
    std::vector<int> i0, ix, iy, i2, iq;
    std::vector<double> det, J, th, ti, fg, u, h, bot;          // bot: empty = no topography
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
    if (hd[11] == 1) rd(c.bot, (size_t)c.n2);
    std::fclose(f);
    return c;
}
}  // namespace mimsem_host

// ---- named arrays on disk (tests/cpp/test_horiz.cpp): "MSEMARR1", int32 count, then per array {int32 name length, name, int32 type
// (0 = int32, 1 = float64), int64 entries, data}; written by mimsem_amd/workloads.py::write_arrays --------------------------------------
namespace mimsem_host {
struct ArrayFile {
    std::map<std::string, std::vector<int>> i;
    std::map<std::string, std::vector<double>> d;
    const std::vector<int>& ints(const std::string& k) const { auto it = i.find(k); if (it == i.end()) throw std::runtime_error("no int array " + k); return it->second; }
    const std::vector<double>& reals(const std::string& k) const { auto it = d.find(k); if (it == d.end()) throw std::runtime_error("no real array " + k); return it->second; }
    bool has(const std::string& k) const { return d.count(k) || i.count(k); }
};
inline ArrayFile read_arrays(const char* path) {
    FILE* f = std::fopen(path, "rb");
    if (!f) throw std::runtime_error(std::string("cannot open ") + path);
    auto need = [&](void* p, size_t sz, size_t n) { if (n && std::fread(p, sz, n, f) != n) { std::fclose(f); throw std::runtime_error(std::string("short read: ") + path); } };
    char magic[8]; int count = 0;
    need(magic, 1, 8); need(&count, 4, 1);
    if (std::memcmp(magic, "MSEMARR1", 8) != 0 || count < 0 || count > 4096) { std::fclose(f); throw std::runtime_error("not an array file"); }
    ArrayFile a;
    for (int k = 0; k < count; k++) {
        int nl = 0, type = 0; long long n = 0;
        need(&nl, 4, 1);
        if (nl < 1 || nl > 255) { std::fclose(f); throw std::runtime_error("bad name length"); }
        std::string name((size_t)nl, ' ');
        need(&name[0], 1, (size_t)nl); need(&type, 4, 1); need(&n, 8, 1);
        if (n < 0 || (type != 0 && type != 1)) { std::fclose(f); throw std::runtime_error("bad array header"); }
        if (type == 0) { auto& v = a.i[name]; v.resize((size_t)n); need(v.data(), 4, (size_t)n); }
        else { auto& v = a.d[name]; v.resize((size_t)n); need(v.data(), 8, (size_t)n); }
    }
    std::fclose(f);
    return a;
}
// the mesh tables of a DeviceMesh stored under their mimsem_mesh_desc names + "sizes" = {elOrd quadOrd nEl nk n0 n1 n2 nq}
inline mimsem_mesh_desc desc_of(const ArrayFile& a) {
    const auto& s = a.ints("sizes");
    mimsem_mesh_desc d{};
    d.elOrd = s.at(0); d.quadOrd = s.at(1); d.nEl = s.at(2); d.nk = s.at(3); d.n0 = s.at(4); d.n1 = s.at(5); d.n2 = s.at(6); d.nq = s.at(7);
    d.inds0 = a.ints("inds0").data(); d.inds1x = a.ints("inds1x").data(); d.inds1y = a.ints("inds1y").data(); d.inds2 = a.ints("inds2").data();
    d.indsq = a.ints("indsq").data(); d.det = a.reals("det").data(); d.J = a.reals("J").data(); d.thick = a.reals("thick").data();
    d.thickInv = a.reals("thickInv").data();
    return d;
}
}  // namespace mimsem_host
