// mimsem_amd/host/mimsem_shim.hpp -- header-only C++ host layer over the C ABI (include/mimsem_hip.h).
//
// The reference's operator classes (eul/Assembly.h:1-384) are constructed from (Topo*, Geom*, LagrangeNode*,
// LagrangeEdge*) and expose  assemble(...)  + a public PETSc  Mat M  that callers feed to MatMult.  PETSc is not in
// this image, so this shim keeps the reference's class NAMES, constructor shape and assemble() signatures but
// works on raw device pointers (what VecGetArray + a device mirror gives): `mult(x, y)` stands where the reference
// has  MatMult(X->M, x, y).  INTEGRATION.md shows the same code wrapped in a MATSHELL inside eul/Assembly.cpp.
//
// Topo / Geom here are minimal stand-ins exposing exactly the public members the operator classes read
// (eul/Topo.h:5-51, eul/Geom.h:8-36); a maintainer passes the real objects instead.
#pragma once
#include <cstddef>
#include <stdexcept>
#include <string>
#include <vector>
#include "../../include/mimsem_hip.h"

namespace mimsem_host {

inline void check(int rc, const char* what) {
    if (rc != MIMSEM_OK)
        throw std::runtime_error(std::string(what) + ": " + mimsem_strerror(rc) +
                                 (rc == MIMSEM_ERR_HIP ? std::string(" [") + mimsem_last_hip_error() + "]" : std::string()));
}

struct GaussLobatto { int n; };
struct LagrangeNode { int n; GaussLobatto* q; };
struct LagrangeEdge { int n; LagrangeNode* l; };

// the members of eul/Topo.h the hot path reads, with the element->local index formulas of eul/Topo.cpp:200-251
struct Topo {
    int pi = 0, elOrd = 0, nElsX = 0, nDofsX = 0, n0 = 0, n1 = 0, n2 = 0, nk = 1;
    Topo(int order, int nels, int nk_) : elOrd(order), nElsX(nels), nDofsX(order*nels), nk(nk_) {
        n0 = (nDofsX + 1)*(nDofsX + 1); n1 = 2*(nDofsX + 1)*nDofsX; n2 = nDofsX*nDofsX;
    }
    void elInds0_l(int ex, int ey, int* out) const {
        int k = 0;
        for (int iy = 0; iy <= elOrd; iy++) for (int ix = 0; ix <= elOrd; ix++)
            out[k++] = (ey*elOrd + iy)*(nDofsX + 1) + ex*elOrd + ix;
    }
    void elInds1x_l(int ex, int ey, int* out) const {
        int k = 0;
        for (int iy = 0; iy < elOrd; iy++) for (int ix = 0; ix <= elOrd; ix++)
            out[k++] = 2*((ey*elOrd + iy)*(nDofsX + 1) + ex*elOrd + ix);
    }
    void elInds1y_l(int ex, int ey, int* out) const {
        int k = 0;
        for (int iy = 0; iy <= elOrd; iy++) for (int ix = 0; ix < elOrd; ix++)
            out[k++] = 2*((ey*elOrd + iy)*nDofsX + ex*elOrd + ix) + 1;
    }
};

// the members of eul/Geom.h the hot path reads (flat arrays instead of double**** )
struct Geom {
    int nk = 1, quad_n = 0, nDofsX = 0;            // Geom::nk, quad->n, quad-grid points per side - 1
    std::vector<double> det;                        // [nEl][mp12]
    std::vector<double> J;                          // [nEl][mp12][4]
    std::vector<double> thick, thickInv;            // [nk][n0q]   on the quad-point grid (eul/Geom.cpp:743-764)
    void elInds0_l(int nElsX, int ex, int ey, int* out) const {    // eul/Geom.cpp:799-811
        (void)nElsX; int k = 0;
        for (int iy = 0; iy <= quad_n; iy++) for (int ix = 0; ix <= quad_n; ix++)
            out[k++] = (ey*quad_n + iy)*(nDofsX + 1) + ex*quad_n + ix;
    }
};

// one device context per (Topo, Geom) pair, shared by every operator object built from them
class Mesh {
public:
    Mesh(const Topo* t, const Geom* g, int device = 0) : topo(t), geom(g) {
        const int n = t->elOrd, np1 = n + 1, mp12 = (g->quad_n + 1)*(g->quad_n + 1), nEl = t->nElsX*t->nElsX;
        std::vector<int> i0((size_t)nEl*np1*np1), ix((size_t)nEl*np1*n), iy((size_t)nEl*np1*n), iq(mp12);
        std::vector<double> th((size_t)g->nk*nEl*mp12, 1.0), ti((size_t)g->nk*nEl*mp12, 1.0);
        const size_t n0q = (size_t)(g->nDofsX + 1)*(g->nDofsX + 1);
        for (int ey = 0; ey < t->nElsX; ey++) for (int ex = 0; ex < t->nElsX; ex++) {
            const int e = ey*t->nElsX + ex;
            t->elInds0_l(ex, ey, &i0[(size_t)e*np1*np1]);
            t->elInds1x_l(ex, ey, &ix[(size_t)e*np1*n]);
            t->elInds1y_l(ex, ey, &iy[(size_t)e*np1*n]);
            g->elInds0_l(t->nElsX, ex, ey, iq.data());
            if (!g->thick.empty())
                for (int k = 0; k < g->nk; k++) for (int q = 0; q < mp12; q++) {
                    th[((size_t)k*nEl + e)*mp12 + q] = g->thick[(size_t)k*n0q + iq[q]];
                    ti[((size_t)k*nEl + e)*mp12 + q] = g->thickInv[(size_t)k*n0q + iq[q]];
                }
        }
        mimsem_mesh_desc d{};
        d.elOrd = n; d.quadOrd = g->quad_n; d.nEl = nEl; d.nk = g->nk; d.n0 = t->n0; d.n1 = t->n1; d.n2 = t->n2;
        d.inds0 = i0.data(); d.inds1x = ix.data(); d.inds1y = iy.data(); d.inds2 = nullptr;
        d.det = g->det.data(); d.J = g->J.data(); d.thick = th.data(); d.thickInv = ti.data();
        check(mimsem_ctx_create(&d, device, &ctx), "mimsem_ctx_create");
    }
    ~Mesh() { mimsem_ctx_destroy(ctx); }
    Mesh(const Mesh&) = delete;
    Mesh& operator=(const Mesh&) = delete;
    double* to_device(const double* host, size_t n) {
        void* p = nullptr;
        check(mimsem_malloc(&p, (long long)(n*sizeof(double))), "mimsem_malloc");
        check(mimsem_memcpy_h2d(ctx, p, host, (long long)(n*sizeof(double))), "h2d");
        return (double*)p;
    }
    void to_host(double* host, const double* dev, size_t n) { check(mimsem_memcpy_d2h(ctx, host, dev, (long long)(n*sizeof(double))), "d2h"); }
    const Topo* topo; const Geom* geom; mimsem_ctx* ctx = nullptr;
};

// common part of every operator class: remembers what assemble() was given, mult() issues the fused launch
class OperatorBase {
protected:
    OperatorBase(Mesh* m, int op_) : mesh(m), op(op_) {}
    Mesh* mesh; int op; int lev = 0; double scale = 1.0; unsigned flags = 0; const double* field = nullptr;
public:
    // MatMult(X->M, x, y) on device vectors (single level, like the reference)
    void mult(const double* x, double* y) const {
        check(mimsem_op_apply(mesh->ctx, op, lev, 1, scale, flags, field, 0, x, 0, y, 0, 1.0), "mimsem_op_apply");
    }
    // the dense element blocks the reference hands to MatSetValues (device, [nEl][esz])
    void element_matrices(double* out) const {
        check(mimsem_op_element_matrices(mesh->ctx, op, lev, scale, flags, field, out), "mimsem_op_element_matrices");
    }
    int elmat_size() const { return mimsem_op_elmat_size(mesh->ctx, op); }
};

// ---- eul/Assembly.h classes (same names, same assemble() argument order) ---------------------------
struct Umat : OperatorBase {     // eul/Assembly.h:1-16
    Umat(Mesh* m, LagrangeNode*, LagrangeEdge*) : OperatorBase(m, MIMSEM_OP_UMAT) {}
    void assemble(int lev_, double scale_, bool vert_scale) { lev = lev_; scale = scale_; flags = vert_scale ? MIMSEM_FLAG_VERT : 0; }
};
struct Wmat : OperatorBase {     // :18-30
    Wmat(Mesh* m, LagrangeEdge*) : OperatorBase(m, MIMSEM_OP_WMAT) {}
    void assemble(int lev_, double scale_, bool vert_scale) { lev = lev_; scale = scale_; flags = vert_scale ? MIMSEM_FLAG_VERT : 0; }
};
struct Uhmat : OperatorBase {    // :32-60
    Uhmat(Mesh* m, LagrangeNode*, LagrangeEdge*) : OperatorBase(m, MIMSEM_OP_UHMAT) {}
    void assemble(const double* h2, int lev_, bool const_vert, double scale_) { field = h2; lev = lev_; scale = scale_; flags = const_vert ? MIMSEM_FLAG_VERT : 0; }
};
struct Pmat : OperatorBase {
    Pmat(Mesh* m, LagrangeNode*) : OperatorBase(m, MIMSEM_OP_PMAT) {}
    void assemble(int lev_, double scale_) { op = MIMSEM_OP_PMAT; field = nullptr; lev = lev_; scale = scale_; }
    void assemble_h(int lev_, double scale_, const double* h2) { op = MIMSEM_OP_PHMAT; field = h2; lev = lev_; scale = scale_; }
};
struct WtQUmat : OperatorBase {
    WtQUmat(Mesh* m, LagrangeNode*, LagrangeEdge*) : OperatorBase(m, MIMSEM_OP_WTQUMAT) {}
    void assemble(const double* u1, int lev_, double scale_) { field = u1; lev = lev_; scale = scale_; }
};
struct RotMat : OperatorBase {
    RotMat(Mesh* m, LagrangeNode*, LagrangeEdge*) : OperatorBase(m, MIMSEM_OP_ROTMAT) {}
    void assemble(const double* q0, int lev_, double scale_) { field = q0; lev = lev_; scale = scale_; }
};
struct Whmat : OperatorBase {
    Whmat(Mesh* m, LagrangeEdge*) : OperatorBase(m, MIMSEM_OP_WHMAT) {}
    void assemble(const double* rho, int lev_, double scale_, bool vert_scale_rho) { field = rho; lev = lev_; scale = scale_; flags = vert_scale_rho ? MIMSEM_FLAG_VERT : 0; }
};
struct Ut_mat : OperatorBase {
    Ut_mat(Mesh* m, LagrangeNode*, LagrangeEdge*) : OperatorBase(m, MIMSEM_OP_UTMAT) {}
    void assemble(int lev_, double scale_) { op = MIMSEM_OP_UTMAT; field = nullptr; lev = lev_; scale = scale_; }
    void assemble_h(int lev_, double scale_, const double* rho) { op = MIMSEM_OP_UTMAT_H; field = rho; lev = lev_; scale = scale_; }
};
struct UtQWmat : OperatorBase {
    UtQWmat(Mesh* m, LagrangeNode*, LagrangeEdge*) : OperatorBase(m, MIMSEM_OP_UTQWMAT) {}
    void assemble(const double* u1, double scale_) { field = u1; scale = scale_; }
};
struct WtQdUdz_mat : OperatorBase {
    WtQdUdz_mat(Mesh* m, LagrangeNode*, LagrangeEdge*) : OperatorBase(m, MIMSEM_OP_WTQDUDZ) {}
    void assemble(const double* u1, double scale_) { field = u1; scale = scale_; }
};
struct WmatInv : OperatorBase {
    WmatInv(Mesh* m, LagrangeEdge*) : OperatorBase(m, MIMSEM_OP_WMATINV) {}
    void assemble(int lev_, double scale_) { lev = lev_; scale = scale_; }
};
struct WhmatInv : OperatorBase {
    WhmatInv(Mesh* m, LagrangeEdge*) : OperatorBase(m, MIMSEM_OP_WHMATINV) {}
    void assemble(const double* rho, int lev_, double scale_) { field = rho; lev = lev_; scale = scale_; }
};
// Uvec (eul/Assembly.h, Assembly.cpp:2124-2430): the matrix-free vectors are the same kernels applied to `vel`
struct Uvec {
    Uvec(Mesh* m, LagrangeNode*, LagrangeEdge*) : mesh(m) {}
    void assemble(int lev, double scale, bool /*vert_scale: ignored by the reference too*/, const double* vel, double* vl) {
        check(mimsem_op_apply(mesh->ctx, MIMSEM_OP_UMAT, lev, 1, scale, MIMSEM_FLAG_VERT, nullptr, 0, vel, 0, vl, 0, 1.0), "Uvec::assemble");
    }
    void assemble_hu(int lev, double scale, const double* vel, const double* rho, bool zero_and_scatter, double fac, double* vl) {
        check(mimsem_op_apply(mesh->ctx, MIMSEM_OP_UHMAT, lev, 1, scale, MIMSEM_FLAG_VERT | (zero_and_scatter ? 0u : MIMSEM_FLAG_ACCUM),
                              rho, 0, vel, 0, vl, 0, fac), "Uvec::assemble_hu");
    }
    void assemble_wxu(int lev, double scale, const double* vel, const double* vort, double* vl) {
        check(mimsem_op_apply(mesh->ctx, MIMSEM_OP_ROTMAT, lev, 1, scale, 0, vort, 0, vel, 0, vl, 0, 1.0), "Uvec::assemble_wxu");
    }
    Mesh* mesh;
};

}  // namespace mimsem_host
