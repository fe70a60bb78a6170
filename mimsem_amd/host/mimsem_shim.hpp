// mimsem_amd/host/mimsem_shim.hpp -- header-only C++ host layer over the C ABI (include/mimsem_hip.h).
//
// The reference's operator classes (eul/Assembly.h:1-384, src/Assembly.h:1-278) are constructed from (Topo*, Geom*,
// LagrangeNode*, LagrangeEdge*) and expose  assemble(...)  + a public PETSc  Mat M  (and MT) that callers feed to MatMult.
// PETSc is not in this image, so this shim keeps the reference's class NAMES, CONSTRUCTOR SIGNATURES and assemble() /
// assemble_up() signatures but works on raw device pointers (what VecGetArray + a device mirror gives): `mult(x, y)` stands
// where the reference has  MatMult(X->M, x, y),  `mult_MT(x, y)`  for MatMult(X->MT, ...).  namespace mimsem_host holds the
// eul/ flavour, mimsem_host::src the src/ flavour (no lev / scale arguments).  The device context of a
// (Topo, Geom) pair is created on first use and shared by every operator built from that pair (Mesh::of).
// INTEGRATION.md shows the same code wrapped in a MATSHELL inside eul/Assembly.cpp.
//
// Topo / Geom here are minimal stand-ins exposing exactly the public members the operator classes read
// (eul/Topo.h:5-51, eul/Geom.h:8-36); a maintainer passes the real objects instead.
#pragma once
#include <cstddef>
#include <map>
#include <memory>
#include <stdexcept>
#include <utility>
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
    // paired: the co-located local 1-form layout of INTEGRATION.md 2.1 (y-edge (r, c) in the odd slot beside x-edge (r, c)) for which
    // the engine's wave-level plan exists; false = the reference's layout (eul/Topo.cpp:215-240), served by the two-pass kernels
    bool paired = false;
    Topo(int order, int nels, int nk_, bool paired_ = false) : elOrd(order), nElsX(nels), nDofsX(order*nels), nk(nk_), paired(paired_) {
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
        for (int iy = 0; iy <= elOrd; iy++) for (int ix = 0; ix < elOrd; ix++) {
            const int r = ey*elOrd + iy, c = ex*elOrd + ix;
            out[k++] = !paired ? 2*(r*nDofsX + c) + 1 : r < nDofsX ? 2*(r*(nDofsX + 1) + c) + 1 : 2*(c*(nDofsX + 1) + nDofsX) + 1;
        }
    }
    // the slot of entry ii of input/edges_y_%.4u.txt (row-major y-edges) in the local vector: loc1[slot] = loc1y[ii]
    int slot_of_edge_y(int ii) const {
        const int r = ii/nDofsX, c = ii%nDofsX;
        return !paired ? 2*ii + 1 : r < nDofsX ? 2*(r*(nDofsX + 1) + c) + 1 : 2*(c*(nDofsX + 1) + nDofsX) + 1;
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
        std::vector<int> i0((size_t)nEl*np1*np1), ix((size_t)nEl*np1*n), iy((size_t)nEl*np1*n), iq(mp12), iqa((size_t)nEl*mp12);
        std::vector<double> th((size_t)g->nk*nEl*mp12, 1.0), ti((size_t)g->nk*nEl*mp12, 1.0);
        const size_t n0q = (size_t)(g->nDofsX + 1)*(g->nDofsX + 1);
        for (int ey = 0; ey < t->nElsX; ey++) for (int ex = 0; ex < t->nElsX; ex++) {
            const int e = ey*t->nElsX + ex;
            t->elInds0_l(ex, ey, &i0[(size_t)e*np1*np1]);
            t->elInds1x_l(ex, ey, &ix[(size_t)e*np1*n]);
            t->elInds1y_l(ex, ey, &iy[(size_t)e*np1*n]);
            g->elInds0_l(t->nElsX, ex, ey, iq.data());
            for (int q = 0; q < mp12; q++) iqa[(size_t)e*mp12 + q] = iq[q];
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
        d.indsq = iqa.data(); d.nq = (int)n0q;
        nEl_ = nEl; n2e = n*n; n0 = d.n0; n1 = d.n1; n2 = d.n2; nk_ = d.nk;
        check(mimsem_ctx_create(&d, device, &ctx), "mimsem_ctx_create");
        use_own_stream();
    }
    // any set of patches the host numbered itself (several cubed-sphere faces on one GPU: the whole sphere of the src/ drivers): the
    // element -> slot tables and the metric as mimsem_mesh_desc takes them.  topo / geom stay null: only classes built from a Mesh* apply.
    explicit Mesh(const mimsem_mesh_desc& d, int device = 0) : topo(nullptr), geom(nullptr) {
        nEl_ = d.nEl; n2e = d.elOrd*d.elOrd; n0 = d.n0; n1 = d.n1; n2 = d.n2; nk_ = d.nk;
        check(mimsem_ctx_create(&d, device, &ctx), "mimsem_ctx_create");
        use_own_stream();
    }
    ~Mesh() { mimsem_ctx_destroy(ctx); }
    // the context of a (Topo, Geom) pair: built on first use, shared by every operator object constructed from the pair (the
    // reference's objects each keep the two pointers; a process holds one pair per rank).  release_all() before the GPU goes away.
    static Mesh* of(const Topo* t, const Geom* g, int device = 0) {
        auto& reg = registry();
        auto it = reg.find({t, g});
        if (it == reg.end()) it = reg.emplace(std::make_pair(t, g), std::unique_ptr<Mesh>(new Mesh(t, g, device))).first;
        return it->second.get();
    }
    // classes the reference constructs from the Topo alone (E10mat, E21mat): the context some Geom-carrying object registered for it
    static Mesh* of_topo(const Topo* t) {
        for (auto& kv : registry()) if (kv.first.first == t) return kv.second.get();
        throw std::runtime_error("mimsem_host: no device context yet for this Topo (construct an operator with (Topo*, Geom*, ...) first)");
    }
    static void release_all() { registry().clear(); }
    // The context launches on a non-blocking stream of its own (not the legacy default stream, which a host without HIP headers would
    // otherwise be left with): a recorded graph replayed on a blocking stream ran 20 % slower (the shallow-water Picard iteration, 1.37 ms
    // against 1.11 ms; scripts/ab_sw_cpp.sh).  Every transfer of this class goes through the same stream (mimsem_memcpy_*); a host with
    // streams of its own passes one to use_stream().
    void use_own_stream() { check(mimsem_ctx_use_own_stream(ctx), "mimsem_ctx_use_own_stream"); }
    void use_default_stream() { check(mimsem_ctx_set_stream(ctx, nullptr), "mimsem_ctx_set_stream"); }
    void use_stream(void* hip_stream) { check(mimsem_ctx_set_stream(ctx, hip_stream), "mimsem_ctx_set_stream"); }
    Mesh(const Mesh&) = delete;
    Mesh& operator=(const Mesh&) = delete;
    double* to_device(const double* host, size_t n) {
        void* p = nullptr;
        check(mimsem_malloc(&p, (long long)(n*sizeof(double))), "mimsem_malloc");
        check(mimsem_memcpy_h2d(ctx, p, host, (long long)(n*sizeof(double))), "h2d");
        return (double*)p;
    }
    void to_host(double* host, const double* dev, size_t n) { check(mimsem_memcpy_d2h(ctx, host, dev, (long long)(n*sizeof(double))), "d2h"); }
    double* device_alloc(size_t n) {
        void* p = nullptr;
        check(mimsem_malloc(&p, (long long)(n*sizeof(double))), "mimsem_malloc");
        return (double*)p;
    }
    // Geom::interp0 / interp1_l / interp1_g / interp2_l / interp2_g (eul/Geom.cpp:328-417) for EVERY quadrature point of the patch:
    // x = device pointer to the local k-form array the reference passes (VecGetArray of the `*l` Vec); out [nEl][mp12] ([..][2] for 1-forms)
    void interp0(const double* x, double* out) const { check(mimsem_interp_quad(ctx, 0, 0, 1, x, 0, out, 0), "interp0"); }
    void interp1_l(const double* x, double* out) const { check(mimsem_interp_quad(ctx, 1, 0, 1, x, 0, out, 0), "interp1_l"); }
    void interp1_g(const double* x, double* out) const { check(mimsem_interp_quad(ctx, 1, MIMSEM_INTERP_GLOBAL, 1, x, 0, out, 0), "interp1_g"); }
    void interp2_l(const double* x, double* out) const { check(mimsem_interp_quad(ctx, 2, 0, 1, x, 0, out, 0), "interp2_l"); }
    void interp2_g(const double* x, double* out) const { check(mimsem_interp_quad(ctx, 2, MIMSEM_INTERP_GLOBAL, 1, x, 0, out, 0), "interp2_g"); }
    const Topo* topo; const Geom* geom; mimsem_ctx* ctx = nullptr;
    int nEl_ = 0, n2e = 0, n0 = 0, n1 = 0, n2 = 0, nk_ = 1;
private:
    static std::map<std::pair<const Topo*, const Geom*>, std::unique_ptr<Mesh>>& registry() {
        static std::map<std::pair<const Topo*, const Geom*>, std::unique_ptr<Mesh>> r;
        return r;
    }
};

// VecScatterBegin / VecScatterEnd on Topo::gtol_0 / gtol_1 (eul/Topo.cpp:145-155) with device-resident vectors: the slot lists are
// the ones VecScatterCreate is given there (per neighbour rank: my ghost slots owned by it, my owned slots it holds as ghosts).
// reverse_add  = VecScatter(gtol, l, g, ADD_VALUES, SCATTER_REVERSE)  (eul/Assembly.cpp:2194-2195)
// forward_insert = VecScatter(gtol, g, l, INSERT_VALUES, SCATTER_FORWARD) (eul/Euler_2.cpp:1455-1456)
// begin_* / end_* are split so that the interior part of an operator (OperatorBase::mult_part) runs while the messages travel.
class VecScatterHalo {
public:
    VecScatterHalo(Mesh* m, int form, const std::vector<int>& ranks, const std::vector<int>& ghost_idx, const std::vector<int>& ghost_off,
                   const std::vector<int>& mirror_idx, const std::vector<int>& mirror_off) : mesh(m) {
        const int nslots = form == 1 ? m->n1 : m->n0, nn = (int)ranks.size();
        check(mimsem_halo_create(m->ctx, nn, ranks.data(), ghost_idx.data(), ghost_off.data(), mirror_idx.data(), mirror_off.data(),
                                 nslots, m->nk_, &rev), "mimsem_halo_create(reverse)");
        check(mimsem_halo_create(m->ctx, nn, ranks.data(), mirror_idx.data(), mirror_off.data(), ghost_idx.data(), ghost_off.data(),
                                 nslots, m->nk_, &fwd), "mimsem_halo_create(forward)");
        if (form == 1) {                       // the slots that travel: their element groups go first in the operators' plans
            std::vector<int> shared(ghost_idx); shared.insert(shared.end(), mirror_idx.begin(), mirror_idx.end());
            check(mimsem_ctx_set_halo_slots(m->ctx, 1, shared.data(), (int)shared.size()), "mimsem_ctx_set_halo_slots");
        }
    }
    ~VecScatterHalo() { mimsem_halo_destroy(rev); mimsem_halo_destroy(fwd); }
    VecScatterHalo(const VecScatterHalo&) = delete;
    VecScatterHalo& operator=(const VecScatterHalo&) = delete;
    void use_rccl(void* nccl_comm) { check(mimsem_halo_set_rccl(rev, nccl_comm), "set_rccl"); check(mimsem_halo_set_rccl(fwd, nccl_comm), "set_rccl"); }
    // the librccl instance that made the communicator (the host's dlopen handle), before the first use_rccl; optional when the process
    // has librccl.so[.1] loaded under that name.  Returns the ABI's code (MIMSEM_ERR_STATE: resolved earlier, nothing changed)
    static int use_rccl_library(void* dl_handle) { return mimsem_halo_use_rccl_library(dl_handle); }
    void use_transport(mimsem_halo_transport_fn fn, void* user) { check(mimsem_halo_set_transport(rev, fn, user), "set_transport"); check(mimsem_halo_set_transport(fwd, fn, user), "set_transport"); }
    void use_loopback() { check(mimsem_halo_set_loopback(rev), "set_loopback"); check(mimsem_halo_set_loopback(fwd), "set_loopback"); }
    void begin_reverse_add(double* v, int nlev, long long stride) { check(mimsem_halo_begin(rev, MIMSEM_HALO_ADD, nlev, v, stride), "halo_begin"); }
    void end_reverse_add() { check(mimsem_halo_end(rev), "halo_end"); }
    void begin_forward_insert(double* v, int nlev, long long stride) { check(mimsem_halo_begin(fwd, MIMSEM_HALO_INSERT, nlev, v, stride), "halo_begin"); }
    void end_forward_insert() { check(mimsem_halo_end(fwd), "halo_end"); }
    void reverse_add(double* v, int nlev, long long stride) { begin_reverse_add(v, nlev, stride); end_reverse_add(); }
    void forward_insert(double* v, int nlev, long long stride) { begin_forward_insert(v, nlev, stride); end_forward_insert(); }
private:
    Mesh* mesh; mimsem_halo *rev = nullptr, *fwd = nullptr;
};

// A recorded launch sequence (mimsem_graph_*): the reference's per-level loops -- for (kk ...) { M1->assemble(kk, SCALE, true);
// M1->mult(x[kk], y[kk]); } (eul/Euler_2.cpp:1427-1457) -- written once between begin() and end(), replayed by launch() with one
// submission.  The arrays are the ones named while recording (VecGetArray of the same Vecs); run the loop once un-recorded first so that
// the library's workspaces have their size.
class Graph {
public:
    explicit Graph(Mesh* m) : mesh(m) {}
    ~Graph() { mimsem_graph_destroy(g); }
    Graph(const Graph&) = delete; Graph& operator=(const Graph&) = delete;
    void begin() { mimsem_graph_destroy(g); g = nullptr; check(mimsem_graph_begin(mesh->ctx), "mimsem_graph_begin"); }
    void end() { check(mimsem_graph_end(mesh->ctx, &g), "mimsem_graph_end"); }
    void launch() { check(mimsem_graph_launch(g), "mimsem_graph_launch"); }
    int nodes() const { return mimsem_graph_num_nodes(g); }
    // record f() -- the host's own loop -- with the capture closed on every exit path
    template <class F> void record(F&& f) {
        begin();
        try { f(); } catch (...) { mimsem_graph* junk = nullptr; (void)mimsem_graph_end(mesh->ctx, &junk); mimsem_graph_destroy(junk); throw; }
        end();
    }
private:
    Mesh* mesh; mimsem_graph* g = nullptr;
};

// common part of every operator class: remembers what assemble() was given, mult() issues the fused launch
class OperatorBase {
protected:
    OperatorBase(Mesh* m, int op_) : mesh(m), op(op_) {}
    OperatorBase(Topo* t, Geom* g, int op_) : mesh(Mesh::of(t, g)), op(op_) {}
    Mesh* mesh; int op; int lev = 0; double scale = 1.0; unsigned flags = 0; const double* field = nullptr;
    const double* field2 = nullptr; double tau = 0.0; bool up = false;      // the assemble_up variants (second field, departure time)
    void apply(const double* x, double* y, unsigned extra) const {
        if (up) check(mimsem_op_apply_up(mesh->ctx, op, lev, 1, scale, tau, flags | extra, field, 0, field2, 0, x, 0, y, 0, 1.0), "mimsem_op_apply_up");
        else check(mimsem_op_apply(mesh->ctx, op, lev, 1, scale, flags | extra, field, 0, x, 0, y, 0, 1.0), "mimsem_op_apply");
    }
public:
    Mesh* device_mesh() const { return mesh; }
    // what the last assemble() set: KSPSetOperators(ksp, X->M, X->M) reads it (class KSP below)
    int op_id() const { return op; } int level() const { return lev; } double op_scale() const { return scale; } unsigned op_flags() const { return flags; }
    const double* op_field() const { return field; } bool is_up() const { return up; }
    // MatMult(X->M, x, y) on device vectors (single level, like the reference)
    void mult(const double* x, double* y) const { apply(x, y, 0u); }
    // the same MatMult in two parts around a halo exchange: mult_part(x, y, MIMSEM_PART_BOUNDARY); halo.begin_reverse_add(y, ...);
    // mult_part(x, y, MIMSEM_PART_INTERIOR); halo.end_reverse_add()  -- no other operator of this Mesh between the two parts
    // error path of a host: forget a BOUNDARY part whose INTERIOR part will not come (the contract of a split apply, mimsem_hip.h)
    void reset_parts() const { check(mimsem_op_apply_part_reset(mesh->ctx), "apply_part_reset"); }
    void mult_part(const double* x, double* y, int part) const {
        if (up) { if (part != MIMSEM_PART_INTERIOR) apply(x, y, 0u); return; }      // upwinded variants run whole in the boundary part
        check(mimsem_op_apply_part(mesh->ctx, op, lev, 1, scale, flags, field, 0, x, 0, y, 0, 1.0, part), "mimsem_op_apply_part");
    }
    // MatMult(X->MT, x, y): the transpose the assemble_up variants build with MatTranspose (eul/Assembly.cpp:261)
    void mult_MT(const double* x, double* y) const { apply(x, y, MIMSEM_FLAG_TRANSPOSE); }
    // the dense element blocks the reference hands to MatSetValues (device, [nEl][esz])
    void element_matrices(double* out) const {
        check(mimsem_op_element_matrices(mesh->ctx, op, lev, scale, flags, field, out), "mimsem_op_element_matrices");
    }
    int elmat_size() const { return mimsem_op_elmat_size(mesh->ctx, op); }
};

// ---- KSP: the solve that follows an operator assembly (eul/HorizSolve.cpp:77-96, :224, :246, :310, :322) -----------------------------
//   KSPCreate(MPI_COMM_WORLD, &ksp1); KSPSetOperators(ksp1, M1->M, M1->M); KSPSetTolerances(ksp1, 1.0e-16, 1.0e-50, PETSC_DEFAULT, 1000);
//   KSPSetType(ksp1, KSPGMRES); KSPGetPC(ksp1, &pc); PCSetType(pc, PCBJACOBI); PCBJacobiSetTotalBlocks(pc, size*nElsX*nElsX, NULL); ...
//   KSPSolve(ksp1, b, x);
// becomes
//   KSP ksp1(&M1_mesh);  ksp1.setOperators(M1);  ksp1.setTolerances(1.0e-16, 1.0e-50, 1000);  ksp1.setType(KSP::GMRES);  ksp1.setPCBJacobi();     (any order)
//   ksp1.solve(b, x);        // after every M1->assemble(lev, SCALE, true): setOperators(M1) again (the reference re-assembles M1->M in place)
// The loops run inside libmimsem_hip (mimsem_ksp_*): no host code between the iterations except the convergence test.
class KSP {
public:
    enum Type { CG = MIMSEM_KSP_CG, GMRES = MIMSEM_KSP_GMRES };
    explicit KSP(Mesh* m, Type t = GMRES) : mesh(m), type(t) { check(mimsem_ksp_create(mesh->ctx, (int)t, &h), "mimsem_ksp_create"); }
    ~KSP() { mimsem_ksp_destroy(h); }
    KSP(const KSP&) = delete; KSP& operator=(const KSP&) = delete;
    // The setters may come in ANY order, as with PETSc (the reference calls KSPSetOperators before KSPSetType / PCSetType,
    // eul/HorizSolve.cpp:77-84): the object remembers operator, type and preconditioner choice; the library handle is brought up to date
    // -- operator re-attached after a type change, preconditioner (re)built from the current operator: PCSetUp -- at the next solve().
    void setType(Type t) {                                   // KSPSetType (re-creates the library object; operator, tolerances and PC choice are kept)
        if (t == type) return;
        mimsem_ksp_destroy(h); h = nullptr; type = t;
        check(mimsem_ksp_create(mesh->ctx, (int)t, &h), "mimsem_ksp_create");
        check(mimsem_ksp_set_tolerances(h, rtol, atol, maxit, restart, 2), "mimsem_ksp_set_tolerances");
        if (guess) check(mimsem_ksp_set_initial_guess_nonzero(h, 1), "mimsem_ksp_set_initial_guess_nonzero");
        attach_operator();
        pc_dirty = true;
    }
    void setTolerances(double rtol_, double atol_, int maxit_, int restart_ = 30) {       // KSPSetTolerances (dtol unused, as in the reference)
        rtol = rtol_; atol = atol_; maxit = maxit_; restart = restart_;
        check(mimsem_ksp_set_tolerances(h, rtol, atol, maxit, restart, 2), "mimsem_ksp_set_tolerances");
    }
    // KSPSetOperators(ksp, X->M, X->M): the operator in the state its last assemble() left it (one level, like the reference's Mat)
    void setOperators(const OperatorBase& A) {
        if (A.is_up()) fail("KSP::setOperators: the upwinded operators go through setOperatorsShell");
        akind = A_OP; a_op = A.op_id(); a_lev = A.level(); a_scale = A.op_scale(); a_flags = A.op_flags(); a_field = A.op_field();
        attach_operator();
        pc_dirty = true;                                     // PCSetUp on the new matrix, at the next solve
    }
    // the packed [u|h] operator of SWEqn::assemble_operator (src/SWEqn_Picard.cpp:622-725) and its coupled element-block preconditioner
    // (PCBJACOBI: the blocks are built from the operator -- PCSetUp -- unless the caller brings its own)
    void setOperatorsSW(double a, double grav, double H, const double* f0, const double* blocks = nullptr) {
        akind = A_SW; sw_a = a; sw_g = grav; sw_H = H; sw_f0 = f0; sw_blocks = blocks;
        attach_operator();
        pc_dirty = true;
    }
    void setOperatorsShell(long long n, mimsem_ksp_apply_fn fn, void* user) {
        akind = A_SHELL; sh_n = n; sh_fn = fn; sh_user = user;
        attach_operator();
        pc_dirty = true;
    }
    void setPCBJacobi() { pc = PC_BJACOBI; pc_dirty = true; }    // PCSetType(pc, PCBJACOBI) + PCBJacobiSetTotalBlocks(one block per element)
    void setPCNone() { pc = PC_NONE; pc_dirty = true; }
    void setPCJacobi(const double* dinv) { pc = PC_JACOBI; pc_dinv = dinv; pc_dirty = true; }     // PCJACOBI with the caller's inverse diagonal (device)
    void setPCShell(mimsem_ksp_apply_fn fn, void* user) { pc = PC_SHELL; pc_fn = fn; pc_user = user; pc_dirty = true; }
    void setInitialGuessNonzero(bool f) { guess = f; check(mimsem_ksp_set_initial_guess_nonzero(h, f ? 1 : 0), "mimsem_ksp_set_initial_guess_nonzero"); }
    void solve(const double* b, double* x) {                 // KSPSolve(ksp, b, x)
        if (akind == A_NONE) fail("KSP::solve before setOperators");
        if (pc_dirty) setup_pc();
        check(mimsem_ksp_solve(h, b, 0, x, 0), "mimsem_ksp_solve");
        check(mimsem_ksp_get_info(h, &its, &rnorm, &reason), "mimsem_ksp_get_info");
    }
    int iterations() const { return its; } double residualNorm() const { return rnorm; } int convergedReason() const { return reason; }
    // For hosts that replace KSPSolve by a fixed-length Chebyshev iteration (src::SWEqn in mimsem_sweqn.hpp): the region of the spectrum of
    // P A from m Arnoldi steps (mimsem_ksp_ritz), and the element blocks PCSetUp built (device; owned by this object, valid until the next set-up)
    void ritz(int m, double* re_min, double* re_max, double* im_max) {
        if (akind == A_NONE) fail("KSP::ritz before setOperators");
        if (pc_dirty) setup_pc();
        check(mimsem_ksp_ritz(h, m, re_min, re_max, im_max), "mimsem_ksp_ritz");
    }
    void pcBlocks(const double** blocks, const double** elem_scale = nullptr, int* nd = nullptr) {
        if (akind == A_NONE) fail("KSP::pcBlocks before setOperators");
        if (pc_dirty) setup_pc();
        check(mimsem_ksp_get_pc_blocks(h, blocks, elem_scale, nd), "mimsem_ksp_get_pc_blocks");
    }
private:
    static void fail(const char* m) { throw std::runtime_error(m); }
    enum AKind { A_NONE, A_OP, A_SW, A_SHELL };
    enum PKind { PC_NONE, PC_BJACOBI, PC_SHELL, PC_JACOBI };
    void attach_operator() {
        switch (akind) {
        case A_OP: check(mimsem_ksp_set_operator(h, a_op, a_lev, 1, a_scale, a_flags, a_field, 0), "mimsem_ksp_set_operator"); break;
        case A_SW: check(mimsem_ksp_set_operator_sw(h, 1, sw_a, sw_g, sw_H, sw_f0, 0), "mimsem_ksp_set_operator_sw"); break;
        case A_SHELL: check(mimsem_ksp_set_operator_shell(h, 1, sh_n, sh_fn, sh_user), "mimsem_ksp_set_operator_shell"); break;
        case A_NONE: break;
        }
    }
    void setup_pc() {                                        // PCSetUp: from the operator the handle holds NOW
        if (pc == PC_SHELL) check(mimsem_ksp_set_pc_shell(h, pc_fn, pc_user), "mimsem_ksp_set_pc_shell");
        else if (pc == PC_JACOBI) check(mimsem_ksp_set_pc_jacobi(h, pc_dinv, 0), "mimsem_ksp_set_pc_jacobi");
        else if (pc == PC_BJACOBI && akind == A_OP) check(mimsem_ksp_set_pc_bjacobi(h), "mimsem_ksp_set_pc_bjacobi");      // 0-, 1- and 2-form operators
        else if (pc == PC_BJACOBI && akind == A_SW) {
            if (sw_blocks) check(mimsem_ksp_set_pc_sw_blocks(h, sw_blocks), "mimsem_ksp_set_pc_sw_blocks");
            else check(mimsem_ksp_set_pc_sw_bjacobi(h), "mimsem_ksp_set_pc_sw_bjacobi");
        } else if (pc == PC_BJACOBI) fail("KSP: PCBJACOBI needs an operator the library can take element blocks of (setOperators / setOperatorsSW), not a shell");
        else if (akind == A_SW && sw_blocks) check(mimsem_ksp_set_pc_sw_blocks(h, sw_blocks), "mimsem_ksp_set_pc_sw_blocks");
        else check(mimsem_ksp_set_pc_none(h), "mimsem_ksp_set_pc_none");
        pc_dirty = false;
    }
    Mesh* mesh; Type type; mimsem_ksp* h = nullptr;
    AKind akind = A_NONE; PKind pc = PC_NONE; bool pc_dirty = true, guess = false;
    int a_op = 0, a_lev = 0; double a_scale = 1.0; unsigned a_flags = 0; const double* a_field = nullptr;
    double sw_a = 0.0, sw_g = 0.0, sw_H = 0.0; const double* sw_f0 = nullptr; const double* sw_blocks = nullptr;
    long long sh_n = 0; mimsem_ksp_apply_fn sh_fn = nullptr; void* sh_user = nullptr;
    mimsem_ksp_apply_fn pc_fn = nullptr; void* pc_user = nullptr; const double* pc_dinv = nullptr;
    double rtol = 1.0e-16, atol = 1.0e-50; int maxit = 1000, restart = 30;
    int its = 0, reason = 0; double rnorm = 0.0;
};
inline void KSPSolve(KSP& ksp, const double* b, double* x) { ksp.solve(b, x); }

// ---- eul/Assembly.h classes (same names, same assemble() argument order) ---------------------------
struct Umat : OperatorBase {     // eul/Assembly.h:1-16
    Umat(Topo* t, Geom* g, LagrangeNode*, LagrangeEdge*) : OperatorBase(t, g, MIMSEM_OP_UMAT) {}
    Umat(Mesh* m, LagrangeNode*, LagrangeEdge*) : OperatorBase(m, MIMSEM_OP_UMAT) {}
    void assemble(int lev_, double scale_, bool vert_scale) { op = MIMSEM_OP_UMAT; up = false; field = nullptr; lev = lev_; scale = scale_; flags = vert_scale ? MIMSEM_FLAG_VERT : 0; }
    // test functions evaluated at x_q + 0.5 tau (ui + uj); M and MT both available afterwards (eul/Assembly.cpp:156-279)
    void assemble_up(int lev_, double scale_, double tau_, const double* ui, const double* uj) {
        op = MIMSEM_OP_UMAT_UP; up = true; field = ui; field2 = uj; tau = tau_; lev = lev_; scale = scale_; flags = 0;
    }
};
struct Wmat : OperatorBase {     // :18-30
    Wmat(Topo* t, Geom* g, LagrangeEdge*) : OperatorBase(t, g, MIMSEM_OP_WMAT) {}
    Wmat(Mesh* m, LagrangeEdge*) : OperatorBase(m, MIMSEM_OP_WMAT) {}
    void assemble(int lev_, double scale_, bool vert_scale) { lev = lev_; scale = scale_; flags = vert_scale ? MIMSEM_FLAG_VERT : 0; }
};
struct Uhmat : OperatorBase {    // :32-60
    Uhmat(Topo* t, Geom* g, LagrangeNode*, LagrangeEdge*) : OperatorBase(t, g, MIMSEM_OP_UHMAT) {}
    Uhmat(Mesh* m, LagrangeNode*, LagrangeEdge*) : OperatorBase(m, MIMSEM_OP_UHMAT) {}
    void assemble(const double* h2, int lev_, bool const_vert, double scale_) { op = MIMSEM_OP_UHMAT; up = false; field = h2; lev = lev_; scale = scale_; flags = const_vert ? MIMSEM_FLAG_VERT : 0; }
    void assemble_up(const double* h2, int lev_, double scale_, double dt, const double* u1) {       // eul/Assembly.cpp:477-560
        op = MIMSEM_OP_UHMAT_UP; up = true; field = h2; field2 = u1; tau = dt; lev = lev_; scale = scale_; flags = 0;
    }
};
struct Pmat : OperatorBase {
    Pmat(Topo* t, Geom* g, LagrangeNode*) : OperatorBase(t, g, MIMSEM_OP_PMAT) {}
    Pmat(Mesh* m, LagrangeNode*) : OperatorBase(m, MIMSEM_OP_PMAT) {}
    void assemble(int lev_, double scale_) { op = MIMSEM_OP_PMAT; field = nullptr; lev = lev_; scale = scale_; }
    void assemble_h(int lev_, double scale_, const double* h2) { op = MIMSEM_OP_PHMAT; field = h2; lev = lev_; scale = scale_; }
};
struct WtQUmat : OperatorBase {
    WtQUmat(Topo* t, Geom* g, LagrangeNode*, LagrangeEdge*) : OperatorBase(t, g, MIMSEM_OP_WTQUMAT) {}
    WtQUmat(Mesh* m, LagrangeNode*, LagrangeEdge*) : OperatorBase(m, MIMSEM_OP_WTQUMAT) {}
    void assemble(const double* u1, int lev_, double scale_) { field = u1; lev = lev_; scale = scale_; }
};
struct RotMat : OperatorBase {
    RotMat(Topo* t, Geom* g, LagrangeNode*, LagrangeEdge*) : OperatorBase(t, g, MIMSEM_OP_ROTMAT) {}
    RotMat(Mesh* m, LagrangeNode*, LagrangeEdge*) : OperatorBase(m, MIMSEM_OP_ROTMAT) {}
    void assemble(const double* q0, int lev_, double scale_) { field = q0; lev = lev_; scale = scale_; }
};
struct Whmat : OperatorBase {
    Whmat(Topo* t, Geom* g, LagrangeEdge*) : OperatorBase(t, g, MIMSEM_OP_WHMAT) {}
    Whmat(Mesh* m, LagrangeEdge*) : OperatorBase(m, MIMSEM_OP_WHMAT) {}
    void assemble(const double* rho, int lev_, double scale_, bool vert_scale_rho) { field = rho; lev = lev_; scale = scale_; flags = vert_scale_rho ? MIMSEM_FLAG_VERT : 0; }
};
struct Ut_mat : OperatorBase {
    Ut_mat(Topo* t, Geom* g, LagrangeNode*, LagrangeEdge*) : OperatorBase(t, g, MIMSEM_OP_UTMAT) {}
    Ut_mat(Mesh* m, LagrangeNode*, LagrangeEdge*) : OperatorBase(m, MIMSEM_OP_UTMAT) {}
    void assemble(int lev_, double scale_) { op = MIMSEM_OP_UTMAT; field = nullptr; lev = lev_; scale = scale_; }
    void assemble_h(int lev_, double scale_, const double* rho) { op = MIMSEM_OP_UTMAT_H; field = rho; lev = lev_; scale = scale_; }
};
struct UtQWmat : OperatorBase {
    UtQWmat(Topo* t, Geom* g, LagrangeNode*, LagrangeEdge*) : OperatorBase(t, g, MIMSEM_OP_UTQWMAT) {}
    UtQWmat(Mesh* m, LagrangeNode*, LagrangeEdge*) : OperatorBase(m, MIMSEM_OP_UTQWMAT) {}
    void assemble(const double* u1, double scale_) { field = u1; scale = scale_; }
};
struct WtQdUdz_mat : OperatorBase {
    WtQdUdz_mat(Topo* t, Geom* g, LagrangeNode*, LagrangeEdge*) : OperatorBase(t, g, MIMSEM_OP_WTQDUDZ) {}
    WtQdUdz_mat(Mesh* m, LagrangeNode*, LagrangeEdge*) : OperatorBase(m, MIMSEM_OP_WTQDUDZ) {}
    void assemble(const double* u1, double scale_) { field = u1; scale = scale_; }
};
struct WmatInv : OperatorBase {
    WmatInv(Topo* t, Geom* g, LagrangeEdge*) : OperatorBase(t, g, MIMSEM_OP_WMATINV) {}
    WmatInv(Mesh* m, LagrangeEdge*) : OperatorBase(m, MIMSEM_OP_WMATINV) {}
    void assemble(int lev_, double scale_) { lev = lev_; scale = scale_; }
};
struct WhmatInv : OperatorBase {
    WhmatInv(Topo* t, Geom* g, LagrangeEdge*) : OperatorBase(t, g, MIMSEM_OP_WHMATINV) {}
    WhmatInv(Mesh* m, LagrangeEdge*) : OperatorBase(m, MIMSEM_OP_WHMATINV) {}
    void assemble(const double* rho, int lev_, double scale_) { field = rho; lev = lev_; scale = scale_; }
};
// Uvec (eul/Assembly.h, Assembly.cpp:2124-2430): the matrix-free vectors are the same kernels applied to `vel`
struct Uvec {
    Uvec(Topo* t, Geom* g, LagrangeNode*, LagrangeEdge*) : mesh(Mesh::of(t, g)) {}
    Uvec(Mesh* m, LagrangeNode*, LagrangeEdge*) : mesh(m) {}
    // Uvec::assemble_hu_up (eul/Assembly.cpp:2281-2373): accumulates fac * (upwinded-test-function flux of vel weighted by rho) into vl
    void assemble_hu_up(int lev, double scale, const double* vel, const double* rho, double fac, double tau, const double* vel2, double* vl) {
        check(mimsem_op_apply_up(mesh->ctx, MIMSEM_OP_UVEC_HU_UP, lev, 1, scale, tau, MIMSEM_FLAG_ACCUM, rho, 0, vel2, 0, vel, 0, vl, 0, fac), "Uvec::assemble_hu_up");
    }
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

// Wvec (eul/Assembly.h:371-384, Assembly.cpp:2443-2552; row B18).  The reference's Wvec never fills its transpose table Wt and all
// its call sites are commented out (eul/HorizSolve.cpp:222-223, 441-448); what the loops compute with Wt = W^T is Wmat x rho and
// WtQUmat(vel2) x vel1, which is what these two issue.
struct Wvec {
    Wvec(Topo* t, Geom* g, LagrangeEdge*) : mesh(Mesh::of(t, g)) {}
    void assemble(int lev, double scale, bool vert_scale, const double* rho, double* vg) {
        check(mimsem_op_apply(mesh->ctx, MIMSEM_OP_WMAT, lev, 1, scale, vert_scale ? MIMSEM_FLAG_VERT : 0u, nullptr, 0, rho, 0, vg, 0, 1.0), "Wvec::assemble");
    }
    void assemble_K(int lev, double scale, const double* vel1, const double* vel2, double* vg) {
        check(mimsem_op_apply(mesh->ctx, MIMSEM_OP_WTQUMAT, lev, 1, scale, 0u, vel2, 0, vel1, 0, vg, 0, 1.0), "Wvec::assemble_K");
    }
    Mesh* mesh;
};

// Umat_ray (eul/Assembly.h; Assembly.cpp:1858-1979): Held-Suarez friction, same assemble() argument order
struct Umat_ray {
    Umat_ray(Topo* t, Geom* g, LagrangeNode*, LagrangeEdge*) : mesh(Mesh::of(t, g)) {}
    Umat_ray(Mesh* m, LagrangeNode*, LagrangeEdge*) : mesh(m) {}
    void assemble(int lev_, double scale_, double dt_, const double* exner_k_, const double* exner_s_) {
        lev = lev_; scale = scale_; dt = dt_; exner_k = exner_k_; exner_s = exner_s_;
    }
    void mult(const double* x, double* y, bool add = false) const {       // add: MatAXPY(M1->M, 1.0, M1ray->M) then MatMult
        check(mimsem_op_apply_up(mesh->ctx, MIMSEM_OP_UMAT_RAY, lev, 1, scale, dt, add ? MIMSEM_FLAG_ACCUM : 0u,
                                 exner_k, 0, exner_s, 0, x, 0, y, 0, 1.0), "Umat_ray");
    }
    void element_matrices(double* out) const {
        check(mimsem_op_element_matrices_ex(mesh->ctx, MIMSEM_OP_UMAT_RAY, lev, scale, dt, 0, exner_k, exner_s, out), "Umat_ray blocks");
    }
    Mesh* mesh; int lev = 0; double scale = 1.0, dt = 0.0; const double *exner_k = nullptr, *exner_s = nullptr;
};

// Pvec / Phvec (Assembly.cpp:585-689): lumped 0-form mass as a vector
struct Pvec {
    Pvec(Topo* t, Geom* g, LagrangeNode*) : mesh(Mesh::of(t, g)) {}
    Pvec(Mesh* m, LagrangeNode*) : mesh(m) {}
    void assemble(int lev, double scale, double* vl) { check(mimsem_pvec(mesh->ctx, lev, 1, scale, nullptr, 0, vl, 0), "Pvec"); }
    Mesh* mesh;
};
struct Phvec {
    Phvec(Topo* t, Geom* g, LagrangeNode*) : mesh(Mesh::of(t, g)) {}
    Phvec(Mesh* m, LagrangeNode*) : mesh(m) {}
    void assemble(const double* h2, int lev, double scale, double* vl) { check(mimsem_pvec(mesh->ctx, lev, 1, scale, h2, 0, vl, 0), "Phvec"); }
    Mesh* mesh;
};

// projections from the quadrature-point grid (Assembly.cpp:691-902); x indexed like Geom's quad grid
struct WtQmat { WtQmat(Topo* t, Geom* g, LagrangeEdge*) : mesh(Mesh::of(t, g)) {} WtQmat(Mesh* m, LagrangeEdge*) : mesh(m) {}
    void mult(const double* xq, double* y) const { check(mimsem_op_apply(mesh->ctx, MIMSEM_OP_WTQ, 0, 1, 1.0, 0, nullptr, 0, xq, 0, y, 0, 1.0), "WtQmat"); } Mesh* mesh; };
struct PtQmat { PtQmat(Topo* t, Geom* g, LagrangeNode*) : mesh(Mesh::of(t, g)) {} PtQmat(Mesh* m, LagrangeNode*) : mesh(m) {}
    void mult(const double* xq, double* y) const { check(mimsem_op_apply(mesh->ctx, MIMSEM_OP_PTQ, 0, 1, 1.0, 0, nullptr, 0, xq, 0, y, 0, 1.0), "PtQmat"); } Mesh* mesh; };
struct UtQmat { UtQmat(Topo* t, Geom* g, LagrangeNode*, LagrangeEdge*) : mesh(Mesh::of(t, g)) {} UtQmat(Mesh* m, LagrangeNode*, LagrangeEdge*) : mesh(m) {}
    void mult(const double* xq2, double* y) const { check(mimsem_op_apply(mesh->ctx, MIMSEM_OP_UTQ, 0, 1, 1.0, 0, nullptr, 0, xq2, 0, y, 0, 1.0), "UtQmat"); } Mesh* mesh; };

// E10mat / E21mat (Assembly.cpp:1102-1220): public members E10, E01 / E21, E12 become mult functions
struct E10mat { explicit E10mat(Topo* t) : mesh(Mesh::of_topo(t)) {} explicit E10mat(Mesh* m) : mesh(m) {}
    void mult_E10(const double* x0, double* y1) const { check(mimsem_incidence_apply(mesh->ctx, 0, 1, x0, 0, y1, 0), "E10"); }
    void mult_E01(const double* x1, double* y0) const { check(mimsem_incidence_apply(mesh->ctx, 3, 1, x1, 0, y0, 0), "E01"); }
    Mesh* mesh; };
struct E21mat { explicit E21mat(Topo* t) : mesh(Mesh::of_topo(t)) {} explicit E21mat(Mesh* m) : mesh(m) {}
    void mult_E21(const double* x1, double* y2) const { check(mimsem_incidence_apply(mesh->ctx, 1, 1, x1, 0, y2, 0), "E21"); }
    void mult_E12(const double* x2, double* y1) const { check(mimsem_incidence_apply(mesh->ctx, 2, 1, x2, 0, y1, 0), "E12"); }
    Mesh* mesh; };

// L2Vecs (eul/L2Vecs.h:1-25): vh[k] = level k's horizontal 2-form vector, vz[e] = column e's vertical vector.
// Device storage: vh [nk][n2] contiguous, vz [nEl][nk*n2e] contiguous (the per-element Vecs of the reference, concatenated).
struct L2Vecs {
    L2Vecs(int nk_, Topo* t, Geom* g) : L2Vecs(nk_, Mesh::of(t, g)) {}              // eul/L2Vecs.h: L2Vecs(int _nk, Topo*, Geom*)
    L2Vecs(int nk_, Mesh* m) : nk(nk_), mesh(m) {
        n2 = m->n2; nEl = m->nEl_; n2e = m->n2e;
        vh = m->device_alloc((size_t)nk*n2); vz = m->device_alloc((size_t)nEl*nk*n2e);
    }
    ~L2Vecs() { mimsem_free(vh); mimsem_free(vz); }
    L2Vecs(const L2Vecs&) = delete;
    L2Vecs& operator=(const L2Vecs&) = delete;
    void HorizToVert() { check(mimsem_l2_transpose(mesh->ctx, 0, nk, vh, n2, vz), "HorizToVert"); }      // L2Vecs.cpp:55-76
    void VertToHoriz() { check(mimsem_l2_transpose(mesh->ctx, 1, nk, vh, n2, vz), "VertToHoriz"); }      // :78-101
    double* level(int k) { return vh + (size_t)k*n2; }                    // vh[k]
    double* column(int e) { return vz + (size_t)e*nk*n2e; }               // vz[e]
    void CopyFromHoriz(const double* host) { check(mimsem_memcpy_h2d(mesh->ctx, vh, host, (long long)((size_t)nk*n2*sizeof(double))), "h2d"); }
    void CopyFromVert(const double* host) { check(mimsem_memcpy_h2d(mesh->ctx, vz, host, (long long)((size_t)nEl*nk*n2e*sizeof(double))), "h2d"); }
    int nk, n2 = 0, nEl = 0, n2e = 0; Mesh* mesh; double *vh = nullptr, *vz = nullptr;
};

// VertOps (eul/VertOps.h:3-72).  The reference assembles ONE column's matrix into VA/VB/... and the caller loops over the
// columns (`for(ii...) { vo->AssembleX(ex, ey, ..., vo->VB); MatMult(vo->VB, a, b); }`); here Assemble* records the operator for
// ALL columns (fields are the concatenated vz arrays) and mult() is that loop's body for every column in one launch.
struct VertOps {
    VertOps(Topo* t, Geom* g) : mesh(Mesh::of(t, g)) {}                              // eul/VertOps.h: VertOps(Topo*, Geom*)
    explicit VertOps(Mesh* m) : mesh(m) {}
    void AssembleConst()                                  { set(MIMSEM_V_CONST); }
    void AssembleConstInv()                               { set(MIMSEM_V_CONST_INV); }
    void AssembleConstWithRho(const double* rho)          { set(MIMSEM_V_CONST_RHO, rho); }
    void AssembleConstWithRhoInv(const double* rho)       { set(MIMSEM_V_CONST_RHO_INV, rho); }
    void AssembleConstWithTheta(const double* theta)      { set(MIMSEM_V_CONST_THETA, theta); }
    void Assemble_EOS_Block(const double* rt)             { set(MIMSEM_V_EOS_BLOCK, rt); }
    void Assemble_EOS_BlockInv(const double* rt, const double* theta) { set(MIMSEM_V_EOS_BLOCK_INV, rt, theta); }
    void AssembleLinear()                                 { set(MIMSEM_V_LINEAR); }
    void AssembleLinearInv()                              { set(MIMSEM_V_LINEAR_INV); }
    void AssembleLinearWithRT(const double* rt, bool do_internal) { set(MIMSEM_V_LINEAR_RT, rt, nullptr, do_internal ? MIMSEM_FLAG_VERT : 0u); }
    void AssembleLinearWithTheta(const double* theta)     { set(MIMSEM_V_LINEAR_THETA, theta); }
    void AssembleLinearWithRho2(const double* rho)        { set(MIMSEM_V_LINEAR_RHO2, rho); }
    void AssembleLinearWithRayleighInv(double dt_fric)    { set(MIMSEM_V_LINEAR_RAYLEIGH_INV); param = dt_fric; }
    void AssembleRayleigh()                               { set(MIMSEM_V_RAYLEIGH); }
    void AssembleLinCon()                                 { set(MIMSEM_V_LINCON); }
    void AssembleLinCon2()                                { set(MIMSEM_V_LINCON2); }
    void AssembleConLin()                                 { set(MIMSEM_V_CONLIN); }
    void AssembleConLinWithW(const double* velz)          { set(MIMSEM_V_CONLIN_W, velz); }
    void AssembleConLinWithRhodPi(const double* theta, const double* dpi) { set(MIMSEM_V_CONLIN_RHODPI, theta, dpi); }
    void AssembleLinearWithRho2_up(const double* rho, double dt, const double* uhl, long long uhl_stride) {
        set(MIMSEM_V_LINEAR_RHO2_UP, rho); param = dt; uh = uhl; uhs = uhl_stride; }
    void AssembleLinCon2_up(double dt, const double* uhl, long long uhl_stride) { set(MIMSEM_V_LINCON2_UP); param = dt; uh = uhl; uhs = uhl_stride; }
    // MatMult(VX, x, y) / MatMultTranspose for every column
    void mult(const double* x, double* y, bool transpose = false) const {
        check(mimsem_colop_apply_ex(mesh->ctx, colop, flags, transpose ? 1 : 0, param, f1, f2, uh, uhs, x, y), "VertOps::mult");
    }
    int nblocks() const { return mimsem_colop_nblocks(mesh->ctx, colop); }
    void blocks(double* out) const { check(mimsem_colop_blocks_ex(mesh->ctx, colop, flags, param, f1, f2, uh, uhs, out), "VertOps::blocks"); }
    // vectors (VertOps.cpp:732-787, 987-1047, 1204-1305)
    void Assemble_EOS_Residual(const double* rt, const double* exner, double* out) { check(mimsem_column_eos(mesh->ctx, 0, rt, exner, 0, 0, out), "EOS_Residual"); }
    void Assemble_EOS_RHS(const double* rt, double* out, double factor, double exponent) { check(mimsem_column_eos(mesh->ctx, 1, rt, nullptr, factor, exponent, out), "EOS_RHS"); }
    void AssembleConstWithLogThetaPlusEta(const double* theta, const double* eta, double* out) { check(mimsem_column_eos(mesh->ctx, 2, theta, eta, 0, 0, out), "LogThetaPlusEta"); }
    void AssembleConstWithRhoExpEta(const double* rho, const double* eta, double* out) { check(mimsem_column_eos(mesh->ctx, 3, rho, eta, 0, 0, out), "RhoExpEta"); }
    void AssembleTempForcing_HS(const double* lat, const double* exner, const double* theta, const double* rho, double* out) {
        check(mimsem_column_temp_forcing_hs(mesh->ctx, lat, exner, theta, rho, out), "TempForcing_HS"); }
    // V10 / V01 / V10_full (VertOps.cpp:134-182)
    void mult_V10(const double* x, double* y) const { check(mimsem_column_incidence(mesh->ctx, 0, x, y), "V10"); }
    void mult_V01(const double* x, double* y) const { check(mimsem_column_incidence(mesh->ctx, 1, x, y), "V01"); }
    void mult_V10_full(const double* x, double* y) const { check(mimsem_column_incidence(mesh->ctx, 2, x, y), "V10_full"); }
    Mesh* mesh;
private:
    void set(int op, const double* a = nullptr, const double* b = nullptr, unsigned fl = 0) { colop = op; f1 = a; f2 = b; flags = fl; param = 0.0; uh = nullptr; uhs = 0; }
    int colop = MIMSEM_V_CONST; unsigned flags = 0; double param = 0.0; const double *f1 = nullptr, *f2 = nullptr, *uh = nullptr; long long uhs = 0;
};

// VertSolve (eul/VertSolve.h): the column solves, every column at once (the reference loops over ex, ey)
struct VertSolve {
    VertSolve(Topo* t, Geom* g, double dt_) : mesh(Mesh::of(t, g)), dt(dt_) {}        // eul/VertSolve.h: VertSolve(Topo*, Geom*, double dt)
    VertSolve(Mesh* m, double dt_) : mesh(m), dt(dt_) {}
    void solve_schur_column_eta(const double* theta, const double* /*velz: unused by the reference*/, const double* rho, const double* eta, const double* pi,
                                double* F_u, double* F_rho, double* F_eta, double* F_pi, double* d_u, double* d_rho, double* d_eta, double* d_pi) {
        check(mimsem_column_solve_schur_eta(mesh->ctx, dt, theta, rho, eta, pi, F_u, F_rho, F_eta, F_pi, d_u, d_rho, d_eta, d_pi), "solve_schur_column_eta");
    }
    void solve_schur_column_3(const double* theta, const double* velz, const double* rho, const double* rt, const double* pi,
                              double* F_u, double* F_rho, double* F_rt, double* F_pi, double* d_u, double* d_rho, double* d_rt, double* d_pi, bool box_twin = false) {
        check(mimsem_column_solve_schur_3(mesh->ctx, dt, box_twin ? MIMSEM_SCHUR3_BOX : 0u, theta, velz, rho, rt, pi,
                                          F_u, F_rho, F_rt, F_pi, d_u, d_rho, d_rt, d_pi, nullptr), "solve_schur_column_3");
    }
    // what PCLU's band-wide pivoting would have guaranteed (eul/VertSolve.cpp:645-653), asked instead: the number of columns of the last
    // solve_schur_column_eta whose refinement did not reach 1e-10 of their solution (-1: the path in use keeps no status), per-column
    // status (0 / 1 / 2) and achieved |correction| / |solution| on request (host arrays of nEl entries, or nullptr)
    int solve_status(int* column_status = nullptr, double* column_ratio = nullptr) {
        int n = -1; check(mimsem_column_solve_status(mesh->ctx, &n, column_status, column_ratio), "column_solve_status"); return n; }
    // ... and PCLU's pivoting itself, for the columns that need it: later solves re-solve flagged columns by a band LU with partial pivoting (status 3)
    // (mode 1: the flagged columns; 2: every column -- the reference's algorithm throughout, at its price; 0: off)
    void set_pivot_fallback(int mode = 1) { check(mimsem_column_set_pivot_fallback(mesh->ctx, mode), "column_set_pivot_fallback"); }
    void diagTheta2(const double* rho, const double* rt, double* theta) { check(mimsem_column_diag_theta(mesh->ctx, 1, rho, rt, theta), "diagTheta2"); }
    void diagTheta_L2(const double* rho, const double* rt, double* theta) { check(mimsem_column_diag_theta(mesh->ctx, 0, rho, rt, theta), "diagTheta_L2"); }
    void diagTheta_up(const double* rho, const double* rt, double* theta, const double* ul, long long ul_stride) {
        check(mimsem_column_diag_theta_up(mesh->ctx, dt, rho, rt, ul, ul_stride, theta), "diagTheta_up"); }
    Mesh* mesh; double dt;
};

// ---- src/ flavour (shallow water, src/Assembly.h:1-278): no lev / scale arguments, no layer thickness ------------------------------
namespace src {
struct Umat : OperatorBase {      // src/Assembly.h:1-13
    Umat(Topo* t, Geom* g, LagrangeNode*, LagrangeEdge*) : OperatorBase(t, g, MIMSEM_OP_UMAT) {}
    void assemble() { op = MIMSEM_OP_UMAT; up = false; field = nullptr; }
};
struct Wmat : OperatorBase {      // :15-24
    Wmat(Topo* t, Geom* g, LagrangeEdge*) : OperatorBase(t, g, MIMSEM_OP_WMAT) {}
    void assemble() {}
};
struct Pmat : OperatorBase {      // :26-35
    Pmat(Topo* t, Geom* g, LagrangeNode*) : OperatorBase(t, g, MIMSEM_OP_PMAT) {}
    void assemble() {}
};
struct Phmat : OperatorBase {     // :37-49
    Phmat(Topo* t, Geom* g, LagrangeNode*) : OperatorBase(t, g, MIMSEM_OP_PHMAT) {}
    void assemble(const double* h2) { op = MIMSEM_OP_PHMAT; up = false; field = h2; }
    // trial functions at the departure points x_q - tau u, tau = 1/(1/(fac dt))  (src/Assembly.cpp:499-567)
    void assemble_up(const double* ul, const double* hl, double fac, double dt) {
        op = MIMSEM_OP_PHMAT_UP; up = true; field = hl; field2 = ul; tau = 1.0/(1.0/(fac*dt));
    }
};
struct Uhmat : OperatorBase {     // :51-79
    Uhmat(Topo* t, Geom* g, LagrangeNode*, LagrangeEdge*) : OperatorBase(t, g, MIMSEM_OP_UHMAT) {}
    void assemble(const double* h2) { field = h2; }
};
struct WtQUmat : OperatorBase {   // :131-155
    WtQUmat(Topo* t, Geom* g, LagrangeNode*, LagrangeEdge*) : OperatorBase(t, g, MIMSEM_OP_WTQUMAT) {}
    void assemble(const double* u1) { field = u1; }
};
struct RotMat : OperatorBase {    // :157-179
    RotMat(Topo* t, Geom* g, LagrangeNode*, LagrangeEdge*) : OperatorBase(t, g, MIMSEM_OP_ROTMAT) {}
    void assemble(const double* q0) { field = q0; }
};
struct Whmat : OperatorBase {     // :199-208
    Whmat(Topo* t, Geom* g, LagrangeEdge*) : OperatorBase(t, g, MIMSEM_OP_WHMAT) {}
    void assemble(const double* h2) { field = h2; }
};
struct RotMat_up : OperatorBase { // :227-250; assemble(q0, ul, tau, dt): the vorticity at x_q - (tau dt) u  (src/Assembly.cpp:1784-1853)
    RotMat_up(Topo* t, Geom* g, LagrangeNode*, LagrangeEdge*) : OperatorBase(t, g, MIMSEM_OP_ROTMAT_UP) { up = true; }
    void assemble(const double* q0, const double* ul, double tau_, double dt) { field = q0; field2 = ul; tau = 1.0/(1.0/(tau_*dt)); }
};
}  // namespace src

}  // namespace mimsem_host
