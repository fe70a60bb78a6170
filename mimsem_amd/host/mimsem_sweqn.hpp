// mimsem_sweqn.hpp -- the shallow-water Picard step (row N3) driven from C++ over the C ABI: the counterpart of the reference's SWEqn class
// (src/SWEqn_Picard.cpp: diagnose_F :253-284, diagnose_Phi :289-320, diagnose_q :322-341, assemble_residual :402-607, assemble_operator
// :622-725, solve :727-791) for a host that holds its fields in device memory.  Same call shape -- solve(un, hn, dt, save, nits, q_exact, bot) --
// and the same arithmetic; what differs is how the nested linear systems are solved:
//   krylov mode   every KSPSolve of the reference is a KSP object of mimsem_shim.hpp (CG for the 1-form mass, GMRES for the upwinded lumped
//                 0-form mass and the [u|h] system), iterations inside libmimsem_hip;
//   fixed mode    (default) the three systems have spectra known once per dt (KSP::ritz): Chebyshev semi-iterations of FIXED length replace
//                 the Krylov solves -- no inner product, no host round trip -- and one whole Picard iteration (residual assembly, its solves,
//                 the [u|h] solve, the update, the check norms) is recorded ONCE as a hipGraph (class Graph) and replayed: one submission and
//                 one read of a few scalars per Picard iteration.  Every solve logs {|last residual|^2, |P b|^2}; a solve that misses its
//                 tolerance sends THAT Picard iteration through the krylov mode again from the saved state.
//   SHARDED (round 6)  a rank that holds some patches of the sphere passes a Shard (below): the fixed mode then runs with the halo exchanges
//                 inside its solves -- a Chebyshev step needs its operator's and its preconditioner's 1-form (0-form) results completed over
//                 the halo and nothing else: no inner product, hence NO all-reduce inside any solve -- and the check norms of a whole Picard
//                 iteration, ownership-weighted partial sums, are reduced ONCE (Shard::allreduce: the host's MPI_Allreduce).  The spectral
//                 regions come from an Arnoldi process of the host's own (Shard::ritz: weighted, all-reduced inner products, once per dt).
//                 The reference's distributed step: src/SWEqn_Picard.cpp:751-765, gtol_x :131-153, :341-400, ghost updates :422-425.
// Header-only, C++17, no HIP toolchain needed (everything goes through include/mimsem_hip.h).
#pragma once
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <functional>
#include <random>
#include <utility>
#include "mimsem_shard.hpp"

namespace mimsem_host {

// coefficient tables of the Chebyshev iterations (they depend on the spectral region and the step number only)
namespace cheb {
// p_k = z_k + beta_k p_{k-1}; x += alpha_k p_k for a spectrum inside an ellipse with centre d and foci d +- c (Manteuffel 1977); c2 = c^2
// may be negative (foci d +- i|c|: a spectrum stretched along the imaginary direction) -- the recurrence stays real
inline std::vector<std::pair<double, double>> ellipse(double d, double c2, int steps) {
    std::vector<std::pair<double, double>> co;
    double al = 1.0/d;
    co.emplace_back(al, 0.0);
    for (int k = 1; k < steps; k++) {
        const double be = (k == 1 ? 0.5 : 0.25)*c2*al*al;
        al = 1.0/(d - be/al);
        co.emplace_back(al, be);
    }
    return co;
}
// asymptotic convergence factor for an ellipse with centre d > 0 and semi-axes a_re, a_im
inline double ellipse_rate(double d, double a_re, double a_im) {
    const double c2 = a_re*a_re - a_im*a_im;
    return (a_re + a_im)/(d + std::sqrt(std::max(d*d - c2, 0.0)));
}
// contraction per step on a real interval [lmin, lmax]
inline double interval_rate(double lmin, double lmax) { const double s = std::sqrt(lmax/lmin); return (s - 1.0)/(s + 1.0); }
}  // namespace cheb

namespace src {

class SWEqn {
public:
    // constants of src/SWEqn_Picard.cpp:22-30
    static constexpr double RAD_EARTH = 6371220.0, RAD_SPHERE = 6371220.0, H_MEAN = 1.0e+4, ROS_ALPHA = 0.5, UP_TAU = 0.5;
    double grav = 9.80616*(RAD_SPHERE/RAD_EARTH), omega = 7.292e-5;
    double rtol = 1.0e-14;               // tolerance of every nested solve (relative, preconditioned residual)
    bool fixed_length = true;            // Chebyshev solves of fixed length (false: the KSP objects, as the reference)
    bool use_graph = true;               // ... recorded as one hipGraph per Picard iteration kind
    bool dual_solves = true;             // the mass-flux and the potential-vorticity solve of an iteration in shared launches (mimsem_sw_dual_chebyshev: the same bits, ~40 launches fewer)
    bool two_launch_steps = false;       // the [u|h] Chebyshev step in two launches (mimsem_sw_chebyshev_step2) instead of three: correct, and no
                                         // faster -- 445.6 against 450.8-453.0 steps/s (profiles/r05_sw_cpp_ab.txt): the element pass grows by what the epilogue cost
    std::vector<double> history;         // |dx| / |x| of the iterations of the last solve()
    int fallbacks = 0;                   // Picard iterations the fixed mode handed to the krylov mode
    int steps_A = 0, steps_M1 = 0, steps_q = 0;
    int graph_nodes(bool first) const { return have_graph[first ? 0 : 1] ? gr[first ? 0 : 1].nodes() : 0; }      // launches of a recorded Picard iteration (0: not recorded)
    double us_submit = 0.0, us_wait = 0.0; long replays = 0;      // host time inside the graph submissions / waiting for the check norms

    // fg: the Coriolis 0-form (SWEqn::coriolis, src/SWEqn_Picard.cpp:95-140), device, n0 entries; it must outlive the object
    // shard (optional): this rank's exchanges and reductions when the sphere is dealt to several ranks -- fixed-length mode only (the KSP objects
    // of the krylov mode iterate inside the library on one context's operator), eager launches with the exchanges in between (no graph)
    SWEqn(Mesh* m, const double* fg_dev, Shard* shard = nullptr) : mesh(m), fg(fg_dev), sh(shard), ksp1(m, KSP::CG), ksp0(m, KSP::GMRES), kspA(m, KSP::GMRES), M1(m), gr{Graph(m), Graph(m)} {
        n0 = m->n0; n1 = m->n1; n2 = m->n2; N = (long long)n1 + n2;
        if (sh) use_graph = false;
        if (std::getenv("MIMSEM_EXPERIMENTS") && std::atoi(std::getenv("MIMSEM_EXPERIMENTS"))) {      // (closed experiments; A/B: scripts/ab_sw_cpp.sh)
            if (const char* e = std::getenv("MIMSEM_SW_STEP2")) two_launch_steps = std::atoi(e) != 0;
            if (const char* e = std::getenv("MIMSEM_SW_DUAL")) dual_solves = std::atoi(e) != 0;
        }
        try {
            for (double** p : {&ui, &uj_buf, &hu, &F, &fu, &p1, &um, &y1, &z1}) *p = mesh->device_alloc(n1);
            for (double** p : {&hi, &hj_buf, &Phi, &t2, &t2b, &hm}) *p = mesh->device_alloc(n2);
            for (double** p : {&m0, &m0fg, &m0h, &dinv, &ones0, &rhs0, &qi, &qj, &p0, &y0}) *p = mesh->device_alloc(n0);
            for (double** p : {&xsave, &res, &bA, &rA, &dA, &rB, &dB, &yA, &zA}) *p = mesh->device_alloc((size_t)N);
            // the two vectors of a check (last residual | its reference) sit side by side: ONE two-row dot per check instead of two
            // (the second row starts at an even offset: 16-byte aligned like every other vector here)
            pair1 = mesh->device_alloc(2*even(n1)); upd1 = pair1; t1 = pair1 + even(n1);
            pair0 = mesh->device_alloc(2*even(n0)); upd0 = pair0; t0 = pair0 + even(n0);
            pairx = mesh->device_alloc(2*even(N)); dx = pairx; x = pairx + even(N);
            chk = mesh->device_alloc(2*NSLOT);
            mimsem_ctx* c = mesh->ctx;
            check(mimsem_pvec(c, 0, 1, 1.0, nullptr, 0, m0, 0), "mimsem_pvec");                                   // M0 is diagonal (collocated 0-forms)
            done0(m0);
            combine(n0, 1.0, m0, 1, fg, 0.0, nullptr, m0fg);                                                      // M0 f
            combine(n0, 1.0, m0, 2, m0, 0.0, nullptr, ones0);
            // ksp1: the 1-form mass matrix with one exact block per element (src/SWEqn_Picard.cpp:84-92)
            M1.assemble();
            ksp1.setOperators(M1); ksp1.setPCBJacobi(); ksp1.setTolerances(rtol, 1.0e-50, 1000);
            ksp0.setTolerances(rtol, 1.0e-50, 1000, 30);
            kspA.setTolerances(rtol, 1.0e-50, 1000, 30);
        } catch (...) { release(); throw; }              // (a constructor that throws runs no destructor)
    }
    ~SWEqn() { release(); }
    SWEqn(const SWEqn&) = delete; SWEqn& operator=(const SWEqn&) = delete;

    // SWEqn::solve (src/SWEqn_Picard.cpp:727-791): un, hn (device) are advanced in place by one time step; `save` (field output) is the
    // host's business and ignored here
    void solve(double* un, double* hn, double dt_, bool /*save*/, int nits, bool q_exact = false, const double* bot = nullptr) {
        if (dt_ != dt || q_exact != qx || bot != bt || fixed_length != set_fixed) setup(dt_, q_exact, bot, un, hn);
        copy(n1, un, ui); copy(n2, hn, hi);
        copy(n1, un, x); copy(n2, hn, x + n1);
        history.clear();
        int it = 0; double norm = 1.0e+9;
        do {
            norm = iteration(it == 0);
            history.push_back(norm);
            it++;
        } while (norm > 1.0e-14 && it < nits);
        copy(n1, x, un); copy(n2, x + n1, hn);
    }

    // the diagnostics on their own (device pointers; results in the caller's arrays)
    void diagnose_F(const double* ui_, const double* uj_, const double* hi_, const double* hj_, double* F_) {          // :253-284
        F_rhs(ui_, uj_, hi_, hj_);
        done1(hu);                                           // (sharded: the four local partial sums, ONE exchange)
        solve_M1(hu, F_);
    }
    void diagnose_Phi(const double* ui_, const double* uj_, const double* hi_, const double* hj_, double* Phi_) {     // :289-320
        mimsem_ctx* c = mesh->ctx;
        check(mimsem_op_apply(c, MIMSEM_OP_WTQUMAT, 0, 1, 1.0, 0, ui_, 0, ui_, 0, Phi_, 0, 1.0/3.0), "WTQUMAT");
        check(mimsem_op_apply(c, MIMSEM_OP_WTQUMAT, 0, 1, 1.0, MIMSEM_FLAG_ACCUM, ui_, 0, uj_, 0, Phi_, 0, 1.0/3.0), "WTQUMAT");
        check(mimsem_op_apply(c, MIMSEM_OP_WTQUMAT, 0, 1, 1.0, MIMSEM_FLAG_ACCUM, uj_, 0, uj_, 0, Phi_, 0, 1.0/3.0), "WTQUMAT");
        combine(n2, 1.0, hi_, 0, nullptr, 1.0, hj_, t2);
        check(mimsem_op_apply(c, MIMSEM_OP_WMAT, 0, 1, 1.0, MIMSEM_FLAG_ACCUM, nullptr, 0, t2, 0, Phi_, 0, grav/2.0), "WMAT");
    }
    // M0h q = M0 f + E01 M1 u; M0h upwinded (Phmat::assemble_up) when dt > 1e-6                                       // :322-341
    void diagnose_q(double dt_, const double* u_, const double* h_, double* q_) {
        mimsem_ctx* c = mesh->ctx;
        check(mimsem_op_apply(c, MIMSEM_OP_UMAT, 0, 1, 1.0, 0, nullptr, 0, u_, 0, t1, 0, 1.0), "UMAT");
        done1(t1);
        check(mimsem_incidence_apply(c, 3, 1, t1, 0, rhs0, 0), "E01");                                         // (every edge counted by the element that owns it)
        done0(rhs0);
        combine(n0, 1.0, m0fg, 0, nullptr, 1.0, rhs0, rhs0);
        check(mimsem_pvec(c, 0, 1, 1.0, h_, 0, m0h, 0), "mimsem_pvec");                                       // Phmat::assemble(h) is diagonal
        done0(m0h);
        if (!(dt_ > 1.0e-6)) { combine(n0, 1.0, rhs0, 2, m0h, 0.0, nullptr, q_); return; }
        combine(n0, 1.0, ones0, 2, m0h, 0.0, nullptr, dinv);
        const double tau = 1.0/(1.0/(UP_TAU*dt_));
        if (inline_fixed && !qcoef.empty()) {
            zero(n0, q_); zero(n0, p0);
            for (size_t k = 0; k < qcoef.size(); k++) {
                if (!sh) {
                    check(mimsem_op_chebyshev_sweep(c, MIMSEM_OP_PHMAT_UP, 0, 1, 1.0, tau, 0, h_, 0, u_, 0, rhs0, 0, dinv, 0, qcoef[k].first, qcoef[k].second,
                                                    p0, 0, q_, 0, k + 1 == qcoef.size() ? upd0 : nullptr, 0), "mimsem_op_chebyshev_sweep");
                    continue;
                }
                // sharded: the operator pass, its result completed over the halo, then the sweep's algebra  z = dinv (b - Op x); p = z + beta p; x += alpha p
                check(mimsem_op_apply_up(c, MIMSEM_OP_PHMAT_UP, 0, 1, 1.0, tau, 0, h_, 0, u_, 0, q_, 0, y0, 0, 1.0), "PHMAT_UP");
                sh->complete0(y0);
                check(mimsem_krylov_chebyshev_px(c, 1, n0, qcoef[k].first, qcoef[k].second, y0, n0, rhs0, n0, dinv, n0, p0, n0, q_, n0,
                                                 k + 1 == qcoef.size() ? upd0 : nullptr, n0), "mimsem_krylov_chebyshev_px");      // one launch for five
            }
            combine(n0, 1.0, rhs0, 1, dinv, 0.0, nullptr, t0);
            log(K_MASS, upd0, t0, n0, sh ? sh->own0 : nullptr);
            return;
        }
        if (sh) throw std::runtime_error("SWEqn (sharded): the upwinded potential-vorticity solve exists in the fixed-length mode only");
        q_h = h_; q_u = u_; q_tau = tau;
        ksp0.setOperatorsShell(n0, &SWEqn::apply_m0h_up, this); ksp0.setPCJacobi(dinv);
        ksp0.solve(rhs0, q_);
    }
    int recalibrations = 0;              // (sharded) times the spectral regions were estimated again after a missed check

private:
    void release() {
        for (double** p : {&ui, &uj_buf, &hu, &F, &fu, &p1, &um, &hi, &hj_buf, &Phi, &t2, &t2b, &hm, &m0, &m0fg, &m0h, &dinv, &ones0, &rhs0, &qi, &qj,
                           &p0, &xsave, &res, &bA, &rA, &dA, &rB, &dB, &chk, &pair1, &pair0, &pairx, &y1, &z1, &y0, &yA, &zA}) { if (*p) mimsem_free(*p); *p = nullptr; }
        t1 = upd1 = t0 = upd0 = x = dx = nullptr;
    }
    static constexpr int NSLOT = 16;
    enum LogKind { K_MASS = 1, K_A = 2, K_PICARD = 3 };
    Mesh* mesh; const double* fg; Shard* sh = nullptr;
    double *y1 = nullptr, *z1 = nullptr, *y0 = nullptr, *yA = nullptr, *zA = nullptr; double widen = 1.0;
    double marginA[2] = {0.90, 1.05}, marginM[2] = {0.90, 1.05};      // the safety margins in force around the two Ritz intervals (setup)
    void done1(double* v) { if (sh) sh->complete1(v); }      // element-local partial sums of a 1-form / 0-form result completed over the halo (one rank: nothing to do)
    void done0(double* v) { if (sh) sh->complete0(v); }
    // (src/Assembly.h's Umat is built from (Topo*, Geom*); this one from the Mesh of a raw descriptor)
    struct MassOp : OperatorBase { explicit MassOp(Mesh* m) : OperatorBase(m, MIMSEM_OP_UMAT) {} void assemble() { up = false; field = nullptr; } };
    KSP ksp1, ksp0, kspA;
    MassOp M1;
    Graph gr[2]; bool have_graph[2] = {false, false}; bool warm[2] = {false, false};
    int n0 = 0, n1 = 0, n2 = 0; long long N = 0;
    double dt = -1.0; bool qx = false; const double* bt = nullptr; bool set_fixed = true; int misses = 0;
    double *ui = nullptr, *uj_buf = nullptr, *hu = nullptr, *F = nullptr, *fu = nullptr, *t1 = nullptr, *p1 = nullptr, *upd1 = nullptr, *um = nullptr;
    double *hi = nullptr, *hj_buf = nullptr, *Phi = nullptr, *t2 = nullptr, *t2b = nullptr, *hm = nullptr;
    double *m0 = nullptr, *m0fg = nullptr, *m0h = nullptr, *dinv = nullptr, *ones0 = nullptr, *rhs0 = nullptr, *t0 = nullptr, *qi = nullptr, *qj = nullptr,
           *p0 = nullptr, *upd0 = nullptr;
    double *x = nullptr, *xsave = nullptr, *res = nullptr, *bA = nullptr, *rA = nullptr, *dA = nullptr, *dx = nullptr, *chk = nullptr;
    double *pair1 = nullptr, *pair0 = nullptr, *pairx = nullptr, *rB = nullptr, *dB = nullptr;
    const double *blocksA = nullptr, *blocks1 = nullptr, *escale1 = nullptr;
    std::vector<std::pair<double, double>> coefM, qcoef; double thetaA = 1.0, deltaA = 1.0;
    bool inline_fixed = false, can_fix = false;
    int slot = 0; int kinds[NSLOT] = {0}; int kinds_of[2][NSLOT] = {{0}}; int nslots_of[2] = {0, 0};
    const double *q_h = nullptr, *q_u = nullptr; double q_tau = 0.0;

    static size_t even(long long n) { return (size_t)((n + 1) & ~1LL); }
    void combine(long long n, double a, const double* A, int op, const double* B, double b, const double* C, double* out) {
        check(mimsem_vec_combine(mesh->ctx, 1, n, a, A, 0, op, B, 0, b, C, 0, out, 0), "mimsem_vec_combine");
    }
    void copy(long long n, const double* a, double* out) { combine(n, 1.0, a, 0, nullptr, 0.0, nullptr, out); }
    void zero(long long n, double* a) { check(mimsem_memset(mesh->ctx, a, 0, n*(long long)sizeof(double)), "mimsem_memset"); }
    void log(int kind, const double* r, const double* ref, long long n, const double* wgt = nullptr) {
        if (slot >= NSLOT) throw std::runtime_error("SWEqn: check-norm slots exhausted");
        kinds[slot] = kind;
        if (wgt) {           // sharded: this rank's ownership-weighted part of both norms; the whole log is all-reduced once per Picard iteration
            double* tmp = n == n0 ? y0 : (n == n1 ? y1 : yA);
            combine(n, 1.0, r, 1, wgt, 0.0, nullptr, tmp);
            check(mimsem_krylov_rowdot(mesh->ctx, 1, n, tmp, n, r, n, chk + 2*slot), "mimsem_krylov_rowdot");
            combine(n, 1.0, ref, 1, wgt, 0.0, nullptr, tmp);
            check(mimsem_krylov_rowdot(mesh->ctx, 1, n, tmp, n, ref, n, chk + 2*slot + 1), "mimsem_krylov_rowdot");
        } else if (ref == r + even(n)) check(mimsem_krylov_rowdot(mesh->ctx, 2, n, r, (long long)even(n), r, (long long)even(n), chk + 2*slot), "mimsem_krylov_rowdot");     // (side by side: both norms in one call)
        else {
            check(mimsem_krylov_rowdot(mesh->ctx, 1, n, r, n, r, n, chk + 2*slot), "mimsem_krylov_rowdot");
            check(mimsem_krylov_rowdot(mesh->ctx, 1, n, ref, n, ref, n, chk + 2*slot + 1), "mimsem_krylov_rowdot");
        }
        slot++;
    }
    static int pc_block_diagonal(void* user, int, const double* r, long long, double* z, long long) {
        SWEqn* s = (SWEqn*)user;
        mimsem_ctx* c = s->mesh->ctx;
        const int rc = mimsem_elem_blocks_apply(c, 1, 1, 0, s->blocks1, 0, s->escale1, 0, r, 0, z, 0, 1.0);
        return rc ? rc : mimsem_op_apply(c, MIMSEM_OP_WMATINV, 0, 1, 1.0, 0, nullptr, 0, r + s->n1, 0, z + s->n1, 0, 1.0);
    }
    static int apply_m0h_up(void* user, int, const double* xin, long long, double* y, long long) {
        SWEqn* s = (SWEqn*)user;
        return mimsem_op_apply_up(s->mesh->ctx, MIMSEM_OP_PHMAT_UP, 0, 1, 1.0, s->q_tau, 0, s->q_h, 0, s->q_u, 0, xin, 0, y, 0, 1.0);
    }

    // hu = 1/3 M1h(hi) ui + 1/6 M1h(hi) uj + 1/6 M1h(hj) ui + 1/3 M1h(hj) uj (:253-284).  M1h is LINEAR in its thickness field: two applies on the
    // combined fields hi/3 + hj/6 and hi/6 + hj/3 (two small combines on 2-forms) instead of four applies -- 6 launches instead of 8
    void F_rhs(const double* ui_, const double* uj_, const double* hi_, const double* hj_) {
        mimsem_ctx* c = mesh->ctx;
        combine(n2, 1.0/3.0, hi_, 0, nullptr, 1.0/6.0, hj_, t2);
        combine(n2, 1.0/6.0, hi_, 0, nullptr, 1.0/3.0, hj_, t2b);
        check(mimsem_op_apply(c, MIMSEM_OP_UHMAT, 0, 1, 1.0, 0, t2, 0, ui_, 0, hu, 0, 1.0), "UHMAT");
        check(mimsem_op_apply(c, MIMSEM_OP_UHMAT, 0, 1, 1.0, MIMSEM_FLAG_ACCUM, t2b, 0, uj_, 0, hu, 0, 1.0), "UHMAT");
    }

    // KSPSolve(ksp1, b, x): the 1-form mass
    void solve_M1(const double* b, double* out) {
        mimsem_ctx* c = mesh->ctx;
        if (inline_fixed && !sh && coefM.size() > 1) {
            // one context: the whole solve from x = 0 as ONE call -- no operator pass in its first step, nothing cleared, P b (the reference norm of
            // the check) is the first step's update (mimsem_block_chebyshev_solve, round 6: 5 launches fewer than the sweeps + the extra preconditioner)
            std::vector<double> flat;
            for (const auto& ab : coefM) { flat.push_back(ab.first); flat.push_back(ab.second); }
            const int rc = mimsem_block_chebyshev_solve(c, MIMSEM_OP_UMAT, 0, 1, 1.0, 0, nullptr, 0, blocks1, escale1, 0, b, 0, (int)coefM.size(), flat.data(),
                                                        out, 0, t1, 0, upd1, 0);
            if (rc != MIMSEM_ERR_UNSUPPORTED) {
                check(rc, "mimsem_block_chebyshev_solve");
                log(K_MASS, upd1, t1, n1);
                return;
            }
        }
        if (inline_fixed) {
            zero(n1, out); zero(n1, p1);
            for (size_t k = 0; k < coefM.size(); k++) {
                if (!sh) {
                    check(mimsem_block_chebyshev_sweep(c, MIMSEM_OP_UMAT, 0, 1, 1.0, 0, nullptr, 0, blocks1, escale1, 0, b, 0, coefM[k].first, coefM[k].second,
                                                       p1, 0, out, 0, k + 1 == coefM.size() ? upd1 : nullptr, 0), "mimsem_block_chebyshev_sweep");
                    continue;
                }
                // sharded: z = P (b - M1 x) with both element-local sums completed over the halo; p = z + beta p; x += alpha p
                check(mimsem_op_apply(c, MIMSEM_OP_UMAT, 0, 1, 1.0, 0, nullptr, 0, out, 0, y1, 0, 1.0), "UMAT");
                sh->complete1(y1);
                combine(n1, -1.0, y1, 0, nullptr, 1.0, b, y1);
                check(mimsem_elem_blocks_apply(c, 1, 1, 0, blocks1, 0, escale1, 0, y1, 0, z1, 0, 1.0), "mimsem_elem_blocks_apply");
                sh->complete1(z1);
                check(mimsem_krylov_chebyshev_px(c, 1, n1, coefM[k].first, coefM[k].second, z1, n1, nullptr, 0, nullptr, 0, p1, n1, out, n1,
                                                 k + 1 == coefM.size() ? upd1 : nullptr, n1), "mimsem_krylov_chebyshev_px");      // p = z + beta p; x += alpha p; upd = z
            }
            check(mimsem_elem_blocks_apply(c, 1, 1, 0, blocks1, 0, escale1, 0, b, 0, t1, 0, 1.0), "mimsem_elem_blocks_apply");
            done1(t1);
            log(K_MASS, upd1, t1, n1, sh ? sh->own1 : nullptr);
            return;
        }
        if (sh) throw std::runtime_error("SWEqn (sharded): the 1-form mass solve exists in the fixed-length mode only");
        ksp1.solve(b, out);
    }

    // once per (dt, q_exact, bot): the [u|h] operator with its coupled element blocks (assemble_operator, :622-725) and the spectral
    // regions the fixed-length solves are built on
    void setup(double dt_, bool q_exact, const double* bot, const double* un, const double* hn) {
        dt = dt_; qx = q_exact; bt = bot; set_fixed = fixed_length; misses = 0;
        have_graph[0] = have_graph[1] = false; warm[0] = warm[1] = false;
        const double a = ROS_ALPHA*dt;
        kspA.setOperatorsSW(a, grav, H_MEAN, fg); kspA.setPCBJacobi();
        can_fix = false;
        // the coupled [u|h] element blocks exist for orders 1..4 (mimsem_sw_blocks_apply: one wavefront per element); above that the
        // preconditioner is block diagonal -- the element blocks of ksp1 on the velocity rows, the exact element-wise inverse of M2 (WmatInv)
        // on the depth rows -- as a PCSHELL, and the solves stay with the KSP objects
        try { kspA.pcBlocks(&blocksA); }
        catch (const std::runtime_error&) {
            ksp1.pcBlocks(&blocks1, &escale1);
            kspA.setPCShell(&SWEqn::pc_block_diagonal, this);
            return;
        }
        if (!fixed_length) return;
        double lo = 0.0, hi_ = 0.0, im = 0.0;
        // (sharded: the host's own Arnoldi process on the COMPLETED operator -- the library's Ritz estimate sees one context's elements only)
        // safety margins around a Ritz interval (round 6; mimsem_amd/krylov.py::ritz_margins): three times what its ends moved between a 25- and
        // a 40-step estimate, at least 1 % -- round 5 took 10 % / 5 % whatever the estimate's quality, which cost 2 of 31 and 2 of 15 steps; a
        // re-estimate after a missed check (widen > 1) opens them by 10 % / 5 % per unit
        double mlo = 0.99, mhi = 1.01, lo25 = 0.0, hi25 = 0.0;
        auto margins = [&]() {
            mlo = 1.0 - std::min(0.4, std::max({0.01, 3.0*std::fabs(lo - lo25)/lo, 0.1*(widen - 1.0)}));
            mhi = 1.0 + std::max({0.01, 3.0*std::fabs(hi_ - hi25)/hi_, 0.05*(widen - 1.0)});
        };
        for (const int m : {25, 40}) {
            lo25 = lo; hi25 = hi_;
            if (sh) sh->ritz(N, m, sh->ownx, [&](const double* v, double* w) { apply_PA(a, v, w); }, [&](double* v) { sh->complete1(v); }, &lo, &hi_, &im);
            else kspA.ritz(m, &lo, &hi_, &im);
        }
        if (!(lo > 0.02 && im <= 0.15*(hi_ - lo))) return;
        margins(); marginA[0] = mlo; marginA[1] = mhi;
        kspA.pcBlocks(&blocksA);
        const double lminA = mlo*lo, lmaxA = mhi*hi_;
        thetaA = 0.5*(lmaxA + lminA); deltaA = 0.5*(lmaxA - lminA);
        steps_A = std::max(2, (int)std::ceil(std::log(0.5*rtol)/std::log(cheb::interval_rate(lminA, lmaxA))) + 1);
        ksp1.pcBlocks(&blocks1, &escale1);
        for (const int m : {25, 40}) {
        lo25 = lo; hi25 = hi_;
        if (sh) sh->ritz(n1, m, sh->own1, [&](const double* v, double* w) {
                             check(mimsem_op_apply(mesh->ctx, MIMSEM_OP_UMAT, 0, 1, 1.0, 0, nullptr, 0, v, 0, y1, 0, 1.0), "UMAT"); sh->complete1(y1);
                             check(mimsem_elem_blocks_apply(mesh->ctx, 1, 1, 0, blocks1, 0, escale1, 0, y1, 0, w, 0, 1.0), "mimsem_elem_blocks_apply"); sh->complete1(w); },
                         [&](double* v) { sh->complete1(v); }, &lo, &hi_, &im, 4321);
        else ksp1.ritz(m, &lo, &hi_, &im);
        }
        if (!(lo > 0.02)) return;
        margins(); marginM[0] = mlo; marginM[1] = mhi;
        const double l1 = mlo*lo, l2 = mhi*hi_;
        steps_M1 = std::max(2, (int)std::ceil(std::log(2.0/rtol)/std::log(1.0/cheb::interval_rate(l1, l2))));
        coefM = cheb::ellipse(0.5*(l1 + l2), 0.25*(l2 - l1)*(l2 - l1), steps_M1);
        qcoef.clear(); steps_q = 0;
        if (!q_exact) {
            // the upwinded lumped 0-form mass under its diagonal: 1 +- i sigma (the upwinding is a skew perturbation of the identity)
            check(mimsem_pvec(mesh->ctx, 0, 1, 1.0, hn, 0, m0h, 0), "mimsem_pvec");
            done0(m0h);
            combine(n0, 1.0, ones0, 2, m0h, 0.0, nullptr, dinv);
            q_h = hn; q_u = un; q_tau = 1.0/(1.0/(UP_TAU*dt));
            if (sh) sh->ritz(n0, 40, sh->own0, [&](const double* v, double* w) {
                                 check(mimsem_op_apply_up(mesh->ctx, MIMSEM_OP_PHMAT_UP, 0, 1, 1.0, q_tau, 0, q_h, 0, q_u, 0, v, 0, w, 0, 1.0), "PHMAT_UP"); sh->complete0(w);
                                 combine(n0, 1.0, w, 1, dinv, 0.0, nullptr, w); },
                             [&](double* v) { sh->complete0(v); }, &lo, &hi_, &im);
            else {
                ksp0.setOperatorsShell(n0, &SWEqn::apply_m0h_up, this); ksp0.setPCJacobi(dinv);
                ksp0.ritz(40, &lo, &hi_, &im);
            }
            const double d0 = 0.5*(hi_ + lo), a_re = 0.5*(hi_ - lo)*1.5*widen + 0.01, a_im = im*1.2*widen + 0.01, rate = cheb::ellipse_rate(d0, a_re, a_im);
            if (!(d0 > 0.2 && rate < 0.6)) return;
            steps_q = std::max(2, (int)std::ceil(std::log(0.5*rtol)/std::log(rate)) + 1);
            qcoef = cheb::ellipse(d0, a_re*a_re - a_im*a_im, steps_q);
        }
        can_fix = true;
    }

    // sharded: w = P A v on packed [u|h] rows with both element-local sums completed over the halo (element pass + gather, EXCHANGE, block pass +
    // gather, EXCHANGE: what one context does in three launches without ever assembling A v)
    void apply_PA(double a, const double* v, double* w) {
        mimsem_ctx* c = mesh->ctx;
        check(mimsem_sw_operator_apply(c, 1, a, grav, H_MEAN, fg, 0, v, 0, yA, 0), "mimsem_sw_operator_apply");
        sh->complete1(yA);
        check(mimsem_sw_blocks_apply(c, 1, blocksA, yA, 0, w, 0), "mimsem_sw_blocks_apply");
        sh->complete1(w);
    }

    // assemble_residual (:402-607) + KSPSolve(kspA, -f, dx) + x += dx (:751-757) on the member arrays
    void body(bool first) {
        mimsem_ctx* c = mesh->ctx;
        slot = 0;
        copy(N, x, xsave);
        // the current iterate's halves: views of x where the depth half starts 16-byte aligned (x changes only at the end of the body), else copies
        const double *uj = x, *hj = x + n1;
        if (n1 & 1) { copy(n1, x, uj_buf); copy(n2, x + n1, hj_buf); uj = uj_buf; hj = hj_buf; }
        // round 6: the mass-flux solve (diagnose_F: steps_M1 sweeps x 3 launches) and the potential-vorticity solve (diagnose_q: steps_q x 2) read
        // nothing of each other -- mimsem_sw_dual_chebyshev issues launch k of both chains as ONE grid (the same kernels' bodies, the same bits)
        const bool dual = dual_solves && inline_fixed && !sh && !qx && !qcoef.empty() && !escale1;
        bool q_done = false;
        if (dual) {
            const double* uq = first ? ui : uj; const double* hq = first ? hi : hj; double* qdst = first ? qi : qj;
            F_rhs(ui, uj, hi, hj);                                                                                                   // the right-hand side of diagnose_F
            check(mimsem_op_apply(c, MIMSEM_OP_UMAT, 0, 1, 1.0, 0, nullptr, 0, uq, 0, t1, 0, 1.0), "UMAT");                          // ... and of diagnose_q
            check(mimsem_incidence_apply(c, 3, 1, t1, 0, rhs0, 0), "E01");
            combine(n0, 1.0, m0fg, 0, nullptr, 1.0, rhs0, rhs0);
            check(mimsem_pvec(c, 0, 1, 1.0, hq, 0, m0h, 0), "mimsem_pvec");
            combine(n0, 1.0, ones0, 2, m0h, 0.0, nullptr, dinv);
            std::vector<double> ca, cb;      // (F, p1, q, p0: outputs / workspaces of solves from x = 0 -- nothing to clear)
            for (auto& pr : coefM) { ca.push_back(pr.first); ca.push_back(pr.second); }
            for (auto& pr : qcoef) { cb.push_back(pr.first); cb.push_back(pr.second); }
            // (the checks' reference vectors, P hu and dinv rhs0, are the first steps' updates: t1, t0)
            check(mimsem_sw_dual_chebyshev(c, (int)coefM.size(), ca.data(), blocks1, hu, p1, F, upd1, t1, (int)qcoef.size(), cb.data(), 1.0/(1.0/(UP_TAU*dt)), hq, uq, rhs0, dinv,
                                           p0, qdst, upd0, t0), "mimsem_sw_dual_chebyshev");
            log(K_MASS, upd1, t1, n1);
            log(K_MASS, upd0, t0, n0);
            q_done = true;
        } else {
            if (first && !qx) diagnose_q(dt, ui, hi, qi);
            diagnose_F(ui, uj, hi, hj, F);
        }
        diagnose_Phi(ui, uj, hi, hj, Phi);
        if (bt) check(mimsem_op_apply(c, MIMSEM_OP_WMAT, 0, 1, 1.0, MIMSEM_FLAG_ACCUM, nullptr, 0, bt, 0, Phi, 0, grav), "WMAT");
        check(mimsem_incidence_apply(c, 2, 1, Phi, 0, fu, 0), "E12");
        if (qx) {
            combine(n1, 0.5, ui, 0, nullptr, 0.5, uj, um); combine(n2, 0.5, hi, 0, nullptr, 0.5, hj, hm);
            diagnose_q(0.0, um, hm, qj);
            check(mimsem_op_apply(c, MIMSEM_OP_ROTMAT, 0, 1, 1.0, MIMSEM_FLAG_ACCUM, qj, 0, F, 0, fu, 0, 1.0), "ROTMAT");              // fu += R(q) F
        } else {
            const double tau = 1.0/(1.0/(UP_TAU*dt));
            const double* qj_ = qi;
            if (!first) { if (!q_done) diagnose_q(dt, uj, hj, qj); qj_ = qj; }
            check(mimsem_op_apply_up(c, MIMSEM_OP_ROTMAT_UP, 0, 1, 1.0, tau, MIMSEM_FLAG_ACCUM, qi, 0, ui, 0, F, 0, fu, 0, 0.5), "ROTMAT_UP");
            check(mimsem_op_apply_up(c, MIMSEM_OP_ROTMAT_UP, 0, 1, 1.0, tau, MIMSEM_FLAG_ACCUM, qj_, 0, uj, 0, F, 0, fu, 0, 0.5), "ROTMAT_UP");
        }
        // the mass terms are linear: M1 (uj - ui) + dt fu and M2 (hj - hi + dt E21 F)
        combine(n1, 1.0, uj, 0, nullptr, -1.0, ui, t1);
        check(mimsem_op_apply(c, MIMSEM_OP_UMAT, 0, 1, 1.0, 0, nullptr, 0, t1, 0, res, 0, 1.0), "UMAT");
        combine(n1, dt, fu, 0, nullptr, 1.0, res, res);
        done1(res);                                          // (sharded: E12 Phi, the rotational terms and M1 (uj - ui) were all LOCAL partial sums: one exchange for the lot)
        check(mimsem_incidence_apply(c, 1, 1, F, 0, t2, 0), "E21");
        combine(n2, 1.0, hj, 0, nullptr, -1.0, hi, t2b);
        combine(n2, dt, t2, 0, nullptr, 1.0, t2b, t2b);
        check(mimsem_op_apply(c, MIMSEM_OP_WMAT, 0, 1, 1.0, 0, nullptr, 0, t2b, 0, res + n1, 0, 1.0), "WMAT");
        if (inline_fixed) {
            // A dx = -f: P f first, the sign rides in the start kernel (r = -P f; d = r / theta; dx = 0: one launch for what were a negation, a clear
            // and a scaling); |P f| = |-P f| is the reference norm of the check
            const double a = ROS_ALPHA*dt, sigma1 = thetaA/deltaA;
            check(mimsem_sw_blocks_apply(c, 1, blocksA, res, 0, rA, 0), "mimsem_sw_blocks_apply");
            done1(rA);
            if (sh) { combine(N, 1.0, rA, 1, sh->ownx, 0.0, nullptr, zA); check(mimsem_krylov_rowdot(c, 1, N, zA, N, rA, N, chk + 2*slot + 1), "mimsem_krylov_rowdot"); }
            else check(mimsem_krylov_rowdot(c, 1, N, rA, N, rA, N, chk + 2*slot + 1), "mimsem_krylov_rowdot");
            check(mimsem_krylov_chebyshev_start(c, 1, N, -1.0, thetaA, rA, N, rA, N, dA, N, dx, N), "mimsem_krylov_chebyshev_start");
            double rho = 1.0/sigma1;
            if (sh) {
                // sharded: a step = P A d with its two exchanges, then ONE update launch  x += d; r -= P A d; d = ca d + cb r  -- no inner product
                for (int k = 0; k < steps_A; k++) {
                    const double rho_new = 1.0/(2.0*sigma1 - rho);
                    apply_PA(a, dA, zA);
                    check(mimsem_krylov_chebyshev_update(c, 1, N, rho_new*rho, 2.0*rho_new/deltaA, zA, N, dx, N, rA, N, dA, N), "mimsem_krylov_chebyshev_update");
                    rho = rho_new;
                }
            } else if (two_launch_steps) {
                // the 1-form part of a step's update rides in the element pass of the next step: r and d alternate between two pairs of arrays
                double *rin = rA, *din = dA, *rout = rB, *dout = dB, pca = 0.0, pcb = 0.0;
                for (int k = 0; k < steps_A; k++) {
                    const double rho_new = 1.0/(2.0*sigma1 - rho), ca = rho_new*rho, cb = 2.0*rho_new/deltaA;
                    check(mimsem_sw_chebyshev_step2(c, 1, a, grav, H_MEAN, fg, 0, blocksA, k > 0, pca, pcb, ca, cb, dx, 0, rin, din, rout, dout, rA, dA, 0),
                          "mimsem_sw_chebyshev_step2");
                    if (k > 0) { std::swap(rin, rout); std::swap(din, dout); }
                    pca = ca; pcb = cb; rho = rho_new;
                }
                check(mimsem_sw_chebyshev_flush(c, 1, pca, pcb, dx, 0, rin, din, 0), "mimsem_sw_chebyshev_flush");
                if (rin != rA) copy(n1, rin, rA);                     // (the norm below is taken over the packed residual)
            } else
            for (int k = 0; k < steps_A; k++) {
                const double rho_new = 1.0/(2.0*sigma1 - rho);
                check(mimsem_sw_operator_precond_chebyshev(c, 1, a, grav, H_MEAN, fg, 0, blocksA, rho_new*rho, 2.0*rho_new/deltaA, dx, 0, rA, 0, dA, 0),
                      "mimsem_sw_operator_precond_chebyshev");
                rho = rho_new;
            }
            if (sh) { combine(N, 1.0, rA, 1, sh->ownx, 0.0, nullptr, zA); check(mimsem_krylov_rowdot(c, 1, N, zA, N, rA, N, chk + 2*slot), "mimsem_krylov_rowdot"); }
            else check(mimsem_krylov_rowdot(c, 1, N, rA, N, rA, N, chk + 2*slot), "mimsem_krylov_rowdot");
            kinds[slot++] = K_A;
        } else {
            if (sh) throw std::runtime_error("SWEqn (sharded): the [u|h] solve exists in the fixed-length mode only");
            combine(N, -1.0, res, 0, nullptr, 0.0, nullptr, bA);
            kspA.solve(bA, dx);
        }
        if (!sh) {
            // x += dx and both norms of the stopping test in ONE launch
            if (slot >= NSLOT) throw std::runtime_error("SWEqn: check-norm slots exhausted");
            check(mimsem_krylov_axpy_dots(c, N, dx, x, chk + 2*slot), "mimsem_krylov_axpy_dots");
            kinds[slot++] = K_PICARD;
            return;
        }
        combine(N, 1.0, dx, 0, nullptr, 1.0, x, x);
        log(K_PICARD, dx, x, N, sh->ownx);
    }

    // one Picard iteration; returns |dx| / |x|
    double iteration(bool first) {
        const int g = first ? 0 : 1;
        double v[2*NSLOT];
        if (fixed_length && can_fix) {
            inline_fixed = true;
            const auto t0 = std::chrono::steady_clock::now();
            auto t1 = t0;
            const bool replay = use_graph && have_graph[g];
            if (replay) { gr[g].launch(); t1 = std::chrono::steady_clock::now(); }
            else {
                body(first);                                   // eagerly the first time (the library's workspaces get their sizes) ...
                for (int k = 0; k < slot; k++) kinds_of[g][k] = kinds[k];
                nslots_of[g] = slot;
                warm[g] = true;
            }
            inline_fixed = false;
            mesh->to_host(v, chk, 2*NSLOT);
            if (sh) sh->allreduce(v, 2*nslots_of[g]);          // the ONE all-reduce of a Picard iteration: every check norm was a rank-local, ownership-weighted sum
            if (replay) {
                const auto t2 = std::chrono::steady_clock::now();
                us_submit += std::chrono::duration<double, std::micro>(t1 - t0).count(); us_wait += std::chrono::duration<double, std::micro>(t2 - t1).count();
                replays++;
            }
            double norm = 0.0; bool ok = true;
            for (int k = 0; k < nslots_of[g]; k++) {
                const double r2 = v[2*k], ref2 = v[2*k + 1], rel = ref2 > 0.0 ? std::sqrt(r2/ref2) : 0.0;
                if (kinds_of[g][k] == K_PICARD) { norm = rel; ok = ok && norm == norm; continue; }
                // the mass solves log the residual the LAST sweep saw (one more contraction lies between it and the result): a factor 30 of
                // slack; the [u|h] system logs the recurrence residual of the result itself
                if (!(rel <= rtol*(kinds_of[g][k] == K_MASS ? 30.0 : 3.0))) ok = false;
            }
            if (ok) {
                misses = 0;
                if (use_graph && !have_graph[g] && warm[g]) {  // ... and recorded for every later call (recording executes nothing)
                    inline_fixed = true;
                    try { gr[g].record([&] { body(first); }); } catch (...) { inline_fixed = false; throw; }
                    inline_fixed = false;
                    have_graph[g] = true;
                }
                return norm;
            }
            fallbacks++;
            copy(N, xsave, x);
            if (sh) {
                // sharded: no Krylov mode to hand the iteration to.  The regions were estimated on an earlier state: estimate them again from the
                // start-of-step state with wider margins and take the iteration again (every rank takes this branch: the norms were all-reduced)
                if (++misses >= 3) throw std::runtime_error("SWEqn (sharded): three Picard iterations in a row missed their checks after re-estimating the spectral regions");
                recalibrations++;
                widen += 0.5;
                const int keep = misses;
                setup(dt, qx, bt, ui, hi);
                misses = keep;
                if (!can_fix) throw std::runtime_error("SWEqn (sharded): the spectral regions do not admit the fixed-length solves");
                return iteration(first);
            }
            if (++misses >= 3) can_fix = false;                // three iterations in a row: the spectral regions no longer hold -- the KSP objects from here on
        }
        if (sh) throw std::runtime_error("SWEqn (sharded): the spectral regions do not admit the fixed-length solves (no Krylov mode on a shard)");
        body(first);
        mesh->to_host(v, chk, 2*NSLOT);
        const int k = slot - 1;                                // the Picard norms are the last slot logged
        return v[2*k + 1] > 0.0 ? std::sqrt(v[2*k]/v[2*k + 1]) : 0.0;
    }
};

}  // namespace src
}  // namespace mimsem_host
